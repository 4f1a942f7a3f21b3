"""HBM streaming rates seen from a kernel (GPU box): fill (write only), copy (read + write), sum (read only), on 1 GiB."""
import torch, time
n = 1 << 28
x = torch.empty(n, dtype=torch.float32, device="cuda"); y = torch.empty_like(x)
def t(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3
for name, fn, bytes_ in (("fill (write)", lambda: x.fill_(1.0), 4 * n), ("copy (read+write)", lambda: y.copy_(x), 8 * n),
                         ("sum (read)", lambda: x.sum(), 4 * n)):
    s = t(fn)
    print(f"{name:20s} {bytes_ / s / 1e12:.2f} TB/s  ({s * 1e6:.0f} us)")
