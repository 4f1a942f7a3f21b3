import sys, os, time, numpy as np, torch
sys.path.insert(0, os.getcwd())
from topo4d_amd import texture
res = 8192
dev = torch.device("cuda")
v = torch.tensor([[1., 1., 0.], [9., 1., 0.], [1., 9., 0.]], device=dev)
t = torch.tensor([[0, 1, 2]], dtype=torch.int32, device=dev)
c = torch.rand(3, 3, device=dev)
img = texture.render_colors(v, t, c, res, res); torch.cuda.synchronize()
best = 1e9
for _ in range(5):
    t0 = time.perf_counter()
    for _ in range(10): img = texture.render_colors(v, t, c, res, res)
    torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / 10)
print("empty-mesh bake ms", round(best * 1e3, 3), "GB/s", round(res * res * 16 / best / 1e9, 1))
# a plain torch fill of the same bytes for comparison
a = torch.empty(res, res, 4, device=dev)
for _ in range(3): a.zero_()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): a.zero_()
torch.cuda.synchronize(); print("torch zero_ of 1.07 GB ms", round((time.perf_counter() - t0) / 20 * 1e3, 3))
