import os, sys, time, torch
sys.path.insert(0, "/root/repo")
from topo4d_amd import loss
V,H,W=24,512,512
g=torch.Generator().manual_seed(0)
im=torch.rand(V,3,H,W,generator=g).cuda(); gt=torch.rand(V,3,H,W,generator=g).cuda()
cm=(torch.randn(V,3,generator=g)*0.1).cuda(); cc=(torch.randn(V,3,generator=g)*0.05).cuda()
def f():
    with torch.no_grad():
        loss._FusedPhotometric.apply(im,gt,cm,cc)
for _ in range(5): f()
torch.cuda.synchronize(); t0=time.perf_counter()
for _ in range(100): f()
torch.cuda.synchronize(); print("kernels only: %.3f ms per 24 views"%((time.perf_counter()-t0)*10))
