import torch, time
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/n
N=32*128*160*256
x=torch.randn(N,device='cuda').bfloat16(); y=torch.empty_like(x); z=torch.randn(N,device='cuda').bfloat16()
small=torch.randn(N//4,device='cuda').bfloat16()
mb=N*2/1e6
ms=t(lambda: y.copy_(x)); print("copy 336MB->336MB: %.1f us  %.0f GB/s total"%(ms*1e3, 2*mb/ms/1e3))
ms=t(lambda: y.zero_()); print("fill 336MB: %.1f us  %.0f GB/s"%(ms*1e3, mb/ms/1e3))
ms=t(lambda: torch.add(x,z,out=y)); print("add 2x336 -> 336: %.1f us %.0f GB/s total"%(ms*1e3, 3*mb/ms/1e3))
ms=t(lambda: torch.relu_(y)); print("relu_ inplace 336 r + 336 w: %.1f us %.0f GB/s"%(ms*1e3, 2*mb/ms/1e3))
ms=t(lambda: x.sum()); print("read-only sum 336MB: %.1f us %.0f GB/s"%(ms*1e3, mb/ms/1e3))
xf=torch.randn(N,device='cuda'); yf=torch.empty_like(xf)
ms=t(lambda: yf.copy_(xf)); print("copy f32 671MB->671MB: %.1f us %.0f GB/s total"%(ms*1e3, 4*mb/ms/1e3))
