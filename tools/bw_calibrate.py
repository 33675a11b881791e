import torch, time
torch.cuda.set_device(0)
n = 18399744  # 264^3
def bench(f, nbytes, name, it=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(it): f()
    torch.cuda.synchronize(); dt=(time.perf_counter()-t)/it
    print("%-40s %.3f ms  %.2f TB/s" % (name, dt*1e3, nbytes/dt/1e12))
a = torch.randn(8, n, dtype=torch.float64, device='cuda'); b = torch.empty_like(a)
bench(lambda: b.copy_(a), 2*a.numel()*8, "copy 8 planes (1.18GB) f64")
a15 = torch.randn(15, n, dtype=torch.float64, device='cuda'); o8 = torch.empty(8, n, dtype=torch.float64, device='cuda')
bench(lambda: torch.sum(a15, dim=0, out=o8[0]), 16*n*8, "sum over 15 planes -> 1 plane")
x = torch.randn(n*8, dtype=torch.float64, device='cuda'); y = torch.empty_like(x)
bench(lambda: torch.mul(x, 2.0, out=y), 2*x.numel()*8, "y = 2x (1.18GB each)")
bench(lambda: x.mul_(2.0), 2*x.numel()*8, "x *= 2 in place")
big = torch.empty(4*1024**3//8, dtype=torch.float64, device='cuda')
bench(lambda: big.fill_(1.0), big.numel()*8, "fill 4GB")
bench(lambda: big.sum(), big.numel()*8, "sum 4GB (read only)")
