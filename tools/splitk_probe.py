import torch, time
dev="cuda:0"
def bench(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter()-t)/n*1e3
for (R, m, k) in [(1048576, 32, 256), (1048576, 7, 32), (9437184, 16, 27), (1048576, 1024, 149), (1048576,1024,256)]:
    A=torch.randn(R,m,device=dev); B=torch.randn(R,k,device=dev)
    ref = A.t() @ B
    t0=bench(lambda: A.t() @ B)
    res=[]
    for S in (256, 1024, 4096):
        if R % S: continue
        f=lambda: torch.bmm(A.view(S, R//S, m).transpose(1,2), B.view(S, R//S, k)).sum(0)
        err=(f()-ref).abs().max().item()/ref.abs().max().item()
        res.append((S, round(bench(f),3), f"{err:.1e}"))
    print((R,m,k), "plain ms", round(t0,3), "splitK", res)
# forward skinny: [1M,256]@[256,32]
X=torch.randn(1048576,256,device=dev); W=torch.randn(32,256,device=dev)
print("fwd 1Mx256 @ 256x32 ms", round(bench(lambda: X @ W.t()),3), " 64-wide", round(bench(lambda: X @ torch.randn(64,256,device=dev).t()),3))
G=torch.randn(1048576,32,device=dev)
print("dX 1Mx32 @ 32x256 ms", round(bench(lambda: G @ W),3))
