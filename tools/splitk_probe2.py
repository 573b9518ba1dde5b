"""Split-K slice count for the PPO batch's weight gradients a^T b ([R, m]^T [R, k], R = T*B = 1M): time of bmm + sum per S."""
import torch, time
dev = "cuda:0"
def bench(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
for (R, m, k) in [(1048576, 1024, 256), (1048576, 1024, 149), (1048576, 256, 7), (1048576, 32, 256), (110592, 1024, 1024)]:
    A = torch.randn(R, m, device=dev); B = torch.randn(R, k, device=dev)
    res = [("plain", round(bench(lambda: A.t() @ B), 3))]
    for S in (16, 32, 64, 128, 256):
        if R % S: continue
        f = lambda: torch.bmm(A.view(S, R // S, m).transpose(1, 2), B.view(S, R // S, k)).sum(0)
        res.append((S, round(bench(f), 3)))
    print((R, m, k), res, flush=True)
