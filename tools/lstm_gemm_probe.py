"""Which form of the recurrent step's GEMM is fastest on this stack?  gates[4096,1024] += hm[4096,256] @ W_hh^T (forward) and
dhm[4096,256] = dg[4096,1024] @ W_hh (backward), as MaskedLSTMSequence issues them (models/common.py), against the same products with
the weight laid out the other way.  Event-timed over 400 back-to-back calls each."""
import torch
dev = "cuda:0"
N, H, G = 4096, 256, 1024
torch.manual_seed(0)
hm = torch.randn(N, H, device=dev); w_hh = torch.randn(G, H, device=dev) * 0.05; xp = torch.randn(N, G, device=dev)
dg = torch.randn(N, G, device=dev); out = torch.empty(N, H, device=dev)
w_hh_t_view, w_hh_t_c = w_hh.t(), w_hh.t().contiguous()
w_view2 = w_hh_t_c.t()                        # [G,H] view of a contiguous [H,G]


def timeit(fn, n=400):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


acc = xp.clone()
tmp = torch.empty(N, G, device=dev)
cases = {
    "fwd addmm_(hm, w_hh.t() view)            [current]": lambda: acc.addmm_(hm, w_hh_t_view),
    "fwd addmm_(hm, w_hh_t contiguous)": lambda: acc.addmm_(hm, w_hh_t_c),
    "fwd mm(hm, w_hh.t() view, out=tmp) (beta 0)": lambda: torch.mm(hm, w_hh_t_view, out=tmp),
    "fwd mm(hm, w_hh_t contiguous, out=tmp) (beta 0)": lambda: torch.mm(hm, w_hh_t_c, out=tmp),
    "bwd mm(dg, w_hh, out)                    [current]": lambda: torch.mm(dg, w_hh, out=out),
    "bwd mm(dg, (w_hh_t contiguous).t() view, out)": lambda: torch.mm(dg, w_view2, out=out),
}
for name, fn in cases.items():
    us = timeit(fn)
    print(f"{name:58s} {us:7.2f} us   {2 * N * H * G / us / 1e6:6.1f} TFLOP/s", flush=True)

# the same products through the other BLAS backend torch can route to (default on this stack: see the first line printed)
print("default preferred_blas_library:", torch.backends.cuda.preferred_blas_library())
for lib in ("cublas", "cublaslt"):                 # (= rocBLAS, hipBLASLt on ROCm)
    try:
        torch.backends.cuda.preferred_blas_library(lib)
    except Exception as e:                          # noqa: BLE001
        print(lib, "not selectable:", e)
        continue
    print("---- preferred_blas_library =", torch.backends.cuda.preferred_blas_library())
    for name in ("fwd addmm_(hm, w_hh.t() view)            [current]", "bwd mm(dg, w_hh, out)                    [current]",
                 "bwd mm(dg, (w_hh_t contiguous).t() view, out)"):
        us = timeit(cases[name])
        print(f"{name:58s} {us:7.2f} us   {2 * N * H * G / us / 1e6:6.1f} TFLOP/s", flush=True)
# the big contractions of a student update
R = 1048576
x1 = torch.randn(R, 152, device=dev); w1 = torch.randn(1024, 152, device=dev); gbig = torch.randn(R, 1024, device=dev)
for lib in ("cublas", "cublaslt"):
    try:
        torch.backends.cuda.preferred_blas_library(lib)
    except Exception:
        continue
    big = {"proj fwd  x1 @ w1.t()": lambda: x1 @ w1.t(), "proj dx   g @ w1": lambda: gbig @ w1}
    for name, fn in big.items():
        us = timeit(fn, n=20)
        print(f"[{lib}] {name:40s} {us / 1e3:7.3f} ms   {2 * R * 152 * 1024 / us / 1e6:6.1f} TFLOP/s", flush=True)
