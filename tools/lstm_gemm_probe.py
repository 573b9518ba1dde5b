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
