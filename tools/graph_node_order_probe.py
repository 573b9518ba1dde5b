"""Which node kinds of a captured HIP graph lose their ordering against the neighbouring kernel nodes on replay?
(root-cause probe for the round-1 abort in at::native::_assert_async_cuda_kernel)

Each variant captures N repetitions of one small pattern into ONE graph, replays it R times with eager work in between and
counts wrong results (every pattern has a known answer):

  reduce      out[i] <- x.max()            PyTorch's multi-block full reduction: hipMemsetAsync of a semaphore (MEMSET NODE)
                                            + reduce kernel whose last block combines the per-block partials
  rowreduce   out[i] <- x.amax(1).amax()   the same maximum from two single-pass reductions (no semaphore, no memset)
  memset      buf.zero_(); buf += 1; out[i] <- buf[0]        an explicit memset node followed by a kernel on the same buffer
  fillk       buf.fill_(0) via a kernel (mul by 0), then += 1 the same hand-off without a memset node (control)
  memcpy      tmp = x * 2 (2.7 MB temporary); dst[i % 8].copy_(tmp)  a D2D MEMCPY NODE draining a temporary whose block the
                                            next repetition reuses; checked by reading dst after the replay
Usage: python tools/graph_node_order_probe.py [N] [R]
"""
import sys
import time

import torch

N = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
R = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = "cuda:0"
torch.manual_seed(0)
x = torch.rand(4096, 169, device=dev) + 1.0
y = torch.rand(4096, 169, device=dev) - 5.0            # a second tensor so that recycled partial buffers hold foreign data
xmax, ymin = x.max().item(), y.min().item()


def run(name, body, check):
    body()
    torch.cuda.synchronize()
    bad0 = check()
    g = torch.cuda.CUDAGraph()
    t0 = time.time()
    with torch.cuda.graph(g):
        body()
    cap = time.time() - t0
    res = []
    for rep in range(R):
        (x @ x.t()[:, :64]).sum()                      # eager work between replays, as a training loop has
        g.replay()
        torch.cuda.synchronize()
        res.append(check())
    print(f"{name:10s} N={N}: eager wrong {bad0}; wrong results per replay {res}; capture {cap:.2f}s", flush=True)


out = torch.zeros(N, 2, device=dev)


def reduce_body():
    for i in range(N):
        torch.add(x.max(), 0, out=out[i, 0])
        torch.add(y.min(), 0, out=out[i, 1])


def rowreduce_body():
    for i in range(N):
        torch.add(x.amax(dim=1).amax(), 0, out=out[i, 0])
        torch.add(y.amin(dim=1).amin(), 0, out=out[i, 1])


def check_out():
    o = out.cpu()
    n = int((o[:, 0] != xmax).sum() + (o[:, 1] != ymin).sum())
    out.zero_()
    return n


buf = torch.zeros(1 << 16, device=dev)
out1 = torch.zeros(N, device=dev)


def memset_body():
    for i in range(N):
        buf.zero_()                                    # contiguous zero_: hipMemsetAsync -> memset node
        buf.add_(1.0)
        torch.add(buf[0], 0, out=out1[i])
        buf.add_(float(i % 7))                         # leave something else behind for the next repetition


def fillk_body():
    for i in range(N):
        buf.mul_(0.0)                                  # the same zeroing by a kernel
        buf.add_(1.0)
        torch.add(buf[0], 0, out=out1[i])
        buf.add_(float(i % 7))


def check_out1():
    n = int((out1.cpu() != 1.0).sum())
    out1.zero_()
    return n


dst = torch.zeros(8, 4096, 169, device=dev)
scale = [float(k + 2) for k in range(8)]


def memcpy_body():
    for i in range(N):
        tmp = x * scale[i % 8]                         # temporary in the graph's pool
        dst[i % 8].copy_(tmp)                          # contiguous, same dtype: hipMemcpyAsync D2D -> memcpy node
        del tmp                                        # its block goes to the next repetition's temporary


def check_dst():
    n = 0
    for k in range(8):
        n += int((dst[k] != x * scale[k]).sum())
    dst.zero_()
    return n


run("reduce", reduce_body, check_out)
run("rowreduce", rowreduce_body, check_out)
run("memset", memset_body, check_out1)
run("fillk", fillk_body, check_out1)
run("memcpy", memcpy_body, check_dst)
