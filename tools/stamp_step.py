"""Diagnostic: phase breakdown of mg_step_kernel from in-kernel s_memtime stamps.
Builds a SEPARATE library with -DDCD_STAMPS (never the shipped one) and prints median cycles per phase."""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dcd-isaac_amd")); sys.path.insert(0, ROOT)
import torch
csrc = os.path.join(ROOT, "dcd-isaac_amd", "csrc")
out = os.path.join(ROOT, "gpurun_out", "libdcd_hip_stamps.so")
os.makedirs(os.path.dirname(out), exist_ok=True)
subprocess.check_call(["make", "-C", csrc], stdout=subprocess.DEVNULL)            # the other objects, unchanged
obj = out.replace(".so", "_mg_kernels.o")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
                       "-DDCD_STAMPS", "-c", "-o", obj, os.path.join(csrc, "mg_kernels.hip")])
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-fPIC", "-shared", "-o", out, obj] +
                      [os.path.join(csrc, f) for f in ("scan_kernels.o", "policy_kernels.o", "host_utils.o", "mg_big_p32.o", "mg_big_p64.o", "mg_big_p128.o")])
import dcd_hip
dcd_hip.LIB_PATH = out
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
U8 = len(sys.argv) > 2 and sys.argv[2] == "u8"
eng = dcd_hip.MultiGridEngine(B, n_clutter=25, choose_goal_last=1)
eng.reset_random()
nblk = (B + 15) // 16 if B < 32 * 512 else ((B + 31) // 32 if B < 64 * 512 else (B + 63) // 64)
st = torch.zeros((nblk, 8), dtype=torch.int64, device="cuda")
dcd_hip.lib().dcd_internal_set_stamps(eng._h, C.c_void_p(st.data_ptr()))
outb = eng.step_buffers(u8=U8, f32=not U8)
names = ["init_luts", "stage_in", "logic", "dyn store+writeback", "render_emit", "term emit"]
import numpy as np
acc = []
spans = []
events = []
for t in range(300):
    a = dcd_hip.engine.random_actions(B, 0, 3, t, 7, "cuda")
    eng.step(a, 0, outb)
    torch.cuda.synchronize()
    s = st.cpu().numpy()
    acc.append(np.diff(s[:, :7], axis=1))
    dn = outb["done"].view(nblk, -1).any(1).cpu().numpy() if B % nblk == 0 else None
    rw = (outb["reward"].view(nblk, -1) > 0).any(1).cpu().numpy() if B % nblk == 0 else None
    events.append((dn, rw))
    span = (s[:, 1:7].max() - s[:, 0].min())
    spans.append(span)
spans = np.array(spans)
print('grid span (first stamp to last stamp) per launch: median', np.median(spans[5:]), 'p90', np.percentile(spans[5:], 90), 'max', spans[5:].max(), 'at step', 5 + spans[5:].argmax())
print('spans around step 249:', spans[245:255])
acc = np.stack(acc[5:])
med = np.median(acc, axis=(0, 1))
for n, m in zip(names, med):
    print(f"{n:24s} {m:9.0f} cycles")
print("per-block total median", np.median(acc.sum(-1)), "cycles; grid span last launch", span)
tot = acc.sum(-1)                                   # [launch, block]
worst = tot.argmax(1)
print("slowest block per launch: mean total", tot.max(1).mean(), "cycles; phases (mean over launches):")
for i, n in enumerate(names):
    print(f"  {n:24s} {np.mean([acc[l, worst[l], i] for l in range(len(worst))]):9.0f}")
wl = tot.max(1).argmax(); wb = tot[wl].argmax()
print("worst launch: step", 5 + wl, "block", wb, "phases", dict(zip(names, acc[wl, wb].tolist())), "median phases of that launch", dict(zip(names, np.median(acc[wl], axis=0).tolist())))
print("percentiles of per-block total (50/90/99/100):", np.percentile(tot, [50, 90, 99, 100]))
if events[0][0] is not None:
    ev = events[5:]
    logic = acc[:, :, 2]
    dn = np.stack([e[0] for e in ev]); rw = np.stack([e[1] for e in ev])
    print("logic cycles, blocks with no episode end: median", np.median(logic[~dn]), "p99", np.percentile(logic[~dn], 99), "max", logic[~dn].max())
    print("logic cycles, blocks with a goal reached:", "median", np.median(logic[rw]) if rw.any() else None, "n", int(rw.sum()))
    print("logic cycles, blocks with an episode end w/o goal:", "median", np.median(logic[dn & ~rw]) if (dn & ~rw).any() else None, "n", int((dn & ~rw).sum()))
    slow = [(dn[l, worst[l]], rw[l, worst[l]]) for l in range(len(worst))]
    print("slowest block of a launch: has episode end in", np.mean([a for a, _ in slow]), "of launches, a goal in", np.mean([b for _, b in slow]))
spans = []

