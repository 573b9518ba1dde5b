"""Host-side timing of dcd_sampler_draws (no GPU): N buffered levels, n sequential draws; checks the result against
the numpy statement of LevelSampler.sample_replay_levels on a prefix of the draws."""
import ctypes as C, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
L = C.CDLL(os.path.join(ROOT, "dcd-isaac_amd", "dcd_hip", "libdcd_hip.so"))
L.dcd_sampler_draws.restype = C.c_int64
N, n = int(sys.argv[1]) if len(sys.argv) > 1 else 4000, int(sys.argv[2]) if len(sys.argv) > 2 else 4096
rng = np.random.default_rng(0)
wsc = rng.random(N); wsc = wsc / wsc.sum() * 0.7
st0 = rng.integers(0, 50, N).astype(np.float64)
keep = (rng.random(N) < 0.95).astype(np.float64)
u = rng.random(n)
p_ = lambda a: a.ctypes.data_as(C.c_void_p)
best = 1e9
for rep in range(5):
    st = st0.copy(); idx = np.zeros(n, dtype=np.int64)
    t = time.perf_counter()
    k = L.dcd_sampler_draws(C.c_int64(N), C.c_int64(n), p_(wsc), C.c_double(0.3), p_(st), p_(keep), p_(u), C.c_int32(1), p_(idx))
    best = min(best, time.perf_counter() - t)
print(f"N={N} n={n}: {best / n * 1e6:.2f} us per draw ({k} draws)")
# numpy statement, first 200 draws
st = st0.copy(); ref = []
for j in range(min(200, n)):
    sw = st * keep; z = np.sum(sw); sw = sw / z
    w = wsc + 0.3 * sw; tot = np.sum(w)
    if tot != 1.0: w = w / tot
    cdf = np.cumsum(w); cdf /= cdf[-1]
    i = int(np.searchsorted(cdf, u[j], side="right")); ref.append(i)
    st = st + 1; st[i] = 0
assert ref == idx[:len(ref)].tolist(), "mismatch vs numpy"
st_c = st0.copy(); idx_c = np.zeros(len(ref), dtype=np.int64)
L.dcd_sampler_draws(C.c_int64(N), C.c_int64(len(ref)), p_(wsc), C.c_double(0.3), p_(st_c), p_(keep), p_(u), C.c_int32(1), p_(idx_c))
assert np.array_equal(st_c, st) and idx_c.tolist() == ref, "staleness after the draws differs from numpy"
# early stop (an unseen-only buffer: z == 0) and non-integer staleness (general path) leave consistent state
st_z = np.zeros(N); k0 = L.dcd_sampler_draws(C.c_int64(N), C.c_int64(5), p_(wsc), C.c_double(0.3), p_(st_z), p_(keep), p_(u), C.c_int32(1), p_(idx_c))
assert k0 == 0 and not st_z.any()
st_f = st0 + 0.5; st_n = st_f.copy(); ref_f = []
for j in range(20):
    sw = st_n * keep; sw = sw / np.sum(sw); w = wsc + 0.3 * sw; tot = np.sum(w); w = w / tot if tot != 1.0 else w
    cdf = np.cumsum(w); cdf /= cdf[-1]; i = int(np.searchsorted(cdf, u[j], side="right")); ref_f.append(i); st_n = st_n + 1; st_n[i] = 0
k1 = L.dcd_sampler_draws(C.c_int64(N), C.c_int64(20), p_(wsc), C.c_double(0.3), p_(st_f), p_(keep), p_(u), C.c_int32(1), p_(idx_c))
assert k1 == 20 and idx_c[:20].tolist() == ref_f and np.array_equal(st_f, st_n), "general path differs from numpy"
print("first", len(ref), "draws equal the numpy statement; checksum", int(idx.sum()), float(st0.sum()))
