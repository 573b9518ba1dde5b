"""Diagnostic: shader clock during (a) a tight stream of tiny kernels and (b) a busy-loop kernel."""
import os, sys, time, subprocess, ctypes as C
import torch
src = r'''
#include <hip/hip_runtime.h>
extern "C" __global__ void probe(unsigned long long* out, int spin) {
  unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  float x = threadIdx.x;
  for (int i = 0; i < spin; ++i) x = x * 1.0001f + 0.5f;
  unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) { out[blockIdx.x * 4 + 0] = c1 - c0; out[blockIdx.x * 4 + 1] = r1 - r0; out[blockIdx.x*4+2] = (unsigned long long)x; }
}
extern "C" void launch(unsigned long long* out, int blocks, int spin, void* stream) {
  hipLaunchKernelGGL(probe, dim3(blocks), dim3(64), 0, (hipStream_t)stream, out, spin);
}
'''
d = "/tmp/clockprobe"; os.makedirs(d, exist_ok=True)
open(d + "/p.hip", "w").write(src)
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-fPIC", "-shared", "-o", d + "/p.so", d + "/p.hip"])
lib = C.CDLL(d + "/p.so")
out = torch.zeros(256 * 4, dtype=torch.int64, device="cuda")
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for blocks, spin, reps in [(256, 2000, 2000), (256, 2000, 20), (4096, 200000, 20), (256, 2000, 2000)]:
    t0 = time.perf_counter()
    for _ in range(reps):
        lib.launch(C.c_void_p(out.data_ptr()), blocks if blocks <= 256 else 256, spin, s)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    o = out.cpu().numpy().reshape(-1, 4)
    mhz = (o[:, 0] / o[:, 1].clip(1)) * 100.0
    print(f"blocks={blocks} spin={spin} reps={reps}: clock median {sorted(mhz)[len(mhz)//2]:.0f} MHz  ({dt/reps*1e6:.1f} us/launch)")
