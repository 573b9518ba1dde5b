// Diagnostic: per-launch cost of back-to-back dependent kernels on one stream (eager), MI355X.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <chrono>
__global__ void k_empty(int* p) { if (p && threadIdx.x == 9999) p[0] = 1; }
__global__ void k_spin(float* out, int spin) { float x = threadIdx.x; for (int i = 0; i < spin; ++i) x = x * 1.0001f + 0.5f; if (x == 12345.f) out[0] = x; }
__global__ void k_lds(float* out, int spin) { __shared__ float s[4096]; s[threadIdx.x] = threadIdx.x; __syncthreads(); float x = s[(threadIdx.x * 7) & 63]; for (int i = 0; i < spin; ++i) x = x * 1.0001f + 0.5f; if (x == 12345.f) out[0] = x; }
__global__ void k_scratch(float* out, int spin, int idx) { float a[32]; for (int i = 0; i < 32; ++i) a[i] = i * spin; a[idx & 31] = 3.f; float x = a[(idx * 7) & 31]; for (int i = 0; i < spin; ++i) x = x * 1.0001f + 0.5f; if (x == 12345.f) out[0] = x; }
template <class F> void timeit(const char* name, F f, int reps) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 20; ++i) f();
  hipDeviceSynchronize();
  auto t0 = std::chrono::high_resolution_clock::now();
  hipEventRecord(e0, 0);
  for (int i = 0; i < reps; ++i) f();
  hipEventRecord(e1, 0);
  auto t1 = std::chrono::high_resolution_clock::now();
  hipDeviceSynchronize();
  auto t2 = std::chrono::high_resolution_clock::now();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%-28s gpu %.2f us/launch   host-issue %.2f us/launch   wall %.2f us/launch\n", name, ms * 1e3 / reps,
         std::chrono::duration<double, std::micro>(t1 - t0).count() / reps, std::chrono::duration<double, std::micro>(t2 - t0).count() / reps);
}
int main() {
  float* out; hipMalloc(&out, 1024);
  const int R = 2000;
  timeit("empty 16x256", [&] { hipLaunchKernelGGL(k_empty, dim3(16), dim3(256), 0, 0, nullptr); }, R);
  timeit("empty 256x64", [&] { hipLaunchKernelGGL(k_empty, dim3(256), dim3(64), 0, 0, nullptr); }, R);
  timeit("empty 4096x64", [&] { hipLaunchKernelGGL(k_empty, dim3(4096), dim3(64), 0, 0, nullptr); }, R);
  timeit("spin2000 256x64", [&] { hipLaunchKernelGGL(k_spin, dim3(256), dim3(64), 0, 0, out, 2000); }, R);
  timeit("spin2000 64x64", [&] { hipLaunchKernelGGL(k_spin, dim3(64), dim3(64), 0, 0, out, 2000); }, R);
  timeit("spin2000 1x64", [&] { hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, 0, out, 2000); }, R);
  timeit("spin200 256x64", [&] { hipLaunchKernelGGL(k_spin, dim3(256), dim3(64), 0, 0, out, 200); }, R);
  timeit("lds spin2000 256x64", [&] { hipLaunchKernelGGL(k_lds, dim3(256), dim3(64), 0, 0, out, 2000); }, R);
  timeit("scratch spin2000 256x64", [&] { hipLaunchKernelGGL(k_scratch, dim3(256), dim3(64), 0, 0, out, 2000, 5); }, R);
  // graph of 200 spin kernels
  hipStream_t s; hipStreamCreate(&s);
  hipGraph_t g; hipGraphExec_t ge;
  hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
  for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(k_spin, dim3(256), dim3(64), 0, s, out, 2000);
  hipStreamEndCapture(s, &g); hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  hipGraphLaunch(ge, s); hipStreamSynchronize(s);
  auto t0 = std::chrono::high_resolution_clock::now();
  for (int i = 0; i < 10; ++i) hipGraphLaunch(ge, s);
  hipStreamSynchronize(s);
  auto t1 = std::chrono::high_resolution_clock::now();
  printf("graph 200x spin2000 256x64: %.2f us/kernel\n", std::chrono::duration<double, std::micro>(t1 - t0).count() / 2000);
  return 0;
}
