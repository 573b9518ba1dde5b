// mg_big.hip — the MultiGrid env logic of mg_env.h for grids LARGER than 15 x 15: the fixed and procedurally
// generated evaluation mazes of the reference (LargeCorridor 21, FourRooms 19, PerfectMaze 21 / 51 / 101:
// envs/multigrid/maze.py:175-206, fourrooms.py, mst_maze.py:118-136; driven by eval.py:250-329).
//
// This is NOT the hot path: zero-shot evaluation runs a few hundred envs.  The same per-env functions are compiled
// with a wider record pitch (DCD_PITCH = 32 / 64 / 128, one translation unit each: -DDCD_PITCH=NN) and generic
// pointers (-DDCD_REC_GENERIC), and run one env per thread directly on the records in global memory — no LDS
// staging, no cooperative render.  Semantics (and the oracle they are tested against) are those of the 15 x 15
// engine: dcd_mg_step etc. forward here when the handle was created with size > 15.
// Teacher-side operations (reset / step_adversary / reset_random / mutate_level) are not offered: every
// AdversarialEnv id of the reference is at most 15 x 15.
#include <hip/hip_runtime.h>
#include <string.h>
#include "../../include/dcd_hip.h"
#include "dcd_internal.h"
#include "mg_env.h"

#if DCD_PITCH == 16
#error "mg_big.hip is built with -DDCD_PITCH=32|64|128"
#endif
#define DCD_CAT2(a, b) a##b
#define DCD_CAT(a, b) DCD_CAT2(a, b)
#if DCD_PITCH == 32
#define SFX(name) DCD_CAT(name, _p32)
#elif DCD_PITCH == 64
#define SFX(name) DCD_CAT(name, _p64)
#else
#define SFX(name) DCD_CAT(name, _p128)
#endif

using namespace DCD_NS;

namespace {

struct Big {
  DcdMgConfig cfg; Params p; int device;
  uint8_t* rec = nullptr; uint32_t* seed = nullptr; Metrics* met = nullptr; uint32_t* errs = nullptr;
};

constexpr int TB = 64;
constexpr int MAXVV3 = DCD_MAX_VIEW * DCD_MAX_VIEW * 3;

__device__ __forceinline__ void raise(uint32_t* errs, uint32_t bits, int env) {
  if (bits) { atomicOr(&errs[0], bits); atomicMin(reinterpret_cast<int*>(&errs[1]), env); }
}

// obs bytes [vx][vy][c] (+ dir) -> DcdStudentObs row (u8 as is; fp32 CHW = byte / 10.0f, obs_wrappers.py:88-115)
__device__ void emit_obs(const uint8_t* img, int dir, const DcdStudentObs& o, size_t row, int V) {
  const int VV = V * V;
  if (o.image_u8) for (int k = 0; k < VV * 3; ++k) o.image_u8[row * VV * 3 + k] = img[k];
  if (o.image_f32)
    for (int c = 0; c < 3; ++c) for (int q = 0; q < VV; ++q) o.image_f32[row * VV * 3 + c * VV + q] = (float)img[q * 3 + c] / 10.0f;
  if (o.dir_u8) o.dir_u8[row] = (uint8_t)dir;
  if (o.dir_f32) o.dir_f32[row] = (float)dir;
}
__device__ void emit_current_obs(Env& e, const Params& p, const DcdStudentObs& o, size_t row) {
  uint8_t img[MAXVV3];
  const int n = p.V * p.V * 3;
  if (e.has_pos()) render_view(e, p, img);
  else for (int k = 0; k < n; ++k) img[k] = 0;                       // no agent placed yet: zero observation
  emit_obs(img, e.has_pos() ? e.adir : 0, o, row, p.V);
}

__global__ void __launch_bounds__(TB) big_init_kernel(Params p, uint8_t* rec, uint32_t* seed, Metrics* met, long long base) {
  const int i = blockIdx.x * TB + threadIdx.x;
  if (i >= p.B) return;
  uint8_t* r = rec + (size_t)i * REC_BYTES;
  for (int k = 0; k < REC_BYTES; k += 16) *reinterpret_cast<uint4*>(r + k) = make_uint4(0, 0, 0, 0);
  seed[i] = (uint32_t)(base + i);
  Env e = make_env(r, seed[i]);
  e.sb(7) = (uint8_t)(p.n_clutter + 2);
  Metrics m;
  env_reset(e, p, m, nullptr);                                       // same start state (and draw count) as the 15 x 15 engine
  e.store_dyn();
  met[i] = m;
}

__global__ void big_set_seed_kernel(uint8_t* rec, uint32_t* seed, const int64_t* seeds, int B) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B) return;
  seed[i] = (uint32_t)seeds[i];
  uint8_t* r = rec + (size_t)i * REC_BYTES;
  r[8 * ROW + SIDE] = 0; r[9 * ROW + SIDE] = 0; r[10 * ROW + SIDE] = 0; r[11 * ROW + SIDE] = 0;
}

// op 0: MultiGridEnv.reset() of a fixed level (env_load_fixed_level); op 1: reset_agent (+ TimeLimit / VecMonitor)
__global__ void __launch_bounds__(TB) big_level_kernel(Params p, int op, uint8_t* rec, const uint32_t* seed, Metrics* met,
                                                       uint32_t* errs, const uint8_t* enc, const int32_t* idx, int n_rows,
                                                       DcdStudentObs out) {
  const int row = blockIdx.x * TB + threadIdx.x;
  if (row >= n_rows) return;
  const int env = idx ? idx[row] : row;
  Env e = make_env(rec + (size_t)env * REC_BYTES, seed[env]);
  Metrics m = met[env];
  if (op == 0) env_load_fixed_level(e, p, m, enc + (size_t)row * p.S * p.S * 3);
  else reset_agent(e);
  e.elapsed = 0; e.ep_ret = 0.f; e.ep_len = 0;                       // TimeLimit.reset, VecMonitor.reset
  e.store_dyn();
  met[env] = m;
  raise(errs, e.err, env);
  emit_current_obs(e, p, out, (size_t)row);
}

// dcd_mg_step: env_step_pre / terminal render / env_step_post, then every StepOut field
__global__ void __launch_bounds__(TB) big_step_kernel(Params p, uint8_t* rec, const uint32_t* seed, Metrics* met, uint32_t* errs,
                                                      const int64_t* actions, int flags, DcdStepOut o) {
  const int env = blockIdx.x * TB + threadIdx.x;
  if (env >= p.B) return;
  Env e = make_env(rec + (size_t)env * REC_BYTES, seed[env]);
  Metrics m = met[env]; bool mdirty = false;
  uint8_t term[MAXVV3 + 1];
  const StepResult r = env_step(e, p, m, mdirty, actions[env], flags & ~DCD_STEP_RESET_RANDOM, term);
  e.store_dyn();
  raise(errs, e.err, env);
  if (o.reward) o.reward[env] = r.reward;
  if (o.done) o.done[env] = r.done;
  if (o.info) o.info[env] = r.info;
  if (o.ep_ret) o.ep_ret[env] = r.ep_ret;
  if (o.ep_len) o.ep_len[env] = r.ep_len;
  if (o.masks) o.masks[env] = r.done ? 0.f : 1.f;
  if (o.bad_masks) o.bad_masks[env] = (r.info & DCD_INFO_TRUNC_KEY) ? 0.f : 1.f;
  if (o.cliffhanger_masks) o.cliffhanger_masks[env] = (r.info & DCD_INFO_CLIFFHANGER) ? 0.f : 1.f;
  emit_current_obs(e, p, o.obs, (size_t)env);
  if (r.have_term) emit_obs(term, term[p.V * p.V * 3], o.term_obs, (size_t)env, p.V);
}

__global__ void __launch_bounds__(TB) big_get_encodings_kernel(Params p, const uint8_t* rec, uint8_t* out) {
  const int env = blockIdx.x * TB + threadIdx.x;
  if (env >= p.B) return;
  const uint8_t* r = rec + (size_t)env * REC_BYTES;
  uint8_t* o = out + (size_t)env * p.S * p.S * 3;
  for (int x = 0; x < p.S; ++x) for (int y = 0; y < p.S; ++y) {
    const uint32_t t = enc3(r[y * ROW + x]);
    uint8_t* c = o + (x * p.S + y) * 3;
    c[0] = (uint8_t)t; c[1] = (uint8_t)(t >> 8); c[2] = (uint8_t)(t >> 16);
  }
}

__global__ void __launch_bounds__(TB) big_get_state_kernel(Params p, uint8_t* rec, int32_t* out) {
  const int i = blockIdx.x * TB + threadIdx.x;
  if (i >= p.B) return;
  Env e = make_env(rec + (size_t)i * REC_BYTES, 0);
  int32_t* o = out + (size_t)i * 16;
  o[0] = e.has_pos() ? e.ax : -1; o[1] = e.has_pos() ? e.ay : -1; o[2] = e.adir;
  o[3] = e.carrying ? (int32_t)enc3((uint8_t)e.carrying) : 0;
  o[4] = e.step_count; o[5] = e.elapsed; o[6] = e.ep_len;
  float f = e.ep_ret; memcpy(&o[7], &f, 4);
  o[8] = e.sb(0) != NOPOS ? e.sb(0) : -1; o[9] = e.sb(0) != NOPOS ? e.sb(1) : -1; o[10] = e.sb(2);
  o[11] = e.sb(3) != NOPOS ? e.sb(3) : -1; o[12] = e.sb(3) != NOPOS ? e.sb(4) : -1;
  o[13] = e.sb(6); o[14] = e.sb(7); o[15] = (int32_t)e.ctr();
}

__global__ void big_get_metrics_kernel(int B, const Metrics* met, int32_t* ncl, int32_t* pas, int32_t* spl, int32_t* dist) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B) return;
  const Metrics m = met[i];
  if (ncl) ncl[i] = m.n_clutter_placed;
  if (pas) pas[i] = m.passable;
  if (spl) spl[i] = m.shortest_path_length;
  if (dist) dist[i] = m.distance_to_goal;
}

#define HIP_OK(x) do { hipError_t _e = (x); if (_e != hipSuccess) return dcd::fail("%s: %s", #x, hipGetErrorString(_e)); } while (0)
inline int blocks(int n) { return (n + TB - 1) / TB; }

}  // namespace

extern "C" {

int SFX(dcd_big_destroy)(void* hv);

int SFX(dcd_big_create)(const DcdMgConfig* cfg, int device, void** out) {
  if (cfg->size > MAX_SIZE) return dcd::fail("size %d exceeds this build's pitch", cfg->size);
  HIP_OK(hipSetDevice(device));
  Big* h = new Big();
  h->cfg = *cfg; h->device = device;
  Params& p = h->p;
  memset(&p, 0, sizeof p);
  static const int NEDIT[3] = {2, 3, 4};
  p.B = cfg->num_envs; p.S = cfg->size; p.V = cfg->agent_view_size; p.max_steps = cfg->max_steps;
  p.max_episode_steps = cfg->max_episode_steps; p.n_clutter = cfg->n_clutter;
  p.resample_n_clutter = cfg->resample_n_clutter; p.choose_goal_last = cfg->choose_goal_last;
  p.random_z_dim = cfg->random_z_dim; p.n_editor_actions = NEDIT[cfg->editor_actions];
  p.fixed_environment = cfg->fixed_environment; p.goal_noise = cfg->goal_noise; p.engine_seed = cfg->engine_seed;
  p.see_through_walls = cfg->see_through_walls != 0;
  init_reward_division(p);
  const size_t B = (size_t)p.B;
  const uint32_t init[2] = {0u, 0x7fffffffu};
  hipError_t e = hipMalloc(&h->rec, B * REC_BYTES);
  if (e == hipSuccess) e = hipMalloc(&h->seed, B * 4);
  if (e == hipSuccess) e = hipMalloc(&h->met, B * sizeof(Metrics));
  if (e == hipSuccess) e = hipMalloc(&h->errs, 8);
  if (e == hipSuccess) e = hipMemcpy(h->errs, init, 8, hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(big_init_kernel, dim3(blocks(p.B)), dim3(TB), 0, 0, p, h->rec, h->seed, h->met, (long long)cfg->env_index_base);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipDeviceSynchronize();
  if (e != hipSuccess) {                                          // nothing of a failed creation survives
    SFX(dcd_big_destroy)(h);
    return dcd::fail("dcd_mg_create: %zu envs of %d bytes: %s", B, REC_BYTES, hipGetErrorString(e));
  }
  *out = h;
  return 0;
}

int SFX(dcd_big_destroy)(void* hv) {
  Big* h = (Big*)hv;
  if (!h) return 0;
  (void)hipSetDevice(h->device);
  (void)hipFree(h->rec); (void)hipFree(h->seed); (void)hipFree(h->met); (void)hipFree(h->errs);
  delete h;
  return 0;
}

int SFX(dcd_big_set_seed)(void* hv, const int64_t* seeds, void* stream) {
  Big* h = (Big*)hv;
  hipLaunchKernelGGL(big_set_seed_kernel, dim3((h->p.B + 255) / 256), dim3(256), 0, (hipStream_t)stream, h->rec, h->seed, seeds, h->p.B);
  HIP_OK(hipGetLastError());
  return 0;
}

int SFX(dcd_big_poll_errors)(void* hv, uint32_t* err_bits, int32_t* first_env, void* stream) {
  Big* h = (Big*)hv;
  uint32_t host[2];
  HIP_OK(hipMemcpyAsync(host, h->errs, 8, hipMemcpyDeviceToHost, (hipStream_t)stream));
  HIP_OK(hipStreamSynchronize((hipStream_t)stream));
  if (host[0]) {
    const uint32_t init[2] = {0u, 0x7fffffffu};
    HIP_OK(hipMemcpyAsync(h->errs, init, 8, hipMemcpyHostToDevice, (hipStream_t)stream));
    HIP_OK(hipStreamSynchronize((hipStream_t)stream));
  }
  if (err_bits) *err_bits = host[0];
  if (first_env) *first_env = host[0] ? (int32_t)host[1] : -1;
  return 0;
}

int SFX(dcd_big_level)(void* hv, int op, const uint8_t* enc, const int32_t* idx, int32_t n, const DcdStudentObs* out, void* stream) {
  Big* h = (Big*)hv;
  DcdStudentObs o; memset(&o, 0, sizeof o); if (out) o = *out;
  if (n <= 0) return 0;
  hipLaunchKernelGGL(big_level_kernel, dim3(blocks(n)), dim3(TB), 0, (hipStream_t)stream, h->p, op, h->rec, h->seed, h->met, h->errs,
                     enc, idx, n, o);
  HIP_OK(hipGetLastError());
  return 0;
}

int SFX(dcd_big_step)(void* hv, const int64_t* actions, int32_t flags, const DcdStepOut* out, void* stream) {
  Big* h = (Big*)hv;
  DcdStepOut o; memset(&o, 0, sizeof o); if (out) o = *out;
  hipLaunchKernelGGL(big_step_kernel, dim3(blocks(h->p.B)), dim3(TB), 0, (hipStream_t)stream, h->p, h->rec, h->seed, h->met, h->errs,
                     actions, flags, o);
  HIP_OK(hipGetLastError());
  return 0;
}

int SFX(dcd_big_get_encodings)(void* hv, uint8_t* out, void* stream) {
  Big* h = (Big*)hv;
  hipLaunchKernelGGL(big_get_encodings_kernel, dim3(blocks(h->p.B)), dim3(TB), 0, (hipStream_t)stream, h->p, h->rec, out);
  HIP_OK(hipGetLastError());
  return 0;
}

int SFX(dcd_big_get_state)(void* hv, int32_t* out, void* stream) {
  Big* h = (Big*)hv;
  hipLaunchKernelGGL(big_get_state_kernel, dim3(blocks(h->p.B)), dim3(TB), 0, (hipStream_t)stream, h->p, h->rec, out);
  HIP_OK(hipGetLastError());
  return 0;
}

int SFX(dcd_big_get_metrics)(void* hv, int32_t* ncl, int32_t* pas, int32_t* spl, int32_t* dist, void* stream) {
  Big* h = (Big*)hv;
  hipLaunchKernelGGL(big_get_metrics_kernel, dim3((h->p.B + 255) / 256), dim3(256), 0, (hipStream_t)stream, h->p.B, h->met, ncl, pas, spl, dist);
  HIP_OK(hipGetLastError());
  return 0;
}

}  // extern "C"
