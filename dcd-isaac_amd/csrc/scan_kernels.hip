// scan_kernels.hip — RolloutStorage scans and LevelSampler episode scoring for gfx950 + C ABI.
//
// All buffers are time-major with the env axis contiguous ([T,B] / [T+1,B]; algos/storage.py:64-109),
// so "one env per lane" makes every wave load/store a contiguous 256-byte row segment.  The scans
// are sequential in t per env (the reference's evaluation order is kept so results are
// bit-identical to it, see include/dcd_hip.h); loads are independent of the recurrence and are
// issued UNROLL steps ahead so the dependent fp32 chain never waits on HBM.
//
// Build with -ffp-contract=off: `a*b+c` must stay two roundings (storage.py:249-256).
#include <hip/hip_runtime.h>
#include <math.h>
#include "../../include/dcd_hip.h"
#include "dcd_internal.h"

namespace {

constexpr int SCAN_BLOCK = 64;
// keeps a batch of loads ahead of the dependent chain: without it the scheduler sinks every load next to
// its use (one s_waitcnt vmcnt(0) per time step: 685 ns/step measured)
#define LOADS_ISSUED() __builtin_amdgcn_sched_barrier(0)
constexpr int UNROLL = 32;   // time steps whose loads are in flight ahead of the dependent chain (one wave per CU at B=4096)

// compute_gae_returns (algos/storage.py:249-256)
__global__ void __launch_bounds__(SCAN_BLOCK) gae_kernel(const float* __restrict__ rew, const float* __restrict__ val,
                                                         const float* __restrict__ msk, float* __restrict__ ret,
                                                         int T, int B, float g, float gl) {
  const int b = blockIdx.x * SCAN_BLOCK + threadIdx.x;
  if (b >= B) return;
  float gae = 0.0f;
  float v1 = val[(size_t)T * B + b];
  int t = T - 1;
  for (; t - (UNROLL - 1) >= 0; t -= UNROLL) {
    float r[UNROLL], v0[UNROLL], m1[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      r[u] = rew[(size_t)(t - u) * B + b];
      v0[u] = val[(size_t)(t - u) * B + b];
      m1[u] = msk[(size_t)(t - u + 1) * B + b];
    }
    LOADS_ISSUED();
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const float delta = (r[u] + ((g * v1) * m1[u])) - v0[u];
      gae = delta + ((gl * m1[u]) * gae);
      ret[(size_t)(t - u) * B + b] = gae + v0[u];
      v1 = v0[u];
    }
  }
  for (; t >= 0; --t) {
    const float r = rew[(size_t)t * B + b], v0 = val[(size_t)t * B + b], m1 = msk[(size_t)(t + 1) * B + b];
    const float delta = (r + ((g * v1) * m1)) - v0;
    gae = delta + ((gl * m1) * gae);
    ret[(size_t)t * B + b] = gae + v0;
    v1 = v0;
  }
}

// compute_discounted_returns (algos/storage.py:271-276)
__global__ void __launch_bounds__(SCAN_BLOCK) disc_kernel(const float* __restrict__ rew, const float* __restrict__ val,
                                                          const float* __restrict__ msk, float* __restrict__ ret,
                                                          int T, int B, float g) {
  const int b = blockIdx.x * SCAN_BLOCK + threadIdx.x;
  if (b >= B) return;
  float R = val[(size_t)T * B + b];
  ret[(size_t)T * B + b] = R;
  int t = T - 1;
  for (; t - (UNROLL - 1) >= 0; t -= UNROLL) {
    float r[UNROLL], m1[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) { r[u] = rew[(size_t)(t - u) * B + b]; m1[u] = msk[(size_t)(t - u + 1) * B + b]; }
    LOADS_ISSUED();
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) { R = ((R * g) * m1[u]) + r[u]; ret[(size_t)(t - u) * B + b] = R; }
  }
  for (; t >= 0; --t) { R = ((R * g) * msk[(size_t)(t + 1) * B + b]) + rew[(size_t)t * B + b]; ret[(size_t)t * B + b] = R; }
}

// get_batched_value_loss (algos/storage.py:290-327)
__global__ void __launch_bounds__(SCAN_BLOCK) value_loss_kernel(const float* __restrict__ ret, const float* __restrict__ val,
                                                                float* __restrict__ out, int T, int B, int mode, int power,
                                                                int clipped) {
  const int b = blockIdx.x * SCAN_BLOCK + threadIdx.x;
  if (b >= B) return;
  double acc = 0.0;
  for (int t0 = 0; t0 < T; t0 += UNROLL) {
    float r[UNROLL], v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) { const int tc = min(t0 + u, T - 1); r[u] = ret[(size_t)tc * B + b]; v[u] = val[(size_t)tc * B + b]; }   // clamped: loads stay unconditional
    LOADS_ISSUED();
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) if (t0 + u < T) {
      float td = r[u] - v[u];
      if (mode == 0) td = fabsf(td); else if (mode == 2) td = td > 0.f ? td : 0.f;
      if (power > 1) { float pw = td; for (int k = 1; k < power; ++k) pw = pw * td; td = pw; }
      acc += (double)td;
    }
  }
  float m = (float)(acc / (double)T);
  if (clipped) m = m < -1.f ? -1.f : (m > 1.f ? 1.f : m);
  out[b] = m;
}

// ---- episode segmentation (level_replay/level_sampler.py:497-578) ---------------------------------
__global__ void __launch_bounds__(SCAN_BLOCK) seg_count_kernel(const float* __restrict__ msk, int T, int B, int* __restrict__ counts) {
  const int b = blockIdx.x * SCAN_BLOCK + threadIdx.x;
  if (b >= B) return;
  int n = 0, start_t = 0;
  for (int t0 = 1; t0 <= T; t0 += UNROLL) {        // a done at t == 0 is skipped (:504-505)
    float m[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) m[u] = msk[(size_t)min(t0 + u, T) * B + b];       // clamped: loads stay unconditional
    LOADS_ISSUED();
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) if (t0 + u <= T && !(m[u] > 0.f) && start_t < T) { ++n; start_t = t0 + u; }
  }
  if (start_t < T) ++n;                          // trailing partial segment (:551)
  counts[b] = n;
}

// exclusive scan of counts -> offsets (single workgroup, carry over chunks); total -> n_recs[0]
__global__ void __launch_bounds__(1024) seg_scan_kernel(const int* __restrict__ counts, int* __restrict__ offsets, int B,
                                                        int* __restrict__ n_recs) {
  __shared__ int wsum[16];
  __shared__ int carry;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  if (tid == 0) carry = 0;
  __syncthreads();
  for (int base = 0; base < B; base += 1024) {
    const int i = base + tid;
    const int v = i < B ? counts[i] : 0;
    int x = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const int y = __shfl_up(x, off); if (lane >= off) x += y; }
    if (lane == 63) wsum[w] = x;
    __syncthreads();
    int wbase = 0;
    for (int k = 0; k < w; ++k) wbase += wsum[k];
    const int c = carry;
    if (i < B) offsets[i] = c + wbase + x - v;
    __syncthreads();
    if (tid == 1023) carry = c + wbase + x;
    __syncthreads();
  }
  if (tid == 0) n_recs[0] = carry;
}

__device__ __forceinline__ void log_softmax(const float* __restrict__ x, int A, float* out) {
  float mx = x[0];
  for (int a = 1; a < A; ++a) mx = x[a] > mx ? x[a] : mx;
  float s = 0.f;
  for (int a = 0; a < A; ++a) s += expf(x[a] - mx);
  const float ls = logf(s);
  for (int a = 0; a < A; ++a) out[a] = (x[a] - mx) - ls;
}

struct ScoreArgs {
  const float *ret, *val, *rew, *msk, *cliff, *logits;
  const int* seeds;
  int T, B, A, strategy; float g;
  DcdEpisodeRec* recs; int max_recs; const int* offsets;
};

// Single pass over t per env (one env per lane): every step's loads are issued CH steps ahead, the
// per-segment accumulators (fp64 sum of fp32 terms in step order, fp32 max) are flushed into a record
// whenever masks[t+1] == 0.  Same accumulation order as oracle/scan_oracle.c -> bit-identical records.
// score functions: level_sampler.py:275-349, 426-439.
struct SegAcc {
  double acc, rs; float mx, mn; float r0, v0; int start;
  __device__ __forceinline__ void reset(int s) { acc = 0.0; rs = 0.0; mx = -INFINITY; mn = INFINITY; r0 = 0.f; v0 = 0.f; start = s; }
};

template <int ST>
__device__ __forceinline__ void flush_segment(const ScoreArgs& a, int b, SegAcc& sg, int end, int flags, int& k) {
  const int L = end - sg.start; constexpr int st = ST;
  if (k < a.max_recs) {
    DcdEpisodeRec r;
    r.actor = b; r.seed = a.seeds[(size_t)sg.start * a.B + b]; r.start_t = sg.start; r.num_steps = L; r.flags = flags;
    r.reward_sum = (float)sg.rs;
    if (st == DCD_SCORE_UNIFORM) { r.mean_score = 1.f; r.max_score = 1.f; }
    else if (st == DCD_SCORE_RETURN_SUM) { r.mean_score = 0.f; r.max_score = 0.f; }
    else if (st == DCD_SCORE_ONE_STEP_TD) {
      if (L > 1) { r.mean_score = (float)(sg.acc / (double)(L - 1)); r.max_score = sg.mx; }
      else { r.mean_score = sg.r0 - sg.v0; r.max_score = r.mean_score; }                  // signed (:433)
    } else if (st == DCD_SCORE_MIN_MARGIN) {                                              // 1 - mean, 1 - min (:302-303)
      r.mean_score = (float)(1.0 - (double)(float)(sg.acc / (double)L)); r.max_score = (float)(1.0 - (double)sg.mn);
    } else { r.mean_score = (float)(sg.acc / (double)L); r.max_score = sg.mx; }
    a.recs[k] = r;
  }
  ++k;
}

template <int ST>                                           // strategy is compile-time: straight-line load batches
__global__ void __launch_bounds__(SCAN_BLOCK) seg_score_kernel(ScoreArgs a) {
  constexpr int CH = 16;
  const int b = blockIdx.x * SCAN_BLOCK + threadIdx.x;
  if (b >= a.B) return;
  const int T = a.T, B = a.B, A = a.A;
  constexpr int st = ST;
  constexpr bool use_ret = st <= DCD_SCORE_SIGNED_VALUE_LOSS, use_logits = st == DCD_SCORE_POLICY_ENTROPY || st == DCD_SCORE_LEAST_CONFIDENCE || st == DCD_SCORE_MIN_MARGIN;
  constexpr bool use_val = use_ret || st == DCD_SCORE_ONE_STEP_TD;
  const float max_entropy = (float)log((double)(A > 0 ? A : 1));
  int k = a.offsets[b];
  SegAcc sg; sg.reset(0);
  float vnext = use_val ? a.val[b] : 0.f;                  // v[t] carried across chunks (TD needs v[t+1])
  for (int t0 = 0; t0 < T; t0 += CH) {
    float m1[CH], cl[CH], rw[CH], rt[CH], v1[CH];
#pragma unroll
    for (int u = 0; u < CH; ++u) {
      const int t = min(t0 + u, T - 1);                    // clamped: loads stay unconditional, tail masked below
      m1[u] = a.msk[(size_t)(t + 1) * B + b];
      cl[u] = a.cliff[(size_t)(t + 1) * B + b];
      rw[u] = a.rew[(size_t)t * B + b];
      rt[u] = use_ret ? a.ret[(size_t)t * B + b] : 0.f;
      v1[u] = use_val ? a.val[(size_t)(t + 1) * B + b] : 0.f;
    }
    LOADS_ISSUED();
#pragma unroll
    for (int u = 0; u < CH; ++u) {
      const int t = t0 + u;
      if (t < T && sg.start < T) {
        const float v0 = vnext;
        const bool seg_ends = !(m1[u] > 0.f);                // done at t+1 closes [start, t+1)
        const bool last_in_seg = seg_ends || t == T - 1;
        sg.rs += (double)rw[u];
        if (t == sg.start) { sg.r0 = rw[u]; sg.v0 = v0; }
        if (use_ret) {
          float adv = rt[u] - v0;
          if (st == DCD_SCORE_VALUE_L1) adv = fabsf(adv); else if (st == DCD_SCORE_POSITIVE_VALUE_LOSS) adv = adv > 0.f ? adv : 0.f;
          sg.acc += adv; sg.mx = adv > sg.mx ? adv : sg.mx;
        } else if (st == DCD_SCORE_ONE_STEP_TD) {
          if (!last_in_seg) { const float td = fabsf((rw[u] + (a.g * v1[u])) - v0); sg.acc += td; sg.mx = td > sg.mx ? td : sg.mx; }
        } else if (use_logits) {
          float ls[16];                                     // A <= 16 on this path (checked by the launcher)
          log_softmax(a.logits + ((size_t)t * B + b) * A, A, ls);
          float v;
          if (st == DCD_SCORE_POLICY_ENTROPY) { float h = 0.f; for (int q = 0; q < A; ++q) h += -expf(ls[q]) * ls[q]; v = h / max_entropy; }
          else if (st == DCD_SCORE_LEAST_CONFIDENCE) { float m = ls[0]; for (int q = 1; q < A; ++q) m = ls[q] > m ? ls[q] : m; v = 1.f - expf(m); }
          else { float q1 = -INFINITY, q2 = -INFINITY; for (int q = 0; q < A; ++q) { if (ls[q] > q1) { q2 = q1; q1 = ls[q]; } else if (ls[q] > q2) q2 = ls[q]; } v = expf(q1) - expf(q2); }
          sg.acc += v; sg.mx = v > sg.mx ? v : sg.mx; sg.mn = v < sg.mn ? v : sg.mn;
        }
        if (seg_ends) { flush_segment<ST>(a, b, sg, t + 1, 1 | ((cl[u] > 0.f) ? 0 : 2), k); sg.reset(t + 1); }
      }
      if (t < T) vnext = v1[u];
    }
  }
  if (sg.start < T) flush_segment<ST>(a, b, sg, T, 0, k);        // trailing partial segment (:551-578)
}

// grounded_* strategies (level_sampler.py:351-424), one lane per record
__global__ void __launch_bounds__(SCAN_BLOCK) grounded_kernel(const DcdEpisodeRec* __restrict__ recs, int n,
                                                              const float* __restrict__ grounded, const float* __restrict__ val,
                                                              int B, int positive_only, int dense, float* __restrict__ mean_out,
                                                              float* __restrict__ max_out) {
  const int k = blockIdx.x * SCAN_BLOCK + threadIdx.x;
  if (k >= n) return;
  const DcdEpisodeRec r = recs[k];
  const int L = dense ? 1 : r.num_steps;
  const float g = grounded[k];
  double acc = 0.0; float mx = -INFINITY;
  for (int t = r.start_t; t < r.start_t + L; ++t) {
    float adv = g - val[(size_t)t * B + r.actor];
    if (positive_only) adv = adv > 0.f ? adv : 0.f;
    acc += adv; mx = adv > mx ? adv : mx;
  }
  mean_out[k] = (float)(acc / (double)L); max_out[k] = mx;
}

// _get_rollout_return_stats (envs/runners/adversarial_runner.py:257-271)
__global__ void __launch_bounds__(SCAN_BLOCK) return_stats_kernel(const float* __restrict__ ep_ret, const uint8_t* __restrict__ info,
                                                                  int T, int B, float* __restrict__ mean_r, float* __restrict__ max_r,
                                                                  int* __restrict__ n_eps) {
  const int b = blockIdx.x * SCAN_BLOCK + threadIdx.x;
  if (b >= B) return;
  double acc = 0.0; float mx = -INFINITY; int n = 0;
  for (int t0 = 0; t0 < T; t0 += UNROLL) {
    float r[UNROLL]; uint8_t f[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) { const int tc = min(t0 + u, T - 1); f[u] = info[(size_t)tc * B + b]; r[u] = ep_ret[(size_t)tc * B + b]; }
    LOADS_ISSUED();
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) if (t0 + u < T && (f[u] & DCD_INFO_EPISODE)) { acc += r[u]; mx = r[u] > mx ? r[u] : mx; ++n; }
  }
  mean_r[b] = n ? (float)(acc / (double)n) : 0.f;
  max_r[b] = n ? mx : 0.f;
  if (n_eps) n_eps[b] = n;
}

// _compute_env_return (envs/runners/adversarial_runner.py:637-664)
__global__ void env_return_kernel(const float* __restrict__ am, const float* __restrict__ ax, const float* __restrict__ vm,
                                  const float* __restrict__ vx, float* __restrict__ out, int B, int mode) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  float v;
  if (mode == 0) { v = vx[b] - am[b]; v = v > 0.f ? v : 0.f; }
  else if (mode == 1) {
    float er, em;
    if (vx[b] > ax[b]) { er = vx[b]; em = am[b]; } else { er = ax[b]; em = vm[b]; }
    v = er - em; v = v > 0.f ? v : 0.f;
  } else v = -ax[b];
  out[b] = v;
}

}  // namespace

#define HIP_OK(x) do { hipError_t _e = (x); if (_e != hipSuccess) return dcd::fail("%s: %s", #x, hipGetErrorString(_e)); } while (0)
static inline int nblk(int n) { return (n + SCAN_BLOCK - 1) / SCAN_BLOCK; }

extern "C" {

int dcd_gae_returns(const float* rewards, const float* value_preds, const float* masks, float* returns, int32_t T, int32_t B,
                    double gamma, double gae_lambda, void* stream) {
  if (!rewards || !value_preds || !masks || !returns || T <= 0 || B <= 0) return dcd::fail("dcd_gae_returns: bad arguments");
  hipLaunchKernelGGL(gae_kernel, dim3(nblk(B)), dim3(SCAN_BLOCK), 0, (hipStream_t)stream, rewards, value_preds, masks, returns, T, B,
                     (float)gamma, (float)(gamma * gae_lambda));
  HIP_OK(hipGetLastError());
  return 0;
}
int dcd_discounted_returns(const float* rewards, const float* value_preds, const float* masks, float* returns, int32_t T, int32_t B,
                           double gamma, void* stream) {
  if (!rewards || !value_preds || !masks || !returns || T <= 0 || B <= 0) return dcd::fail("dcd_discounted_returns: bad arguments");
  hipLaunchKernelGGL(disc_kernel, dim3(nblk(B)), dim3(SCAN_BLOCK), 0, (hipStream_t)stream, rewards, value_preds, masks, returns, T, B, (float)gamma);
  HIP_OK(hipGetLastError());
  return 0;
}
int dcd_batched_value_loss(const float* returns, const float* value_preds, float* out, int32_t T, int32_t B, int32_t mode,
                           int32_t power, int32_t clipped, void* stream) {
  if (!returns || !value_preds || !out || T <= 0 || B <= 0 || mode < 0 || mode > 2) return dcd::fail("dcd_batched_value_loss: bad arguments");
  hipLaunchKernelGGL(value_loss_kernel, dim3(nblk(B)), dim3(SCAN_BLOCK), 0, (hipStream_t)stream, returns, value_preds, out, T, B, mode, power, clipped);
  HIP_OK(hipGetLastError());
  return 0;
}

int dcd_episode_scores(const float* returns, const float* value_preds, const float* rewards, const float* masks,
                       const float* cliffhanger_masks, const int32_t* level_seeds, const float* action_log_dist, int32_t T, int32_t B,
                       int32_t A, int32_t strategy, double gamma, DcdEpisodeRec* recs, int32_t max_recs, int32_t* n_recs,
                       int32_t* workspace, void* stream) {
  if (!value_preds || !rewards || !masks || !cliffhanger_masks || !level_seeds || !recs || !n_recs || !workspace || T <= 0 || B <= 0)
    return dcd::fail("dcd_episode_scores: bad arguments");
  if (strategy < 0 || strategy > DCD_SCORE_RETURN_SUM) return dcd::fail("dcd_episode_scores: unknown strategy %d", strategy);
  const bool needs_logits = strategy == DCD_SCORE_POLICY_ENTROPY || strategy == DCD_SCORE_LEAST_CONFIDENCE || strategy == DCD_SCORE_MIN_MARGIN;
  if (needs_logits && (!action_log_dist || A <= 0 || A > 16)) return dcd::fail("dcd_episode_scores: strategy %d needs action_log_dist with 1..16 actions", strategy);
  const bool needs_returns = strategy <= DCD_SCORE_SIGNED_VALUE_LOSS;
  if (needs_returns && !returns) return dcd::fail("dcd_episode_scores: strategy %d needs returns", strategy);
  int* counts = workspace; int* offsets = workspace + B;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(seg_count_kernel, dim3(nblk(B)), dim3(SCAN_BLOCK), 0, s, masks, T, B, counts);
  hipLaunchKernelGGL(seg_scan_kernel, dim3(1), dim3(1024), 0, s, counts, offsets, B, n_recs);
  ScoreArgs a;
  a.ret = returns; a.val = value_preds; a.rew = rewards; a.msk = masks; a.cliff = cliffhanger_masks; a.logits = action_log_dist;
  a.seeds = level_seeds; a.T = T; a.B = B; a.A = A; a.strategy = strategy; a.g = (float)gamma; a.recs = recs; a.max_recs = max_recs;
  a.offsets = offsets;
  switch (strategy) {
#define DCD_SCORE_CASE(ST) case ST: hipLaunchKernelGGL(seg_score_kernel<ST>, dim3(nblk(B)), dim3(SCAN_BLOCK), 0, s, a); break;
    DCD_SCORE_CASE(0) DCD_SCORE_CASE(1) DCD_SCORE_CASE(2) DCD_SCORE_CASE(3) DCD_SCORE_CASE(4)
    DCD_SCORE_CASE(5) DCD_SCORE_CASE(6) DCD_SCORE_CASE(7) DCD_SCORE_CASE(8)
#undef DCD_SCORE_CASE
  }
  HIP_OK(hipGetLastError());
  return 0;
}

int dcd_grounded_scores(const DcdEpisodeRec* recs, int32_t n, const float* grounded, const float* value_preds, int32_t B,
                        int32_t positive_only, int32_t dense, float* mean_out, float* max_out, void* stream) {
  if (n == 0) return 0;
  if (!recs || !grounded || !value_preds || !mean_out || !max_out || n < 0) return dcd::fail("dcd_grounded_scores: bad arguments");
  hipLaunchKernelGGL(grounded_kernel, dim3(nblk(n)), dim3(SCAN_BLOCK), 0, (hipStream_t)stream, recs, n, grounded, value_preds, B, positive_only, dense, mean_out, max_out);
  HIP_OK(hipGetLastError());
  return 0;
}

int dcd_rollout_return_stats(const float* ep_ret, const uint8_t* info, int32_t T, int32_t B, float* mean_return, float* max_return,
                             int32_t* n_episodes, void* stream) {
  if (!ep_ret || !info || !mean_return || !max_return || T <= 0 || B <= 0) return dcd::fail("dcd_rollout_return_stats: bad arguments");
  hipLaunchKernelGGL(return_stats_kernel, dim3(nblk(B)), dim3(SCAN_BLOCK), 0, (hipStream_t)stream, ep_ret, info, T, B, mean_return, max_return, n_episodes);
  HIP_OK(hipGetLastError());
  return 0;
}

int dcd_env_return(const float* agent_mean, const float* agent_max, const float* adv_mean, const float* adv_max, float* out, int32_t B,
                   int32_t mode, void* stream) {
  if (!agent_max || !out || B <= 0 || mode < 0 || mode > 2) return dcd::fail("dcd_env_return: bad arguments");
  if (mode != 2 && (!agent_mean || !adv_max)) return dcd::fail("dcd_env_return: paired modes need agent_mean and adv_max");
  if (mode == 1 && !adv_mean) return dcd::fail("dcd_env_return: flexible_paired needs adv_mean");
  hipLaunchKernelGGL(env_return_kernel, dim3((B + 255) / 256), dim3(256), 0, (hipStream_t)stream, agent_mean, agent_max, adv_mean, adv_max, out, B, mode);
  HIP_OK(hipGetLastError());
  return 0;
}

}  // extern "C"
