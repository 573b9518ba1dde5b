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
#include <string.h>
#include "../../include/dcd_hip.h"
#include <hip/hip_ext.h>
#include "dcd_internal.h"

namespace {

constexpr int SCAN_BLOCK = 64;
// keeps a batch of loads ahead of the dependent chain: without it the scheduler sinks every load next to
// its use (one s_waitcnt vmcnt(0) per time step: 685 ns/step measured)
#define LOADS_ISSUED() __builtin_amdgcn_sched_barrier(0)
constexpr int UNROLL = 32;   // time steps whose loads are in flight ahead of the dependent chain (one wave per CU at B=4096)

// Batched forms (the *_batch entry points): blockIdx.y selects one of up to DCD_MAX_SCAN_SETS independent rollouts of the
// same shape — PAIRED's protagonist and antagonist — so their scans are ONE launch each: a 4096-env scan is 64 waves on 256
// CUs and latency-bound, two of them side by side cost the time of one.
struct GaeSets { const float* rew[DCD_MAX_SCAN_SETS]; const float* val[DCD_MAX_SCAN_SETS]; const float* msk[DCD_MAX_SCAN_SETS]; float* ret[DCD_MAX_SCAN_SETS]; };

// compute_gae_returns (algos/storage.py:249-256)
__device__ __forceinline__ void gae_body(const GaeSets& sets, int set, int b, int T, int B, float g, float gl) {
  const float* __restrict__ rew = sets.rew[set]; const float* __restrict__ val = sets.val[set];
  const float* __restrict__ msk = sets.msk[set]; float* __restrict__ ret = sets.ret[set];
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
__global__ void __launch_bounds__(SCAN_BLOCK) gae_kernel(GaeSets sets, int T, int B, float g, float gl) {
  gae_body(sets, blockIdx.y, blockIdx.x * SCAN_BLOCK + threadIdx.x, T, B, g, gl);
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
  int T, B, A, strategy; float g, max_entropy;
  DcdEpisodeRec* recs; int max_recs; const int* offsets;
};

// Single pass over t per env (one env per lane): every step's loads are issued CH steps ahead, the
// per-segment accumulators (fp64 sum of fp32 terms in step order, fp32 max) are flushed into a record
// whenever masks[t+1] == 0.  Same accumulation order as oracle/scan_oracle.c -> bit-identical records.
// score functions: level_sampler.py:275-349, 426-439.
struct SegAcc {
  double acc, rs; float mx, mn; float r0, v0; int start; int seed;
  __device__ __forceinline__ void reset(int s) { acc = 0.0; rs = 0.0; mx = -INFINITY; mn = INFINITY; r0 = 0.f; v0 = 0.f; start = s; }
};

// policy_entropy on the episode lengths where the reference's expression is defined (level_sampler.py:283):
// `-exp(l) * l.sum(-1) / max_entropy` broadcasts the [L,A] probabilities against the [L] row sums, which numpy/torch only
// accept for L == 1 (every entry times the one row sum) and L == A (entry (i,j) times the sum of row j); mean and max
// then run over all L*A entries.  Those two lengths reproduce the reference; every other length (where the reference
// raises) scores the per-step entropy it evidently meant, accumulated by the caller.  Rare: re-reads the L rows.
__device__ __noinline__ void entropy_reference_lengths_strided(const float* __restrict__ logits, size_t row_stride, int A, int L,
                                                               float max_entropy, float* mean_out, float* max_out) {
  float S[16], ls[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    S[j] = 0.f;
    if (j < L) { log_softmax(logits + (size_t)j * row_stride, A, ls); float t = 0.f; for (int q = 0; q < A; ++q) t += ls[q]; S[j] = t; }
  }
  double acc = 0.0; float mx = -INFINITY;
  for (int i = 0; i < L; ++i) {
    log_softmax(logits + (size_t)i * row_stride, A, ls);
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      if (j < A) {
        const float v = ((-expf(ls[j])) * (L == 1 ? S[0] : S[j])) / max_entropy;
        acc += (double)v; mx = v > mx ? v : mx;
      }
    }
  }
  *mean_out = (float)(acc / (double)(L * A)); *max_out = mx;
}

// The trailing partial segment (flags == 0, level_sampler.py:551-578) slices `returns[start_t:]` / `value_preds[start_t:]`, which
// have T + 1 rows where logits and rewards have T: the advantage strategies average over the segment's steps AND row T, while
// num_steps stays T - start_t.  tail_row adds that element, tail_rows is what it adds to the divisor.
template <int ST>
__device__ __forceinline__ int tail_rows(int flags) { return (ST <= DCD_SCORE_SIGNED_VALUE_LOSS && flags == 0) ? 1 : 0; }
template <int ST>
__device__ __forceinline__ void tail_row(const ScoreArgs& a, int b, float v_T, SegAcc& sg) {
  if (ST <= DCD_SCORE_SIGNED_VALUE_LOSS) {
    float adv = a.ret[(size_t)a.T * a.B + b] - v_T;
    if (ST == DCD_SCORE_VALUE_L1) adv = fabsf(adv); else if (ST == DCD_SCORE_POSITIVE_VALUE_LOSS) adv = adv > 0.f ? adv : 0.f;
    sg.acc += adv; sg.mx = adv > sg.mx ? adv : sg.mx;
  }
}

template <int ST, bool HAVE_SEED = false>
__device__ __forceinline__ void flush_segment(const ScoreArgs& a, int b, SegAcc& sg, int end, int flags, int& k) {
  const int L = end - sg.start; constexpr int st = ST;
  if (k < a.max_recs) {
    DcdEpisodeRec r;
    r.actor = b; r.seed = HAVE_SEED ? sg.seed : a.seeds[(size_t)sg.start * a.B + b]; r.start_t = sg.start; r.num_steps = L; r.flags = flags;
    r.reward_sum = (float)sg.rs;
    if (st == DCD_SCORE_UNIFORM) { r.mean_score = 1.f; r.max_score = 1.f; }
    else if (st == DCD_SCORE_RETURN_SUM) { r.mean_score = 0.f; r.max_score = 0.f; }
    else if (st == DCD_SCORE_ONE_STEP_TD) {
      if (L > 1) { r.mean_score = (float)(sg.acc / (double)(L - 1)); r.max_score = sg.mx; }
      else { r.mean_score = sg.r0 - sg.v0; r.max_score = r.mean_score; }                  // signed (:433)
    } else if (st == DCD_SCORE_MIN_MARGIN) {                                              // 1 - mean, 1 - min (:302-303)
      r.mean_score = (float)(1.0 - (double)(float)(sg.acc / (double)L)); r.max_score = (float)(1.0 - (double)sg.mn);
    } else if (st == DCD_SCORE_POLICY_ENTROPY && (L == 1 || L == a.A)) {
      // rows of one env are B*A floats apart: gather the episode's rows through a strided view
      entropy_reference_lengths_strided(a.logits + ((size_t)sg.start * a.B + b) * a.A, (size_t)a.B * a.A, a.A, L,
                                        a.max_entropy, &r.mean_score, &r.max_score);
    } else { r.mean_score = (float)(sg.acc / (double)(L + tail_rows<ST>(flags))); r.max_score = sg.mx; }
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
  const float max_entropy = a.max_entropy;
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
  if (sg.start < T) { tail_row<ST>(a, b, vnext, sg); flush_segment<ST>(a, b, sg, T, 0, k); }      // trailing partial segment (:551-578)
}

// ---- the same segmentation and scoring with the rollout staged in LDS ------------------------------------------
// One env per lane bounds a scan by its loads: a wave holds at most 63 loads in flight, so 256 steps of 5 arrays are
// 16+ dependent HBM round trips (37 us at B = 4096).  Here a 512-thread workgroup owns SE = 32 envs: every thread
// issues its share of a [<= 256 steps] x [32 envs] tile of all six arrays at once (24 x 16-byte loads per thread, ONE
// round trip), the tile is committed to LDS, and 32 lanes run the unchanged sequential pass from LDS.  Segment counts
// come from a time-parallel count (a segment starts at t = 0 and after every masks[t] == 0, 1 <= t < T), record
// offsets from per-32-env block sums that every workgroup adds up itself — no single-workgroup scan launch.
constexpr int SE = 32;            // envs per workgroup (one 128-byte line per row of a tile)
constexpr int SNT = 512;          // threads per workgroup
constexpr int STILE = 256;        // time steps per LDS tile
constexpr int SWEEPS = STILE / (SNT / (SE / 4));   // rows a thread loads per array and tile (8)

// counts[b] = 1 + #{t in [1, T) : masks[t][b] == 0}; block_sums[b / 32] = sum of counts over the 32-env block.
// Workgroup = 64 envs x 4 time quarters (one wave each), every wave's loads in flight together.
struct CountSets { const float* msk[DCD_MAX_SCAN_SETS]; int* counts[DCD_MAX_SCAN_SETS]; int* block_sums[DCD_MAX_SCAN_SETS]; };
__device__ __forceinline__ void seg_count_par_body(const CountSets& sets, int set, int bx, int T, int B) {
  const float* __restrict__ msk = sets.msk[set]; int* __restrict__ counts = sets.counts[set];
  int* __restrict__ block_sums = sets.block_sums[set];
  __shared__ int part[4][64];
  const int lane = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int b = bx * 64 + lane;
  const int per = (T - 1 + 3) / 4, t_lo = 1 + q * per, t_hi = min(t_lo + per, T);
  int n = 0;
  if (b < B) {
    for (int t0 = t_lo; t0 < t_hi; t0 += UNROLL) {
      float m[UNROLL];
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) m[u] = msk[(size_t)min(t0 + u, T) * B + b];     // clamped: loads stay unconditional
      LOADS_ISSUED();
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) if (t0 + u < t_hi && !(m[u] > 0.f)) ++n;
    }
  }
  part[q][lane] = n;
  __syncthreads();
  if (q == 0) {
    n = (b < B) ? part[0][lane] + part[1][lane] + part[2][lane] + part[3][lane] + 1 : 0;
    if (b < B) counts[b] = n;
    int sum = n;
#pragma unroll
    for (int off = 16; off >= 1; off >>= 1) sum += __shfl_xor(sum, off);                 // within each half-wave of 32
    if ((lane & 31) == 0 && bx * 64 + lane < B) block_sums[(bx * 64 + lane) >> 5] = sum;
  }
}
__global__ void __launch_bounds__(256) seg_count_par_kernel(CountSets sets, int T, int B) {
  seg_count_par_body(sets, blockIdx.y, blockIdx.x, T, B);
}

// The staged kernel's sequential lanes only PARK a finished segment in its record slot (sums as accumulated, no
// division): a flush is divergent — the whole wave walks through it whenever one lane closes a segment — and the
// fp64 division in it cost more than the scan itself (38 us of 256 steps at B = 4096 went there).  All threads of the
// workgroup then turn the parked slots of their block into records (same expressions as flush_segment, bit for bit).
struct RawRec { int32_t actor, seed, start_t, len_flags; double acc; float reward_sum, mx; };   // 32 bytes, as DcdEpisodeRec
static_assert(sizeof(RawRec) == sizeof(DcdEpisodeRec), "raw slot must overlay the record");
constexpr int RAW_LEN_BITS = 24;                         // len_flags = L | flags << 24 | single-step-TD << 30

template <int ST>
__device__ __forceinline__ void park_segment(const ScoreArgs& a, int b, const SegAcc& sg, int end, int flags, int& k) {
  const int L = end - sg.start;
  if (k < a.max_recs) {
    RawRec r;
    r.actor = b; r.seed = sg.seed; r.start_t = sg.start; r.len_flags = L | (flags << RAW_LEN_BITS);
    r.acc = sg.acc; r.mx = sg.mx; r.reward_sum = (float)sg.rs;
    if (ST == DCD_SCORE_ONE_STEP_TD && L <= 1) { r.acc = (double)(sg.r0 - sg.v0); r.len_flags |= 1 << 30; }   // signed (:433)
    *reinterpret_cast<RawRec*>(a.recs + k) = r;
  }
  ++k;
}
template <int ST>
__device__ __forceinline__ void finish_record(DcdEpisodeRec* slot) {
  const RawRec w = *reinterpret_cast<const RawRec*>(slot);
  const int L = w.len_flags & ((1 << RAW_LEN_BITS) - 1);
  DcdEpisodeRec r;
  r.actor = w.actor; r.seed = w.seed; r.start_t = w.start_t; r.num_steps = L; r.flags = (w.len_flags >> RAW_LEN_BITS) & 3;
  r.reward_sum = w.reward_sum;
  if (ST == DCD_SCORE_UNIFORM) { r.mean_score = 1.f; r.max_score = 1.f; }
  else if (ST == DCD_SCORE_RETURN_SUM) { r.mean_score = 0.f; r.max_score = 0.f; }
  else if (ST == DCD_SCORE_ONE_STEP_TD) {
    if (w.len_flags & (1 << 30)) { r.mean_score = (float)w.acc; r.max_score = r.mean_score; }
    else { r.mean_score = (float)(w.acc / (double)(L - 1)); r.max_score = w.mx; }
  } else { r.mean_score = (float)(w.acc / (double)(L + tail_rows<ST>(r.flags))); r.max_score = w.mx; }
  *slot = r;
}

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));     // a plain vector: arrays of it stay in registers

struct ScoreSets { ScoreArgs a[DCD_MAX_SCAN_SETS]; const int* counts[DCD_MAX_SCAN_SETS]; const int* block_sums[DCD_MAX_SCAN_SETS]; int* n_recs[DCD_MAX_SCAN_SETS]; };
template <int ST>
__global__ void __launch_bounds__(SNT) seg_score_lds_kernel(ScoreSets sets) {
  const ScoreArgs& a = sets.a[blockIdx.y];
  const int* __restrict__ counts = sets.counts[blockIdx.y]; const int* __restrict__ block_sums = sets.block_sums[blockIdx.y];
  int* __restrict__ n_recs = sets.n_recs[blockIdx.y];
  __shared__ __attribute__((aligned(16))) float s_rew[STILE][SE];
  __shared__ __attribute__((aligned(16))) float s_ret[STILE][SE];
  __shared__ __attribute__((aligned(16))) float s_val[STILE][SE];     // value_preds[t + 1]
  __shared__ __attribute__((aligned(16))) int s_seed[STILE][SE];
  __shared__ __attribute__((aligned(16))) uint8_t s_flag[STILE][SE];  // bit 0: masks[t + 1] > 0, bit 1: cliffhanger[t + 1] > 0
  __shared__ int s_part[SNT / 64];
  __shared__ int s_range[2];
  constexpr int st = ST;
  constexpr bool use_ret = st <= DCD_SCORE_SIGNED_VALUE_LOSS, use_val = use_ret || st == DCD_SCORE_ONE_STEP_TD;
  const int tid = threadIdx.x, lane = tid & 63, T = a.T, B = a.B;
  const int b0 = blockIdx.x * SE;                          // B % SE == 0 on this path (checked by the launcher)
  // ---- record offset of this block = sum of the block sums before it; of a lane = + exclusive scan of its block's counts
  int acc = 0;
  for (int j = tid; j < (int)blockIdx.x; j += SNT) acc += block_sums[j];
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off);
  if (lane == 0) s_part[tid >> 6] = acc;
  int cnt = 0, k = 0;
  float vnext = 0.f;
  if (tid < SE) { cnt = counts[b0 + tid]; if (use_val) vnext = a.val[b0 + tid]; }
  __syncthreads();
  if (tid < 64) {
    int x = (tid < SE) ? cnt : 0;
#pragma unroll
    for (int off = 1; off < SE; off <<= 1) { const int y = __shfl_up(x, off); if (lane >= off) x += y; }
    int base = 0;
#pragma unroll
    for (int w = 0; w < SNT / 64; ++w) base += s_part[w];
    k = base + x - cnt;
    if (tid == SE - 1) { s_range[0] = base; s_range[1] = base + x; if (blockIdx.x == gridDim.x - 1) n_recs[0] = base + x; }
  }
  SegAcc sg; sg.reset(0);
  const int c4 = tid & 7, rr = tid >> 3;                   // this thread's 4-env column group and first row of a sweep
  for (int t0 = 0; t0 < T; t0 += STILE) {
    const int rows = min(STILE, T - t0);
    {
      u32x4 q_rew[SWEEPS], q_ret[SWEEPS], q_val[SWEEPS], q_seed[SWEEPS], q_msk[SWEEPS], q_cl[SWEEPS];
#pragma unroll
      for (int w = 0; w < SWEEPS; ++w) {
        const int r = min(w * (SNT / 8) + rr, rows - 1);   // clamped: loads stay unconditional
        const size_t o0 = (size_t)(t0 + r) * B + b0 + c4 * 4, o1 = o0 + B;
        q_rew[w] = *reinterpret_cast<const u32x4*>(a.rew + o0);
        q_seed[w] = *reinterpret_cast<const u32x4*>(a.seeds + o0);
        q_msk[w] = *reinterpret_cast<const u32x4*>(a.msk + o1);
        q_cl[w] = *reinterpret_cast<const u32x4*>(a.cliff + o1);
        if (use_ret) q_ret[w] = *reinterpret_cast<const u32x4*>(a.ret + o0);
        if (use_val) q_val[w] = *reinterpret_cast<const u32x4*>(a.val + o1);
      }
      LOADS_ISSUED();
      if (t0 > 0) __syncthreads();                         // the previous tile has been consumed
#pragma unroll
      for (int w = 0; w < SWEEPS; ++w) {
        const int r = w * (SNT / 8) + rr;                  // rows beyond the tile hold copies of its last row, never read
        {
          *reinterpret_cast<u32x4*>(&s_rew[r][c4 * 4]) = q_rew[w];
          *reinterpret_cast<u32x4*>(&s_seed[r][c4 * 4]) = q_seed[w];
          if (use_ret) *reinterpret_cast<u32x4*>(&s_ret[r][c4 * 4]) = q_ret[w];
          if (use_val) *reinterpret_cast<u32x4*>(&s_val[r][c4 * 4]) = q_val[w];
          const uint32_t mm[4] = {q_msk[w].x, q_msk[w].y, q_msk[w].z, q_msk[w].w}, cc[4] = {q_cl[w].x, q_cl[w].y, q_cl[w].z, q_cl[w].w};
          uint32_t f = 0;
#pragma unroll
          for (int e = 0; e < 4; ++e)
            f |= (uint32_t)((__uint_as_float(mm[e]) > 0.f ? 1u : 0u) | (__uint_as_float(cc[e]) > 0.f ? 2u : 0u)) << (8 * e);
          *reinterpret_cast<uint32_t*>(&s_flag[r][c4 * 4]) = f;
        }
      }
    }
    __syncthreads();
    if (tid < SE) {
      const int b = b0 + tid;
      constexpr int CH = 16;                               // steps whose LDS reads are issued ahead of the dependent chain
      for (int r0 = 0; r0 < rows; r0 += CH) {
        float rw[CH], rt[CH], vl[CH]; uint32_t fl[CH]; int sd[CH];
#pragma unroll
        for (int u = 0; u < CH; ++u) {
          const int r = min(r0 + u, STILE - 1);            // rows beyond the tile: valid LDS, masked below
          rw[u] = s_rew[r][tid]; fl[u] = s_flag[r][tid]; sd[u] = s_seed[r][tid];
          rt[u] = use_ret ? s_ret[r][tid] : 0.f;
          vl[u] = use_val ? s_val[r][tid] : 0.f;
        }
        LOADS_ISSUED();
#pragma unroll
        for (int u = 0; u < CH; ++u) {
          const int t = t0 + r0 + u;
          if (r0 + u < rows && sg.start < T) {
            const float v0 = vnext;
            const bool seg_ends = !(fl[u] & 1u);
            const bool last_in_seg = seg_ends || t == T - 1;
            sg.rs += (double)rw[u];
            if (t == sg.start) { sg.r0 = rw[u]; sg.v0 = v0; sg.seed = sd[u]; }
            if (use_ret) {
              float adv = rt[u] - v0;
              if (st == DCD_SCORE_VALUE_L1) adv = fabsf(adv); else if (st == DCD_SCORE_POSITIVE_VALUE_LOSS) adv = adv > 0.f ? adv : 0.f;
              sg.acc += adv; sg.mx = adv > sg.mx ? adv : sg.mx;
            } else if (st == DCD_SCORE_ONE_STEP_TD) {
              if (!last_in_seg) { const float td = fabsf((rw[u] + (a.g * vl[u])) - v0); sg.acc += td; sg.mx = td > sg.mx ? td : sg.mx; }
            }
            if (seg_ends) { park_segment<ST>(a, b, sg, t + 1, 1 | ((fl[u] & 2u) ? 0 : 2), k); sg.reset(t + 1); }
          }
          if (use_val && r0 + u < rows) vnext = vl[u];
        }
      }
    }
  }
  if (tid < SE && sg.start < T) { tail_row<ST>(a, b0 + tid, vnext, sg); park_segment<ST>(a, b0 + tid, sg, T, 0, k); }   // trailing partial segment (:551-578)
  __syncthreads();                                         // the block's slots [s_range[0], s_range[1]) are parked
  for (int j = s_range[0] + tid; j < min(s_range[1], a.max_recs); j += SNT) finish_record<ST>(a.recs + j);
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
struct StatSets { const float* ep_ret[DCD_MAX_SCAN_SETS]; const uint8_t* info[DCD_MAX_SCAN_SETS]; float* mean_r[DCD_MAX_SCAN_SETS]; float* max_r[DCD_MAX_SCAN_SETS]; int* n_eps[DCD_MAX_SCAN_SETS]; };
__device__ __forceinline__ void return_stats_body(const StatSets& sets, int set, int b, int T, int B) {
  const float* __restrict__ ep_ret = sets.ep_ret[set]; const uint8_t* __restrict__ info = sets.info[set];
  float* __restrict__ mean_r = sets.mean_r[set]; float* __restrict__ max_r = sets.max_r[set];
  int* __restrict__ n_eps = sets.n_eps[set];
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
__global__ void __launch_bounds__(SCAN_BLOCK) return_stats_kernel(StatSets sets, int T, int B) {
  return_stats_body(sets, blockIdx.y, blockIdx.x * SCAN_BLOCK + threadIdx.x, T, B);
}

// The scans that follow a pair of rollouts and do not depend on each other — GAE, return statistics, segment counts, each for
// up to two rollouts — side by side in ONE launch: blockIdx.y = kind * n + set.  Each is latency-bound on 64 ... 256 waves, so
// the launch takes the time of the longest (GAE) instead of their sum.  256-thread workgroups: GAE / statistics take one env
// per thread (the first ceil(B / 256) workgroups of their rows), the count 64 envs x 4 time quarters per workgroup.
__global__ void __launch_bounds__(256) rollout_tail_kernel(GaeSets gs, StatSets ss, CountSets cs, int n, int n_count, int T, int B, float g, float gl) {
  const int y = blockIdx.y;
  if (y < n) { if (blockIdx.x * 256 < B) gae_body(gs, y, blockIdx.x * 256 + threadIdx.x, T, B, g, gl); }
  else if (y < 2 * n) { if (blockIdx.x * 256 < B) return_stats_body(ss, y - n, blockIdx.x * 256 + threadIdx.x, T, B); }
  else if (y - 2 * n < n_count) { if (blockIdx.x * 64 < B) seg_count_par_body(cs, y - 2 * n, blockIdx.x, T, B); }
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

// ---- VecNormalize(ret=True).step_wait (envs/wrappers/vec_normalize.py:37-45) --------------------------------------------------
// ret = ret*gamma + rews; ret_rms.update(ret); rews = clip(rews / sqrt(ret_rms.var + eps), -cliprew, cliprew); ret[news] = 0.
// fp64 throughout like the numpy statement (rews promoted from fp32); ret_rms is baselines' RunningMeanStd: batch mean and
// population variance merged into (mean, var, count) by the parallel-variance formula.  One workgroup: the batch moments are
// two deterministic tree reductions (numpy sums pairwise: the last bits of the moments may differ, tolerance in the tests).
__global__ void __launch_bounds__(1024) return_normalize_kernel(float* __restrict__ rew, const uint8_t* __restrict__ done, double* __restrict__ ret,
                                                                double* __restrict__ rms, double gamma, double eps, double clip, int B) {
  __shared__ double red[1024];
  __shared__ double bcast;
  const int tid = threadIdx.x;
  auto reduce = [&](double v) {
    red[tid] = v;
    __syncthreads();
    for (int s = 512; s > 0; s >>= 1) { if (tid < s) red[tid] += red[tid + s]; __syncthreads(); }
    const double r = red[0];
    __syncthreads();
    return r;
  };
  double a = 0.0;
  for (int i = tid; i < B; i += 1024) { const double r = ret[i] * gamma + (double)rew[i]; ret[i] = r; a += r; }
  const double bmean = reduce(a) / (double)B;
  double q = 0.0;
  for (int i = tid; i < B; i += 1024) { const double d = ret[i] - bmean; q += d * d; }
  const double bvar = reduce(q) / (double)B;
  if (tid == 0) {                                   // RunningMeanStd.update_from_moments
    const double mean = rms[0], var = rms[1], count = rms[2], bc = (double)B;
    const double delta = bmean - mean, tot = count + bc;
    const double m2 = var * count + bvar * bc + delta * delta * count * bc / tot;
    rms[0] = mean + delta * bc / tot; rms[1] = m2 / tot; rms[2] = tot;
    bcast = sqrt(m2 / tot + eps);
  }
  __syncthreads();
  const double sd = bcast;
  for (int i = tid; i < B; i += 1024) {
    double r = (double)rew[i] / sd;
    r = r < -clip ? -clip : (r > clip ? clip : r);
    rew[i] = (float)r;
    if (done[i]) ret[i] = 0.0;
  }
}

extern "C" {

int dcd_gae_returns_batch(int32_t n, const float* const* rewards, const float* const* value_preds, const float* const* masks,
                          float* const* returns, int32_t T, int32_t B, double gamma, double gae_lambda, void* stream) {
  if (n < 1 || n > DCD_MAX_SCAN_SETS || !rewards || !value_preds || !masks || !returns || T <= 0 || B <= 0)
    return dcd::fail("dcd_gae_returns_batch: bad arguments (1 <= n <= %d)", DCD_MAX_SCAN_SETS);
  GaeSets sets; memset(&sets, 0, sizeof sets);
  for (int k = 0; k < n; ++k) {
    if (!rewards[k] || !value_preds[k] || !masks[k] || !returns[k]) return dcd::fail("dcd_gae_returns_batch: null pointer in set %d", k);
    sets.rew[k] = rewards[k]; sets.val[k] = value_preds[k]; sets.msk[k] = masks[k]; sets.ret[k] = returns[k];
  }
  DCD_LAUNCH_TAGGED(dcd::PROF_TAIL, gae_kernel, dim3(nblk(B), n), dim3(SCAN_BLOCK), (hipStream_t)stream, sets, T, B, (float)gamma,
                     (float)(gamma * gae_lambda));
  HIP_OK(hipGetLastError());
  return 0;
}
int dcd_gae_returns(const float* rewards, const float* value_preds, const float* masks, float* returns, int32_t T, int32_t B,
                    double gamma, double gae_lambda, void* stream) {
  return dcd_gae_returns_batch(1, &rewards, &value_preds, &masks, &returns, T, B, gamma, gae_lambda, stream);
}

int dcd_discounted_returns(const float* rewards, const float* value_preds, const float* masks, float* returns, int32_t T, int32_t B,
                           double gamma, void* stream) {
  if (!rewards || !value_preds || !masks || !returns || T <= 0 || B <= 0) return dcd::fail("dcd_discounted_returns: bad arguments");
  DCD_LAUNCH_TAGGED(dcd::PROF_TAIL, disc_kernel, dim3(nblk(B)), dim3(SCAN_BLOCK), (hipStream_t)stream, rewards, value_preds, masks, returns, T, B, (float)gamma);
  HIP_OK(hipGetLastError());
  return 0;
}
int dcd_batched_value_loss(const float* returns, const float* value_preds, float* out, int32_t T, int32_t B, int32_t mode,
                           int32_t power, int32_t clipped, void* stream) {
  if (!returns || !value_preds || !out || T <= 0 || B <= 0 || mode < 0 || mode > 2) return dcd::fail("dcd_batched_value_loss: bad arguments");
  DCD_LAUNCH_TAGGED(dcd::PROF_OTHER, value_loss_kernel, dim3(nblk(B)), dim3(SCAN_BLOCK), (hipStream_t)stream, returns, value_preds, out, T, B, mode, power, clipped);
  HIP_OK(hipGetLastError());
  return 0;
}

// one rollout's scoring arguments (dcd_episode_scores_batch)
static int check_score_set(const DcdEpisodeScoreArgs& q, int32_t T, int32_t B, int32_t A, int32_t strategy, bool needs_logits, bool needs_returns) {
  if (!q.value_preds || !q.rewards || !q.masks || !q.cliffhanger_masks || !q.level_seeds || !q.recs || !q.n_recs || !q.workspace || T <= 0 || B <= 0)
    return dcd::fail("dcd_episode_scores: bad arguments");
  if (needs_logits && (!q.action_log_dist || A <= 0 || A > 16)) return dcd::fail("dcd_episode_scores: strategy %d needs action_log_dist with 1..16 actions", strategy);
  if (needs_returns && !q.returns) return dcd::fail("dcd_episode_scores: strategy %d needs returns", strategy);
  return 0;
}

// the LDS-staged scoring path (time-parallel counts + tile kernel) applies: value / TD / uniform strategies, B % 32 == 0, aligned
static bool scores_staged(int32_t n, const DcdEpisodeScoreArgs* q, int32_t B, int32_t strategy) {
  const bool needs_logits = strategy == DCD_SCORE_POLICY_ENTROPY || strategy == DCD_SCORE_LEAST_CONFIDENCE || strategy == DCD_SCORE_MIN_MARGIN;
  const bool needs_returns = strategy <= DCD_SCORE_SIGNED_VALUE_LOSS;
  auto aligned16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; };
  bool staged = !needs_logits && B % SE == 0 && B <= (1 << 18);
  for (int k = 0; k < n && staged; ++k)
    staged = aligned16(q[k].rewards) && aligned16(q[k].masks) && aligned16(q[k].cliffhanger_masks) && aligned16(q[k].level_seeds) &&
             aligned16(q[k].value_preds) && (!needs_returns || aligned16(q[k].returns));
  return staged;
}

static int episode_scores_impl(int32_t n, const DcdEpisodeScoreArgs* q, int32_t T, int32_t B, int32_t A, int32_t strategy, double gamma,
                               void* stream, bool counts_done) {
  if (n < 1 || n > DCD_MAX_SCAN_SETS || !q) return dcd::fail("dcd_episode_scores_batch: 1 <= n <= %d sets", DCD_MAX_SCAN_SETS);
  if (strategy < 0 || strategy > DCD_SCORE_RETURN_SUM) return dcd::fail("dcd_episode_scores: unknown strategy %d", strategy);
  const bool needs_logits = strategy == DCD_SCORE_POLICY_ENTROPY || strategy == DCD_SCORE_LEAST_CONFIDENCE || strategy == DCD_SCORE_MIN_MARGIN;
  const bool needs_returns = strategy <= DCD_SCORE_SIGNED_VALUE_LOSS;
  for (int k = 0; k < n; ++k) if (check_score_set(q[k], T, B, A, strategy, needs_logits, needs_returns)) return 1;
  hipStream_t s = (hipStream_t)stream;
  const bool staged = scores_staged(n, q, B, strategy);
  if (counts_done && !staged) return dcd::fail("dcd_episode_scores: internal: counts_done on the unstaged path");
  auto fill = [&](ScoreArgs& a, const DcdEpisodeScoreArgs& x) {
    a.ret = x.returns; a.val = x.value_preds; a.rew = x.rewards; a.msk = x.masks; a.cliff = x.cliffhanger_masks; a.logits = x.action_log_dist;
    a.seeds = x.level_seeds; a.T = T; a.B = B; a.A = A; a.strategy = strategy; a.g = (float)gamma; a.recs = x.recs; a.max_recs = x.max_recs;
    a.max_entropy = A > 1 ? (float)(-(1.0 / A) * log(1.0 / A) * A) : 1.f;       // level_sampler.py:281, evaluated in fp64 like numpy
    a.offsets = nullptr;
  };
  if (staged) {                                       // [T, 32-env] tiles in LDS, 2 launches for all n sets
    ScoreSets sets; memset(&sets, 0, sizeof sets);
    CountSets cs; memset(&cs, 0, sizeof cs);
    for (int k = 0; k < n; ++k) {
      int* counts = q[k].workspace; int* block_sums = q[k].workspace + B;
      fill(sets.a[k], q[k]); sets.a[k].logits = nullptr;
      sets.counts[k] = counts; sets.block_sums[k] = block_sums; sets.n_recs[k] = q[k].n_recs;
      cs.msk[k] = q[k].masks; cs.counts[k] = counts; cs.block_sums[k] = block_sums;
    }
    if (!counts_done) DCD_LAUNCH_TAGGED(dcd::PROF_COUNT, seg_count_par_kernel, dim3((B + 63) / 64, n), dim3(256), s, cs, T, B);
    switch (strategy) {
#define DCD_SCORE_CASE(ST) case ST: DCD_LAUNCH_TAGGED(dcd::PROF_SCORE, seg_score_lds_kernel<ST>, dim3(B / SE, n), dim3(SNT), s, sets); break;
      DCD_SCORE_CASE(0) DCD_SCORE_CASE(1) DCD_SCORE_CASE(2) DCD_SCORE_CASE(6) DCD_SCORE_CASE(7) DCD_SCORE_CASE(8)
#undef DCD_SCORE_CASE
    }
    HIP_OK(hipGetLastError());
    return 0;
  }
  for (int k = 0; k < n; ++k) {                       // one env per lane (logit strategies, odd batch sizes): per set
    int* counts = q[k].workspace; int* offsets = q[k].workspace + B;
    DCD_LAUNCH_TAGGED(dcd::PROF_COUNT, seg_count_kernel, dim3(nblk(B)), dim3(SCAN_BLOCK), s, q[k].masks, T, B, counts);
    DCD_LAUNCH_TAGGED(dcd::PROF_COUNT, seg_scan_kernel, dim3(1), dim3(1024), s, counts, offsets, B, q[k].n_recs);
    ScoreArgs a; fill(a, q[k]); a.offsets = offsets;
    switch (strategy) {
#define DCD_SCORE_CASE(ST) case ST: DCD_LAUNCH_TAGGED(dcd::PROF_SCORE, seg_score_kernel<ST>, dim3(nblk(B)), dim3(SCAN_BLOCK), s, a); break;
      DCD_SCORE_CASE(0) DCD_SCORE_CASE(1) DCD_SCORE_CASE(2) DCD_SCORE_CASE(3) DCD_SCORE_CASE(4)
      DCD_SCORE_CASE(5) DCD_SCORE_CASE(6) DCD_SCORE_CASE(7) DCD_SCORE_CASE(8)
#undef DCD_SCORE_CASE
    }
  }
  HIP_OK(hipGetLastError());
  return 0;
}

int dcd_episode_scores_batch(int32_t n, const DcdEpisodeScoreArgs* q, int32_t T, int32_t B, int32_t A, int32_t strategy, double gamma,
                             void* stream) {
  return episode_scores_impl(n, q, T, B, A, strategy, gamma, stream, false);
}

int dcd_rollout_tail_batch(int32_t n, const float* const* rewards, const float* const* value_preds, const float* const* masks,
                           float* const* returns, const float* const* ep_ret, const uint8_t* const* info, float* const* mean_return,
                           float* const* max_return, int32_t* const* n_episodes, const DcdEpisodeScoreArgs* score, int32_t T, int32_t B,
                           int32_t A, int32_t strategy, double gamma, double gae_lambda, void* stream) {
  if (n < 1 || n > DCD_MAX_SCAN_SETS || !rewards || !value_preds || !masks || !returns || !ep_ret || !info || !mean_return || !max_return ||
      T <= 0 || B <= 0)
    return dcd::fail("dcd_rollout_tail_batch: bad arguments (1 <= n <= %d)", DCD_MAX_SCAN_SETS);
  GaeSets gs; memset(&gs, 0, sizeof gs);
  StatSets ss; memset(&ss, 0, sizeof ss);
  CountSets cs; memset(&cs, 0, sizeof cs);
  for (int k = 0; k < n; ++k) {
    if (!rewards[k] || !value_preds[k] || !masks[k] || !returns[k] || !ep_ret[k] || !info[k] || !mean_return[k] || !max_return[k])
      return dcd::fail("dcd_rollout_tail_batch: null pointer in set %d", k);
    gs.rew[k] = rewards[k]; gs.val[k] = value_preds[k]; gs.msk[k] = masks[k]; gs.ret[k] = returns[k];
    ss.ep_ret[k] = ep_ret[k]; ss.info[k] = info[k]; ss.mean_r[k] = mean_return[k]; ss.max_r[k] = max_return[k];
    ss.n_eps[k] = n_episodes ? n_episodes[k] : nullptr;
  }
  int n_count = 0;                                                   // counts ride along when the scoring takes the staged path
  if (score) {
    if (strategy < 0 || strategy > DCD_SCORE_RETURN_SUM) return dcd::fail("dcd_rollout_tail_batch: unknown strategy %d", strategy);
    const bool needs_logits = strategy == DCD_SCORE_POLICY_ENTROPY || strategy == DCD_SCORE_LEAST_CONFIDENCE || strategy == DCD_SCORE_MIN_MARGIN;
    for (int k = 0; k < n; ++k) if (check_score_set(score[k], T, B, A, strategy, needs_logits, strategy <= DCD_SCORE_SIGNED_VALUE_LOSS)) return 1;
    if (scores_staged(n, score, B, strategy)) {
      n_count = n;
      for (int k = 0; k < n; ++k) { cs.msk[k] = score[k].masks; cs.counts[k] = score[k].workspace; cs.block_sums[k] = score[k].workspace + B; }
    }
  }
  const unsigned gx = (unsigned)((B + 63) / 64);                     // >= ceil(B / 256): the GAE / statistics rows use a prefix of it
  DCD_LAUNCH_TAGGED(dcd::PROF_TAIL, rollout_tail_kernel, dim3(n_count ? gx : (unsigned)((B + 255) / 256), 2 * n + n_count), dim3(256), (hipStream_t)stream,
                     gs, ss, cs, n, n_count, T, B, (float)gamma, (float)(gamma * gae_lambda));
  HIP_OK(hipGetLastError());
  if (score) return episode_scores_impl(n, score, T, B, A, strategy, gamma, stream, n_count > 0);
  return 0;
}

int dcd_episode_scores(const float* returns, const float* value_preds, const float* rewards, const float* masks,
                       const float* cliffhanger_masks, const int32_t* level_seeds, const float* action_log_dist, int32_t T, int32_t B,
                       int32_t A, int32_t strategy, double gamma, DcdEpisodeRec* recs, int32_t max_recs, int32_t* n_recs,
                       int32_t* workspace, void* stream) {
  DcdEpisodeScoreArgs q = {returns, value_preds, rewards, masks, cliffhanger_masks, level_seeds, action_log_dist, recs, max_recs, n_recs, workspace};
  return dcd_episode_scores_batch(1, &q, T, B, A, strategy, gamma, stream);
}

int dcd_return_normalize(float* rewards, const uint8_t* done, double* ret, double* rms, double gamma, double epsilon, double cliprew,
                         int32_t B, void* stream) {
  if (!rewards || !done || !ret || !rms || B <= 0) return dcd::fail("dcd_return_normalize: bad arguments");
  hipLaunchKernelGGL(return_normalize_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, rewards, done, ret, rms, gamma, epsilon, cliprew, B);
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

int dcd_rollout_return_stats_batch(int32_t n, const float* const* ep_ret, const uint8_t* const* info, int32_t T, int32_t B,
                                   float* const* mean_return, float* const* max_return, int32_t* const* n_episodes, void* stream) {
  if (n < 1 || n > DCD_MAX_SCAN_SETS || !ep_ret || !info || !mean_return || !max_return || T <= 0 || B <= 0)
    return dcd::fail("dcd_rollout_return_stats_batch: bad arguments (1 <= n <= %d)", DCD_MAX_SCAN_SETS);
  StatSets sets; memset(&sets, 0, sizeof sets);
  for (int k = 0; k < n; ++k) {
    if (!ep_ret[k] || !info[k] || !mean_return[k] || !max_return[k]) return dcd::fail("dcd_rollout_return_stats_batch: null pointer in set %d", k);
    sets.ep_ret[k] = ep_ret[k]; sets.info[k] = info[k]; sets.mean_r[k] = mean_return[k]; sets.max_r[k] = max_return[k];
    sets.n_eps[k] = n_episodes ? n_episodes[k] : nullptr;
  }
  DCD_LAUNCH_TAGGED(dcd::PROF_OTHER, return_stats_kernel, dim3(nblk(B), n), dim3(SCAN_BLOCK), (hipStream_t)stream, sets, T, B);
  HIP_OK(hipGetLastError());
  return 0;
}
int dcd_rollout_return_stats(const float* ep_ret, const uint8_t* info, int32_t T, int32_t B, float* mean_return, float* max_return,
                             int32_t* n_episodes, void* stream) {
  if (!ep_ret || !info || !mean_return || !max_return || T <= 0 || B <= 0) return dcd::fail("dcd_rollout_return_stats: bad arguments");
  return dcd_rollout_return_stats_batch(1, &ep_ret, &info, T, B, &mean_return, &max_return, n_episodes ? &n_episodes : nullptr, stream);
}

int dcd_env_return(const float* agent_mean, const float* agent_max, const float* adv_mean, const float* adv_max, float* out, int32_t B,
                   int32_t mode, void* stream) {
  if (!agent_max || !out || B <= 0 || mode < 0 || mode > 2) return dcd::fail("dcd_env_return: bad arguments");
  if (mode != 2 && (!agent_mean || !adv_max)) return dcd::fail("dcd_env_return: paired modes need agent_mean and adv_max");
  if (mode == 1 && !adv_mean) return dcd::fail("dcd_env_return: flexible_paired needs adv_mean");
  DCD_LAUNCH_TAGGED(dcd::PROF_OTHER, env_return_kernel, dim3((B + 255) / 256), dim3(256), (hipStream_t)stream, agent_mean, agent_max, adv_mean, adv_max, out, B, mode);
  HIP_OK(hipGetLastError());
  return 0;
}

}  // extern "C"
