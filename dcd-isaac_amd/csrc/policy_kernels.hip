// policy_kernels.hip — the student's rollout-time policy step (no grad) for gfx950 + C ABI.
//
// What the reference does per rollout step in eager PyTorch (models/multigrid_models.py:141-167 `act`,
// :109-139 base forward; models/common.py:149-156 masked single-step LSTM; models/distributions.py:18-56):
//   conv3x3(3->F)+ReLU | one-hot(direction)->Linear | (h,c)*mask -> LSTM cell -> actor MLP -> Categorical
//   .sample()/.log_probs(), critic MLP -> value                           (~45 kernel launches, host-bound)
// here: TWO kernels around ONE GEMM (the only dense contraction: [B,K]x[K,4H], K = 9F+E+H = 405 — left to
// rocBLAS through torch.addmm, already at ~90 TFLOP/s fp32 for this shape):
//   policy_features_kernel   X[b] = [ relu(conv(image[b])) | embed_w[:,dir[b]] + embed_b | h[b]*mask[b] ]
//   (GEMM)                   gates = X @ [W_ih | W_hh]^T + (b_ih + b_hh)
//   policy_cell_heads_kernel LSTM pointwise -> (h',c') straight into storage[step+1]; actor/critic hidden
//                            layers from LDS-resident weights (one output unit per lane); logits, log-softmax,
//                            Philox inverse-CDF sample (or argmax), log-prob, value -> storage[step]
// fp32 throughout; results agree with the PyTorch module to fp32 round-off (different summation order), the
// tolerance is written in tests/test_gpu_policy.py.  The weights stay PyTorch parameters: the kernels read
// them through the pointers of DcdStudentNet, nothing is copied or cached across optimiser steps.
#include <hip/hip_runtime.h>
#include <math.h>
#include "../../include/dcd_hip.h"
#include "dcd_internal.h"
#include "mg_env.h"

namespace {

constexpr int PB = 256;          // threads per workgroup (4 waves)
constexpr int FE = 8;            // envs per workgroup, features kernel
constexpr int HE = 16;           // envs per workgroup, cell+heads kernel
constexpr int CB = 512, CW = CB / 64, EW = HE / CW;   // its workgroup: 8 waves x 2 envs (two waves per SIMD hide each other's latency)
constexpr int MAXV = 7, MAXF = 32, MAXH = 256, MAXA = 16;

// ---- features ---------------------------------------------------------------------------------
// VV/FF/EE/HH > 0: compile-time geometry (the registered student: V=5, F=16, E=5, H=256 -> K=405), so the
// (env, k) index split is multiplications by constants; 0: runtime geometry.
// U8: the observation arrives as the env wrote it — raw encoding bytes [vx][vy][c] and a direction byte (RolloutStorage
// in u8 mode: 76 instead of 304 bytes per env-step through HBM) — and VecPreprocessImageWrapper's /10 + HWC->CHW
// (obs_wrappers.py:88-115) happens while the image is staged into LDS: dcd::div10f(k) is the exact fp32 of k / 10.0.
// A launch serves up to two networks (PAIRED: protagonist and antagonist act on their own copies of the levels at the same
// step): blockIdx.y picks the set — own parameters, inputs and operand rows; the launch is latency-bound, the second
// network's workgroups ride along (9.3 us for one, see DESIGN.md §3 for both).
struct FeatSets {
  DcdStudentNet n[2];
  const float* image[2]; const float* direction[2]; const float* hc[2]; const float* masks[2]; float* x[2];
};

template <int VV, int FF, int EE, int HH, bool U8 = false>
__global__ void __launch_bounds__(PB) policy_features_kernel(FeatSets S, int KP, int B) {
  const int set = blockIdx.y;
  const DcdStudentNet n = S.n[set];
  const float* __restrict__ image = S.image[set];
  const float* __restrict__ direction = S.direction[set];
  const float* __restrict__ hc = S.hc[set];
  const float* __restrict__ masks = S.masks[set];
  float* __restrict__ x = S.x[set];
  __shared__ float s_img[FE][3 * MAXV * MAXV];
  __shared__ float s_w[MAXF * 27 + MAXF];
  __shared__ float s_emb[FE][16];
  const int tid = threadIdx.x, env0 = blockIdx.x * FE, ne = min(FE, B - env0);
  const int V = VV ? VV : n.V, F = FF ? FF : n.F, E = VV ? EE : n.E, H = HH ? HH : n.H;
  const int O = V - 2, OO = O * O, CF = F * OO, IMG = 3 * V * V;
  // every global load of the workgroup is issued before the first wait: one memory round trip in all.
  // (a) h * masks (common.py:151-153): ne*H values, HB per thread (H = 256 = PB: thread <-> hidden unit, i <-> env)
  constexpr int HB = FE * MAXH / PB;
  float hv[HB];
#pragma unroll
  for (int i = 0; i < HB; ++i) {
    const int q = i * PB + tid, e = q / H, j = q - e * H;
    hv[i] = q < ne * H ? hc[(size_t)(env0 + e) * 2 * H + j] * masks[env0 + e] : 0.f;
  }
  // (c) one_hot(dim, direction) @ W^T + b (:123-125) == column `direction` of W + b: a dependent gather, started now
  if (tid < ne * E) {
    const int e = tid / E, j = tid - e * E;
    int d = U8 ? (int)reinterpret_cast<const uint8_t*>(direction)[env0 + e] : (int)direction[env0 + e];
    d = d < 0 ? 0 : (d >= n.D ? n.D - 1 : d);
    s_emb[e][j] = n.embed_w[j * n.D + d] + n.embed_b[j];
  }
  if (U8) {
    const uint8_t* img8 = reinterpret_cast<const uint8_t*>(image);
    for (int q = tid; q < ne * IMG; q += PB) {                    // byte q of the batch = env e, cell (vx,vy), channel c
      const int e = q / IMG, r = q - e * IMG, cell = r / 3, c = r - cell * 3;
      s_img[e][c * V * V + cell] = dcd::div10f(img8[(size_t)env0 * IMG + q]);
    }
  } else {
    for (int q = tid; q < ne * IMG; q += PB) s_img[q / IMG][q % IMG] = image[(size_t)env0 * IMG + q];
  }
  for (int q = tid; q < F * 27; q += PB) s_w[q] = n.conv_w[q];
  for (int q = tid; q < F; q += PB) s_w[MAXF * 27 + q] = n.conv_b[q];
  __builtin_amdgcn_sched_barrier(0);                               // loads above stay batched ahead of the stores
#pragma unroll
  for (int i = 0; i < HB; ++i) {
    const int q = i * PB + tid, e = q / H, j = q - e * H;
    if (q < ne * H) x[(size_t)(env0 + e) * KP + CF + E + j] = hv[i];
  }
  __syncthreads();
  if (tid < ne * E) { const int e = tid / E, j = tid - e * E; x[(size_t)(env0 + e) * KP + CF + j] = s_emb[e][j]; }
  // (b) Conv2d 'valid' + Flatten + ReLU (multigrid_models.py:45-47), image and weights in LDS
  for (int q = tid; q < ne * CF; q += PB) {
    const int e = q / CF, k = q - e * CF, f = k / OO, o = k - f * OO, oy = o / O, ox = o - oy * O;
    const float* w = s_w + f * 27;
    float acc = s_w[MAXF * 27 + f];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) acc = fmaf(w[c * 9 + ky * 3 + kx], s_img[e][c * V * V + (oy + ky) * V + ox + kx], acc);
    x[(size_t)(env0 + e) * KP + k] = fmaxf(acc, 0.f);
  }
}

// VecPreprocessImageWrapper._preprocess (obs_wrappers.py:88-115) for N stored observations: image_f32[n][c][vx][vy] =
// image_u8[n][vx][vy][c] / 10.0 (exact: dcd::div10f), dir_f32[n] = dir_u8[n].  One output element per thread.
__global__ void obs_u8_to_f32_kernel(const uint8_t* __restrict__ img8, const uint8_t* __restrict__ dir8, float* __restrict__ img, float* __restrict__ dir,
                                     long long N, int V) {
  const long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int VV = V * V, IMG = 3 * VV;
  if (q >= N * IMG) return;
  const long long n = q / IMG;
  const int r = (int)(q - n * IMG), c = r / VV, cell = r - c * VV;
  img[q] = dcd::div10f(img8[n * IMG + cell * 3 + c]);
  if (dir8 && dir && r == 0) dir[n] = (float)dir8[n];
}

// ---- features of a grid-sized observation (the teacher: 15 x 15 image, F = 128 filters, time-step embedding, z) -----
// One workgroup per env.  Thread <-> conv output position (P = (S-2)^2 <= 256): its 27-value window sits in registers,
// the loop runs over the F filters with the filter's 27 weights as LDS broadcasts; for a fixed filter the P threads
// write P consecutive floats of the row.  Row layout as the module's cat (multigrid_models.py:119-131):
// [relu(conv) flattened [F][P] | scalar_embed(one_hot(scalar)) | random_z | h * mask], pitch KP.
constexpr int GMAXS = 17, GMAXF = 128;
// U8: image is the raw full-grid encoding uint8[B,S,S,3] ([x][y][c], what the teacher's rollout buffer holds in u8 mode);
// the /10 and HWC->CHW of the preprocessing wrapper happen while it is staged (dcd::div10f: exact).
template <bool U8>
__global__ void __launch_bounds__(PB) grid_features_kernel(DcdStudentNet n, const float* __restrict__ image, const float* __restrict__ scalar,
                                                           const float* __restrict__ z, int Z, const float* __restrict__ hc,
                                                           const float* __restrict__ masks, float* __restrict__ x, int KP, int B) {
  __shared__ float s_img[3 * GMAXS * GMAXS];
  __shared__ float s_w[GMAXF * 27 + GMAXF];
  const int tid = threadIdx.x, b = blockIdx.x;
  const int S = n.V, O = S - 2, P = O * O, F = n.F, E = n.E, H = n.H, CF = F * P, IMG = 3 * S * S;
  float* row = x + (size_t)b * KP;
  // independent global loads first: h, mask, embedding gather, z
  float hv = 0.f, ev = 0.f, zv = 0.f;
  const float m = masks[b];
  if (tid < H) hv = hc[(size_t)b * 2 * H + tid] * m;
  if (tid < E) { int d = (int)scalar[b]; d = d < 0 ? 0 : (d >= n.D ? n.D - 1 : d); ev = n.embed_w[tid * n.D + d] + n.embed_b[tid]; }
  if (tid < Z) zv = z[(size_t)b * Z + tid];
  if (U8) {
    const uint8_t* img8 = reinterpret_cast<const uint8_t*>(image) + (size_t)b * IMG;
    for (int q = tid; q < IMG; q += PB) { const int cell = q / 3, c = q - cell * 3; s_img[c * S * S + cell] = dcd::div10f(img8[q]); }
  } else {
    for (int q = tid; q < IMG; q += PB) s_img[q] = image[(size_t)b * IMG + q];
  }
  for (int q = tid; q < F * 27; q += PB) s_w[q] = n.conv_w[q];
  for (int q = tid; q < F; q += PB) s_w[GMAXF * 27 + q] = n.conv_b[q];
  if (tid < E) row[CF + tid] = ev;
  if (tid < Z) row[CF + E + tid] = zv;
  if (tid < H) row[CF + E + Z + tid] = hv;
  __syncthreads();
  if (tid < P) {
    const int oy = tid / O, ox = tid - oy * O;
    float win[27];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) win[c * 9 + ky * 3 + kx] = s_img[c * S * S + (oy + ky) * S + ox + kx];
    for (int f = 0; f < F; ++f) {
      const float* w = s_w + f * 27;
      float acc = s_w[GMAXF * 27 + f];
#pragma unroll
      for (int k = 0; k < 27; ++k) acc = fmaf(w[k], win[k], acc);
      row[f * P + tid] = fmaxf(acc, 0.f);
    }
  }
}

// ---- LSTM pointwise + heads + sampling ----------------------------------------------------------
// rollout-time activations on the hardware exp/rcp (v_exp_f32 / v_rcp_f32, ~1 ulp each): |error| < 3e-7 on outputs in
// [-1,1], inside the 3e-6 tolerance against the PyTorch module (tests/test_gpu_policy.py).  The TRAINING kernels
// below keep the precise libm forms.
__device__ __forceinline__ float fast_sigmoid(float v) { return __frcp_rn(1.f + __expf(-v)); }
__device__ __forceinline__ float fast_tanh(float v) { const float a = fminf(fmaxf(v, -15.f), 15.f); return 1.f - 2.f * __frcp_rn(__expf(2.f * a) + 1.f); }
__device__ __forceinline__ float sigmoidf_(float v) { return 1.f / (1.f + expf(-v)); }

struct CellArgs {
  const float* gates; const float* hc_in; const float* masks; float* hc_out;
  float* value; int64_t* action; float* logp; float* log_dist; float* u_out;
  uint32_t seed, counter; const uint32_t* counter_base; int64_t env_index_base; int32_t deterministic, B;
};

// Packed head weights (built by heads_pack_kernel whenever the optimiser stepped): w1t [H][64] = stacked
// [actor | critic] first layers TRANSPOSED (a wave reads 64 consecutive floats per k), b1 [64],
// w2 [16][68] (row a < NA: actor output a over units [0,A1); row NA: critic output over units [A1,U)), b2 [16].
constexpr int W2P = 68;   // row pitch of w2: 68 mod 32 banks = 4, the 8 output lanes of an env hit distinct banks
constexpr int PK_W1 = 0, PK_B1 = MAXH * 64, PK_W2 = PK_B1 + 64, PK_B2 = PK_W2 + MAXA * W2P, PK_FLOATS = PK_B2 + MAXA;
static_assert(PK_FLOATS % 4 == 0, "packed heads are staged as float4");

__global__ void __launch_bounds__(PB) heads_pack_kernel(DcdStudentNet n, float* __restrict__ pk) {
  const int q = blockIdx.x * PB + threadIdx.x, H = n.H, U = n.A1 + n.C1;
  if (q >= PK_FLOATS) return;
  float w = 0.f;
  if (q < PK_B1) { const int k = q >> 6, u = q & 63; w = u < n.A1 ? n.actor_w1[u * H + k] : (u < U ? n.critic_w1[(u - n.A1) * H + k] : 0.f); }
  else if (q < PK_W2) { const int u = q - PK_B1; w = u < n.A1 ? n.actor_b1[u] : (u < U ? n.critic_b1[u - n.A1] : 0.f); }
  else if (q < PK_B2) {
    const int r = (q - PK_W2) / W2P, u = (q - PK_W2) % W2P;
    if (r < n.NA && u < n.A1) w = n.actor_w2[r * n.A1 + u];
    else if (r == n.NA && u >= n.A1 && u < U) w = n.critic_w2[u - n.A1];
  } else { const int r = q - PK_B2; w = r < n.NA ? n.actor_b2[r] : (r == n.NA ? n.critic_b2[0] : 0.f); }
  pk[q] = w;
}

// Workgroup = CW waves x EW envs.  Cell: lane l owns hidden units l, l+64, l+128, l+192 of each of its wave's envs.
// Heads layer 1: lane l = stacked unit l, the wave's envs advance together (one weight read feeds EW FMAs, their h
// values come as one LDS broadcast).  Layer 2, log-softmax and sampling: 16 lanes per env.
struct CellSets { DcdStudentNet n[2]; const float* packed[2]; CellArgs a[2]; };

template <int H>
__global__ void __launch_bounds__(CB) policy_cell_heads_kernel(CellSets S) {
  const DcdStudentNet n = S.n[blockIdx.y];
  const float* __restrict__ packed = S.packed[blockIdx.y];
  const CellArgs a = S.a[blockIdx.y];
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* s_pk = smem;                               // PK_FLOATS
  float* s_h = s_pk + PK_FLOATS;                    // [CW waves][H][EW envs]
  float* s_t = s_h + CW * H * EW;                   // [CW waves][EW envs][64]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int q = tid; q < PK_FLOATS / 4; q += CB) reinterpret_cast<float4*>(s_pk)[q] = reinterpret_cast<const float4*>(packed)[q];
  const uint32_t ctr = a.counter + (a.counter_base ? *a.counter_base : 0u);   // device-resident base: graph replays draw fresh numbers
  float* hb = s_h + wave * H * EW;
  float* tb = s_t + wave * EW * 64;
  const int b0 = blockIdx.x * HE + wave * EW;
#pragma unroll
  for (int e = 0; e < EW; ++e) {                                       // nn.LSTM cell, gate order i,f,g,o
    const int b = b0 + e;
    if (b < a.B) {
      const float m = a.masks[b];
      const float* g = a.gates + (size_t)b * 4 * H;
      const float* ci = a.hc_in + (size_t)b * 2 * H + H;
      float* ho = a.hc_out + (size_t)b * 2 * H;
#pragma unroll
      for (int q = 0; q < H / 64; ++q) {
        const int j = lane + 64 * q;
        const float gi = g[j], gf = g[H + j], gg = g[2 * H + j], go = g[3 * H + j];
        const float c = fast_sigmoid(gf) * (ci[j] * m) + fast_sigmoid(gi) * fast_tanh(gg);
        const float h = fast_sigmoid(go) * fast_tanh(c);
        ho[j] = h; ho[H + j] = c; hb[j * EW + e] = h;
      }
    } else {
#pragma unroll
      for (int q = 0; q < H / 64; ++q) hb[(lane + 64 * q) * EW + e] = 0.f;
    }
  }
  __syncthreads();
  {                                                                   // Linear(H -> A1 | C1) + Tanh (common.py:215-224)
    const float* w1 = s_pk + PK_W1;
    const float bias = s_pk[PK_B1 + lane];
    static_assert(EW == 2, "the layer-1 loop reads the wave's h values as one float2");
    float a0 = bias, a1 = bias;
#pragma unroll 8
    for (int k = 0; k < H; ++k) {
      const float w = w1[k * 64 + lane];
      const float2 h2 = *reinterpret_cast<const float2*>(hb + k * 2);
      a0 = fmaf(w, h2.x, a0); a1 = fmaf(w, h2.y, a1);
    }
    tb[lane] = fast_tanh(a0); tb[64 + lane] = fast_tanh(a1);
  }
  __syncthreads();
  const int e = lane >> 4, o = lane & 15, b = b0 + e, U = n.A1 + n.C1, grp = lane & ~15;
  const bool live = e < EW && b < a.B;
  float out = 0.f;                                                    // lanes o < NA: logits; o == NA: value
  if (o <= n.NA && e < EW) {
    out = s_pk[PK_B2 + o];
    const float* w2 = s_pk + PK_W2 + o * W2P;
    const float* t = tb + e * 64;
    for (int u = 0; u < U; ++u) out = fmaf(w2[u], t[u], out);
  }
  // log-softmax over the NA lanes of this env's 16-lane group (Categorical(logits=...): logits - logsumexp)
  const float lg = o < n.NA ? out : -INFINITY;
  float mx = lg;
#pragma unroll
  for (int s = 1; s < 16; s <<= 1) mx = fmaxf(mx, __shfl_xor(mx, s));
  const float ex = o < n.NA ? expf(lg - mx) : 0.f;
  float sum = ex;
#pragma unroll
  for (int s = 1; s < 16; s <<= 1) sum += __shfl_xor(sum, s);
  const float lsm = lg - mx - logf(sum);
  const float p = ex / sum;
  // action: inverse CDF of one Philox uniform per (env, counter), or the first arg max
  uint32_t r4[4];
  dcd::philox4x32_10(ctr, (uint32_t)(a.env_index_base + b), (uint32_t)((a.env_index_base + b) >> 32), 0x504F4C31u,
                     a.seed, 0x41435421u, r4);
  const float u01 = (float)(r4[0] >> 8) * (1.0f / 16777216.0f);
  float cum = 0.f, best = -INFINITY; int arg = 0, chosen = -1;
  for (int q = 0; q < n.NA; ++q) {
    const float pq = __shfl(p, grp + q), lq = __shfl(lg, grp + q);
    cum += pq;
    if (chosen < 0 && u01 < cum) chosen = q;
    if (lq > best) { best = lq; arg = q; }
  }
  if (chosen < 0) chosen = n.NA - 1;                                  // cum rounded below u
  const int act = a.deterministic ? arg : chosen;
  const float lp = __shfl(lsm, grp + act), val = __shfl(out, grp + n.NA);
  if (live) {
    if (o < n.NA) a.log_dist[(size_t)b * n.NA + o] = lsm;
    if (o == 0) {
      a.value[b] = val; a.action[b] = act; a.logp[b] = lp;
      if (a.u_out) a.u_out[b] = u01;
    }
  }
}

// ---- masked LSTM sequence for the PPO update (training path) ---------------------------------------
// models/common.py:157-207: the flattened (T*N) sequence is cut wherever ANY env has mask 0 and the state is
// re-masked at each cut == multiplying (h, c) by masks[t] before EVERY step.  Per step the recurrence is one
// GEMM (gates = xp[t] + hm @ W_hh^T, rocBLAS) and ONE pointwise kernel here; backward likewise.  Everything
// the backward needs sits in [T,N,*] buffers after the forward (pre-activation gates — the GEMMs' outputs —, cell state, masked h).
constexpr int LB = 256;

// gates: [N,4H] pre-activations (i,f,g,o), left as they are — the backward kernel recomputes the four activations from them
// (round 4: writing the activated gates back was a third of this kernel's 50 MB per step; the recomputation is 4 transcendentals
// per element on values the backward reads anyway, bit-identical to what was stored).  c_prev: C[t-1] (or c0),
// m_t: masks[t] (applied to c_prev here; h was masked when hm[t] was formed).  m_next: masks[t+1] or NULL.
__global__ void __launch_bounds__(LB) lstm_seq_fwd_kernel(const float* __restrict__ gates, const float* __restrict__ c_prev,
                                                          const float* __restrict__ m_t, const float* __restrict__ m_next,
                                                          float* __restrict__ c_out, float* __restrict__ h_out,
                                                          float* __restrict__ hm_next, int N, int H) {
  const int q = blockIdx.x * LB + threadIdx.x;
  if (q >= N * H) return;
  const int b = q / H, j = q - b * H;
  const float* g = gates + (size_t)b * 4 * H;
  const float i = sigmoidf_(g[j]), f = sigmoidf_(g[H + j]), gg = tanhf(g[2 * H + j]), o = sigmoidf_(g[3 * H + j]);
  const float c = f * (c_prev[q] * m_t[b]) + i * gg;
  const float h = o * tanhf(c);
  c_out[q] = c; h_out[q] = h;
  if (hm_next) hm_next[q] = h * m_next[b];
}

// step t of the backward.  act: PRE-activation gates of step t (activated here, as the forward did); c_t, c_prev (C[t-1] or c0); m_t = masks[t];
// dout: grad wrt h_t from the heads; dhm_next: DG[t+1] @ W_hh (raw; NULL at t = T-1), m_next = masks[t+1];
// dcf_next: dc_{t+1} * f_{t+1} (raw; NULL at T-1).  Writes dgates (pre-activation grads) and dcf = dc_t * f_t.
__global__ void __launch_bounds__(LB) lstm_seq_bwd_kernel(const float* __restrict__ act, const float* __restrict__ c_t,
                                                          const float* __restrict__ c_prev, const float* __restrict__ m_t,
                                                          const float* __restrict__ dout, const float* __restrict__ dhm_next,
                                                          const float* __restrict__ dcf_next, const float* __restrict__ m_next,
                                                          const float* __restrict__ dh_last, const float* __restrict__ dc_last,
                                                          float* __restrict__ dgates, float* __restrict__ dcf, int N, int H) {
  const int q = blockIdx.x * LB + threadIdx.x;
  if (q >= N * H) return;
  const int b = q / H, j = q - b * H;
  const float* a = act + (size_t)b * 4 * H;
  const float i = sigmoidf_(a[j]), f = sigmoidf_(a[H + j]), g = tanhf(a[2 * H + j]), o = sigmoidf_(a[3 * H + j]);
  const float mn = m_next ? m_next[b] : 0.f;
  float dh = dout ? dout[q] : 0.f, dc = 0.f;
  if (dhm_next) dh += dhm_next[q] * mn;
  if (dcf_next) dc += dcf_next[q] * mn;
  if (dh_last) dh += dh_last[q];
  if (dc_last) dc += dc_last[q];
  const float tc = tanhf(c_t[q]);
  dc += dh * o * (1.f - tc * tc);
  const float cm = c_prev[q] * m_t[b];
  float* d = dgates + (size_t)b * 4 * H;
  d[j] = dc * g * i * (1.f - i);
  d[H + j] = dc * cm * f * (1.f - f);
  d[2 * H + j] = dc * i * (1.f - g * g);
  d[3 * H + j] = dh * tc * o * (1.f - o);
  dcf[q] = dc * f;
}

// out[r, :] += sum over k < K of y[src[r, k], :] (src < 0: no term), terms added in k order: the scatter side of the
// teacher's incremental input projection without atomics — every y row is read once, every out row written once,
// and the result does not depend on scheduling.  Workgroup = one destination row x 256 float4 columns.
__global__ void __launch_bounds__(256) gather_sum_rows_kernel(const float* __restrict__ y, const int32_t* __restrict__ src,
                                                              float* __restrict__ out, int K, int G4) {
  const int r = blockIdx.x, c = blockIdx.y * 256 + threadIdx.x;
  if (c >= G4) return;
  const int32_t* s = src + (size_t)r * K;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  bool any = false;
  for (int k = 0; k < K; ++k) {
    const int i = s[k];                                    // wave-uniform
    if (i < 0) continue;
    const float4 v = reinterpret_cast<const float4*>(y)[(size_t)i * G4 + c];
    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    any = true;
  }
  if (!any) return;
  float4* o = reinterpret_cast<float4*>(out) + (size_t)r * G4 + c;
  float4 w = *o;
  w.x += acc.x; w.y += acc.y; w.z += acc.z; w.w += acc.w;
  *o = w;
}

// ---- the teacher's carried input projection, device-resident (no host synchronisation, fixed shapes: graph-capturable) --------
// A constructive teacher step changes at most ONE cell of the full-grid observation (adversarial.py:466-530: a wall on a
// free cell, the goal, or the agent), i.e. at most nine of the P = (S-2)^2 conv positions see a different 3x3 window.
// Per step:  xp[b] += sum over those positions p of (relu(conv)(new window) - relu(conv)(old window)) @ Wp[p],
// Wp[p] = W_ih[:, f*P + p]^T [F, 4H].  The (env, position) rows are binned by POSITION so that one weight block serves a
// whole tile of rows: bins are padded to DR-row tiles, the GEMM runs over a FIXED number of tiles (dcd_grid_delta_tiles)
// whose weight block index comes from a device table, and unused tiles multiply garbage nobody reads.
//   grid_changed_cell_kernel  one wave per env: the changed cell (or -1), error bit if more than one cell changed
//   grid_bin_rank_kernel      one workgroup per position: members in env order -> rank within the bin, bin size
//   grid_delta_rows_kernel    bin bases (prefix over padded sizes), feature-difference rows, (env, slot) -> row table,
//                             tile -> position table
// Ranks follow env order, so rows, sums and results do not depend on scheduling.
constexpr int DR = 128;           // rows per GEMM tile (rocBLAS runs 128-row batches at twice the rate of 64-row ones)
constexpr int DE = 8;             // envs per workgroup, grid_delta_rows_kernel
constexpr int DMAXP = 256;        // (GMAXS - 2)^2 = 225 positions at most

__global__ void __launch_bounds__(256) grid_changed_cell_kernel(const uint8_t* __restrict__ cur, const uint8_t* __restrict__ prev, long long prev_stride,
                                                                int S, int B, int32_t* __restrict__ cell, int32_t* __restrict__ src,
                                                                int32_t* __restrict__ err) {
  const int lane = threadIdx.x & 63, b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  const int NC = S * S;
  const uint8_t* a = cur + (size_t)b * NC * 3;
  const uint8_t* o = prev + (size_t)b * prev_stride;
  int first = -1, n = 0;
  for (int c0 = 0; c0 < NC; c0 += 64) {                                // wave-uniform trip count
    const int c = c0 + lane;
    bool ch = false;
    if (c < NC) ch = a[c * 3] != o[c * 3] || a[c * 3 + 1] != o[c * 3 + 1] || a[c * 3 + 2] != o[c * 3 + 2];
    const unsigned long long m = __ballot(ch);
    if (m && first < 0) first = c0 + __ffsll((long long)m) - 1;
    n += __popcll(m);
  }
  if (lane == 0) {
    cell[b] = first;
    if (n > 1) atomicOr(err, 1);
  }
  if (lane < 9) src[(size_t)b * 9 + lane] = -1;
}

// position p = (r, q) sees cell (cr, cq) iff r <= cr <= r+2 and q <= cq <= q+2; slot of the pair = (cr-r)*3 + (cq-q)
__global__ void __launch_bounds__(256) grid_bin_rank_kernel(const int32_t* __restrict__ cell, int S, int B, int32_t* __restrict__ src,
                                                            int32_t* __restrict__ count) {
  __shared__ int s_part[256];
  const int p = blockIdx.x, O = S - 2, r = p / O, q = p - r * O, tid = threadIdx.x;
  const int per = (B + 255) / 256, b0 = tid * per, b1 = min(B, b0 + per);
  int mine = 0;
  for (int b = b0; b < b1; ++b) {
    const int c = cell[b];
    if (c >= 0) { const int cr = c / S, cq = c - cr * S; mine += (cr >= r && cr <= r + 2 && cq >= q && cq <= q + 2); }
  }
  s_part[tid] = mine;
  __syncthreads();
  for (int d = 1; d < 256; d <<= 1) {                                  // inclusive scan
    const int v = tid >= d ? s_part[tid - d] : 0;
    __syncthreads();
    s_part[tid] += v;
    __syncthreads();
  }
  int rank = s_part[tid] - mine;
  if (tid == 255) count[p] = s_part[255];
  for (int b = b0; b < b1; ++b) {
    const int c = cell[b];
    if (c < 0) continue;
    const int cr = c / S, cq = c - cr * S;
    if (cr >= r && cr <= r + 2 && cq >= q && cq <= q + 2) src[(size_t)b * 9 + (cr - r) * 3 + (cq - q)] = rank++;
  }
}

__global__ void __launch_bounds__(256) grid_delta_rows_kernel(DcdStudentNet n, const uint8_t* __restrict__ cur, const uint8_t* __restrict__ prev,
                                                              long long prev_stride, const int32_t* __restrict__ cell,
                                                              const int32_t* __restrict__ count, int32_t* __restrict__ src,
                                                              float* __restrict__ rows, int32_t* __restrict__ tile_pos, int n_tiles,
                                                              int32_t* __restrict__ n_used, int B, int32_t* __restrict__ err) {
  __shared__ float s_w[GMAXF * 27 + GMAXF];
  __shared__ float s_new[DE][3][25], s_old[DE][3][25];                  // the 5 x 5 cells around each env's changed cell
  __shared__ int s_base[DMAXP + 1], s_row[DE][9], s_cell[DE];
  const int tid = threadIdx.x, S = n.V, O = S - 2, P = O * O, F = n.F, env0 = blockIdx.x * DE, ne = min(DE, B - env0);
  if (tid < ne) s_cell[tid] = cell[env0 + tid];
  for (int q = tid; q < F * 27; q += 256) s_w[q] = n.conv_w[q];
  for (int q = tid; q < F; q += 256) s_w[GMAXF * 27 + q] = n.conv_b[q];
  s_base[tid] = tid < P ? (count[tid] + DR - 1) / DR * DR : 0;           // padded bin sizes -> exclusive prefix
  __syncthreads();
  for (int d = 1; d < 256; d <<= 1) {
    const int v = tid >= d ? s_base[tid - d] : 0;
    __syncthreads();
    s_base[tid] += v;
    __syncthreads();
  }
  const int incl = s_base[tid];
  __syncthreads();
  s_base[tid + 1 <= DMAXP ? tid + 1 : DMAXP] = incl;                     // s_base[p] = rows before bin p (tid 255 -> slot 256)
  if (tid == 0) s_base[0] = 0;
  __syncthreads();
  if (blockIdx.x == 0) {                                               // tile -> position table for the GEMM's weight blocks
    if (tid == 0 && s_base[P] > n_tiles * DR) atomicOr(err, 2);
    if (tid == 0) *n_used = min(s_base[P] / DR, n_tiles);
    for (int t = tid; t < n_tiles; t += 256) {
      int lo = 0, hi = P;                                              // last p with base[p] <= t*DR
      while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (s_base[mid] <= t * DR) lo = mid; else hi = mid; }
      tile_pos[t] = lo;
    }
  }
  // windows: thread -> (env e, channel c, neighbourhood cell (i, j)), cells outside the grid read as 0 (never used)
  for (int w = tid; w < ne * 75; w += 256) {
    const int e = w / 75, r75 = w - e * 75, c = r75 / 25, ij = r75 - c * 25, i = ij / 5, j = ij - i * 5;
    const int cc = s_cell[e];
    float vn = 0.f, vo = 0.f;
    if (cc >= 0) {
      const int cr = cc / S + i - 2, cq = cc % S + j - 2;
      if (cr >= 0 && cr < S && cq >= 0 && cq < S) {
        const size_t off = (size_t)(cr * S + cq) * 3 + c;
        vn = dcd::div10f(cur[(size_t)(env0 + e) * S * S * 3 + off]);
        vo = dcd::div10f(prev[(size_t)(env0 + e) * prev_stride + off]);
      }
    }
    s_new[e][c][ij] = vn; s_old[e][c][ij] = vo;
  }
  if (tid < ne * 9) {                                                  // (env, slot) -> row of the binned operand
    const int e = tid / 9, k = tid - e * 9, cc = s_cell[e];
    int row = -1;
    if (cc >= 0) {
      const int r = cc / S - k / 3, q = cc % S - k % 3;                // slot k = (cr-r)*3 + (cq-q)
      if (r >= 0 && r < O && q >= 0 && q < O) {
        row = s_base[r * O + q] + src[(size_t)(env0 + e) * 9 + k];
        if (row >= n_tiles * DR) row = -1;                             // cannot happen (error bit 2 above); never write outside
      }
    }
    s_row[e][k] = row;
    src[(size_t)(env0 + e) * 9 + k] = row;
  }
  __syncthreads();
  // thread <-> filter f (its 27 weights in registers), the workgroup's 256 / FP2 pair lanes walk the (env, slot) pairs:
  // a wave reads one pair's window values as LDS broadcasts and writes 64 consecutive floats of its row.  VALU-bound
  // (36.8k rows x 128 filters x 54 FMAs at 4096 envs: ~30 us); fetching a pair's 54 window values with one LDS read and
  // handing them round by v_readlane measured the same (the readlane takes the issue slot the LDS read freed).
  const int FP2 = F <= 32 ? 32 : (F <= 64 ? 64 : 128), f = tid & (FP2 - 1), lanes = 256 / FP2;
  if (f >= F) return;
  float w[27];
#pragma unroll
  for (int k = 0; k < 27; ++k) w[k] = s_w[f * 27 + k];
  const float bias = s_w[GMAXF * 27 + f];
  for (int pair = tid / FP2; pair < ne * 9; pair += lanes) {
    const int e = pair / 9, k = pair - e * 9;
    const int row = s_row[e][k];
    if (row < 0) continue;
    // position (r, q) = (cr - k/3, cq - k%3): its window cell (k1, k2) is neighbourhood cell (2 - k/3 + k1, 2 - k%3 + k2)
    const int i0 = 2 - k / 3, j0 = 2 - k % 3;
    float an = bias, ao = bias;
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int k1 = 0; k1 < 3; ++k1)
#pragma unroll
        for (int k2 = 0; k2 < 3; ++k2) {
          an = fmaf(w[c * 9 + k1 * 3 + k2], s_new[e][c][(i0 + k1) * 5 + j0 + k2], an);
          ao = fmaf(w[c * 9 + k1 * 3 + k2], s_old[e][c][(i0 + k1) * 5 + j0 + k2], ao);
        }
    rows[(size_t)row * F + f] = fmaxf(an, 0.f) - fmaxf(ao, 0.f);
  }
}

// ---- the teacher's rollout step around its two GEMMs: operand row of everything but the image, then cell + heads ------
// x_rest[b] = [scalar_embed(one_hot(scalar[b])) | z[b] | h[b]*mask[b] | 1 | 0...]  (multigrid_models.py:119-131 minus the
// conv features, whose projection is carried: dcd_grid_delta_rows).  The 1 multiplies the bias row of the stacked weight.
__global__ void __launch_bounds__(256) grid_rest_features_kernel(DcdStudentNet n, const float* __restrict__ scalar, const float* __restrict__ z,
                                                                 int Z, const float* __restrict__ hc, const float* __restrict__ masks,
                                                                 float* __restrict__ x, int KP, int B) {
  const long long q = (long long)blockIdx.x * 256 + threadIdx.x;
  const int b = (int)(q / KP), k = (int)(q - (long long)b * KP);
  if (b >= B) return;
  const int E = n.E, H = n.H;
  float v = 0.f;
  if (k < E) { int d = (int)scalar[b]; d = d < 0 ? 0 : (d >= n.D ? n.D - 1 : d); v = n.embed_w[k * n.D + d] + n.embed_b[k]; }
  else if (k < E + Z) v = z[(size_t)b * Z + (k - E)];
  else if (k < E + Z + H) v = hc[(size_t)b * 2 * H + (k - E - Z)] * masks[b];
  else if (k == E + Z + H) v = 1.f;
  x[(size_t)b * KP + k] = v;
}

// Wide heads (the teacher: 169 actions).  Packed layout: w1t [H][64] stacked [actor | critic] first layers transposed, b1 [64],
// w2 [NA + 1][WW2P] (row a < NA: actor output a over the A1 actor units; row NA: the critic output over the C1 critic units,
// both starting at column 0; pitch 33: the 64 lanes of a wave read 64 different rows conflict-free), b2 [NA + 1].
constexpr int WW2P = 33, WMAXA = 192;
__host__ __device__ constexpr int wide_w2(int) { return MAXH * 64 + 64; }
__host__ __device__ inline int wide_b2(int NA) { return MAXH * 64 + 64 + (NA + 1) * WW2P; }
__host__ __device__ inline int wide_floats(int NA) { return (wide_b2(NA) + NA + 1 + 3) / 4 * 4; }

__global__ void __launch_bounds__(PB) wide_heads_pack_kernel(DcdStudentNet n, float* __restrict__ pk) {
  const int q = blockIdx.x * PB + threadIdx.x, H = n.H, U = n.A1 + n.C1, NA = n.NA;
  if (q >= wide_floats(NA)) return;
  float w = 0.f;
  if (q < MAXH * 64) { const int k = q >> 6, u = q & 63; w = u < n.A1 ? n.actor_w1[u * H + k] : (u < U ? n.critic_w1[(u - n.A1) * H + k] : 0.f); }
  else if (q < wide_w2(0)) { const int u = q - MAXH * 64; w = u < n.A1 ? n.actor_b1[u] : (u < U ? n.critic_b1[u - n.A1] : 0.f); }
  else if (q < wide_b2(NA)) {
    const int r = (q - wide_w2(0)) / WW2P, u = (q - wide_w2(0)) % WW2P;
    if (r < NA && u < n.A1) w = n.actor_w2[r * n.A1 + u];
    else if (r == NA && u < n.C1) w = n.critic_w2[u];
  } else { const int r = q - wide_b2(NA); if (r < NA) w = n.actor_b2[r]; else if (r == NA) w = n.critic_b2[0]; }
  pk[q] = w;
}

// Same cell and first head layer as policy_cell_heads_kernel (8 waves x 2 envs); the second layer, the log-softmax and the
// draw take a whole wave per env: lane l owns outputs l, l + 64, l + 128 (< NA; output NA is the value).  Inverse CDF in
// index order: per 64-group an inclusive wave scan of the probabilities, group offsets added, first output whose cumulative
// probability exceeds the uniform (ballot per group).  One Philox uniform per (env, counter), as for the students.
template <int H>
__global__ void __launch_bounds__(CB) wide_cell_heads_kernel(DcdStudentNet n, const float* __restrict__ packed, CellArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int NA = n.NA, PKF = wide_floats(NA);
  float* s_pk = smem;
  float* s_h = s_pk + PKF;                          // [CW waves][H][EW envs]
  float* s_t = s_h + CW * H * EW;                   // [CW waves][EW envs][64]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int q = tid; q < PKF / 4; q += CB) reinterpret_cast<float4*>(s_pk)[q] = reinterpret_cast<const float4*>(packed)[q];
  const uint32_t ctr = a.counter + (a.counter_base ? *a.counter_base : 0u);
  float* hb = s_h + wave * H * EW;
  float* tb = s_t + wave * EW * 64;
  const int b0 = blockIdx.x * HE + wave * EW;
#pragma unroll
  for (int e = 0; e < EW; ++e) {                                       // nn.LSTM cell, gate order i,f,g,o
    const int b = b0 + e;
    if (b < a.B) {
      const float m = a.masks[b];
      const float* g = a.gates + (size_t)b * 4 * H;
      const float* ci = a.hc_in + (size_t)b * 2 * H + H;
      float* ho = a.hc_out + (size_t)b * 2 * H;
#pragma unroll
      for (int q = 0; q < H / 64; ++q) {
        const int j = lane + 64 * q;
        const float gi = g[j], gf = g[H + j], gg = g[2 * H + j], go = g[3 * H + j];
        const float c = fast_sigmoid(gf) * (ci[j] * m) + fast_sigmoid(gi) * fast_tanh(gg);
        const float h = fast_sigmoid(go) * fast_tanh(c);
        ho[j] = h; ho[H + j] = c; hb[j * EW + e] = h;
      }
    } else {
#pragma unroll
      for (int q = 0; q < H / 64; ++q) hb[(lane + 64 * q) * EW + e] = 0.f;
    }
  }
  __syncthreads();
  {                                                                   // Linear(H -> A1 | C1) + Tanh
    const float* w1 = s_pk;
    const float bias = s_pk[MAXH * 64 + lane];
    static_assert(EW == 2, "the layer-1 loop reads the wave's h values as one float2");
    float a0 = bias, a1 = bias;
#pragma unroll 8
    for (int k = 0; k < H; ++k) {
      const float w = w1[k * 64 + lane];
      const float2 h2 = *reinterpret_cast<const float2*>(hb + k * 2);
      a0 = fmaf(w, h2.x, a0); a1 = fmaf(w, h2.y, a1);
    }
    tb[lane] = fast_tanh(a0); tb[64 + lane] = fast_tanh(a1);
  }
  __syncthreads();
  const float* w2 = s_pk + wide_w2(0);
  const float* b2 = s_pk + wide_b2(NA);
  for (int e = 0; e < EW; ++e) {
    const int b = b0 + e;
    if (b >= a.B) break;                                               // wave-uniform
    const float* t = tb + e * 64;
    float lg[3], out[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const int o = lane + 64 * r;
      float acc = 0.f;
      if (o <= NA) {
        acc = b2[o];
        const float* wr = w2 + o * WW2P;
        const float* tt = o < NA ? t : t + n.A1;
        const int nu = o < NA ? n.A1 : n.C1;
        for (int u = 0; u < nu; ++u) acc = fmaf(wr[u], tt[u], acc);
      }
      out[r] = acc;
      lg[r] = o < NA ? acc : -INFINITY;
    }
    float mx = fmaxf(fmaxf(lg[0], lg[1]), lg[2]);
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) mx = fmaxf(mx, __shfl_xor(mx, s));
    float ex[3], sum = 0.f;
#pragma unroll
    for (int r = 0; r < 3; ++r) { ex[r] = (lane + 64 * r) < NA ? expf(lg[r] - mx) : 0.f; sum += ex[r]; }
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) sum += __shfl_xor(sum, s);
    const float lse = logf(sum);
    float lsm[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      lsm[r] = lg[r] - mx - lse;
      if (lane + 64 * r < NA) a.log_dist[(size_t)b * NA + lane + 64 * r] = lsm[r];
    }
    uint32_t r4[4];
    dcd::philox4x32_10(ctr, (uint32_t)(a.env_index_base + b), (uint32_t)((a.env_index_base + b) >> 32), 0x504F4C31u,
                       a.seed, 0x41435421u, r4);
    const float u01 = (float)(r4[0] >> 8) * (1.0f / 16777216.0f);
    int chosen = -1, arg = -1;
    float base = 0.f;
#pragma unroll
    for (int r = 0; r < 3; ++r) {                                      // cumulative probabilities in index order
      float cum = ex[r] / sum;
#pragma unroll
      for (int s = 1; s < 64; s <<= 1) { const float v = __shfl_up(cum, s); if (lane >= s) cum += v; }
      cum += base;
      const unsigned long long hit = __ballot((lane + 64 * r) < NA && u01 < cum);
      if (chosen < 0 && hit) chosen = 64 * r + __ffsll((long long)hit) - 1;
      base = __shfl(cum, 63);
      const unsigned long long top = __ballot((lane + 64 * r) < NA && lg[r] == mx);
      if (arg < 0 && top) arg = 64 * r + __ffsll((long long)top) - 1;
    }
    if (chosen < 0) chosen = NA - 1;                                   // cumulative sum rounded below u
    const int act = a.deterministic ? arg : chosen;
    const int ar = act >> 6, al = act & 63;
    const float lp = __shfl(ar == 0 ? lsm[0] : (ar == 1 ? lsm[1] : lsm[2]), al);
    const int vr = NA >> 6, vl = NA & 63;
    const float val = __shfl(vr == 0 ? out[0] : (vr == 1 ? out[1] : out[2]), vl);
    if (lane == 0) {
      a.value[b] = val; a.action[b] = act; a.logp[b] = lp;
      if (a.u_out) a.u_out[b] = u01;
    }
  }
}

// ---- Y[t] = rows[t] @ Wp[tile_pos[t]]: the tiled product of the carried projection, fp32 MFMA ------------------------
// One workgroup = one 128 x 128 block of Y: tile t (DR = 128 binned rows, K = F <= 128 features) x column block nb of the
// position's weight block Wp[p] [F, G].  The whole K extent sits in LDS at once — A 128 x F (pitch 132: float4 stores,
// 4-way conflicts on the fragment reads, which are 1/12 of the MFMA time), B F x 128 — so there is no K loop to
// pipeline: load, barrier, 64 k-steps of v_mfma_f32_32x32x2_f32 (exact fp32: a k-ordered fmaf chain per element), store.
// 4 waves as 2 x 2, each 64 x 64 = four 32 x 32 accumulators (4 independent MFMAs per 4 LDS reads).  Tiles past the
// last bin (n_used, written by grid_delta_rows_kernel) exit at once: the padding of the fixed-shape launch costs nothing.
// Replaces {index_select of 456 weight blocks (234 MB) + rocBLAS batched GEMM}: 308 -> see DESIGN.md §3.
// Tried and dropped: K in two halves of 64 (67 KB of LDS -> two workgroups per CU, second half prefetched into registers
// under the first half's MFMAs): 145 -> 181 us per step, same-box A/B.
constexpr int GA_P = 132;          // LDS pitch of the A tile (floats)
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int FF>                      // FF = 128: the teacher's filter count, loads fully unrolled; 0: runtime F
__global__ void __launch_bounds__(256) grid_tile_gemm_kernel(const float* __restrict__ rows, const float* __restrict__ wp,
                                                             const int32_t* __restrict__ tile_pos, const int32_t* __restrict__ n_used,
                                                             float* __restrict__ y, int F_in, int G) {
  const int F = FF ? FF : F_in;
  extern __shared__ __attribute__((aligned(16))) float gsm[];
  float* As = gsm;                       // [DR][GA_P]
  float* Bs = gsm + DR * GA_P;           // [F][128]
  const int nbk = G / 128, t = blockIdx.x / nbk, nb = blockIdx.x - t * nbk, tid = threadIdx.x;
  if (t >= *n_used) return;              // workgroup-uniform
  const int p = tile_pos[t];
  const float* a = rows + (size_t)t * DR * F;
  const float* b = wp + (size_t)p * F * G + nb * 128;
  const int F4 = F / 4;
  if (FF == 128) {
    // every global load of the workgroup in flight before the first LDS store: ONE memory round trip for the 128 KB of
    // operands (the load -> store loop below waited 32 times: 255 us per step against 110)
    float4 ra[16], rb[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) ra[i] = reinterpret_cast<const float4*>(a)[i * 256 + tid];
#pragma unroll
    for (int i = 0; i < 16; ++i) { const int q = i * 256 + tid; rb[i] = *reinterpret_cast<const float4*>(b + (size_t)(q >> 5) * G + (q & 31) * 4); }
#pragma unroll
    for (int i = 0; i < 16; ++i) { const int q = i * 256 + tid; *reinterpret_cast<float4*>(As + (q >> 5) * GA_P + (q & 31) * 4) = ra[i]; }
#pragma unroll
    for (int i = 0; i < 16; ++i) { const int q = i * 256 + tid; *reinterpret_cast<float4*>(Bs + (q >> 5) * 128 + (q & 31) * 4) = rb[i]; }
  } else {
    for (int q = tid; q < DR * F4; q += 256) {                         // A: rows contiguous in k
      const int r = q / F4, k4 = q - r * F4;
      *reinterpret_cast<float4*>(As + r * GA_P + k4 * 4) = reinterpret_cast<const float4*>(a)[q];
    }
    for (int q = tid; q < F * 32; q += 256) {                          // B: 128 columns of row k
      const int k = q >> 5, c4 = q & 31;
      *reinterpret_cast<float4*>(Bs + k * 128 + c4 * 4) = *reinterpret_cast<const float4*>(b + (size_t)k * G + c4 * 4);
    }
  }
  __syncthreads();
  const int lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1, li = lane & 31, lk = lane >> 5;
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const float* a0 = As + (wm * 64 + li) * GA_P + lk;                   // A[i = li][k = k0 + lk]
  const float* b0 = Bs + lk * 128 + wn * 64 + li;                      // B[k = k0 + lk][j = li]
  for (int k0 = 0; k0 < F; k0 += 2) {
    const float av0 = a0[k0], av1 = a0[32 * GA_P + k0];
    const float bv0 = b0[k0 * 128], bv1 = b0[k0 * 128 + 32];
    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av0, bv0, acc[0][0], 0, 0, 0);
    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av0, bv1, acc[0][1], 0, 0, 0);
    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1, bv0, acc[1][0], 0, 0, 0);
    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1, bv1, acc[1][1], 0, 0, 0);
  }
  // C/D map of the 32 x 32 forms: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
  float* yo = y + ((size_t)t * DR + wm * 64) * G + nb * 128 + wn * 64 + li;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
        yo[(size_t)row * G + j * 32] = acc[i][j][r];
      }
}

}  // namespace

extern "C" {

static int check_net(const DcdStudentNet* n) {
  if (!n) return dcd::fail("dcd_policy: net is NULL");
  if (n->V < 3 || n->V > MAXV || n->F < 1 || n->F > MAXF) return dcd::fail("dcd_policy: unsupported conv geometry V=%d F=%d", n->V, n->F);
  if (n->H != MAXH) return dcd::fail("dcd_policy: hidden size %d unsupported (256)", n->H);
  if (n->A1 < 1 || n->C1 < 1 || n->A1 + n->C1 > 64) return dcd::fail("dcd_policy: head widths %d+%d exceed 64", n->A1, n->C1);
  if (n->NA < 1 || n->NA >= MAXA) return dcd::fail("dcd_policy: %d actions unsupported (<16)", n->NA);
  if (n->E < 0 || n->E > 16 || n->D < 1) return dcd::fail("dcd_policy: bad scalar embedding E=%d D=%d", n->E, n->D);
  return 0;
}

int dcd_policy_input_width(const DcdStudentNet* n) { return n ? n->F * (n->V - 2) * (n->V - 2) + n->E + n->H : 0; }

int dcd_policy_features(const DcdStudentNet* net, const float* image, const float* direction, const float* hc,
                        const float* masks, float* x, int32_t x_pitch, int32_t B, void* stream) {
  if (check_net(net)) return 1;
  if (x_pitch < dcd_policy_input_width(net)) return dcd::fail("dcd_policy_features: x_pitch %d < K", x_pitch);
  if (!image || !direction || !hc || !masks || !x || B <= 0) return dcd::fail("dcd_policy_features: NULL pointer or B <= 0");
  const dim3 grid((B + FE - 1) / FE), block(PB);
  FeatSets S{};
  S.n[0] = *net; S.image[0] = image; S.direction[0] = direction; S.hc[0] = hc; S.masks[0] = masks; S.x[0] = x;
  if (net->V == 5 && net->F == 16 && net->E == 5 && net->H == 256)
    hipLaunchKernelGGL((policy_features_kernel<5, 16, 5, 256>), grid, block, 0, (hipStream_t)stream, S, x_pitch, B);
  else
    hipLaunchKernelGGL((policy_features_kernel<0, 0, 0, 0>), grid, block, 0, (hipStream_t)stream, S, x_pitch, B);
  if (hipGetLastError() != hipSuccess) return dcd::fail("dcd_policy_features: launch failed");
  return 0;
}

static int grid_features(const DcdStudentNet* net, const void* image, bool u8, const float* scalar, const float* z, int32_t z_dim,
                         const float* hc, const float* masks, float* x, int32_t x_pitch, int32_t B, void* stream) {
  if (!net) return dcd::fail("dcd_policy_grid_features: net is NULL");
  const int P = (net->V - 2) * (net->V - 2);
  if (net->V < 3 || net->V > GMAXS || P > PB || net->F < 1 || net->F > GMAXF) return dcd::fail("dcd_policy_grid_features: unsupported conv geometry S=%d F=%d", net->V, net->F);
  if (net->H < 1 || net->H > PB || net->E < 0 || net->E > PB || net->D < 1 || z_dim < 0 || z_dim > PB) return dcd::fail("dcd_policy_grid_features: unsupported widths");
  if (!image || !scalar || !hc || !masks || !x || (z_dim && !z) || B <= 0) return dcd::fail("dcd_policy_grid_features: NULL pointer or B <= 0");
  if (x_pitch < net->F * P + net->E + z_dim + net->H) return dcd::fail("dcd_policy_grid_features: x_pitch %d too small", x_pitch);
  const float* img = reinterpret_cast<const float*>(image);
  if (u8) hipLaunchKernelGGL(grid_features_kernel<true>, dim3(B), dim3(PB), 0, (hipStream_t)stream, *net, img, scalar, z, z_dim, hc, masks, x, x_pitch, B);
  else hipLaunchKernelGGL(grid_features_kernel<false>, dim3(B), dim3(PB), 0, (hipStream_t)stream, *net, img, scalar, z, z_dim, hc, masks, x, x_pitch, B);
  if (hipGetLastError() != hipSuccess) return dcd::fail("dcd_policy_grid_features: launch failed");
  return 0;
}
int dcd_policy_grid_features(const DcdStudentNet* net, const float* image, const float* scalar, const float* z, int32_t z_dim,
                             const float* hc, const float* masks, float* x, int32_t x_pitch, int32_t B, void* stream) {
  return grid_features(net, image, false, scalar, z, z_dim, hc, masks, x, x_pitch, B, stream);
}
int dcd_policy_grid_features_u8(const DcdStudentNet* net, const uint8_t* image_u8, const float* scalar, const float* z, int32_t z_dim,
                                const float* hc, const float* masks, float* x, int32_t x_pitch, int32_t B, void* stream) {
  return grid_features(net, image_u8, true, scalar, z, z_dim, hc, masks, x, x_pitch, B, stream);
}
int dcd_policy_packed_heads_floats(void) { return PK_FLOATS; }

int dcd_policy_pack_heads(const DcdStudentNet* net, float* packed, void* stream) {
  if (check_net(net)) return 1;
  if (!packed) return dcd::fail("dcd_policy_pack_heads: NULL output");
  hipLaunchKernelGGL(heads_pack_kernel, dim3((PK_FLOATS + PB - 1) / PB), dim3(PB), 0, (hipStream_t)stream, *net, packed);
  if (hipGetLastError() != hipSuccess) return dcd::fail("dcd_policy_pack_heads: launch failed");
  return 0;
}

// one or two agents per launch (a1 may be NULL); the two networks must have one geometry
static int features_u8_sets(const DcdPolicyAgent* a0, const DcdPolicyAgent* a1, int32_t x_pitch, int32_t B, void* stream, const char* who) {
  FeatSets S{};
  const DcdPolicyAgent* as[2] = {a0, a1};
  const int ns = a1 ? 2 : 1;
  for (int k = 0; k < ns; ++k) {
    const DcdPolicyAgent* a = as[k];
    if (check_net(a->net)) return 1;
    if (x_pitch < dcd_policy_input_width(a->net)) return dcd::fail("%s: x_pitch %d < K", who, x_pitch);
    if (!a->image_u8 || !a->dir_u8 || !a->hc_in || !a->masks || !a->x || B <= 0) return dcd::fail("%s: NULL pointer or B <= 0", who);
    S.n[k] = *a->net; S.image[k] = reinterpret_cast<const float*>(a->image_u8); S.direction[k] = reinterpret_cast<const float*>(a->dir_u8);
    S.hc[k] = a->hc_in; S.masks[k] = a->masks; S.x[k] = a->x;
  }
  const DcdStudentNet* n0 = a0->net;
  if (a1 && (a1->net->V != n0->V || a1->net->F != n0->F || a1->net->E != n0->E || a1->net->H != n0->H || a1->net->D != n0->D))
    return dcd::fail("%s: the two networks differ in geometry", who);
  const dim3 grid((B + FE - 1) / FE, ns), block(PB);
  if (n0->V == 5 && n0->F == 16 && n0->E == 5 && n0->H == 256)
    hipLaunchKernelGGL((policy_features_kernel<5, 16, 5, 256, true>), grid, block, 0, (hipStream_t)stream, S, x_pitch, B);
  else
    hipLaunchKernelGGL((policy_features_kernel<0, 0, 0, 0, true>), grid, block, 0, (hipStream_t)stream, S, x_pitch, B);
  if (hipGetLastError() != hipSuccess) return dcd::fail("%s: launch failed", who);
  return 0;
}

int dcd_policy_features_u8(const DcdStudentNet* net, const uint8_t* image_u8, const uint8_t* dir_u8, const float* hc, const float* masks,
                           float* x, int32_t x_pitch, int32_t B, void* stream) {
  const DcdPolicyAgent a{net, nullptr, image_u8, dir_u8, hc, masks, x};
  return features_u8_sets(&a, nullptr, x_pitch, B, stream, "dcd_policy_features_u8");
}

int dcd_policy_features_u8_pair(const DcdPolicyAgent* a0, const DcdPolicyAgent* a1, int32_t x_pitch, int32_t B, void* stream) {
  if (!a0 || !a1) return dcd::fail("dcd_policy_features_u8_pair: NULL agent");
  return features_u8_sets(a0, a1, x_pitch, B, stream, "dcd_policy_features_u8_pair");
}

int dcd_obs_u8_to_f32(const uint8_t* image_u8, const uint8_t* dir_u8, float* image_f32, float* dir_f32, int64_t N, int32_t V, void* stream) {
  if (!image_u8 || !image_f32 || N < 0 || V < 1 || V > DCD_MAX_GRID) return dcd::fail("dcd_obs_u8_to_f32: bad arguments");
  if (N == 0) return 0;
  const long long total = (long long)N * 3 * V * V;
  hipLaunchKernelGGL(obs_u8_to_f32_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, image_u8, dir_u8, image_f32,
                     dir_f32, (long long)N, V);
  if (hipGetLastError() != hipSuccess) return dcd::fail("dcd_obs_u8_to_f32: launch failed");
  return 0;
}

static int cell_heads_sets(const DcdPolicyAgent* a0, const DcdPolicyAgent* a1, int32_t B, void* stream, const char* who) {
  CellSets S{};
  const DcdPolicyAgent* as[2] = {a0, a1};
  const int ns = a1 ? 2 : 1;
  for (int k = 0; k < ns; ++k) {
    const DcdPolicyAgent* a = as[k];
    if (check_net(a->net)) return 1;
    if (!a->packed || !a->gates || !a->hc_in || !a->masks || !a->hc_out || !a->value || !a->action || !a->log_prob || !a->log_dist || B <= 0)
      return dcd::fail("%s: NULL pointer or B <= 0", who);
    S.n[k] = *a->net; S.packed[k] = a->packed;
    S.a[k] = CellArgs{a->gates, a->hc_in, a->masks, a->hc_out, a->value, a->action, a->log_prob, a->log_dist, a->u_out, a->seed, a->counter,
                      a->counter_base, a->env_index_base, a->deterministic, B};
  }
  const size_t lds = sizeof(float) * (PK_FLOATS + CW * MAXH * EW + CW * EW * 64);
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&policy_cell_heads_kernel<MAXH>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds) != hipSuccess)
      return dcd::fail("%s: cannot reserve %zu bytes of LDS", who, lds);
    attr_set = true;
  }
  hipLaunchKernelGGL(policy_cell_heads_kernel<MAXH>, dim3((B + HE - 1) / HE, ns), dim3(CB), lds, (hipStream_t)stream, S);
  if (hipGetLastError() != hipSuccess) return dcd::fail("%s: launch failed", who);
  return 0;
}

int dcd_policy_cell_heads(const DcdStudentNet* net, const float* packed, const float* gates, const float* hc_in, const float* masks, float* hc_out,
                          float* value, int64_t* action, float* action_log_prob, float* action_log_dist, float* u_out,
                          uint32_t seed, uint32_t counter, const uint32_t* counter_base, int64_t env_index_base, int32_t deterministic, int32_t B, void* stream) {
  DcdPolicyAgent a{};
  a.net = net; a.packed = packed; a.gates = gates; a.hc_in = hc_in; a.masks = masks; a.hc_out = hc_out; a.value = value; a.action = action;
  a.log_prob = action_log_prob; a.log_dist = action_log_dist; a.u_out = u_out; a.seed = seed; a.counter = counter; a.counter_base = counter_base;
  a.env_index_base = env_index_base; a.deterministic = deterministic;
  return cell_heads_sets(&a, nullptr, B, stream, "dcd_policy_cell_heads");
}

int dcd_policy_cell_heads_pair(const DcdPolicyAgent* a0, const DcdPolicyAgent* a1, int32_t B, void* stream) {
  if (!a0 || !a1) return dcd::fail("dcd_policy_cell_heads_pair: NULL agent");
  return cell_heads_sets(a0, a1, B, stream, "dcd_policy_cell_heads_pair");
}

int dcd_lstm_seq_fwd_step(const float* gates, const float* c_prev, const float* m_t, const float* m_next, float* c_out, float* h_out,
                          float* hm_next, int32_t N, int32_t H, void* stream) {
  if (!gates || !c_prev || !m_t || !c_out || !h_out || N <= 0 || H <= 0) return dcd::fail("dcd_lstm_seq_fwd_step: NULL pointer or empty");
  if (hm_next && !m_next) return dcd::fail("dcd_lstm_seq_fwd_step: hm_next needs m_next");
  hipLaunchKernelGGL(lstm_seq_fwd_kernel, dim3((N * H + LB - 1) / LB), dim3(LB), 0, (hipStream_t)stream, gates, c_prev, m_t, m_next,
                     c_out, h_out, hm_next, N, H);
  if (hipGetLastError() != hipSuccess) return dcd::fail("dcd_lstm_seq_fwd_step: launch failed");
  return 0;
}

int dcd_lstm_seq_bwd_step(const float* act, const float* c_t, const float* c_prev, const float* m_t, const float* dout,
                          const float* dhm_next, const float* dcf_next, const float* m_next, const float* dh_last,
                          const float* dc_last, float* dgates, float* dcf, int32_t N, int32_t H, void* stream) {
  if (!act || !c_t || !c_prev || !m_t || !dgates || !dcf || N <= 0 || H <= 0) return dcd::fail("dcd_lstm_seq_bwd_step: NULL pointer or empty");
  if ((dhm_next || dcf_next) && !m_next) return dcd::fail("dcd_lstm_seq_bwd_step: recurrent grads need m_next");
  hipLaunchKernelGGL(lstm_seq_bwd_kernel, dim3((N * H + LB - 1) / LB), dim3(LB), 0, (hipStream_t)stream, act, c_t, c_prev, m_t, dout,
                     dhm_next, dcf_next, m_next, dh_last, dc_last, dgates, dcf, N, H);
  if (hipGetLastError() != hipSuccess) return dcd::fail("dcd_lstm_seq_bwd_step: launch failed");
  return 0;
}

int dcd_gather_sum_rows(const float* y, const int32_t* src, float* out, int32_t R, int32_t K, int32_t G, void* stream) {
  if (!y || !src || !out || R <= 0 || K <= 0 || G <= 0 || (G & 3)) return dcd::fail("dcd_gather_sum_rows: NULL pointer, empty, or G not a multiple of 4");
  if (((uintptr_t)y | (uintptr_t)out) & 15) return dcd::fail("dcd_gather_sum_rows: y / out must be 16-byte aligned");
  hipLaunchKernelGGL(gather_sum_rows_kernel, dim3((unsigned)R, (unsigned)((G / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, y, src, out, K, G / 4);
  if (hipGetLastError() != hipSuccess) return dcd::fail("dcd_gather_sum_rows: launch failed");
  return 0;
}

int dcd_grid_delta_tile_rows(void) { return DR; }

int dcd_grid_delta_tiles(int32_t S, int32_t B) {
  if (S < 3 || S > GMAXS || B <= 0) return 0;
  const long long P = (long long)(S - 2) * (S - 2);
  return (int)((9LL * B + P * (DR - 1) + DR - 1) / DR);
}

int dcd_grid_delta_rows(const DcdStudentNet* net, const uint8_t* image_u8, const uint8_t* prev_u8, int64_t prev_env_stride,
                        int32_t* cell, int32_t* count, int32_t* src, float* rows, int32_t* tile_pos, int32_t n_tiles,
                        int32_t* n_used, int32_t* err, int32_t B, void* stream) {
  if (!net || !image_u8 || !prev_u8 || !cell || !count || !src || !rows || !tile_pos || !n_used || !err || B <= 0)
    return dcd::fail("dcd_grid_delta_rows: NULL pointer or empty batch");
  if (net->V < 3 || net->V > GMAXS || net->F < 1 || net->F > GMAXF || !net->conv_w || !net->conv_b)
    return dcd::fail("dcd_grid_delta_rows: grid size must be 3..17, F <= 128, conv weights set");
  if (n_tiles < dcd_grid_delta_tiles(net->V, B)) return dcd::fail("dcd_grid_delta_rows: n_tiles below dcd_grid_delta_tiles(S, B)");
  const int S = net->V, P = (S - 2) * (S - 2);
  if (prev_env_stride != 0 && prev_env_stride != (int64_t)S * S * 3)
    return dcd::fail("dcd_grid_delta_rows: prev_env_stride must be 0 (one image for all envs) or S*S*3");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(grid_changed_cell_kernel, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, st, image_u8, prev_u8, (long long)prev_env_stride, S, B,
                     cell, src, err);
  hipLaunchKernelGGL(grid_bin_rank_kernel, dim3((unsigned)P), dim3(256), 0, st, (const int32_t*)cell, S, B, src, count);
  hipLaunchKernelGGL(grid_delta_rows_kernel, dim3((unsigned)((B + DE - 1) / DE)), dim3(256), 0, st, *net, image_u8, prev_u8,
                     (long long)prev_env_stride, (const int32_t*)cell, (const int32_t*)count, src, rows, tile_pos, n_tiles, n_used, B, err);
  if (hipGetLastError() != hipSuccess) return dcd::fail("dcd_grid_delta_rows: launch failed");
  return 0;
}
static int check_wide_net(const DcdStudentNet* n, const char* who) {
  if (!n) return dcd::fail("%s: net is NULL", who);
  if (n->H != MAXH) return dcd::fail("%s: hidden size %d unsupported (256)", who, n->H);
  if (n->A1 < 1 || n->C1 < 1 || n->A1 + n->C1 > 64 || n->A1 >= WW2P || n->C1 >= WW2P) return dcd::fail("%s: head widths %d+%d unsupported (<= 32 each)", who, n->A1, n->C1);
  if (n->NA < 1 || n->NA >= WMAXA) return dcd::fail("%s: %d actions unsupported (< 192)", who, n->NA);
  if (!n->actor_w1 || !n->actor_b1 || !n->actor_w2 || !n->actor_b2 || !n->critic_w1 || !n->critic_b1 || !n->critic_w2 || !n->critic_b2)
    return dcd::fail("%s: head weights not set", who);
  return 0;
}

int dcd_policy_grid_rest_features(const DcdStudentNet* net, const float* scalar, const float* z, int32_t z_dim, const float* hc,
                                  const float* masks, float* x, int32_t x_pitch, int32_t B, void* stream) {
  if (!net || !scalar || !hc || !masks || !x || (z_dim && !z) || B <= 0) return dcd::fail("dcd_policy_grid_rest_features: NULL pointer or B <= 0");
  if (net->E < 0 || net->D < 1 || z_dim < 0 || net->H < 1 || x_pitch < net->E + z_dim + net->H + 1 || !net->embed_w || !net->embed_b)
    return dcd::fail("dcd_policy_grid_rest_features: x_pitch %d < E + z_dim + H + 1, or embedding weights not set", x_pitch);
  const long long total = (long long)B * x_pitch;
  hipLaunchKernelGGL(grid_rest_features_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, *net, scalar, z, z_dim, hc,
                     masks, x, x_pitch, B);
  if (hipGetLastError() != hipSuccess) return dcd::fail("dcd_policy_grid_rest_features: launch failed");
  return 0;
}

int dcd_policy_wide_packed_heads_floats(const DcdStudentNet* net) { return check_wide_net(net, "dcd_policy_wide_packed_heads_floats") ? 0 : wide_floats(net->NA); }

int dcd_policy_wide_pack_heads(const DcdStudentNet* net, float* packed, void* stream) {
  if (check_wide_net(net, "dcd_policy_wide_pack_heads")) return 1;
  if (!packed) return dcd::fail("dcd_policy_wide_pack_heads: NULL output");
  hipLaunchKernelGGL(wide_heads_pack_kernel, dim3((wide_floats(net->NA) + PB - 1) / PB), dim3(PB), 0, (hipStream_t)stream, *net, packed);
  if (hipGetLastError() != hipSuccess) return dcd::fail("dcd_policy_wide_pack_heads: launch failed");
  return 0;
}

int dcd_policy_wide_cell_heads(const DcdStudentNet* net, const float* packed, const float* gates, const float* hc_in, const float* masks,
                               float* hc_out, float* value, int64_t* action, float* action_log_prob, float* action_log_dist, float* u_out,
                               uint32_t seed, uint32_t counter, const uint32_t* counter_base, int64_t env_index_base, int32_t deterministic,
                               int32_t B, void* stream) {
  if (check_wide_net(net, "dcd_policy_wide_cell_heads")) return 1;
  if (!packed || !gates || !hc_in || !masks || !hc_out || !value || !action || !action_log_prob || !action_log_dist || B <= 0)
    return dcd::fail("dcd_policy_wide_cell_heads: NULL pointer or B <= 0");
  CellArgs a{gates, hc_in, masks, hc_out, value, action, action_log_prob, action_log_dist, u_out, seed, counter, counter_base, env_index_base,
             deterministic, B};
  const size_t lds = sizeof(float) * ((size_t)wide_floats(net->NA) + CW * MAXH * EW + CW * EW * 64);
  static bool attr_set = false;
  if (!attr_set) {
    const size_t cap = sizeof(float) * ((size_t)wide_floats(WMAXA - 1) + CW * MAXH * EW + CW * EW * 64);
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&wide_cell_heads_kernel<MAXH>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)cap) != hipSuccess)
      return dcd::fail("dcd_policy_wide_cell_heads: cannot reserve %zu bytes of LDS", cap);
    attr_set = true;
  }
  hipLaunchKernelGGL(wide_cell_heads_kernel<MAXH>, dim3((B + HE - 1) / HE), dim3(CB), lds, (hipStream_t)stream, *net, packed, a);
  if (hipGetLastError() != hipSuccess) return dcd::fail("dcd_policy_wide_cell_heads: launch failed");
  return 0;
}

int dcd_grid_tile_gemm(const float* rows, const float* wp, const int32_t* tile_pos, const int32_t* n_used, float* y, int32_t n_tiles,
                       int32_t F, int32_t G, void* stream) {
  if (!rows || !wp || !tile_pos || !n_used || !y || n_tiles <= 0) return dcd::fail("dcd_grid_tile_gemm: NULL pointer or no tiles");
  if (F < 4 || F > GMAXF || (F & 3) || G < 128 || (G & 127)) return dcd::fail("dcd_grid_tile_gemm: F must be a multiple of 4 up to 128, G a multiple of 128");
  if (((uintptr_t)rows | (uintptr_t)wp | (uintptr_t)y) & 15) return dcd::fail("dcd_grid_tile_gemm: rows / wp / y must be 16-byte aligned");
  static_assert(DR == 128, "grid_tile_gemm_kernel computes 128-row tiles");
  const size_t lds = (size_t)(DR * GA_P + F * 128) * sizeof(float);
  static bool once = false;
  if (!once) {
    const int cap = (int)(DR * GA_P + GMAXF * 128) * (int)sizeof(float);
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&grid_tile_gemm_kernel<128>), hipFuncAttributeMaxDynamicSharedMemorySize, cap) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(&grid_tile_gemm_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, cap) != hipSuccess)
      return dcd::fail("dcd_grid_tile_gemm: cannot reserve %zu bytes of LDS", lds);
    once = true;
  }
  const dim3 grid((unsigned)n_tiles * (unsigned)(G / 128));
  if (F == 128) hipLaunchKernelGGL(grid_tile_gemm_kernel<128>, grid, dim3(256), lds, (hipStream_t)stream, rows, wp, tile_pos, n_used, y, F, G);
  else hipLaunchKernelGGL(grid_tile_gemm_kernel<0>, grid, dim3(256), lds, (hipStream_t)stream, rows, wp, tile_pos, n_used, y, F, G);
  if (hipGetLastError() != hipSuccess) return dcd::fail("dcd_grid_tile_gemm: launch failed");
  return 0;
}

}  // extern "C"
