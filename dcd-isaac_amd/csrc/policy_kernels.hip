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
template <int VV, int FF, int EE, int HH, bool U8 = false>
__global__ void __launch_bounds__(PB) policy_features_kernel(DcdStudentNet n, const float* __restrict__ image,
                                                             const float* __restrict__ direction, const float* __restrict__ hc,
                                                             const float* __restrict__ masks, float* __restrict__ x, int KP, int B) {
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
template <int H>
__global__ void __launch_bounds__(CB) policy_cell_heads_kernel(DcdStudentNet n, const float* __restrict__ packed, CellArgs a) {
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
// the backward needs is written by the forward into [T,N,*] buffers (activated gates, cell state, masked h).
constexpr int LB = 256;

// gates: [N,4H] pre-activation in, ACTIVATED (i,f,g,o) out (saved for backward).  c_prev: C[t-1] (or c0),
// m_t: masks[t] (applied to c_prev here; h was masked when hm[t] was formed).  m_next: masks[t+1] or NULL.
__global__ void __launch_bounds__(LB) lstm_seq_fwd_kernel(float* __restrict__ gates, const float* __restrict__ c_prev,
                                                          const float* __restrict__ m_t, const float* __restrict__ m_next,
                                                          float* __restrict__ c_out, float* __restrict__ h_out,
                                                          float* __restrict__ hm_next, int N, int H) {
  const int q = blockIdx.x * LB + threadIdx.x;
  if (q >= N * H) return;
  const int b = q / H, j = q - b * H;
  float* g = gates + (size_t)b * 4 * H;
  const float i = sigmoidf_(g[j]), f = sigmoidf_(g[H + j]), gg = tanhf(g[2 * H + j]), o = sigmoidf_(g[3 * H + j]);
  const float c = f * (c_prev[q] * m_t[b]) + i * gg;
  const float h = o * tanhf(c);
  g[j] = i; g[H + j] = f; g[2 * H + j] = gg; g[3 * H + j] = o;
  c_out[q] = c; h_out[q] = h;
  if (hm_next) hm_next[q] = h * m_next[b];
}

// step t of the backward.  act: activated gates of step t; c_t, c_prev (C[t-1] or c0); m_t = masks[t];
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
  const float i = a[j], f = a[H + j], g = a[2 * H + j], o = a[3 * H + j];
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
                                                              int B, int32_t* __restrict__ err) {
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
  if (net->V == 5 && net->F == 16 && net->E == 5 && net->H == 256)
    hipLaunchKernelGGL((policy_features_kernel<5, 16, 5, 256>), grid, block, 0, (hipStream_t)stream, *net, image, direction, hc, masks, x, x_pitch, B);
  else
    hipLaunchKernelGGL((policy_features_kernel<0, 0, 0, 0>), grid, block, 0, (hipStream_t)stream, *net, image, direction, hc, masks, x, x_pitch, B);
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

int dcd_policy_features_u8(const DcdStudentNet* net, const uint8_t* image_u8, const uint8_t* dir_u8, const float* hc, const float* masks,
                           float* x, int32_t x_pitch, int32_t B, void* stream) {
  if (check_net(net)) return 1;
  if (x_pitch < dcd_policy_input_width(net)) return dcd::fail("dcd_policy_features_u8: x_pitch %d < K", x_pitch);
  if (!image_u8 || !dir_u8 || !hc || !masks || !x || B <= 0) return dcd::fail("dcd_policy_features_u8: NULL pointer or B <= 0");
  const dim3 grid((B + FE - 1) / FE), block(PB);
  const float* img = reinterpret_cast<const float*>(image_u8); const float* dir = reinterpret_cast<const float*>(dir_u8);
  if (net->V == 5 && net->F == 16 && net->E == 5 && net->H == 256)
    hipLaunchKernelGGL((policy_features_kernel<5, 16, 5, 256, true>), grid, block, 0, (hipStream_t)stream, *net, img, dir, hc, masks, x, x_pitch, B);
  else
    hipLaunchKernelGGL((policy_features_kernel<0, 0, 0, 0, true>), grid, block, 0, (hipStream_t)stream, *net, img, dir, hc, masks, x, x_pitch, B);
  if (hipGetLastError() != hipSuccess) return dcd::fail("dcd_policy_features_u8: launch failed");
  return 0;
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

int dcd_policy_cell_heads(const DcdStudentNet* net, const float* packed, const float* gates, const float* hc_in, const float* masks, float* hc_out,
                          float* value, int64_t* action, float* action_log_prob, float* action_log_dist, float* u_out,
                          uint32_t seed, uint32_t counter, const uint32_t* counter_base, int64_t env_index_base, int32_t deterministic, int32_t B, void* stream) {
  if (check_net(net)) return 1;
  if (!packed || !gates || !hc_in || !masks || !hc_out || !value || !action || !action_log_prob || !action_log_dist || B <= 0)
    return dcd::fail("dcd_policy_cell_heads: NULL pointer or B <= 0");
  CellArgs a{gates, hc_in, masks, hc_out, value, action, action_log_prob, action_log_dist, u_out, seed, counter, counter_base, env_index_base,
             deterministic, B};
  const size_t lds = sizeof(float) * (PK_FLOATS + CW * MAXH * EW + CW * EW * 64);
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&policy_cell_heads_kernel<MAXH>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds) != hipSuccess)
      return dcd::fail("dcd_policy_cell_heads: cannot reserve %zu bytes of LDS", lds);
    attr_set = true;
  }
  hipLaunchKernelGGL(policy_cell_heads_kernel<MAXH>, dim3((B + HE - 1) / HE), dim3(CB), lds, (hipStream_t)stream, *net, packed, a);
  if (hipGetLastError() != hipSuccess) return dcd::fail("dcd_policy_cell_heads: launch failed");
  return 0;
}

int dcd_lstm_seq_fwd_step(float* gates, const float* c_prev, const float* m_t, const float* m_next, float* c_out, float* h_out,
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
                        int32_t* err, int32_t B, void* stream) {
  if (!net || !image_u8 || !prev_u8 || !cell || !count || !src || !rows || !tile_pos || !err || B <= 0)
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
                     (long long)prev_env_stride, (const int32_t*)cell, (const int32_t*)count, src, rows, tile_pos, n_tiles, B, err);
  if (hipGetLastError() != hipSuccess) return dcd::fail("dcd_grid_delta_rows: launch failed");
  return 0;
}

}  // extern "C"
