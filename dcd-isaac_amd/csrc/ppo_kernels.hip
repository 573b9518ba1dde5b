// ppo_kernels.hip — the student network's PPO-batch glue as four fused kernels (gfx950) + C ABI.
//
// What `PPO.update` evaluates per minibatch (reference: algos/ppo.py:83-116 -> models/multigrid_models.py:169-186
// `evaluate_actions`, :109-139 base forward) is, around the LSTM and three dense contractions,
//   image -> Conv2d(3->16, 3x3, valid) -> Flatten -> ReLU | one_hot(direction) -> Linear(4->5) | cat          (features)
//   h -> Linear(256->32)+Tanh -> Linear(32->7) -> log-softmax -> log_prob(action), entropy                      (actor)
//   h -> Linear(256->32)+Tanh -> Linear(32->1)                                                                  (critic)
// At T*B = 1 048 576 rows PyTorch runs these as ~90 launches per epoch whose traffic is copies, not arithmetic: an im2col
// stack + two transposing copies (3 GB), a 0.6 GB `cat`, row-sums of narrow [1M, 5] / [1M, 7] gradients at 0.7 ms each,
// two dx GEMMs + an add for the heads (profiles/r04_ppo_shapes_before.log: ~45 of an update's 207 ms).  Here:
//   ppo_features_fwd_kernel   x[r] = [relu(conv(image[r])) | embed_w[:, dir[r]] + embed_b | 1 | 0..]  — the GEMM operand row of
//                             the LSTM input projection written directly; the constant 1 meets the bias column of the stacked
//                             weight, so neither the projection nor its backward needs a bias pass
//   ppo_features_bwd_kernel   d conv.weight / conv.bias / scalar_embed.weight from dx, per-workgroup partial sums (fixed order)
//   ppo_heads_fwd_kernel      [R,64] pre-activations of BOTH first layers (one stacked GEMM outside) -> tanh, second layers,
//                             log-softmax, log_prob(action), per-row entropy, value
//   ppo_heads_bwd_kernel      the reverse: d pre-activations [R,64] (one GEMM each for dh and dW outside) + partial sums of the
//                             second layers' weight / bias gradients and of the first layers' bias gradients
// fp32 throughout, no atomics, results independent of scheduling.  Checked against the composed PyTorch ops
// (tests/test_gpu_fused_ppo.py) and, through PPO.update, against the reference's own numbers (tests/test_gpu_ppo_vs_reference.py).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include "../../include/dcd_hip.h"
#include "dcd_internal.h"
#include "mg_env.h"

namespace {

constexpr int V5 = 5, VV = 25, IMG = 75, F16 = 16, CF = 144;       // the registered student: 5x5 view, 16 filters, 3x3 outputs
constexpr int FB = 256, FR = 16;                                    // features kernels: 256 threads = 16 rows x 16 filters
constexpr int IP = 76;                                              // LDS pitch of one image (75 floats + 1)
constexpr int WP = 29;                                              // LDS pitch of one filter's 27 weights + bias (odd: 16 filters, 16 banks)
constexpr int FPART = 512;                                          // floats per partial row of the features backward

template <bool U8>
__device__ __forceinline__ void stage_images(float* s_img, const void* image, long long row0, int n, int tid) {
  if (U8) {
    const uint8_t* g = reinterpret_cast<const uint8_t*>(image) + row0 * IMG;          // [vx][vy][c] bytes
    for (int q = tid; q < n * IMG; q += FB) {
      const int e = q / IMG, r = q - e * IMG, cell = r / 3, c = r - cell * 3;
      s_img[e * IP + c * VV + cell] = dcd::div10f(g[q]);                               // k / 10.0, CHW (obs_wrappers.py:88-115)
    }
  } else {
    const float* g = reinterpret_cast<const float*>(image) + row0 * IMG;              // CHW floats
    for (int q = tid; q < n * IMG; q += FB) { const int e = q / IMG; s_img[e * IP + (q - e * IMG)] = g[q]; }
  }
}

__device__ __forceinline__ int load_dir(const void* dir, bool u8, long long row, int D) {
  int d = u8 ? (int)reinterpret_cast<const uint8_t*>(dir)[row] : (int)reinterpret_cast<const float*>(dir)[row];
  return d < 0 ? 0 : (d >= D ? D - 1 : d);
}

// ---- features, forward ----------------------------------------------------------------------------------------------------
template <bool U8>
__global__ void __launch_bounds__(FB) ppo_features_fwd_kernel(DcdStudentNet n, const void* __restrict__ image, const void* __restrict__ dir,
                                                              float* __restrict__ x, int KP, long long R) {
  __shared__ float s_img[FR * IP];
  __shared__ float s_w[F16 * WP];
  const int tid = threadIdx.x, e = tid >> 4, f = tid & 15;
  const long long row0 = (long long)blockIdx.x * FR;
  const int nr = (int)min((long long)FR, R - row0);
  stage_images<U8>(s_img, image, row0, nr, tid);
  for (int q = tid; q < F16 * 27; q += FB) s_w[(q / 27) * WP + q % 27] = n.conv_w[q];
  if (tid < F16) s_w[tid * WP + 27] = n.conv_b[tid];
  __syncthreads();
  if (e >= nr) return;
  const long long row = row0 + e;
  float img[IMG], w[27];
#pragma unroll
  for (int i = 0; i < IMG; ++i) img[i] = s_img[e * IP + i];
#pragma unroll
  for (int k = 0; k < 27; ++k) w[k] = s_w[f * WP + k];
  const float b = s_w[f * WP + 27];
  float* xr = x + row * KP;
#pragma unroll
  for (int o = 0; o < 9; ++o) {                                       // Conv2d 'valid' + Flatten + ReLU (multigrid_models.py:45-47)
    const int oy = o / 3, ox = o - oy * 3;
    float acc = b;
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) acc = fmaf(w[c * 9 + ky * 3 + kx], img[c * VV + (oy + ky) * V5 + ox + kx], acc);
    xr[f * 9 + o] = fmaxf(acc, 0.f);
  }
  // one_hot(direction) @ W^T + b == column `direction` of W + b (multigrid_models.py:123-125); then the constant 1 and the padding
  const int E = n.E;
  for (int j = f; j < KP - CF; j += F16) {
    float v = 0.f;
    if (j < E) { const int d = load_dir(dir, U8, row, n.D); v = n.embed_w[j * n.D + d] + n.embed_b[j]; }
    else if (j == E) v = 1.f;
    xr[CF + j] = v;
  }
}

// ---- features, backward ---------------------------------------------------------------------------------------------------
// Thread (e, f): row lane e of 16, filter f.  Over the workgroup's rows e, e + 16, ... it accumulates its filter's 27 weight
// gradients + the bias gradient (the row's image in registers: 243 FMAs per 75 LDS reads) and, for f < E, the embedding
// gradients of output j = f per direction; then the 16 row lanes are summed (wave shuffles, then LDS over the 4 waves) in a
// fixed order.  partial[wg]: [0,432) d conv.weight [f][c][ky][kx] | [432,448) d conv.bias | [448 + j*8 + d] d embed.weight[j][d].
template <bool U8>
__global__ void __launch_bounds__(FB) ppo_features_bwd_kernel(DcdStudentNet n, const void* __restrict__ image, const void* __restrict__ dir,
                                                              const float* __restrict__ x, const float* __restrict__ dx, int KP, long long R,
                                                              int rows_per_wg, float* __restrict__ partial) {
  __shared__ float s_img[FR * IP];
  __shared__ float s_red[4 * F16 * 37];
  const int tid = threadIdx.x, e = tid >> 4, f = tid & 15;
  const long long wg0 = (long long)blockIdx.x * rows_per_wg;
  const long long wg1 = min(R, wg0 + rows_per_wg);
  float acc[27], accb = 0.f, acce[8];
#pragma unroll
  for (int k = 0; k < 27; ++k) acc[k] = 0.f;
#pragma unroll
  for (int d = 0; d < 8; ++d) acce[d] = 0.f;
  const int E = n.E, D = n.D;
  for (long long row0 = wg0; row0 < wg1; row0 += FR) {
    const int nr = (int)min((long long)FR, wg1 - row0);
    __syncthreads();                                                   // the previous chunk's images are consumed
    stage_images<U8>(s_img, image, row0, nr, tid);
    __syncthreads();
    if (e < nr) {
      const long long row = row0 + e;
      float g[9];
#pragma unroll
      for (int o = 0; o < 9; ++o) {                                    // ReLU backward: the saved output is its own mask
        const float xv = x[row * KP + f * 9 + o], gv = dx[row * KP + f * 9 + o];
        g[o] = xv > 0.f ? gv : 0.f;
        accb += g[o];
      }
      float img[IMG];
#pragma unroll
      for (int i = 0; i < IMG; ++i) img[i] = s_img[e * IP + i];
#pragma unroll
      for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) {
            float a = acc[c * 9 + ky * 3 + kx];
#pragma unroll
            for (int o = 0; o < 9; ++o) a = fmaf(g[o], img[c * VV + (o / 3 + ky) * V5 + (o % 3) + kx], a);
            acc[c * 9 + ky * 3 + kx] = a;
          }
      if (f < E) {
        const float ge = dx[row * KP + CF + f];
        const int d = load_dir(dir, U8, row, D);
#pragma unroll
        for (int q = 0; q < 8; ++q) acce[q] += (q == d) ? ge : 0.f;
      }
    }
  }
  // sum over the 16 row lanes: lanes (e & 3) of a wave by shuffles, then the 4 waves through LDS
#pragma unroll
  for (int k = 0; k < 27; ++k) { acc[k] += __shfl_xor(acc[k], 16); acc[k] += __shfl_xor(acc[k], 32); }
  accb += __shfl_xor(accb, 16); accb += __shfl_xor(accb, 32);
#pragma unroll
  for (int d = 0; d < 8; ++d) { acce[d] += __shfl_xor(acce[d], 16); acce[d] += __shfl_xor(acce[d], 32); }
  const int wave = tid >> 6;
  __syncthreads();
  if ((tid & 63) < F16) {
    float* r = s_red + (wave * F16 + f) * 37;
#pragma unroll
    for (int k = 0; k < 27; ++k) r[k] = acc[k];
    r[27] = accb;
#pragma unroll
    for (int d = 0; d < 8; ++d) r[28 + d] = acce[d];
  }
  __syncthreads();
  float* out = partial + (size_t)blockIdx.x * FPART;
  for (int q = tid; q < FPART; q += FB) {
    int ff, slot;
    if (q < 432) { ff = q / 27; slot = q - ff * 27; }
    else if (q < 448) { ff = q - 432; slot = 27; }
    else { ff = (q - 448) >> 3; slot = 28 + ((q - 448) & 7); }
    float s = 0.f;
    if (ff < F16) s = ((s_red[(0 * F16 + ff) * 37 + slot] + s_red[(1 * F16 + ff) * 37 + slot]) + s_red[(2 * F16 + ff) * 37 + slot]) +
                      s_red[(3 * F16 + ff) * 37 + slot];
    out[q] = (q >= 448 && ff >= 8) ? 0.f : s;
  }
}

// ---- heads ----------------------------------------------------------------------------------------------------------------
constexpr int HR = 128;            // rows per tile = threads per workgroup
constexpr int HPW = 64;            // width of the stacked first layers (32 actor + 32 critic)
constexpr int TP = 65;             // LDS pitch of one row of the tile (conflict-free: consecutive rows, consecutive banks)
constexpr int MAXNA = 8;
constexpr int HPART = 384;         // partial row of the heads backward: [0,256) dW_a2[k][i] | [256,264) db_a2 | [264,296) dW_c2 | 296 db_c2 | [304,368) db1

__device__ __forceinline__ void tile_load(float* s, const float* __restrict__ g, long long row0, int nr, int tid) {
  const float4* g4 = reinterpret_cast<const float4*>(g + row0 * HPW);
  for (int q = tid; q < nr * (HPW / 4); q += HR) {
    const int r = q >> 4, c4 = q & 15;
    const float4 v = g4[q];
    float* d = s + r * TP + c4 * 4;
    d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
  }
}
__device__ __forceinline__ void tile_store(const float* s, float* __restrict__ g, long long row0, int nr, int tid) {
  float4* g4 = reinterpret_cast<float4*>(g + row0 * HPW);
  for (int q = tid; q < nr * (HPW / 4); q += HR) {
    const int r = q >> 4, c4 = q & 15;
    const float* d = s + r * TP + c4 * 4;
    g4[q] = make_float4(d[0], d[1], d[2], d[3]);
  }
}

// pre: [R,64] = h @ [actor.0.0.weight ; critic.0.0.weight]^T (no bias) in, tanh(pre + bias) out (saved for the backward).
// FixedCategorical (models/distributions.py:18-56): logits - logsumexp; log_probs(action); entropy = -sum p * logp.
__global__ void __launch_bounds__(HR) ppo_heads_fwd_kernel(DcdStudentNet n, float* __restrict__ pre, const int64_t* __restrict__ actions,
                                                           float* __restrict__ logp_all, float* __restrict__ value, float* __restrict__ logp_act,
                                                           float* __restrict__ ent, long long R) {
  __shared__ float s_t[HR * TP];
  __shared__ float s_w2[MAXNA * 32 + MAXNA + 32 + 1 + HPW];
  const int tid = threadIdx.x, NA = n.NA;
  const long long row0 = (long long)blockIdx.x * HR;
  const int nr = (int)min((long long)HR, R - row0);
  tile_load(s_t, pre, row0, nr, tid);
  float* w_a2 = s_w2; float* b_a2 = s_w2 + MAXNA * 32; float* w_c2 = b_a2 + MAXNA; float* b_c2 = w_c2 + 32; float* b1 = b_c2 + 1;
  for (int q = tid; q < NA * 32; q += HR) w_a2[q] = n.actor_w2[q];
  if (tid < NA) b_a2[tid] = n.actor_b2[tid];
  if (tid < 32) { w_c2[tid] = n.critic_w2[tid]; b1[tid] = n.actor_b1[tid]; b1[32 + tid] = n.critic_b1[tid]; }
  if (tid == 0) b_c2[0] = n.critic_b2[0];
  __syncthreads();
  if (tid < nr) {
    float* t = s_t + tid * TP;
    float l[MAXNA];
#pragma unroll
    for (int k = 0; k < MAXNA; ++k) l[k] = k < NA ? b_a2[k] : -INFINITY;
    float v = b_c2[0];
#pragma unroll 8
    for (int i = 0; i < 32; ++i) {
      const float a = tanhf(t[i] + b1[i]), c = tanhf(t[32 + i] + b1[32 + i]);
      t[i] = a; t[32 + i] = c;
#pragma unroll
      for (int k = 0; k < MAXNA; ++k) if (k < NA) l[k] = fmaf(w_a2[k * 32 + i], a, l[k]);
      v = fmaf(w_c2[i], c, v);
    }
    float m = l[0];
#pragma unroll
    for (int k = 1; k < MAXNA; ++k) m = fmaxf(m, l[k]);
    float se = 0.f;
#pragma unroll
    for (int k = 0; k < MAXNA; ++k) if (k < NA) se += expf(l[k] - m);
    const float lse = m + logf(se);
    const long long row = row0 + tid;
    int a = (int)actions[row];
    a = a < 0 ? 0 : (a >= NA ? NA - 1 : a);
    float h = 0.f, la = 0.f;
    float lp[MAXNA];
#pragma unroll
    for (int k = 0; k < MAXNA; ++k) {
      lp[k] = k < NA ? l[k] - lse : 0.f;
      if (k < NA) { h -= expf(lp[k]) * lp[k]; la = (k == a) ? lp[k] : la; }
    }
    float4* o = reinterpret_cast<float4*>(logp_all + row * MAXNA);
    o[0] = make_float4(lp[0], lp[1], lp[2], lp[3]); o[1] = make_float4(lp[4], lp[5], lp[6], lp[7]);
    value[row] = v; logp_act[row] = la; ent[row] = h;
  }
  __syncthreads();
  tile_store(s_t, pre, row0, nr, tid);
}

// act: [R,64] tanh outputs in, d pre-activations out.  dvalue / dlogp / dent: upstream gradients per row.
// Per tile of 128 rows: (1) thread <-> row: d logits, dvalue -> LDS; (2) thread <-> accumulator: second-layer weight / bias
// gradients from (d logits, tanh outputs), summed in row order; (3) thread <-> row: the tile's tanh outputs are replaced by the
// d pre-activations; (4) thread <-> accumulator: first-layer bias gradients; the tile goes back to `act`.
__global__ void __launch_bounds__(HR) ppo_heads_bwd_kernel(DcdStudentNet n, float* __restrict__ act, const float* __restrict__ logp_all,
                                                           const int64_t* __restrict__ actions, const float* __restrict__ dvalue,
                                                           const float* __restrict__ dlogp, const float* __restrict__ dent, long long R,
                                                           int rows_per_wg, float* __restrict__ partial) {
  __shared__ float s_t[HR * TP];            // tanh outputs of the tile, then its d pre-activations
  __shared__ float s_dl[HR * 9];            // d logits [8] + dvalue per row
  __shared__ float s_w2[MAXNA * 32 + 32];
  const int tid = threadIdx.x, NA = n.NA;
  float* w_a2 = s_w2; float* w_c2 = s_w2 + MAXNA * 32;
  for (int q = tid; q < NA * 32; q += HR) w_a2[q] = n.actor_w2[q];
  if (tid < 32) w_c2[tid] = n.critic_w2[tid];
  const long long wg0 = (long long)blockIdx.x * rows_per_wg, wg1 = min(R, wg0 + rows_per_wg);
  float acc[3] = {0.f, 0.f, 0.f};            // accumulators tid, tid + 128, tid + 256 of the partial row
  for (long long row0 = wg0; row0 < wg1; row0 += HR) {
    const int nr = (int)min((long long)HR, wg1 - row0);
    __syncthreads();
    tile_load(s_t, act, row0, nr, tid);
    float dl[MAXNA], gv = 0.f;
#pragma unroll
    for (int k = 0; k < MAXNA; ++k) dl[k] = 0.f;
    if (tid < nr) {
      const long long row = row0 + tid;
      const float4* lp4 = reinterpret_cast<const float4*>(logp_all + row * MAXNA);
      const float4 q0 = lp4[0], q1 = lp4[1];
      const float lp[MAXNA] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
      int a = (int)actions[row];
      a = a < 0 ? 0 : (a >= NA ? NA - 1 : a);
      const float gl = dlogp ? dlogp[row] : 0.f, ge = dent ? dent[row] : 0.f;
      gv = dvalue ? dvalue[row] : 0.f;
      float p[MAXNA], h = 0.f;
#pragma unroll
      for (int k = 0; k < MAXNA; ++k) { p[k] = k < NA ? expf(lp[k]) : 0.f; h -= p[k] * lp[k]; }
#pragma unroll
      for (int k = 0; k < MAXNA; ++k) {        // d log-softmax: gl (1[k=a] - p_k);  d entropy: -p_k (logp_k + H)
        dl[k] = k < NA ? gl * ((k == a ? 1.f : 0.f) - p[k]) - ge * p[k] * (lp[k] + h) : 0.f;
        s_dl[tid * 9 + k] = dl[k];
      }
      s_dl[tid * 9 + 8] = gv;
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < 3; ++s) {              // (2) second-layer gradients: sums over the tile's rows in row order
      const int q = tid + s * HR;
      float a = acc[s];
      if (q < 256) { const int k = q >> 5, i = q & 31; for (int r = 0; r < nr; ++r) a = fmaf(s_dl[r * 9 + k], s_t[r * TP + i], a); }
      else if (q < 264) { const int k = q - 256; for (int r = 0; r < nr; ++r) a += s_dl[r * 9 + k]; }
      else if (q < 296) { const int i = q - 264; for (int r = 0; r < nr; ++r) a = fmaf(s_dl[r * 9 + 8], s_t[r * TP + 32 + i], a); }
      else if (q == 296) { for (int r = 0; r < nr; ++r) a += s_dl[r * 9 + 8]; }
      acc[s] = a;
    }
    __syncthreads();
    if (tid < nr) {                            // (3) d pre-activations over the tanh outputs
      float* t = s_t + tid * TP;
#pragma unroll 8
      for (int i = 0; i < 32; ++i) {
        float da = 0.f;
#pragma unroll
        for (int k = 0; k < MAXNA; ++k) if (k < NA) da = fmaf(dl[k], w_a2[k * 32 + i], da);
        const float av = t[i], cv = t[32 + i];
        t[i] = da * (1.f - av * av);
        t[32 + i] = gv * w_c2[i] * (1.f - cv * cv);
      }
    }
    __syncthreads();
    {                                          // (4) first-layer bias gradients (accumulators 304 .. 367 live in slot 2: 304 - 256 = 48)
      const int q = tid + 2 * HR;
      if (q >= 304 && q < 368) { const int i = q - 304; float a = acc[2]; for (int r = 0; r < nr; ++r) a += s_t[r * TP + i]; acc[2] = a; }
    }
    tile_store(s_t, act, row0, nr, tid);
  }
  float* out = partial + (size_t)blockIdx.x * HPART;
#pragma unroll
  for (int s = 0; s < 3; ++s) out[tid + s * HR] = acc[s];
}

int check_student(const DcdStudentNet* n, const char* who) {
  if (!n) return dcd::fail("%s: NULL net", who);
  if (n->V != V5 || n->F != F16) return dcd::fail("%s: built for the registered student (5x5 view, 16 filters), got V=%d F=%d", who, n->V, n->F);
  if (n->E < 1 || n->E > 8 || n->D < 1 || n->D > 8) return dcd::fail("%s: scalar embedding must be at most 8 x 8, got %d x %d", who, n->E, n->D);
  return 0;
}
int check_heads(const DcdStudentNet* n, const char* who) {
  if (!n) return dcd::fail("%s: NULL net", who);
  if (n->A1 != 32 || n->C1 != 32 || n->NA < 1 || n->NA > MAXNA)
    return dcd::fail("%s: built for 32-unit hidden layers and at most %d actions, got A1=%d C1=%d NA=%d", who, MAXNA, n->A1, n->C1, n->NA);
  if (!n->actor_w2 || !n->actor_b2 || !n->critic_w2 || !n->critic_b2 || !n->actor_b1 || !n->critic_b1) return dcd::fail("%s: head weights not set", who);
  return 0;
}

}  // namespace

extern "C" {

int dcd_ppo_features_partial_floats(void) { return FPART; }
int dcd_ppo_heads_partial_floats(void) { return HPART; }

int dcd_ppo_features_fwd(const DcdStudentNet* net, const void* image, const void* direction, int32_t obs_u8, float* x, int32_t x_pitch,
                         int64_t R, void* stream) {
  if (check_student(net, "dcd_ppo_features_fwd")) return 1;
  if (!image || !direction || !x || R <= 0 || !net->conv_w || !net->conv_b || !net->embed_w || !net->embed_b)
    return dcd::fail("dcd_ppo_features_fwd: NULL pointer or empty batch");
  if (x_pitch < CF + net->E + 1) return dcd::fail("dcd_ppo_features_fwd: x_pitch %d below %d (features + embedding + the constant 1)", x_pitch, CF + net->E + 1);
  const dim3 grid((unsigned)((R + FR - 1) / FR));
  if (obs_u8) hipLaunchKernelGGL(ppo_features_fwd_kernel<true>, grid, dim3(FB), 0, (hipStream_t)stream, *net, image, direction, x, x_pitch, (long long)R);
  else hipLaunchKernelGGL(ppo_features_fwd_kernel<false>, grid, dim3(FB), 0, (hipStream_t)stream, *net, image, direction, x, x_pitch, (long long)R);
  if (hipGetLastError() != hipSuccess) return dcd::fail("dcd_ppo_features_fwd: launch failed");
  return 0;
}

int dcd_ppo_features_bwd(const DcdStudentNet* net, const void* image, const void* direction, int32_t obs_u8, const float* x, const float* dx,
                         int32_t x_pitch, int64_t R, int32_t rows_per_wg, float* partial, void* stream) {
  if (check_student(net, "dcd_ppo_features_bwd")) return 1;
  if (!image || !direction || !x || !dx || !partial || R <= 0) return dcd::fail("dcd_ppo_features_bwd: NULL pointer or empty batch");
  if (rows_per_wg < FR || rows_per_wg % FR) return dcd::fail("dcd_ppo_features_bwd: rows_per_wg must be a positive multiple of %d", FR);
  if (x_pitch < CF + net->E + 1) return dcd::fail("dcd_ppo_features_bwd: x_pitch too small");
  const dim3 grid((unsigned)((R + rows_per_wg - 1) / rows_per_wg));
  if (obs_u8) hipLaunchKernelGGL(ppo_features_bwd_kernel<true>, grid, dim3(FB), 0, (hipStream_t)stream, *net, image, direction, x, dx, x_pitch, (long long)R, rows_per_wg, partial);
  else hipLaunchKernelGGL(ppo_features_bwd_kernel<false>, grid, dim3(FB), 0, (hipStream_t)stream, *net, image, direction, x, dx, x_pitch, (long long)R, rows_per_wg, partial);
  if (hipGetLastError() != hipSuccess) return dcd::fail("dcd_ppo_features_bwd: launch failed");
  return 0;
}

int dcd_ppo_heads_fwd(const DcdStudentNet* net, float* pre_act, const int64_t* actions, float* logp_all, float* value, float* logp_act,
                      float* entropy, int64_t R, void* stream) {
  if (check_heads(net, "dcd_ppo_heads_fwd")) return 1;
  if (!pre_act || !actions || !logp_all || !value || !logp_act || !entropy || R <= 0) return dcd::fail("dcd_ppo_heads_fwd: NULL pointer or empty batch");
  if (((uintptr_t)pre_act | (uintptr_t)logp_all) & 15) return dcd::fail("dcd_ppo_heads_fwd: pre_act / logp_all must be 16-byte aligned");
  hipLaunchKernelGGL(ppo_heads_fwd_kernel, dim3((unsigned)((R + HR - 1) / HR)), dim3(HR), 0, (hipStream_t)stream, *net, pre_act, actions, logp_all,
                     value, logp_act, entropy, (long long)R);
  if (hipGetLastError() != hipSuccess) return dcd::fail("dcd_ppo_heads_fwd: launch failed");
  return 0;
}

int dcd_ppo_heads_bwd(const DcdStudentNet* net, float* act_dpre, const float* logp_all, const int64_t* actions, const float* dvalue,
                      const float* dlogp, const float* dentropy, int64_t R, int32_t rows_per_wg, float* partial, void* stream) {
  if (check_heads(net, "dcd_ppo_heads_bwd")) return 1;
  if (!act_dpre || !logp_all || !actions || !partial || R <= 0) return dcd::fail("dcd_ppo_heads_bwd: NULL pointer or empty batch");
  if (rows_per_wg < HR || rows_per_wg % HR) return dcd::fail("dcd_ppo_heads_bwd: rows_per_wg must be a positive multiple of %d", HR);
  if (((uintptr_t)act_dpre | (uintptr_t)logp_all) & 15) return dcd::fail("dcd_ppo_heads_bwd: act_dpre / logp_all must be 16-byte aligned");
  hipLaunchKernelGGL(ppo_heads_bwd_kernel, dim3((unsigned)((R + rows_per_wg - 1) / rows_per_wg)), dim3(HR), 0, (hipStream_t)stream, *net, act_dpre,
                     logp_all, actions, dvalue, dlogp, dentropy, (long long)R, rows_per_wg, partial);
  if (hipGetLastError() != hipSuccess) return dcd::fail("dcd_ppo_heads_bwd: launch failed");
  return 0;
}

}  // extern "C"
