// mg_kernels.hip — MultiGrid engine kernels for gfx950 (MI355X) + their C ABI (include/dcd_hip.h).
//
// Mapping: one env per lane, one 64-lane wavefront per workgroup.  A wave owns 64 consecutive
// 256-byte env records = 16 KiB of contiguous HBM: it pulls them with 16 fully coalesced
// 1-KiB wave loads (16 B per lane) into LDS (row pitch 272 B: 16-B aligned for ds_write_b128,
// 68 dwords so lanes spread over banks), then every lane works on ITS env purely out of LDS —
// the front cell, the V×V neighbourhood gather and all writes are LDS traffic, so a step costs
// exactly one dependent HBM round trip.  Results leave through LDS staging so that every global
// store instruction of a wave is contiguous (obs tensors are [B, ...] row-major).
// The hot step writes back only the 16-byte dynamic state plus the <=2 cells it changed; rare
// paths (episode end, respawn, resets) write the whole record back.
//
// HBM bytes per env-step: read 256 (record) + 8 (action); write 16 (dyn)
// + <=2 (cells) + obs 75+1 (u8) and/or 300+4 (f32) + reward 4 + done/info 2 + masks 12.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>
#include <hip/hip_ext.h>
#include "mg_env.h"
#include "dcd_internal.h"

using namespace dcd;

namespace {

constexpr int WAVE = 64;
// threads per workgroup: wave 0 runs the per-env logic, ALL waves stage records and emit observations.
// Small batches (EPB 16/32: latency-bound, one workgroup per CU) use 512 threads so the fused render is one
// slot per thread; large batches (EPB 64: bandwidth-bound) use 256 to keep more workgroups resident.
template <int EPB> struct Threads { static constexpr int value = EPB == 64 ? 256 : 512; };
constexpr int LP = 272;            // LDS pitch of one env record
constexpr int MAX_OP = 152;        // staged u8 obs per env: V*V*3 + dir, padded (V<=7 -> 148)

__device__ __forceinline__ int obs_pitch(int V) { return (V * V * 3 + 1 + 3) & ~3; }

// Streaming (`nt`) stores for the step kernel's outputs in the BANDWIDTH regime only (64 envs per workgroup, B >= 32 768):
// nothing in the launch reads them back, and they drain to memory while other workgroups still compute instead of sitting
// dirty in the XCD's L2 until the end-of-kernel write-back: -3 % at B = 1M.  At B = 4096 the same stores made the pass
// 5.5 % SLOWER (1.518 -> 1.606 ms, same-box A/B, profiles/r02_ab_nt_stores.log): there the launch is latency-bound and a
// write-through store costs the issuing wave more than the deferred write-back costs the launch.
template <bool NTS, class T> __device__ __forceinline__ void st_out(T* p, T v) {
  if (NTS) __builtin_nontemporal_store(v, p);
  else *p = v;
}
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef DCD_REC uint8_t lds_u8;      // address_space(3) in the device pass, plain in the host pass
__device__ __forceinline__ lds_u8* as_lds(uint8_t* p) { return (lds_u8*)p; }

// LUTs shared by the emission code, indexed by packed cell code: the 3 encoding bytes and their /10 floats
struct Luts { uint32_t enc[64]; float f[3][64]; float div10[16]; };
template <bool FLOATS = true>
__device__ __forceinline__ void init_luts(Luts& l, int tid) {
  if (tid < 64) {
    const uint32_t t = tid < C_END ? enc3_flat((uint32_t)tid) : 0u;
    l.enc[tid] = t;
    if (FLOATS) {                                     // only the fp32 emission reads them
      l.f[0][tid] = div10f(t & 15); l.f[1][tid] = div10f((t >> 8) & 15); l.f[2][tid] = div10f((t >> 16) & 15);
      if (tid < 16) l.div10[tid] = div10f((uint32_t)tid);
    }
  }
}

// ---- workgroup-cooperative record staging: n rows x 16 chunks of 16 B over NT threads ---------
template <int ROWS, int NT>
struct StageBuf { uint4 v[(ROWS * 16 + NT - 1) / NT]; };
template <int ROWS, int NT, class RowToEnv>
__device__ __forceinline__ void stage_load(StageBuf<ROWS, NT>& sb, const uint8_t* __restrict__ rec, RowToEnv env_of, int n, int tid) {
#pragma unroll
  for (int i = 0; i < (ROWS * 16 + NT - 1) / NT; ++i) {
    const int q = i * NT + tid, row = q >> 4, c = q & 15;
    sb.v[i] = make_uint4(0, 0, 0, 0);
    if (row < n) sb.v[i] = *reinterpret_cast<const uint4*>(rec + (size_t)env_of(row) * REC_BYTES + c * 16);
  }
}
template <int ROWS, int NT>
__device__ __forceinline__ void stage_commit(const StageBuf<ROWS, NT>& sb, uint8_t* lds, int n, int tid) {
#pragma unroll
  for (int i = 0; i < (ROWS * 16 + NT - 1) / NT; ++i) {
    const int q = i * NT + tid, row = q >> 4, c = q & 15;
    if (row < n) *reinterpret_cast<uint4*>(lds + row * LP + c * 16) = sb.v[i];
  }
}
template <int NT, class RowToEnv>
__device__ __forceinline__ void stage_out(const uint8_t* lds, uint8_t* __restrict__ rec, RowToEnv env_of, int n, int tid) {
  for (int q = tid; q < n * 16; q += NT) {
    const int row = q >> 4, c = q & 15;
    *reinterpret_cast<uint4*>(rec + (size_t)env_of(row) * REC_BYTES + c * 16) = *reinterpret_cast<const uint4*>(lds + row * LP + c * 16);
  }
}

// ---- fused render + emit of the student observation, cooperative over the workgroup -----------
// One (env row, view cell) pair per thread-iteration: world cell through the closed-form map of
// multigrid.py:1080-1093, packed code from the LDS record, channel values from the code LUTs, stored
// as fp32 CHW (/10) and/or u8 [vx][vy][c].  Consecutive threads hit consecutive view cells of the same
// env, so each channel's store is a run of contiguous floats.  Semantics == dcd::render_view.
// Opaque walls (see_through == 0): the TPE threads of an env sit in one wave, so the env's see-through bit board
// is a slice of a wave ballot; every thread then runs Grid.process_vis on it (dcd::process_vis_mask) and drops
// its cell if unseen — no LDS, no barrier.
// (Tried for the u8 emission and dropped after a same-box A/B: staging the bytes in LDS and writing the workgroup's block
// with 16-byte stores — one more barrier and LDS round trip cost more than the three single-byte stores per view cell:
// 4.40 -> 4.58 us per launch at 8192 envs, 134 -> 144 us at 1M.)
template <int V, int ROWS, int NT>
__device__ __forceinline__ void render_emit(const uint8_t* lds, const Luts& lut, const DcdStudentObs& o, size_t out_row0,
                                            int n, int S, int tid, int see_through) {
  // TPE threads per env (power of two >= V*V): env = slot >> log2(TPE), view cell = slot & (TPE-1) — shifts
  // and masks instead of divisions; all offsets are 32-bit from per-workgroup base pointers.
  constexpr int VV = V * V, H = V / 2, TPE = VV <= 16 ? 16 : (VV <= 32 ? 32 : 64), ITERS = (ROWS * TPE + NT - 1) / NT;
  constexpr bool NTS = ROWS == 64;
  float* __restrict__ f32 = o.image_f32 ? o.image_f32 + out_row0 * (3 * VV) : nullptr;
  uint8_t* __restrict__ u8 = o.image_u8 ? o.image_u8 + out_row0 * (3 * VV) : nullptr;
  if (f32 || u8) {
    int code[ITERS];
#pragma unroll
    for (int i = 0; i < ITERS; ++i) {                         // phase 1: all LDS gathers in flight together
      const int q = i * NT + tid, row = q / TPE, ci = q & (TPE - 1), vx = ci / V, vy = ci - vx * V;
      code[i] = -1;
      if (row < n && ci < VV) {
        const uint32_t dyn = *reinterpret_cast<const uint32_t*>(lds + row * LP + DYN);
        const int ax = dyn & 255, ay = (dyn >> 8) & 255, d = (dyn >> 16) & 255;
        const int fx = (d == 0) - (d == 2), fy = (d == 1) - (d == 3);
        const int wx = ax + fx * (V - 1 - vy) - fy * (vx - H), wy = ay + fy * (V - 1 - vy) + fx * (vx - H);
        const bool own = (vx == H && vy == V - 1), oob = (wx < 0 || wx >= S || wy < 0 || wy >= S);
        int c = lds[row * LP + ((own || oob) ? DYN + 3 : wy * ROW + wx)];        // own cell -> carrying byte
        if (oob && !own) c = C_WALL_GREY;
        else if (!own && c >= C_AGENT0 && c < C_DOOR0) c = C_AGENT0 + (((c - C_AGENT0) - (d + 1)) & 3);
        code[i] = (ax == NOPOS) ? 64 : c;                      // 64: no agent placed yet -> zero observation
      }
    }
    if (DCD_RARE(!see_through)) {
#pragma unroll
      for (int i = 0; i < ITERS; ++i) {
        const int q = i * NT + tid, ci = q & (TPE - 1), vx = ci / V, vy = ci - vx * V;
        const bool tr = code[i] >= 0 && code[i] != 64 && ((vx == H && vy == V - 1) || see_behind((uint8_t)code[i]));
        const unsigned long long bal = __ballot(tr);
        const unsigned long long mine = TPE == 64 ? bal : ((bal >> ((tid & 63) & ~(TPE - 1))) & ((1ull << (TPE & 63)) - 1ull));
        const unsigned long long vis = process_vis_mask(mine, V);
        if (code[i] >= 0 && !((vis >> ci) & 1)) code[i] = 64;
      }
    }
    __builtin_amdgcn_sched_barrier(0);                       // keep the gathers batched ahead of the LUT/store phase
#pragma unroll
    for (int i = 0; i < ITERS; ++i) {                         // phase 2: LUT + stores
      const int q = i * NT + tid, row = q / TPE, ci = q & (TPE - 1);
      if (code[i] < 0) continue;
      const bool none = code[i] == 64;
      const int c = none ? 0 : code[i];
      const unsigned off = (unsigned)(row * (3 * VV) + ci);
      if (f32) {
        const float f0 = lut.f[0][c], f1 = lut.f[1][c], f2 = lut.f[2][c];
        st_out<NTS>(&f32[off], none ? 0.f : f0); st_out<NTS>(&f32[off + VV], none ? 0.f : f1); st_out<NTS>(&f32[off + 2 * VV], none ? 0.f : f2);
      }
      if (u8) {
        const uint32_t t = none ? 0u : lut.enc[c];
        const unsigned uo = (unsigned)(row * (3 * VV) + ci * 3);
        st_out<NTS>(&u8[uo], (uint8_t)t); st_out<NTS>(&u8[uo + 1], (uint8_t)(t >> 8)); st_out<NTS>(&u8[uo + 2], (uint8_t)(t >> 16));
      }
    }
  }
  if (tid < n) {
    const uint32_t dyn = *reinterpret_cast<const uint32_t*>(lds + tid * LP + DYN);
    const int d = ((dyn & 255) == NOPOS) ? 0 : ((dyn >> 16) & 255);
    if (o.dir_u8) st_out<(ROWS == 64)>(&o.dir_u8[out_row0 + tid], (uint8_t)d);
    if (o.dir_f32) st_out<(ROWS == 64)>(&o.dir_f32[out_row0 + tid], (float)d);
  }
}
template <int ROWS, int NT>
__device__ __forceinline__ void render_emit_v(const uint8_t* lds, const Luts& lut, const DcdStudentObs& o, size_t out_row0,
                                              int n, const Params& p, int tid) {
  if (p.V == 5) render_emit<5, ROWS, NT>(lds, lut, o, out_row0, n, p.S, tid, p.see_through_walls);
  else if (p.V == 7) render_emit<7, ROWS, NT>(lds, lut, o, out_row0, n, p.S, tid, p.see_through_walls);
  else render_emit<3, ROWS, NT>(lds, lut, o, out_row0, n, p.S, tid, p.see_through_walls);
}

// ---- emission of per-lane staged observations (terminal obs: rare) ---------------------------
// sobs: [n][OP] bytes ([vx][vy][c] then dir); rows with valid[row]==0 are left untouched.
template <int NT>
__device__ __forceinline__ void emit_staged(const uint8_t* sobs, const Luts& lut, const DcdStudentObs& o,
                                            size_t out_row0, int n, int V, int tid, const uint8_t* valid) {
  const int VV = V * V, E = VV * 3, OP = obs_pitch(V);
  if (o.image_u8) for (int f = tid; f < n * E; f += NT) {
    const int row = f / E, el = f - row * E;
    if (valid[row]) o.image_u8[out_row0 * E + f] = sobs[row * OP + el];
  }
  if (o.image_f32) for (int f = tid; f < n * E; f += NT) {
    const int row = f / E, el = f - row * E, c = el / VV, rem = el - c * VV;
    if (valid[row]) o.image_f32[out_row0 * E + f] = lut.div10[sobs[row * OP + rem * 3 + c] & 15];
  }
  if (tid < n && valid[tid]) {
    const uint8_t d = sobs[tid * OP + E];
    if (o.dir_u8) o.dir_u8[out_row0 + tid] = d;
    if (o.dir_f32) o.dir_f32[out_row0 + tid] = (float)d;
  }
}

__device__ __forceinline__ void raise(uint32_t* errs, uint32_t bits, int env) {
  if (bits) { atomicOr(&errs[0], bits); atomicMin(reinterpret_cast<int*>(&errs[1]), env); }
}

// =================================================================================== hot step
struct StepArgs {
  uint8_t* rec; const uint32_t* seed; Metrics* met; uint32_t* errs;
  const int64_t* actions; int flags;
  DcdStepOut out[2];        // [1]: the second replica's outputs (dcd_mg_step_pair)
  // fused random-action mode (dcd_mg_step_random)
  int K; uint32_t action_seed, t0; int64_t env_index_base; unsigned long long* n_episodes;
  unsigned long long* stamps;   // diagnostic build only (-DDCD_STAMPS): [grid][8] s_memtime per phase
};

#ifdef DCD_STAMPS
#define STAMP(i) do { if (tid == 0 && a.stamps) a.stamps[blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define STAMP(i) do {} while (0)
#endif

// EPB = envs per workgroup.  Threads [0,EPB) of wave 0 run the per-env logic; ALL 256 threads share
// staging and the fused render/emit.  Small batches use EPB=16 so that B=4096 still spreads over all
// 256 CUs; large batches use EPB=64 (every lane of the logic wave busy).
// per-env results handed from the logic lane to the duty waves through LDS
struct StepRes { float reward, ep_ret; int ep_len; uint32_t flags; uint32_t doff; };   // flags: done | info<<8 | full_dirty<<16 | ctr_dirty<<17 | nd<<20

// Workgroup = NT worker threads + 2 duty waves.  Thread roles:
//   wave 0, lanes [0,EPB)      per-env logic (one env per lane), result -> LDS
//   all NT worker threads      record staging before, fused render+emit after the logic
//   duty wave A (tid NT+lane)  per-env scalar outputs (reward, done, info, episode stats, masks) -> global
//   duty wave B (tid NT+64+lane) record write-back: 16-byte dynamic state + the <=2 cells changed (or the whole record)
// so the logic wave's critical path holds no global stores at all.
// RR: the launch may auto-reset finished envs to RANDOM levels (DR without PLR).  A compile-time flag because that
// path inlines reset_random + the BFS metrics: the other variant is a third of the code (instruction-cache misses
// on cold paths were a visible part of the slowest workgroup's time).
// The first four arguments repeat a.rec / a.seed / a.actions / p.B as plain scalars: with
// -amdgpu-kernarg-preload-count they arrive in SGPRs at wave launch (struct arguments are not preloaded), so the
// record / seed / action loads are issued without waiting for a kernarg fetch first.
// PAIR: the launch steps TWO replicas of the batch (dcd_mg_step_pair: the protagonist's and the antagonist's copy of the
// levels, records / seeds / metrics contiguous: replica 1 starts at row B).  Workgroups [0, bps) serve replica 0,
// [bps, 2 bps) replica 1; a replica has its own action array and output set, everything else is indexed by record row.
// OBS8: only the raw u8 observation is emitted (obs.image_u8 / dir_u8 and, for truncations, term_obs.image_u8): the fp32
// tables of the emission are not built.
template <bool RANDOM, int EPB, bool RR, bool PAIR = false, bool OBS8 = false>
__global__ void __launch_bounds__(Threads<EPB>::value + 2 * WAVE) mg_step_kernel(const uint8_t* __restrict__ rec_in,
                                                                                 const uint32_t* __restrict__ seed_in,
                                                                                 const int64_t* __restrict__ actions_in, int B_in,
                                                                                 const int64_t* __restrict__ actions1_in,
                                                                                 Params p, StepArgs a) {
  constexpr int NT = Threads<EPB>::value;
  __shared__ __attribute__((aligned(16))) uint8_t lds[EPB * LP];
  __shared__ __attribute__((aligned(16))) uint8_t sterm[EPB * MAX_OP];
  __shared__ StepRes res[EPB];
  __shared__ uint8_t term_valid[EPB];
  __shared__ int any_term;
  __shared__ Luts lut;
  const int tid = threadIdx.x;
  int set = 0, blk = blockIdx.x;
  if (PAIR) { const int bps = (B_in + EPB - 1) / EPB; if (blk >= bps) { set = 1; blk -= bps; } }
  const int env0 = blk * EPB;                        // first env of this workgroup within its replica
  const int row0 = PAIR ? set * B_in + env0 : env0;  // ... and its record row
  const int n = min(EPB, B_in - env0), env = env0 + tid, row = row0 + tid;
  const DcdStepOut& outs = a.out[PAIR ? set : 0];
  STAMP(0);
  // issue every independent global load up front: records, action, seed (one HBM round trip in all)
  StageBuf<EPB, NT> sb;
  if (tid < NT) stage_load<EPB, NT>(sb, rec_in, [row0](int r) { return row0 + r; }, n, tid);
  int64_t act0 = 0; uint32_t seed = 0;
  if (tid < n) { seed = seed_in[row]; if (!RANDOM) act0 = ((PAIR && set) ? actions1_in : actions_in)[env]; term_valid[tid] = 0; }
  const int OP = obs_pitch(p.V);
  if (tid == 0) any_term = 0;
  init_luts<!OBS8>(lut, tid);
  STAMP(1);
  if (tid < NT) stage_commit<EPB, NT>(sb, lds, n, tid);
  __syncthreads();
  STAMP(2);

  unsigned long long episodes = 0;
  if (tid < WAVE) {                                  // the logic wave: lane <-> env for lanes < n, all 64 lanes for rare cooperative work
    const bool live = tid < n;
    const int step_flags = RR ? a.flags : (a.flags & ~DCD_STEP_RESET_RANDOM);
    Env e = make_env(as_lds(lds + (live ? tid : 0) * LP), seed);
    e.cache_side();
    Metrics m = {0, 0, 0, 0}; bool mdirty = false;
    if (RR && live && (step_flags & DCD_STEP_RESET_RANDOM)) m = a.met[row];
    StepResult r = {};
    const int K = RANDOM ? a.K : 1;
    for (int k = 0; k < K; ++k) {
      int64_t act = act0;
      if (RANDOM) {
        uint32_t o[4];
        philox4x32_10(a.t0 + k, 0, 0, 0, a.action_seed, (uint32_t)(a.env_index_base + env), o);
        act = (int64_t)mulhi32(o[0], 7u);
      }
      const int fl = (RANDOM && k != K - 1) ? (step_flags & ~DCD_STEP_LAST) : step_flags;
      StepPre sp = {};
      if (live) sp = env_step_pre(e, p, act);
      // pre-reset observation of envs whose time limit fired (info['truncated_obs']): one view cell per lane
      unsigned long long tm = __ballot(live && sp.have_term);
      if (DCD_RARE(tm != 0)) {
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");       // the owners' cell writes are visible to the wave
        const int V = p.V, VV = V * V;
        while (tm) {
          const int src = __builtin_ctzll(tm); tm &= tm - 1;
          const int ax = __shfl(e.ax, src), ay = __shfl(e.ay, src), ad = __shfl(e.adir, src), ca = __shfl(e.carrying, src);
          auto rec = as_lds(lds + src * LP);
          auto out = as_lds(sterm + src * OP);
          const int vx = tid / V, vy = tid - vx * V;
          const uint8_t cc = tid < VV ? view_cell_code(rec, p.S, V, ax, ay, ad, ca, vx, vy) : C_WALL_GREY;
          unsigned long long vis = ~0ull;
          if (!p.see_through_walls)
            vis = process_vis_mask(__ballot(tid < VV && ((vx == V / 2 && vy == V - 1) || see_behind(cc))), V);
          if (tid < VV) {
            const uint32_t t = ((vis >> tid) & 1) ? enc3(cc) : 0u;
            out[tid * 3] = (uint8_t)t; out[tid * 3 + 1] = (uint8_t)(t >> 8); out[tid * 3 + 2] = (uint8_t)(t >> 16);
          }
          if (tid == 0) out[VV * 3] = (uint8_t)ad;
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
      }
      if (live) {
        r = env_step_post(e, p, m, mdirty, fl, sp);
        if (RANDOM) episodes += (r.info & DCD_INFO_EPISODE) ? 1 : 0;
      }
    }
    if (live) {
      if (DCD_RARE(r.have_term)) { term_valid[tid] = 1; any_term = 1; }
      e.store_dyn();                                 // LDS copy feeds the render and the write-back wave
      StepRes sr;
      sr.reward = r.reward; sr.ep_ret = r.ep_ret; sr.ep_len = r.ep_len;
      sr.flags = (uint32_t)r.done | ((uint32_t)r.info << 8) | ((uint32_t)(e.full_dirty ? 1 : 0) << 16) | ((uint32_t)(e.ctr_dirty ? 1 : 0) << 17) | ((uint32_t)min(e.nd, 2) << 20);
      sr.doff = (uint32_t)e.doff0 | ((uint32_t)e.doff1 << 16);
      res[tid] = sr;
      if (RR && DCD_RARE(mdirty)) a.met[row] = m;
      if (DCD_RARE(e.err)) raise(a.errs, e.err, env);
    }
    STAMP(3);
  }
  __syncthreads();
  STAMP(4);
  if (tid < NT) {
    render_emit_v<EPB, NT>(lds, lut, outs.obs, (size_t)env0, n, p, tid);      // post-auto-reset observation
    STAMP(5);
    if (any_term) emit_staged<NT>(sterm, lut, outs.term_obs, (size_t)env0, n, p.V, tid, term_valid);
    STAMP(6);
  } else if (tid < NT + WAVE) {                      // duty wave A: per-env scalar outputs
    const int l = tid - NT, ev = env0 + l;
    if (l < n) {
      const StepRes sr = res[l];
      const uint32_t done = sr.flags & 255, info = (sr.flags >> 8) & 255;
      const DcdStepOut& o = outs;
      if (o.reward) st_out<(EPB == 64)>(&o.reward[ev], sr.reward);
      if (o.done) st_out<(EPB == 64)>(&o.done[ev], (uint8_t)done);
      if (o.info) st_out<(EPB == 64)>(&o.info[ev], (uint8_t)info);
      if (o.ep_ret) st_out<(EPB == 64)>(&o.ep_ret[ev], sr.ep_ret);
      if (o.ep_len) st_out<(EPB == 64)>(&o.ep_len[ev], (int32_t)sr.ep_len);
      if (o.masks) st_out<(EPB == 64)>(&o.masks[ev], done ? 0.f : 1.f);                                // runner :566-573
      if (o.bad_masks) st_out<(EPB == 64)>(&o.bad_masks[ev], (info & DCD_INFO_TRUNC_KEY) ? 0.f : 1.f);
      if (o.cliffhanger_masks) st_out<(EPB == 64)>(&o.cliffhanger_masks[ev], (info & DCD_INFO_CLIFFHANGER) ? 0.f : 1.f);
    }
  } else {                                           // duty wave B: record write-back
    const int l = tid - NT - WAVE;
    if (l < n) {
      const StepRes sr = res[l];
      uint8_t* g = a.rec + (size_t)(row0 + l) * REC_BYTES;
      const uint8_t* src = lds + l * LP;
      if ((sr.flags >> 16) & 1) {
        uint4 rows[16];
#pragma unroll
        for (int y = 0; y < 16; ++y) rows[y] = *reinterpret_cast<const uint4*>(src + y * ROW);
#pragma unroll
        for (int y = 0; y < 16; ++y) *reinterpret_cast<uint4*>(g + y * ROW) = rows[y];
      } else {
        const int nd = (sr.flags >> 20) & 3, o0 = sr.doff & 0xFFFF, o1 = sr.doff >> 16;
        const uint4 dyn = *reinterpret_cast<const uint4*>(src + DYN);
        const uint8_t c0 = src[o0], c1 = src[o1];
        if (nd > 0) st_out<(EPB == 64)>(&g[o0], c0);
        if (nd > 1) st_out<(EPB == 64)>(&g[o1], c1);
        if ((sr.flags >> 17) & 1) {                      // the draw counter moved (a goal was reached): side bytes of rows 8..11
#pragma unroll
          for (int k = 8; k < 12; ++k) g[k * ROW + SIDE] = src[k * ROW + SIDE];
        }
        st_out<(EPB == 64)>(reinterpret_cast<u32x4*>(g + DYN), u32x4{dyn.x, dyn.y, dyn.z, dyn.w});
      }
    }
  }
  if (RANDOM && a.n_episodes && tid < WAVE) {
    for (int off = 32; off > 0; off >>= 1) episodes += __shfl_down(episodes, off);
    if (tid == 0 && episodes) atomicAdd(a.n_episodes, episodes);
  }
}

// =================================================================================== level ops
enum LevelOp { OP_RESET = 0, OP_STEP_ADV, OP_RESET_AGENT, OP_RESET_RANDOM, OP_SET_ENCODING, OP_REPLAY, OP_MUTATE,
               OP_READ_ENC, OP_FIXED_LEVEL };

struct LevelArgs {
  uint8_t* rec; const uint32_t* seed; Metrics* met; uint32_t* errs;
  const int32_t* idx; int n_rows;                 // rows -> env (nullptr: identity over B)
  const int64_t* loc;                             // OP_STEP_ADV [B]; OP_REPLAY actions [n,max_len]
  const int32_t* lens; int max_len;               // OP_REPLAY
  const uint8_t* enc;                             // OP_SET_ENCODING [n,S,S,3]
  int num_edits;                                  // OP_MUTATE
  DcdAdvObs adv; uint8_t* adv_done;               // teacher outputs
  DcdStudentObs stu;                              // student outputs
  uint8_t* enc_out;                               // OP_READ_ENC [B,S,S,3]
};

// full-grid observation of the workgroup's n envs: u8 [x][y][c] and/or fp32 CHW /10, contiguous per
// workgroup (identity rows only).  SS = compile-time grid size (0: runtime p.S).
template <int SS, int NT>
__device__ __forceinline__ void emit_grid(const uint8_t* lds, const Luts& lut, uint8_t* u8, float* f32, size_t row0, int n,
                                          int S_rt, int tid) {
  const int S = SS ? SS : S_rt, G = S * S * 3, SQ = S * S;
  if (f32) {
#pragma unroll 4
    for (int f = tid; f < n * G; f += NT) {
      const int r = f / G, el = f - r * G, c = el / SQ, cell = el - c * SQ, x = cell / S, y = cell - x * S;
      f32[row0 * G + f] = lut.f[c][lds[r * LP + y * ROW + x]];
    }
  }
  if (u8) {
    // (one byte per store; assembling dwords from pairs of 24-bit LUT entries — a quarter of the stores and loop trips —
    // measured the same, streaming stores slower: of a teacher step's 6.0 us, 2.1 are this image's 2.8 MB, 0.35 random_z's
    // 0.8 MB, 3.2 the launch with no outputs at all — `tools/time_teacher_steps.py`, round 2)
#pragma unroll 4
    for (int f = tid; f < n * G; f += NT) {
      const int r = f / G, el = f - r * G, cell = el / 3, c = el - cell * 3, x = cell / S, y = cell - x * S;
      u8[row0 * G + f] = (uint8_t)(lut.enc[lds[r * LP + y * ROW + x]] >> (8 * c));
    }
  }
}

template <int OP, int EPB>
__global__ void __launch_bounds__(Threads<EPB>::value) mg_level_kernel(const uint8_t* __restrict__ rec_in, const int32_t* __restrict__ idx_in,
                                                                      int n_rows_in, Params p, LevelArgs a) {   // leading scalars: kernarg preload
  constexpr int NT = Threads<EPB>::value;
  __shared__ __attribute__((aligned(16))) uint8_t lds[EPB * LP];
  __shared__ int env_of[EPB];
  __shared__ uint32_t zctr[EPB];                  // per-row rng counter at which this call's random_z starts
  __shared__ Luts lut;
  constexpr bool DECODE = (OP == OP_SET_ENCODING || OP == OP_FIXED_LEVEL);
  __shared__ uint8_t codes[DECODE ? EPB * MAX_SIZE * MAX_SIZE : 1];   // the rows' levels, decoded cooperatively
  __shared__ uint32_t bad_enc[DECODE ? EPB : 1];
  const int tid = threadIdx.x, row0 = blockIdx.x * EPB;
  const int n = min(EPB, n_rows_in - row0), row = row0 + tid;
  if (tid < EPB) env_of[tid] = (tid < n) ? (idx_in ? idx_in[row] : row) : -1;
  const int S = p.S;
  init_luts(lut, tid);
  if (DECODE && tid < EPB) bad_enc[tid] = 0;
  __syncthreads();
  {
    StageBuf<EPB, NT> sb;
    stage_load<EPB, NT>(sb, rec_in, [&](int r) { return env_of[r]; }, n, tid);
    stage_commit<EPB, NT>(sb, lds, n, tid);
  }
  if (DECODE) {                                   // WorldObj.decode of every cell of every row (multigrid.py:264-280), one per thread
    const int SS = S * S;
    for (int q = tid; q < n * SS; q += NT) {
      const int r = q / SS, cell = q - r * SS;
      const uint8_t* c = a.enc + ((size_t)(row0 + r) * SS + cell) * 3;
      uint8_t code;
      if (!dec3(c[0], c[1], c[2], &code)) { atomicOr(&bad_enc[r], 1u); code = C_NONE; }
      codes[r * (MAX_SIZE * MAX_SIZE) + cell] = code;
    }
  }
  __syncthreads();

  constexpr bool ADV_OUT = (OP == OP_RESET || OP == OP_STEP_ADV || OP == OP_READ_ENC);
  if (tid < n && OP != OP_READ_ENC) {
    const int env = env_of[tid];
    // (key / metrics / teacher action are read here, after the staging barrier: issuing the three loads next to the record
    // loads at kernel entry measured 1.6 % SLOWER on the whole pass, same-box A/B, round 2)
    Env e = make_env(as_lds(lds + tid * LP), a.seed[env]);
    Metrics m = a.met[env];
    bool done = false;
    if (OP == OP_RESET) {
      env_reset(e, p, m, nullptr);                  // random_z floats are produced cooperatively below
      e.elapsed = 0;                                // TimeLimit.reset (time_limit.py:35-37)
      e.ep_ret = 0.f; e.ep_len = 0;                 // VecMonitor.reset (vec_monitor.py:36-40)
    } else if (OP == OP_STEP_ADV) {
      done = env_step_adversary(e, p, m, a.loc[env], nullptr);
    } else if (OP == OP_RESET_AGENT) {
      reset_agent(e);
      e.elapsed = 0;                                // time_limit.py:47-49
      e.ep_ret = 0.f; e.ep_len = 0;                 // vec_monitor.py:42-46
    } else if (OP == OP_RESET_RANDOM) {
      env_reset_random(e, p, m);
      e.elapsed = 0;                                // time_limit.py:39-41
      e.ep_ret = 0.f; e.ep_len = 0;                 // vec_monitor.py:48-52
    } else if (OP == OP_SET_ENCODING) {
      env_reset(e, p, m, nullptr);                  // adversarial.py:272
      e.elapsed = 0;                                // time_limit.py:43-45
      env_set_codes(e, p, m, as_lds(codes + tid * (MAX_SIZE * MAX_SIZE)), bad_enc[tid] != 0);
    } else if (OP == OP_FIXED_LEVEL) {
      e.step_count = 0; e.sb(6) = 0;               // env_load_fixed_level with the level already decoded
      reset_agent_status(e);
      e.sb(0) = NOPOS; e.sb(3) = NOPOS;
      reset_metrics(m, p);
      env_set_codes(e, p, m, as_lds(codes + tid * (MAX_SIZE * MAX_SIZE)), bad_enc[tid] != 0, true);
      e.elapsed = 0;                                // TimeLimit.reset
      e.ep_ret = 0.f; e.ep_len = 0;                 // VecMonitor.reset (vec_monitor.py:36-40)
    } else if (OP == OP_REPLAY) {
      env_reset(e, p, m, nullptr);
      e.elapsed = 0;
      const int L = a.lens[row];
      if (p.resample_n_clutter) e.sb(7) = (uint8_t)L;                       // adversarial.py:277-278
      for (int k = 0; k < L; ++k)
        if (env_step_adversary(e, p, m, a.loc[(size_t)row * a.max_len + k], nullptr)) reset_agent(e);   // :280-283
    } else if (OP == OP_MUTATE) {
      env_mutate_level(e, p, m, a.num_edits);
    }
    e.store_dyn();
    a.met[env] = m;
    raise(a.errs, e.err, env);
    if (ADV_OUT) {
      zctr[tid] = e.ctr() - 13;                     // generate_random_z consumed the last 13 counters
      if (a.adv.time_step) a.adv.time_step[env] = (float)e.sb(6);
      if (a.adv_done) a.adv_done[env] = done ? 1 : 0;
    }
  }
  __syncthreads();
  if (OP != OP_READ_ENC) stage_out<NT>(lds, a.rec, [&](int r) { return env_of[r]; }, n, tid);
  if (ADV_OUT) {
    if (OP != OP_READ_ENC && a.adv.random_z) {
      // generate_random_z (adversarial.py:449-450): (row, q) pairs, one Philox block = 4 floats each
      for (int j = tid; j < n * 13; j += NT) {
        const int r = j / 13, q = j - r * 13, env = env_of[r];
        uint32_t o[4];
        philox4x32_10(zctr[r] + q, 0, 0, 0, a.seed[env], p.engine_seed, o);
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          const int k = q * 4 + w;
          if (k < p.random_z_dim) a.adv.random_z[(size_t)env * p.random_z_dim + k] = (float)(o[w] >> 8) * (1.0f / 16777216.0f);
        }
      }
    }
    uint8_t* u8 = (OP == OP_READ_ENC) ? a.enc_out : a.adv.image_u8;
    float* f32 = (OP == OP_READ_ENC) ? nullptr : a.adv.image_f32;
    if (S == 15) emit_grid<15, NT>(lds, lut, u8, f32, (size_t)row0, n, S, tid);
    else emit_grid<0, NT>(lds, lut, u8, f32, (size_t)row0, n, S, tid);
  } else {
    render_emit_v<EPB, NT>(lds, lut, a.stu, (size_t)row0, n, p, tid);
  }
}

__global__ void set_seed_kernel(uint8_t* rec, uint32_t* seed, const int64_t* seeds, int B) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B) return;
  seed[i] = (uint32_t)seeds[i];
  uint8_t* r = rec + (size_t)i * REC_BYTES;        // draw counter := 0 (SB8..11)
  r[8 * ROW + 15] = 0; r[9 * ROW + 15] = 0; r[10 * ROW + 15] = 0; r[11 * ROW + 15] = 0;
}

__global__ void __launch_bounds__(WAVE) init_records_kernel(Params p, uint8_t* rec, uint32_t* seed, Metrics* met, long long base) {
  // MultiGridEnv.__init__ -> reset() (multigrid.py:463): every env starts as an empty walled grid
  __shared__ __attribute__((aligned(16))) uint8_t lds[WAVE * LP];
  const int i = blockIdx.x * WAVE + threadIdx.x;
  if (i >= p.B) return;
  uint8_t* l = lds + threadIdx.x * LP;
  for (int k = 0; k < REC_BYTES; ++k) l[k] = 0;
  seed[i] = (uint32_t)(base + i);
  Env e = make_env(as_lds(l), seed[i]);
  e.sb(7) = (uint8_t)(p.n_clutter + 2);             // adversary_max_steps (adversarial.py:107)
  Metrics m;
  env_reset(e, p, m, nullptr);
  e.store_dyn();
  met[i] = m;
  for (int k = 0; k < REC_BYTES; k += 16) *reinterpret_cast<uint4*>(rec + (size_t)i * REC_BYTES + k) = *reinterpret_cast<const uint4*>(l + k);
}

__global__ void __launch_bounds__(WAVE) get_state_kernel(Params p, const uint8_t* rec, int32_t* out) {
  __shared__ __attribute__((aligned(16))) uint8_t lds[WAVE * LP];
  const int i = blockIdx.x * WAVE + threadIdx.x;
  if (i >= p.B) return;
  uint8_t* l = lds + threadIdx.x * LP;
  for (int k = 0; k < REC_BYTES; k += 16) *reinterpret_cast<uint4*>(l + k) = *reinterpret_cast<const uint4*>(rec + (size_t)i * REC_BYTES + k);
  Env e = make_env(as_lds(l), 0);
  int32_t* o = out + (size_t)i * 16;
  o[0] = e.has_pos() ? e.ax : -1; o[1] = e.has_pos() ? e.ay : -1; o[2] = e.adir;
  o[3] = e.carrying ? (int32_t)enc3((uint8_t)e.carrying) : 0;
  o[4] = e.step_count; o[5] = e.elapsed; o[6] = e.ep_len;
  float f = e.ep_ret; memcpy(&o[7], &f, 4);
  o[8] = e.sb(0) != NOPOS ? e.sb(0) : -1; o[9] = e.sb(0) != NOPOS ? e.sb(1) : -1; o[10] = e.sb(2);
  o[11] = e.sb(3) != NOPOS ? e.sb(3) : -1; o[12] = e.sb(3) != NOPOS ? e.sb(4) : -1;
  o[13] = e.sb(6); o[14] = e.sb(7); o[15] = (int32_t)e.ctr();
}

__global__ void get_metrics_kernel(int B, const Metrics* met, int32_t* ncl, int32_t* pas, int32_t* spl, int32_t* dist) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B) return;
  const Metrics m = met[i];
  if (ncl) ncl[i] = m.n_clutter_placed;
  if (pas) pas[i] = m.passable;
  if (spl) spl[i] = m.shortest_path_length;
  if (dist) dist[i] = m.distance_to_goal;
}

// records (16 x 16 B per env), Philox keys and metrics (16 B per env) of one replica into another
__global__ void clone_kernel(int B, const uint4* __restrict__ rec_s, uint4* __restrict__ rec_d, const uint32_t* __restrict__ seed_s,
                             uint32_t* __restrict__ seed_d, const uint4* __restrict__ met_s, uint4* __restrict__ met_d) {
  static_assert(sizeof(Metrics) == 16 && REC_BYTES == 256, "clone_kernel copies 16-byte units");
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q < B * 16) rec_d[q] = rec_s[q];
  else if (q < B * 17) { met_d[q - B * 16] = met_s[q - B * 16]; }
  else if (q < B * 18) { seed_d[q - B * 17] = seed_s[q - B * 17]; }
}

__global__ void random_actions_kernel(int64_t* actions, int B, int T, long long base, uint32_t seed, uint32_t t0, int num_actions) {
  const long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= (long long)B * T) return;
  const int t = (int)(q / B), i = (int)(q - (long long)t * B);
  uint32_t o[4];
  philox4x32_10(t0 + t, 0, 0, 0, seed, (uint32_t)(base + i), o);
  actions[q] = (int64_t)mulhi32(o[0], (uint32_t)num_actions);
}

}  // namespace

// =================================================================================== C ABI
// large-grid variants (mg_big.hip, one build per record pitch); a handle created with size > 15 forwards to them
#define DCD_BIG_DECL(P) \
  int dcd_big_create_p##P(const DcdMgConfig*, int, void**); int dcd_big_destroy_p##P(void*); \
  int dcd_big_set_seed_p##P(void*, const int64_t*, void*); int dcd_big_poll_errors_p##P(void*, uint32_t*, int32_t*, void*); \
  int dcd_big_level_p##P(void*, int, const uint8_t*, const int32_t*, int32_t, const DcdStudentObs*, void*); \
  int dcd_big_step_p##P(void*, const int64_t*, int32_t, const DcdStepOut*, void*); \
  int dcd_big_get_encodings_p##P(void*, uint8_t*, void*); int dcd_big_get_state_p##P(void*, int32_t*, void*); \
  int dcd_big_get_metrics_p##P(void*, int32_t*, int32_t*, int32_t*, int32_t*, void*);
extern "C" { DCD_BIG_DECL(32) DCD_BIG_DECL(64) DCD_BIG_DECL(128) }
#define BIG(h, fn, ...) ((h)->pitch == 32 ? fn##_p32((h)->big, __VA_ARGS__) : (h)->pitch == 64 ? fn##_p64((h)->big, __VA_ARGS__) : fn##_p128((h)->big, __VA_ARGS__))
#define NO_BIG(h, what) do { if ((h)->big) return dcd::fail(what ": not available for grids above 15 x 15 (teacher-side op)"); } while (0)

struct DcdMg {
  void* big = nullptr;   // size > 15: the pitch-specific handle of mg_big.hip, everything below is unused
  int pitch = 16;
  // replicas: R copies of the batch in ONE allocation (rec / seed / met of replica k start at row k*B); the handle
  // dcd_mg_create returns is replica 0 and owns the memory, dcd_mg_replica hands out non-owning views of the others
  int replicas = 1, replica = 0;
  DcdMg* owner = nullptr;
  DcdMgConfig cfg;
  Params p;
  int device;
  uint8_t* rec = nullptr;
  uint32_t* seed = nullptr;
  Metrics* met = nullptr;
  uint32_t* errs = nullptr;      // [0] bits, [1] first env
  unsigned long long* stamps = nullptr;   // diagnostic build only
  // per-launch event pairs while profiling (dcd_mg_profile_begin/end)
  std::vector<hipEvent_t> ev;
  int ev_used = 0;
  bool profiling = false;
};

namespace dcd {
thread_local std::string g_last_error;
int fail(const char* fmt, ...) {
  char buf[512];
  va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
  g_last_error = buf;
  return 1;
}
}  // namespace dcd

#define HIP_OK(x) do { hipError_t _e = (x); if (_e != hipSuccess) return dcd::fail("%s: %s", #x, hipGetErrorString(_e)); } while (0)
#define CHECK_H(h) do { if (!(h)) return dcd::fail("null handle"); } while (0)

static inline int nblk(int n) { return (n + 63) / 64; }
// envs per workgroup of the step kernel: keep >= ~2 workgroups per CU (256 CUs) before widening
static inline int step_epb(int B) {
  static const int forced = [] { const char* e = getenv("DCD_STEP_EPB"); return e ? atoi(e) : 0; }();   // diagnostic: same-box A/B
  if (forced == 16 || forced == 32 || forced == 64) return forced;
  return B >= 64 * 512 ? 64 : (B >= 32 * 512 ? 32 : 16);
}

static LevelArgs base_args(DcdMg* h) {
  LevelArgs a; memset(&a, 0, sizeof a);
  a.rec = h->rec; a.seed = h->seed; a.met = h->met; a.errs = h->errs; a.n_rows = h->p.B;
  return a;
}
template <int OP> static int launch_level(DcdMg* h, const LevelArgs& a, void* stream) {
  if (a.n_rows <= 0) return 0;
  const int epb = step_epb(a.n_rows);
  const dim3 grid((a.n_rows + epb - 1) / epb);
  if (epb == 16) DCD_LAUNCH_TAGGED(dcd::PROF_LEVEL, (mg_level_kernel<OP, 16>), grid, dim3(Threads<16>::value), (hipStream_t)stream, (const uint8_t*)a.rec, a.idx, a.n_rows, h->p, a);
  else if (epb == 32) DCD_LAUNCH_TAGGED(dcd::PROF_LEVEL, (mg_level_kernel<OP, 32>), grid, dim3(Threads<32>::value), (hipStream_t)stream, (const uint8_t*)a.rec, a.idx, a.n_rows, h->p, a);
  else DCD_LAUNCH_TAGGED(dcd::PROF_LEVEL, (mg_level_kernel<OP, 64>), grid, dim3(Threads<64>::value), (hipStream_t)stream, (const uint8_t*)a.rec, a.idx, a.n_rows, h->p, a);
  HIP_OK(hipGetLastError());
  return 0;
}


extern "C" {

const char* dcd_last_error(void) { return dcd::g_last_error.c_str(); }
int dcd_abi_version(void) { return DCD_ABI_VERSION; }

int dcd_mg_create(const DcdMgConfig* cfg, int device, DcdMg** out) {
  if (!cfg || !out) return dcd::fail("dcd_mg_create: null argument");
  if (cfg->num_envs <= 0) return dcd::fail("num_envs must be > 0");
  if (cfg->size < 3 || cfg->size > DCD_MAX_GRID) return dcd::fail("size %d outside [3,%d]", cfg->size, DCD_MAX_GRID);
  if (cfg->agent_view_size < 3 || cfg->agent_view_size > DCD_MAX_VIEW || !(cfg->agent_view_size & 1))
    return dcd::fail("agent_view_size %d must be odd in [3,%d]", cfg->agent_view_size, DCD_MAX_VIEW);
  if (cfg->random_z_dim < 0 || cfg->random_z_dim > 52) return dcd::fail("random_z_dim %d outside [0,52]", cfg->random_z_dim);
  if (cfg->editor_actions < 0 || cfg->editor_actions > 2) return dcd::fail("editor_actions must be 0..2");
  if (cfg->n_clutter < 0 || cfg->n_clutter + 2 > 255) return dcd::fail("n_clutter %d out of range", cfg->n_clutter);
  if (cfg->max_steps <= 0 || cfg->max_steps > 65535 || cfg->max_episode_steps > 65535) return dcd::fail("max_steps out of range");
  if (cfg->replicas < 0 || cfg->replicas > DCD_MAX_REPLICAS) return dcd::fail("replicas %d outside [0,%d]", cfg->replicas, DCD_MAX_REPLICAS);
  if (cfg->replicas > 1 && cfg->size > MAX_SIZE) return dcd::fail("replicas are not available for grids above 15 x 15");
  HIP_OK(hipSetDevice(device));
  DcdMg* h = new DcdMg();
  h->cfg = *cfg; h->device = device;
  if (cfg->size > MAX_SIZE) {                                      // large evaluation mazes: mg_big.hip
    h->pitch = cfg->size <= 31 ? 32 : (cfg->size <= 63 ? 64 : 128);
    h->p.B = cfg->num_envs; h->p.S = cfg->size; h->p.V = cfg->agent_view_size;
    const int rc = h->pitch == 32 ? dcd_big_create_p32(cfg, device, &h->big)
                 : h->pitch == 64 ? dcd_big_create_p64(cfg, device, &h->big) : dcd_big_create_p128(cfg, device, &h->big);
    if (rc) { delete h; return rc; }
    *out = h;
    return 0;
  }
  static const int NEDIT[3] = {2, 3, 4};
  Params& p = h->p;
  p.B = cfg->num_envs; p.S = cfg->size; p.V = cfg->agent_view_size; p.max_steps = cfg->max_steps;
  p.max_episode_steps = cfg->max_episode_steps; p.n_clutter = cfg->n_clutter;
  p.resample_n_clutter = cfg->resample_n_clutter; p.choose_goal_last = cfg->choose_goal_last;
  p.random_z_dim = cfg->random_z_dim; p.n_editor_actions = NEDIT[cfg->editor_actions];
  p.fixed_environment = cfg->fixed_environment; p.goal_noise = cfg->goal_noise; p.engine_seed = cfg->engine_seed;
  p.see_through_walls = cfg->see_through_walls != 0;
  init_reward_division(p);
  h->replicas = cfg->replicas > 1 ? cfg->replicas : 1;
  const size_t B = (size_t)p.B * h->replicas;           // every replica starts as the same empty walled grid, same seeds
  const uint32_t init[2] = {0u, 0x7fffffffu};
  hipError_t e = hipMalloc(&h->rec, B * REC_BYTES);
  if (e == hipSuccess) e = hipMalloc(&h->seed, B * 4);
  if (e == hipSuccess) e = hipMalloc(&h->met, B * sizeof(Metrics));
  if (e == hipSuccess) e = hipMalloc(&h->errs, 8);
  if (e == hipSuccess) e = hipMemcpy(h->errs, init, 8, hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    for (int k = 0; k < h->replicas && e == hipSuccess; ++k) {
      hipLaunchKernelGGL(init_records_kernel, dim3((p.B + 63) / 64), dim3(64), 0, 0, p, h->rec + (size_t)k * p.B * REC_BYTES,
                         h->seed + (size_t)k * p.B, h->met + (size_t)k * p.B, (long long)cfg->env_index_base);
      e = hipGetLastError();
    }
  }
  if (e == hipSuccess) e = hipDeviceSynchronize();
  if (e != hipSuccess) {                                          // nothing of a failed creation survives
    dcd_mg_destroy(h);
    return dcd::fail("dcd_mg_create: %zu envs: %s", B, hipGetErrorString(e));
  }
  *out = h;
  return 0;
}

int dcd_mg_destroy(DcdMg* h) {
  if (!h) return 0;
  if (h->big) {
    const int rc = h->pitch == 32 ? dcd_big_destroy_p32(h->big) : h->pitch == 64 ? dcd_big_destroy_p64(h->big) : dcd_big_destroy_p128(h->big);
    delete h;
    return rc;
  }
  if (h->owner) { delete h; return 0; }                 // a replica view owns nothing
  (void)hipSetDevice(h->device);
  (void)hipFree(h->rec); (void)hipFree(h->seed); (void)hipFree(h->met); (void)hipFree(h->errs);
  for (hipEvent_t e : h->ev) (void)hipEventDestroy(e);
  delete h;
  return 0;
}

int dcd_mg_replica(DcdMg* h, int32_t k, DcdMg** out) {
  CHECK_H(h);
  if (!out) return dcd::fail("dcd_mg_replica: null out");
  NO_BIG(h, "dcd_mg_replica");
  if (h->owner) return dcd::fail("dcd_mg_replica: call it on the handle dcd_mg_create returned");
  if (k < 1 || k >= h->replicas) return dcd::fail("dcd_mg_replica: replica %d outside [1,%d)", k, h->replicas);
  DcdMg* v = new DcdMg();
  v->cfg = h->cfg; v->p = h->p; v->device = h->device; v->replicas = 1; v->replica = k; v->owner = h;
  v->rec = h->rec + (size_t)k * h->p.B * REC_BYTES; v->seed = h->seed + (size_t)k * h->p.B; v->met = h->met + (size_t)k * h->p.B;
  v->errs = h->errs;                                     // one error word per allocation
  *out = v;
  return 0;
}

int dcd_mg_clone_levels(DcdMg* dst, DcdMg* src, void* stream) {
  CHECK_H(dst); CHECK_H(src);
  NO_BIG(dst, "dcd_mg_clone_levels"); NO_BIG(src, "dcd_mg_clone_levels");
  if (dst->p.B != src->p.B || dst->p.S != src->p.S || dst->device != src->device)
    return dcd::fail("dcd_mg_clone_levels: handles differ in batch, grid size or device");
  if (dst->rec == src->rec) return 0;
  const int B = src->p.B;
  DCD_LAUNCH_TAGGED(dcd::PROF_OTHER, clone_kernel, dim3((B * 18 + 255) / 256), dim3(256), (hipStream_t)stream, B,
                     reinterpret_cast<const uint4*>(src->rec), reinterpret_cast<uint4*>(dst->rec), src->seed, dst->seed,
                     reinterpret_cast<const uint4*>(src->met), reinterpret_cast<uint4*>(dst->met));
  HIP_OK(hipGetLastError());
  return 0;
}

int dcd_mg_set_seed(DcdMg* h, const int64_t* seeds, void* stream) {
  CHECK_H(h);
  if (!seeds) return dcd::fail("dcd_mg_set_seed: null seeds");
  if (h->big) return BIG(h, dcd_big_set_seed, seeds, stream);
  hipLaunchKernelGGL(set_seed_kernel, dim3((h->p.B + 255) / 256), dim3(256), 0, (hipStream_t)stream, h->rec, h->seed, seeds, h->p.B);
  HIP_OK(hipGetLastError());
  return 0;
}

int dcd_mg_poll_errors(DcdMg* h, uint32_t* err_bits, int32_t* first_env, void* stream) {
  CHECK_H(h);
  if (h->big) return BIG(h, dcd_big_poll_errors, err_bits, first_env, stream);
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

int dcd_mg_reset(DcdMg* h, const DcdAdvObs* out, void* stream) {
  CHECK_H(h);
  NO_BIG(h, "dcd_mg_reset");
  LevelArgs a = base_args(h); if (out) a.adv = *out;
  return launch_level<OP_RESET>(h, a, stream);
}
int dcd_mg_step_adversary(DcdMg* h, const int64_t* loc, const DcdAdvObs* out, uint8_t* done, void* stream) {
  CHECK_H(h);
  NO_BIG(h, "dcd_mg_step_adversary");
  if (!loc) return dcd::fail("dcd_mg_step_adversary: null loc");
  LevelArgs a = base_args(h); a.loc = loc; if (out) a.adv = *out; a.adv_done = done;
  return launch_level<OP_STEP_ADV>(h, a, stream);
}
int dcd_mg_reset_agent(DcdMg* h, const DcdStudentObs* out, void* stream) {
  CHECK_H(h);
  if (h->big) return BIG(h, dcd_big_level, 1, nullptr, nullptr, h->p.B, out, stream);
  LevelArgs a = base_args(h); if (out) a.stu = *out;
  return launch_level<OP_RESET_AGENT>(h, a, stream);
}
int dcd_mg_reset_random(DcdMg* h, const DcdStudentObs* out, void* stream) {
  CHECK_H(h);
  NO_BIG(h, "dcd_mg_reset_random");
  LevelArgs a = base_args(h); if (out) a.stu = *out;
  return launch_level<OP_RESET_RANDOM>(h, a, stream);
}
int dcd_mg_reset_to_encoding(DcdMg* h, const uint8_t* enc, const int32_t* idx, int32_t n, const DcdStudentObs* out, void* stream) {
  CHECK_H(h);
  NO_BIG(h, "dcd_mg_reset_to_encoding");
  if (!enc) return dcd::fail("dcd_mg_reset_to_encoding: null enc");
  if (n < 0 || (!idx && n != h->p.B)) return dcd::fail("dcd_mg_reset_to_encoding: n=%d must equal num_envs=%d when idx is NULL", n, h->p.B);
  LevelArgs a = base_args(h); a.enc = enc; a.idx = idx; a.n_rows = n; if (out) a.stu = *out;
  return launch_level<OP_SET_ENCODING>(h, a, stream);
}
int dcd_mg_reset_to_fixed_level(DcdMg* h, const uint8_t* enc, const int32_t* idx, int32_t n, const DcdStudentObs* out, void* stream) {
  CHECK_H(h);
  if (!enc) return dcd::fail("dcd_mg_reset_to_fixed_level: null enc");
  if (n < 0 || (!idx && n != h->p.B)) return dcd::fail("dcd_mg_reset_to_fixed_level: n=%d must equal num_envs=%d when idx is NULL", n, h->p.B);
  if (h->big) return BIG(h, dcd_big_level, 0, enc, idx, n, out, stream);
  LevelArgs a = base_args(h); a.enc = enc; a.idx = idx; a.n_rows = n; if (out) a.stu = *out;
  return launch_level<OP_FIXED_LEVEL>(h, a, stream);
}
int dcd_mg_reset_to_actions(DcdMg* h, const int64_t* actions, const int32_t* lens, int32_t max_len, const int32_t* idx, int32_t n,
                            const DcdStudentObs* out, void* stream) {
  CHECK_H(h);
  if (!actions || !lens) return dcd::fail("dcd_mg_reset_to_actions: null actions/lens");
  NO_BIG(h, "dcd_mg_reset_to_actions");
  if (n < 0 || (!idx && n != h->p.B)) return dcd::fail("dcd_mg_reset_to_actions: n=%d must equal num_envs=%d when idx is NULL", n, h->p.B);
  LevelArgs a = base_args(h); a.loc = actions; a.lens = lens; a.max_len = max_len; a.idx = idx; a.n_rows = n; if (out) a.stu = *out;
  return launch_level<OP_REPLAY>(h, a, stream);
}
int dcd_mg_mutate_level(DcdMg* h, int32_t num_edits, const DcdStudentObs* out, void* stream) {
  CHECK_H(h);
  NO_BIG(h, "dcd_mg_mutate_level");
  if (num_edits < 0 || num_edits > 64) return dcd::fail("num_edits %d outside [0,64]", num_edits);
  LevelArgs a = base_args(h); a.num_edits = num_edits; if (out) a.stu = *out;
  return launch_level<OP_MUTATE>(h, a, stream);
}

// one launch over `n_sets` replicas starting at handle h (1: the plain step)
static int launch_step(DcdMg* h, int n_sets, const int64_t* actions0, const int64_t* actions1, int32_t flags, const DcdStepOut* out0,
                       const DcdStepOut* out1, void* stream) {
  StepArgs a; memset(&a, 0, sizeof a);
  a.rec = h->rec; a.seed = h->seed; a.met = h->met; a.errs = h->errs; a.actions = actions0; a.flags = flags;
  a.stamps = h->stamps;
  if (out0) a.out[0] = *out0;
  if (out1) a.out[1] = *out1;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (h->profiling && h->ev_used + 2 <= (int)h->ev.size()) { e0 = h->ev[h->ev_used]; e1 = h->ev[h->ev_used + 1]; h->ev_used += 2; }
  else dcd::prof_pair(dcd::PROF_STEP, &e0, &e1);
  const int epb = step_epb(h->p.B * n_sets);
  const dim3 grid(((h->p.B + epb - 1) / epb) * n_sets);
#define DCD_LAUNCH_STEP(EPB_, RR_, PAIR_, O8_) hipExtLaunchKernelGGL((mg_step_kernel<false, EPB_, RR_, PAIR_, O8_>), grid, dim3(Threads<EPB_>::value + 2 * WAVE), 0, (hipStream_t)stream, e0, e1, 0, (const uint8_t*)h->rec, (const uint32_t*)h->seed, actions0, h->p.B, actions1, h->p, a)
#define DCD_LAUNCH_RR(EPB_, PAIR_, O8_) do { if (rr) DCD_LAUNCH_STEP(EPB_, true, PAIR_, O8_); else DCD_LAUNCH_STEP(EPB_, false, PAIR_, O8_); } while (0)
#define DCD_LAUNCH_EPB(EPB_) do { if (n_sets == 2) { if (obs8) DCD_LAUNCH_RR(EPB_, true, true); else DCD_LAUNCH_RR(EPB_, true, false); } \
                                  else { if (obs8) DCD_LAUNCH_RR(EPB_, false, true); else DCD_LAUNCH_RR(EPB_, false, false); } } while (0)
  const bool rr = (flags & DCD_STEP_RESET_RANDOM) != 0;
  // raw-u8-only emission: no set wants an fp32 image
  auto u8_only = [&](const DcdStepOut& o) { return o.obs.image_u8 && !o.obs.image_f32 && !o.term_obs.image_f32; };
  const bool obs8 = u8_only(a.out[0]) && (n_sets == 1 || u8_only(a.out[1]));
  if (epb == 16) DCD_LAUNCH_EPB(16); else if (epb == 32) DCD_LAUNCH_EPB(32); else DCD_LAUNCH_EPB(64);
#undef DCD_LAUNCH_EPB
#undef DCD_LAUNCH_RR
#undef DCD_LAUNCH_STEP
  HIP_OK(hipGetLastError());
  return 0;
}

int dcd_mg_step(DcdMg* h, const int64_t* actions, int32_t flags, const DcdStepOut* out, void* stream) {
  CHECK_H(h);
  if (!actions) return dcd::fail("dcd_mg_step: null actions");
  if (h->big) return BIG(h, dcd_big_step, actions, flags, out, stream);
  return launch_step(h, 1, actions, nullptr, flags, out, nullptr, stream);
}

int dcd_mg_step_pair(DcdMg* h0, DcdMg* h1, const int64_t* actions0, const int64_t* actions1, int32_t flags,
                     const DcdStepOut* out0, const DcdStepOut* out1, void* stream) {
  CHECK_H(h0); CHECK_H(h1);
  if (!actions0 || !actions1) return dcd::fail("dcd_mg_step_pair: null actions");
  NO_BIG(h0, "dcd_mg_step_pair");
  if (h1->owner != (h0->owner ? h0->owner : h0) || h1->replica != h0->replica + 1)
    return dcd::fail("dcd_mg_step_pair: the handles must be consecutive replicas of one dcd_mg_create (replicas >= 2)");
  return launch_step(h0, 2, actions0, actions1, flags, out0, out1, stream);
}

// diagnostic builds (-DDCD_STAMPS) only; not part of include/dcd_hip.h
int dcd_internal_set_stamps(DcdMg* h, unsigned long long* dev_ptr) { if (!h) return 1; h->stamps = dev_ptr; return 0; }

int dcd_mg_profile_begin(DcdMg* h, int32_t capacity) {
  CHECK_H(h);
  NO_BIG(h, "dcd_mg_profile_begin");
  if (capacity <= 0 || capacity > (1 << 20)) return dcd::fail("dcd_mg_profile_begin: bad capacity %d", capacity);
  while ((int)h->ev.size() < 2 * capacity) {
    hipEvent_t e;
    HIP_OK(hipEventCreate(&e));
    h->ev.push_back(e);
  }
  h->ev_used = 0; h->profiling = true;
  return 0;
}

int dcd_mg_profile_end(DcdMg* h, int32_t* n_launches, double* total_ms, void* stream) {
  CHECK_H(h);
  h->profiling = false;
  HIP_OK(hipStreamSynchronize((hipStream_t)stream));
  double tot = 0.0;
  for (int i = 0; i + 1 < h->ev_used; i += 2) {
    float ms = 0.f;
    HIP_OK(hipEventElapsedTime(&ms, h->ev[i], h->ev[i + 1]));
    tot += ms;
  }
  if (n_launches) *n_launches = h->ev_used / 2;
  if (total_ms) *total_ms = tot;
  h->ev_used = 0;
  return 0;
}

int dcd_mg_step_random(DcdMg* h, uint32_t action_seed, uint32_t t0, int32_t K, int32_t flags, const DcdStepOut* out,
                       int64_t* n_episodes, void* stream) {
  CHECK_H(h);
  if (K <= 0) return dcd::fail("dcd_mg_step_random: K must be > 0");
  NO_BIG(h, "dcd_mg_step_random");
  StepArgs a; memset(&a, 0, sizeof a);
  a.rec = h->rec; a.seed = h->seed; a.met = h->met; a.errs = h->errs; a.flags = flags;
  a.K = K; a.action_seed = action_seed; a.t0 = t0; a.env_index_base = h->cfg.env_index_base;
  a.n_episodes = reinterpret_cast<unsigned long long*>(n_episodes);
  if (out) a.out[0] = *out;
  const int epb = step_epb(h->p.B);
  const dim3 grid((h->p.B + epb - 1) / epb);
#define DCD_LAUNCH_STEP(EPB_, RR_) hipLaunchKernelGGL((mg_step_kernel<true, EPB_, RR_>), grid, dim3(Threads<EPB_>::value + 2 * WAVE), 0, (hipStream_t)stream, (const uint8_t*)h->rec, (const uint32_t*)h->seed, (const int64_t*)nullptr, h->p.B, (const int64_t*)nullptr, h->p, a)
  const bool rr = (flags & DCD_STEP_RESET_RANDOM) != 0;
  if (epb == 16) { if (rr) DCD_LAUNCH_STEP(16, true); else DCD_LAUNCH_STEP(16, false); }
  else if (epb == 32) { if (rr) DCD_LAUNCH_STEP(32, true); else DCD_LAUNCH_STEP(32, false); }
  else { if (rr) DCD_LAUNCH_STEP(64, true); else DCD_LAUNCH_STEP(64, false); }
#undef DCD_LAUNCH_STEP
  HIP_OK(hipGetLastError());
  return 0;
}

int dcd_mg_get_encodings(DcdMg* h, uint8_t* out, void* stream) {
  CHECK_H(h);
  if (!out) return dcd::fail("dcd_mg_get_encodings: null out");
  if (h->big) return BIG(h, dcd_big_get_encodings, out, stream);
  LevelArgs a = base_args(h); a.enc_out = out;
  return launch_level<OP_READ_ENC>(h, a, stream);
}
int dcd_mg_get_metrics(DcdMg* h, int32_t* ncl, int32_t* pas, int32_t* spl, int32_t* dist, void* stream) {
  CHECK_H(h);
  if (h->big) return BIG(h, dcd_big_get_metrics, ncl, pas, spl, dist, stream);
  hipLaunchKernelGGL(get_metrics_kernel, dim3((h->p.B + 255) / 256), dim3(256), 0, (hipStream_t)stream, h->p.B, h->met, ncl, pas, spl, dist);
  HIP_OK(hipGetLastError());
  return 0;
}
int dcd_mg_get_state(DcdMg* h, int32_t* out, void* stream) {
  CHECK_H(h);
  if (!out) return dcd::fail("dcd_mg_get_state: null out");
  if (h->big) return BIG(h, dcd_big_get_state, out, stream);
  hipLaunchKernelGGL(get_state_kernel, dim3((h->p.B + 63) / 64), dim3(64), 0, (hipStream_t)stream, h->p, h->rec, out);
  HIP_OK(hipGetLastError());
  return 0;
}
int dcd_random_actions(int64_t* actions, int32_t B, int32_t T, int64_t base, uint32_t seed, uint32_t t0, int32_t num_actions, void* stream) {
  if (!actions || B <= 0 || T <= 0) return dcd::fail("dcd_random_actions: bad arguments");
  const long long total = (long long)B * T;
  hipLaunchKernelGGL(random_actions_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, actions, B, T, (long long)base, seed, t0, num_actions);
  HIP_OK(hipGetLastError());
  return 0;
}

}  // extern "C"
