// mg_env.h — per-environment MultiGrid logic on a packed 256-byte env record.
//
// One env = one 256-byte record (2 × 128-B L2 lines), staged into LDS by the kernels in
// mg_kernels.hip and owned by ONE lane while it is there:
//
//   byte y*16 + x   (x,y < 15)  packed cell code of grid cell (x,y)           rows 0..14
//   byte y*16 + 15  (y < 15)    "side column" SB[y]: per-level scalars
//        SB0 start_x (0xFF none)  SB1 start_y  SB2 start_dir  SB3 goal_x (0xFF none)  SB4 goal_y
//        SB5 flags (bit0 n_clutter_sampled)    SB6 adversary_step_count  SB7 adversary_max_steps
//        SB8..11 rng draw counter (LE u32)     SB12..14 spare
//   bytes 240..255  dynamic state (the only part the hot step always rewrites)
//        240 agent_x (0xFF none) 241 agent_y 242 agent_dir 243 carrying (cell code, 0 none)
//        244 step_count u16   246 elapsed_steps u16 (TimeLimit)   248 ep_len u16 (VecMonitor)
//        250 env.done  251 spare   252 ep_ret f32 (VecMonitor)
//      (held in registers while a lane owns the env: Env::load_dyn / store_dyn)
//
// Packed cell codes (55 values; decode(encode(code)) is the identity, and set_encoding's
// normalisations — Goal() colour, door state>=3, agent id — are built in):
//   0 None | 1+c wall | 7+c floor | 13+c key | 19+c ball | 25+c box | 31 goal | 32 lava |
//   33+d agent facing d | 37+3c+s door (s: 0 open, 1 closed, 2 locked)         c = colour 0..5
//
// Semantics follow the reference file:line cited at each function (envs/multigrid/
// multigrid.py, envs/multigrid/adversarial.py, envs/wrappers/*).  The functions are
// __host__ __device__ so tests/hostsim can run the identical code on the CPU (debugging and
// sanitizers only — the product path has no CPU mode).
#pragma once
#include <stdint.h>
#include "../../include/dcd_hip.h"

#if defined(__HIPCC__)
#define DCD_HD __host__ __device__ __forceinline__
#define DCD_HDM __host__ __device__ __forceinline__
#else
#define DCD_HD static inline
#define DCD_HDM inline
#endif
// On the device an env record is ALWAYS in LDS while a lane works on it; typing the pointers as
// address_space(3) makes every access a ds_read/ds_write (generic "flat" accesses also count on
// vmcnt and would wait for outstanding global stores).  Host builds (tests/hostsim) use plain pointers.
// rare-path hints: the hot step falls through; resets, respawns and error paths are laid out cold
#define DCD_RARE(x) __builtin_expect(!!(x), 0)
#if defined(__HIP_DEVICE_COMPILE__) && !defined(DCD_REC_GENERIC)
#define DCD_REC __attribute__((address_space(3)))
#else
#define DCD_REC                /* host builds, and the large-grid kernels (mg_big.hip) that work on records in global memory */
#endif

// Record geometry.  DCD_PITCH = 16 (default) is the layout described above: the hot path, grids up to 15 x 15.
// The same logic compiled with DCD_PITCH = 32 / 64 / 128 (mg_big.hip, one namespace per pitch) serves the large
// evaluation mazes (21, 51, 101 cells): PITCH rows of PITCH bytes, side column at x = PITCH-1 of rows 0..14,
// dynamic state in the last 16 bytes.
#ifndef DCD_PITCH
#define DCD_PITCH 16
#endif
#if DCD_PITCH == 16
#define DCD_NS dcd
#elif DCD_PITCH == 32
#define DCD_NS dcd_p32
#elif DCD_PITCH == 64
#define DCD_NS dcd_p64
#elif DCD_PITCH == 128
#define DCD_NS dcd_p128
#else
#error "DCD_PITCH must be 16, 32, 64 or 128"
#endif

namespace DCD_NS {

constexpr int ROW = DCD_PITCH;
constexpr int REC_BYTES = ROW * ROW;
constexpr int DYN = REC_BYTES - 16;
constexpr int SIDE = ROW - 1;
constexpr int MAX_SIZE = ROW - 1;                  // largest grid this pitch holds
constexpr uint8_t NOPOS = 0xFF;

enum : uint8_t { C_NONE = 0, C_WALL0 = 1, C_FLOOR0 = 7, C_KEY0 = 13, C_BALL0 = 19, C_BOX0 = 25,
                 C_GOAL = 31, C_LAVA = 32, C_AGENT0 = 33, C_DOOR0 = 37, C_END = 55 };
constexpr uint8_t C_WALL_GREY = C_WALL0 + 5;

// level-constant parameters shared by every env of a handle (kernel argument, by value)
struct Params {
  int32_t B, S, V, max_steps, max_episode_steps, n_clutter, resample_n_clutter, choose_goal_last,
      random_z_dim, n_editor_actions, fixed_environment;
  float goal_noise;
  uint32_t engine_seed;
  int32_t see_through_walls;   // 0: gen_obs_grid masks the view with Grid.process_vis (multigrid.py:1000-1002)
  int32_t fast_div;            // step_count / max_steps may be formed from inv_max_steps (init_reward_division)
  double inv_max_steps;
};

// ---------------------------------------------------------------- Philox4x32-10 (Salmon et al. SC'11)
DCD_HD uint32_t mulhi32(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * (uint64_t)b) >> 32); }
DCD_HD void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                          uint32_t out[4]) {
  const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    uint32_t hi0 = mulhi32(M0, c0), lo0 = M0 * c0, hi1 = mulhi32(M1, c2), lo1 = M1 * c2;
    uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
    c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
    k0 += W0; k1 += W1;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// float(k)/10.0f in IEEE fp32 division equals (float)((double)k/10.0) bit for bit for every byte value
// (checked for k = 0..255; `float(k)*0.1f` does NOT, first mismatch k = 9) — computed in
// registers so that building the LUTs needs no memory round trip.  kDiv10 documents the fp64 definition and is
// what the parity tests compare against through the oracle.
// The quotient is formed from 0.1f with one FMA correction (q0 = k*r; q = fma(fma(-q0, 10, k), r, q0)): equal to the
// IEEE division for every k in 0..255 (checked on the host, tools/ and tests/test_gpu_env_parity compare every
// observation with the oracle's true division), 5 instructions instead of the ~15 of a correctly rounded fp32 division
// — four of them sat on the logic wave's critical path at the start of every launch.
DCD_HD float div10f(uint32_t k) {
  const float n = (float)k, r = 0.1f, q0 = n * r;
  return __builtin_fmaf(__builtin_fmaf(-q0, 10.0f, n), r, q0);
}

// ---------------------------------------------------------------- packed codes
DCD_HD bool is_wall(uint8_t c) { return c >= C_WALL0 && c < C_FLOOR0; }
DCD_HD bool is_agent(uint8_t c) { return c >= C_AGENT0 && c < C_DOOR0; }
DCD_HD bool is_door(uint8_t c) { return c >= C_DOOR0 && c < C_END; }
DCD_HD bool can_overlap(uint8_t c) {   // gym_minigrid WorldObj.can_overlap: Goal, Floor, Lava, open Door
  return (c >= C_FLOOR0 && c < C_KEY0) || c == C_GOAL || c == C_LAVA || (is_door(c) && ((c - C_DOOR0) % 3) == 0);
}
DCD_HD bool can_pickup(uint8_t c) { return c >= C_KEY0 && c < C_GOAL; }   // Key, Ball, Box
// WorldObj.encode / Agent.encode (multigrid.py:138-140); None -> (1,0,0) (Grid.encode)
DCD_HD constexpr uint32_t enc3(uint8_t c) {     // type | colour<<8 | state<<16
  if (c == C_NONE) return DCD_OBJ_EMPTY;
  if (c < C_FLOOR0) return DCD_OBJ_WALL | ((uint32_t)(c - C_WALL0) << 8);
  if (c < C_KEY0) return DCD_OBJ_FLOOR | ((uint32_t)(c - C_FLOOR0) << 8);
  if (c < C_BALL0) return DCD_OBJ_KEY | ((uint32_t)(c - C_KEY0) << 8);
  if (c < C_BOX0) return DCD_OBJ_BALL | ((uint32_t)(c - C_BALL0) << 8);
  if (c < C_GOAL) return DCD_OBJ_BOX | ((uint32_t)(c - C_BOX0) << 8);
  if (c == C_GOAL) return DCD_OBJ_GOAL | (1u << 8);
  if (c == C_LAVA) return DCD_OBJ_LAVA;
  if (c < C_DOOR0) return DCD_OBJ_AGENT | ((uint32_t)(c - C_AGENT0) << 16);
  uint32_t d = c - C_DOOR0;
  return DCD_OBJ_DOOR | ((d / 3) << 8) | ((d % 3) << 16);
}
// The same map without branches (the step kernel builds its code -> encoding LUT with it at the start of every launch,
// one code per lane of the logic wave: as eleven nested range tests that was ~70 instructions of exec-mask juggling on
// the critical path).  Families of six colours: k = c - 1, family k / 6 = (k * 43) >> 8 for k < 30, types 2,3,5,6,7
// packed in 0x76532; doors: d / 3 = (d * 171) >> 9 for d < 18.  Checked against enc3 for every code at compile time.
DCD_HD constexpr uint32_t enc3_flat(uint32_t c) {
  const uint32_t k = c - 1u;
  const uint32_t g = ((k * 43u) >> 8) & 7u;
  const uint32_t fam = ((0x76532u >> (4u * g)) & 15u) | ((k - 6u * g) << 8);
  const uint32_t d = c - (uint32_t)C_DOOR0;
  const uint32_t dc = ((d * 171u) >> 9) & 7u;
  const uint32_t door = (uint32_t)DCD_OBJ_DOOR | (dc << 8) | ((d - 3u * dc) << 16);
  uint32_t r = fam;
  r = c == (uint32_t)C_NONE ? (uint32_t)DCD_OBJ_EMPTY : r;
  r = c == (uint32_t)C_GOAL ? ((uint32_t)DCD_OBJ_GOAL | (1u << 8)) : r;
  r = c == (uint32_t)C_LAVA ? (uint32_t)DCD_OBJ_LAVA : r;
  r = (c >= (uint32_t)C_AGENT0 && c < (uint32_t)C_DOOR0) ? ((uint32_t)DCD_OBJ_AGENT | ((c - (uint32_t)C_AGENT0) << 16)) : r;
  r = c >= (uint32_t)C_DOOR0 ? door : r;
  return r;
}
constexpr bool enc3_flat_matches() {
  for (uint32_t c = 0; c < (uint32_t)C_END; ++c) if (enc3_flat(c) != enc3((uint8_t)c)) return false;
  return true;
}
static_assert(enc3_flat_matches(), "enc3_flat must equal enc3 on every packed code");
// WorldObj.decode (multigrid.py:72-107); returns false on bytes the reference raises on
DCD_HD bool dec3(uint8_t t, uint8_t col, uint8_t s, uint8_t* code) {
  *code = C_NONE;
  if (t > DCD_OBJ_AGENT) return false;
  if (t != DCD_OBJ_AGENT && col > 5) return false;
  switch (t) {
    case DCD_OBJ_UNSEEN: case DCD_OBJ_EMPTY: return true;
    case DCD_OBJ_WALL: *code = C_WALL0 + col; return true;
    case DCD_OBJ_FLOOR: *code = C_FLOOR0 + col; return true;
    case DCD_OBJ_KEY: *code = C_KEY0 + col; return true;
    case DCD_OBJ_BALL: *code = C_BALL0 + col; return true;
    case DCD_OBJ_BOX: *code = C_BOX0 + col; return true;
    case DCD_OBJ_DOOR: *code = C_DOOR0 + col * 3 + (s == 0 ? 0 : (s == 2 ? 2 : 1)); return true;
    case DCD_OBJ_GOAL: *code = C_GOAL; return true;
    case DCD_OBJ_LAVA: *code = C_LAVA; return true;
    default: *code = C_AGENT0 + (s & 3); return s < 4;   // Agent(dir=state); dir>3 would break dir_vec
  }
}

// ---------------------------------------------------------------- env view over the record bytes
// The 16-byte dynamic state is held in REGISTERS while a lane works on its env (one 16-byte load at
// make_env, one 16-byte store at store_dyn); only grid cells and the side column are touched in
// LDS.  has_pos <=> ax != NOPOS.
struct Env {
  DCD_REC uint8_t* r;  // 256-byte record (LDS on device, host memory in tests/hostsim), row pitch 16
  uint32_t seed;       // Philox key word 0 (per-env seed)
  uint32_t err;        // DCD_ERR_* bits raised by this call
  // write-back bookkeeping for the hot step: cells changed (<=2) or everything
  int doff0, doff1;    // scalars, not an array: a dynamically indexed member would pin the whole Env to scratch
  int nd;
  bool full_dirty;
  bool ctr_dirty;     // the draw counter bytes (side column rows 8..11) changed
  // dynamic state (bytes 240..255 of the record)
  int ax, ay, adir, carrying, done;
  int step_count, elapsed, ep_len;
  float ep_ret;

  DCD_HDM DCD_REC uint8_t& cell(int x, int y) { return r[y * ROW + x]; }
  DCD_HDM DCD_REC uint8_t& sb(int k) { return r[k * ROW + SIDE]; }
  DCD_HDM void set_cell(int x, int y, uint8_t c) {
    const int o = y * ROW + x;
    r[o] = c;
    doff1 = (nd == 1) ? o : doff1;            // branch-free: first two writes are remembered, a third
    doff0 = (nd == 0) ? o : doff0;            // marks the whole record dirty
    full_dirty = full_dirty || (nd >= 2);
    nd += 1;
  }
  DCD_HDM bool has_pos() const { return ax != NOPOS; }
  DCD_HDM void pack_dyn(uint32_t w[4]) const {
    w[0] = (uint32_t)ax | ((uint32_t)ay << 8) | ((uint32_t)adir << 16) | ((uint32_t)carrying << 24);
    w[1] = (uint32_t)step_count | ((uint32_t)elapsed << 16);
    w[2] = (uint32_t)ep_len | ((uint32_t)done << 16);
    __builtin_memcpy(&w[3], &ep_ret, 4);
  }
  DCD_HDM void load_dyn() {
    uint32_t w[4];
    const DCD_REC uint32_t* q = reinterpret_cast<const DCD_REC uint32_t*>(r + DYN);   // 16-byte aligned
    w[0] = q[0]; w[1] = q[1]; w[2] = q[2]; w[3] = q[3];
    ax = w[0] & 255; ay = (w[0] >> 8) & 255; adir = (w[0] >> 16) & 255; carrying = w[0] >> 24;
    step_count = w[1] & 0xFFFF; elapsed = w[1] >> 16;
    ep_len = w[2] & 0xFFFF; done = (w[2] >> 16) & 255;
    __builtin_memcpy(&ep_ret, &w[3], 4);
  }
  DCD_HDM void store_dyn() {
    uint32_t w[4]; pack_dyn(w);
    DCD_REC uint32_t* q = reinterpret_cast<DCD_REC uint32_t*>(r + DYN);
    q[0] = w[0]; q[1] = w[1]; q[2] = w[2]; q[3] = w[3];
  }
  // rng draw counter lives in SB8..11
  DCD_HDM uint32_t ctr_bytes() { return sb(8) | (sb(9) << 8) | (sb(10) << 16) | ((uint32_t)sb(11) << 24); }
  DCD_HDM uint32_t ctr() { return cached ? c_ctr : ctr_bytes(); }
  DCD_HDM void set_ctr(uint32_t v) { sb(8) = (uint8_t)v; sb(9) = (uint8_t)(v >> 8); sb(10) = (uint8_t)(v >> 16); sb(11) = (uint8_t)(v >> 24); c_ctr = v; ctr_dirty = true; }
  // Start pose (SB0..2) and draw counter as registers.  The step kernel reads them ONCE next to the dynamic state
  // (cache_side): the end of an episode needs all seven, and as dependent single-byte LDS reads inside the divergent
  // goal path they cost ~1000 cycles in the slowest workgroup of a launch.  Level-building code writes SB0..2 and never
  // sets `cached`; the one in-step writer (env_reset_random) drops the cache first.
  int c_sx, c_sy, c_sd; uint32_t c_ctr; bool cached;
  int pending_done;     // agent_is_done left placements to finish_agent_done (1: carried object, 2: + observable respawn)
  DCD_HDM void cache_side() { c_sx = sb(0); c_sy = sb(1); c_sd = sb(2); c_ctr = ctr_bytes(); cached = true; }
  DCD_HDM int start_x() { return cached ? c_sx : (int)sb(0); }
  DCD_HDM int start_y() { return cached ? c_sy : (int)sb(1); }
  DCD_HDM int start_dir() { return cached ? c_sd : (int)sb(2); }
};

DCD_HD Env make_env(DCD_REC uint8_t* rec, uint32_t seed) {
  Env e; e.r = rec; e.seed = seed; e.err = 0; e.nd = 0; e.full_dirty = false; e.ctr_dirty = false; e.doff0 = e.doff1 = 0;
  e.cached = false; e.c_sx = e.c_sy = e.c_sd = 0; e.c_ctr = 0; e.pending_done = 0;
  e.load_dyn();
  return e;
}

DCD_HD uint32_t rand_u32(Env& e, const Params& p) {
  uint32_t o[4], n = e.ctr();
  philox4x32_10(n, 0, 0, 0, e.seed, p.engine_seed, o);
  e.set_ctr(n + 1);
  return o[0];
}
DCD_HD int randint(Env& e, const Params& p, int n) { return (int)mulhi32(rand_u32(e, p), (uint32_t)n); }
DCD_HD float uniform01(Env& e, const Params& p) { return (float)(rand_u32(e, p) >> 8) * (1.0f / 16777216.0f); }

// generate_random_z (adversarial.py:449-450): always consumes 13 counters
DCD_HD void generate_random_z(Env& e, const Params& p, float* z) {
  uint32_t n = e.ctr();
  for (int q = 0; q < 13; ++q) {
    uint32_t o[4];
    philox4x32_10(n + q, 0, 0, 0, e.seed, p.engine_seed, o);
    if (z) {
#pragma unroll
      for (int w = 0; w < 4; ++w) { int k = q * 4 + w; if (k < p.random_z_dim) z[k] = (float)(o[w] >> 8) * (1.0f / 16777216.0f); }
    }
  }
  e.set_ctr(n + 13);
}

// metrics live outside the record (int32[4] per env): n_clutter_placed, passable, shortest path, distance
struct Metrics { int32_t n_clutter_placed, passable, shortest_path_length, distance_to_goal; };
DCD_HD void reset_metrics(Metrics& m, const Params& p) {           // adversarial.py:184-188
  m.distance_to_goal = -1; m.n_clutter_placed = 0; m.passable = -1;
  m.shortest_path_length = (p.S - 2) * (p.S - 2) + 1;
}

#if DCD_PITCH == 16
// _gen_grid (adversarial.py:166-172): border of grey walls, None inside.  Whole rows as four 32-bit words (the side
// column byte at x = 15 is preserved); cells beyond S stay None.
DCD_HD void gen_grid(Env& e, const Params& p) {
  const int S = p.S;
  for (int y = 0; y < 15; ++y) {
    DCD_REC uint32_t* q = reinterpret_cast<DCD_REC uint32_t*>(e.r + y * ROW);
    const uint32_t side = q[3] & 0xFF000000u;
    uint32_t w[4] = {0, 0, 0, 0};
    if (y < S) {
      if (y == 0 || y == S - 1) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
#pragma unroll
          for (int b = 0; b < 4; ++b) if (4 * k + b < S) w[k] |= (uint32_t)C_WALL_GREY << (8 * b);
        }
      } else {
        w[0] = C_WALL_GREY;
#pragma unroll
        for (int k = 0; k < 4; ++k) if ((S - 1) >> 2 == k) w[k] |= (uint32_t)C_WALL_GREY << (8 * ((S - 1) & 3));
      }
    }
    q[0] = w[0]; q[1] = w[1]; q[2] = w[2]; q[3] = (w[3] & 0x00FFFFFFu) | side;
  }
  e.full_dirty = true;
}
#else
DCD_HD void gen_grid(Env& e, const Params& p) {                          // _gen_grid (adversarial.py:166-172)
  const int S = p.S;
  for (int y = 0; y < S; ++y)
    for (int x = 0; x < S; ++x)
      e.cell(x, y) = (x == 0 || y == 0 || x == S - 1 || y == S - 1) ? C_WALL_GREY : C_NONE;
  e.full_dirty = true;
}
#endif

// bit x of the result: cell (x,y) is a wall (codes 1..6), x < S.  Byte-parallel range test: codes are < 0x80, so the
// per-byte additions cannot carry into the neighbouring byte.
DCD_HD uint32_t row_wall_mask(Env& e, int y, int S) {                     // grids up to 15 wide (first 16 bytes of the row)
  const DCD_REC uint32_t* q = reinterpret_cast<const DCD_REC uint32_t*>(e.r + y * ROW);
  uint32_t m = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const uint32_t w = q[k];
    const uint32_t z = ((w + 0x7F7F7F7Fu) & ~(w + 0x79797979u)) & 0x80808080u;     // byte >= 1 and byte < 7
    m |= ((((z >> 7) * 0x00204081u) >> 21) & 0xFu) << (4 * k);
  }
  return m & ((1u << S) - 1u);
}

// bit x of the result: cell (x,y) is None, x < S.  One 16-byte row read + exact zero-byte detection per dword
// (the side column byte at x = 15 and the padding beyond S are masked off).
// `chunk` selects cells [16*chunk, 16*chunk+16) of the row (always 0 at the default pitch).
DCD_HD uint32_t row_free_mask(Env& e, int y, int S, int chunk = 0) {
  uint32_t w[4];
  const DCD_REC uint32_t* q = reinterpret_cast<const DCD_REC uint32_t*>(e.r + y * ROW + 16 * chunk);   // 16-byte aligned
  w[0] = q[0]; w[1] = q[1]; w[2] = q[2]; w[3] = q[3];
  uint32_t m = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const uint32_t z = ~((((w[k] & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | w[k]) | 0x7F7F7F7Fu);      // 0x80 in every zero byte
    m |= ((((z >> 7) * 0x00204081u) >> 21) & 0xFu) << (4 * k);                                // gather the 4 flags
  }
  const int left = S - 16 * chunk;                                 // cells of this chunk inside the grid
  return left >= 16 ? m : (left <= 0 ? 0u : (m & ((1u << left) - 1u)));
}

// place_obj with top=None,size=None (multigrid.py:565-632): uniform over None cells != agent_pos,
// drawn as the k-th such cell in row-major order.  code==C_NONE: position only (place_one_agent).
// Two passes over the S row masks (count, then locate): ~30 LDS reads instead of 2 x S*S byte reads.
DCD_HD bool place_uniform(Env& e, const Params& p, uint8_t code, int* px, int* py) {
  const int S = p.S;
  const int ex = e.has_pos() ? e.ax : -1, ey = e.ay;
  constexpr int CH = ROW / 16;                               // 16-cell chunks per row (1 at the default pitch)
  int n = 0;
  for (int y = 0; y < S; ++y)
    for (int ch = 0; ch < CH; ++ch) {
      uint32_t m = row_free_mask(e, y, S, ch);
      if (y == ey && ex >= 0 && (ex >> 4) == ch) m &= ~(1u << (ex & 15));
      n += __builtin_popcount(m);
    }
  if (n == 0) { e.err |= DCD_ERR_NO_FREE_CELL; return false; }
  int k = randint(e, p, n);
  for (int y = 0; y < S; ++y)
    for (int ch = 0; ch < CH; ++ch) {
      uint32_t m = row_free_mask(e, y, S, ch);
      if (y == ey && ex >= 0 && (ex >> 4) == ch) m &= ~(1u << (ex & 15));
      const int c = __builtin_popcount(m);
      if (k >= c) { k -= c; continue; }
      for (; k > 0; --k) m &= m - 1;                         // drop the k lowest free cells of this chunk
      const int x = 16 * ch + __builtin_ctz(m);
      if (code != C_NONE) e.set_cell(x, y, code);
      *px = x; *py = y;
      return true;
    }
  return false;
}

// place_agent_at_pos (multigrid.py:668-681); rand_dir=True means dir := 0 (:670-672)
DCD_HD void place_agent_at_pos(Env& e, int x, int y, bool rand_dir) {
  e.ax = x; e.ay = y;
  if (rand_dir) e.adir = 0;
  e.set_cell(x, y, (uint8_t)(C_AGENT0 + e.adir));
}
// place_one_agent with top=None (multigrid.py:651-666)
DCD_HD bool place_one_agent(Env& e, const Params& p, bool rand_dir, int* px, int* py) {
  e.ax = NOPOS;
  if (!place_uniform(e, p, C_NONE, px, py)) return false;
  place_agent_at_pos(e, *px, *py, rand_dir);
  return true;
}
DCD_HD void reset_agent_status(Env& e) {                                  // adversarial.py:231-236
  e.ax = NOPOS; e.adir = e.start_dir(); e.done = 0; e.carrying = C_NONE;
}
// AdversarialEnv.reset_agent (adversarial.py:238-269); TimeLimit counter is the caller's job
DCD_HD void reset_agent(Env& e) {
  if (e.has_pos()) e.set_cell(e.ax, e.ay, C_NONE);
  reset_agent_status(e);
  e.step_count = 0;                                                      // step_count
  const int sx = e.start_x();
  if (sx == NOPOS) { e.err |= DCD_ERR_NO_START_POS; return; }
  place_agent_at_pos(e, sx, e.start_y(), false);
}

// compute_metrics (adversarial.py:190-192,407-447): wall count + BFS on the interior graph with
// wall nodes removed.  Bit-board BFS: one uint16 row mask per interior row (<= 13x13).
DCD_HD void compute_metrics(Env& e, const Params& p, Metrics& m) {
  const int I = p.S - 2;
  if (I > 13) return;      // large evaluation mazes (MazeEnv / MSTMazeEnv define no such metrics): defaults stay
  uint16_t fr[13], rc[13];
  int nw = 0;
#pragma unroll
  for (int y = 0; y < 13; ++y) {
    uint16_t f = 0;
    if (y < I) {                                                   // interior row y+1: columns 1..I
      const uint32_t w = (row_wall_mask(e, y + 1, p.S) >> 1) & ((1u << I) - 1u);
      nw += __builtin_popcount(w);
      f = (uint16_t)(~w & ((1u << I) - 1u));
    }
    fr[y] = f; rc[y] = 0;
  }
  m.n_clutter_placed = nw;
  if (e.sb(0) == NOPOS || e.sb(3) == NOPOS) return;
  const int sx = e.sb(0) - 1, sy = e.sb(1) - 1, gx = e.sb(3) - 1, gy = e.sb(4) - 1;
  int dx = gx - sx, dy = gy - sy;
  m.distance_to_goal = (dx < 0 ? -dx : dx) + (dy < 0 ? -dy : dy);
#pragma unroll
  for (int y = 0; y < 13; ++y) if (y == sy) rc[y] = (uint16_t)(1u << sx);
  int dist = 0, found = (sx == gx && sy == gy) ? 1 : 0;
  for (int it = 0; it < 169 && !found; ++it) {
    uint16_t nr[13]; bool changed = false;
#pragma unroll
    for (int y = 0; y < 13; ++y) {
      uint16_t v = rc[y];
      uint16_t up = (y > 0) ? rc[y - 1] : 0, dn = (y < 12) ? rc[y + 1] : 0;
      uint16_t x = (uint16_t)((v | (uint16_t)(v << 1) | (v >> 1) | up | dn) & fr[y]);
      changed |= (x != v); nr[y] = x;
    }
    if (!changed) break;
#pragma unroll
    for (int y = 0; y < 13; ++y) rc[y] = nr[y];
    dist = it + 1;
    uint16_t grow = 0;
#pragma unroll
    for (int y = 0; y < 13; ++y) if (y == gy) grow = rc[y];
    found = (grow >> gx) & 1;
  }
  m.passable = found;
  m.shortest_path_length = found ? dist : I * I + 1;
}

// AdversarialEnv.reset (adversarial.py:194-229): z == nullptr still consumes the draws
DCD_HD void env_reset(Env& e, const Params& p, Metrics& m, float* z) {
  e.step_count = 0; e.sb(6) = 0;
  if (p.resample_n_clutter) e.sb(5) &= (uint8_t)~1u;
  e.sb(2) = (uint8_t)randint(e, p, 4);
  reset_agent_status(e);
  e.sb(0) = NOPOS; e.sb(3) = NOPOS;
  reset_metrics(m, p);
  gen_grid(e, p);
  generate_random_z(e, p, z);
}

DCD_HD void remove_wall(Env& e, Metrics& m, int x, int y) {               // adversarial.py:399-405
  if (is_wall(e.cell(x, y))) { e.set_cell(x, y, C_NONE); m.n_clutter_placed -= 1; }
}

// step_adversary (adversarial.py:452-539); returns done
DCD_HD bool env_step_adversary(Env& e, const Params& p, Metrics& m, int64_t loc, float* z) {
  const int I = p.S - 2, dim = I * I;
  if (loc < 0 || loc >= dim) { e.err |= DCD_ERR_BAD_ADV_ACTION; loc = 0; }
  if (p.resample_n_clutter && !(e.sb(5) & 1)) {                           // :469-472
    int nc = (int)(((double)loc / (double)dim) * (double)p.n_clutter);
    e.sb(7) = (uint8_t)(nc + 2); e.sb(5) |= 1;
  }
  const int cnt = e.sb(6), mx = e.sb(7);
  if (cnt < mx) {
    const int x = (int)(loc % I) + 1, y = (int)(loc / I) + 1;
    const bool choose_goal = p.choose_goal_last ? (cnt == mx - 2) : (cnt == 0);
    const bool choose_agent = p.choose_goal_last ? (cnt == mx - 1) : (cnt == 1);
    if (choose_goal) {
      bool noisy = false;
      if (p.goal_noise > 0.0f) noisy = uniform01(e, p) < p.goal_noise;     // :493
      if (noisy) {
        int px, py;
        if (place_uniform(e, p, C_GOAL, &px, &py)) { e.sb(3) = (uint8_t)px; e.sb(4) = (uint8_t)py; }
      } else {
        remove_wall(e, m, x, y);
        e.set_cell(x, y, C_GOAL);
        e.sb(3) = (uint8_t)x; e.sb(4) = (uint8_t)y;
      }
    } else if (choose_agent) {
      remove_wall(e, m, x, y);
      if (e.cell(x, y) != C_NONE) {                                        // goal already here :505
        int px, py;
        if (place_one_agent(e, p, false, &px, &py)) { e.sb(0) = (uint8_t)px; e.sb(1) = (uint8_t)py; }
      } else {
        e.sb(0) = (uint8_t)x; e.sb(1) = (uint8_t)y;
        place_agent_at_pos(e, x, y, false);
      }
    } else if (e.cell(x, y) == C_NONE) {                                   // :517-520
      e.set_cell(x, y, C_WALL_GREY); m.n_clutter_placed += 1;
    }
  }
  e.sb(6) = (uint8_t)(cnt + 1);
  bool done = false;
  if (cnt + 1 >= p.n_clutter + 2) { done = true; reset_metrics(m, p); compute_metrics(e, p, m); }   // :525
  generate_random_z(e, p, z);
  e.full_dirty = true;
  return done;
}

// reset_to_encoding body after reset() (adversarial.py:290-294; multigrid.py:264-280).
// keep_dir: fixed-level families (MazeEnv._gen_grid, maze.py:76-94) take the start direction from the
// encoding's agent cell instead of the random draw of AdversarialEnv.reset.
DCD_HD void env_set_encoding(Env& e, const Params& p, Metrics& m, const uint8_t* enc, bool keep_dir = false) {
  const int S = p.S;
  for (int i = 0; i < S; ++i) for (int j = 0; j < S; ++j) {
    const uint8_t* c = enc + (i * S + j) * 3;
    uint8_t code;
    if (!dec3(c[0], c[1], c[2], &code)) { e.err |= DCD_ERR_BAD_ENCODING; code = C_NONE; }
    if (is_agent(code)) { e.sb(0) = (uint8_t)i; e.sb(1) = (uint8_t)j; if (keep_dir) e.sb(2) = (uint8_t)(code - C_AGENT0); }
    else if (code == C_GOAL) { e.sb(3) = (uint8_t)i; e.sb(4) = (uint8_t)j; }
    e.cell(i, j) = code;
  }
  e.full_dirty = true;
  compute_metrics(e, p, m);
  reset_agent(e);
}

// The same from codes that the workgroup has already decoded into LDS (mg_kernels.hip: one (level, cell) pair per thread —
// a lane decoding its 675 encoding bytes from global memory one dependent load at a time took ~0.2 ms, which a replay
// rollout pays at every step for the handful of envs that finished an episode).  codes[i*S + j] = packed code of cell
// (x=i, y=j); bad: this level had an undecodable cell.  Same cell order, hence the same last-agent / last-goal winners.
DCD_HD void env_set_codes(Env& e, const Params& p, Metrics& m, const DCD_REC uint8_t* codes, bool bad, bool keep_dir = false) {
  const int S = p.S;
  if (bad) e.err |= DCD_ERR_BAD_ENCODING;
  for (int i = 0; i < S; ++i) for (int j = 0; j < S; ++j) {
    const uint8_t code = codes[i * S + j];
    if (is_agent(code)) { e.sb(0) = (uint8_t)i; e.sb(1) = (uint8_t)j; if (keep_dir) e.sb(2) = (uint8_t)(code - C_AGENT0); }
    else if (code == C_GOAL) { e.sb(3) = (uint8_t)i; e.sb(4) = (uint8_t)j; }
    e.cell(i, j) = code;
  }
  e.full_dirty = true;
  compute_metrics(e, p, m);
  reset_agent(e);
}

// MultiGridEnv.reset of a FIXED level (multigrid.py:470-502 with maze.py:76-94 as _gen_grid): the grid is
// the given encoding, the agent starts on its cell facing the encoded direction; no random draw.
DCD_HD void env_load_fixed_level(Env& e, const Params& p, Metrics& m, const uint8_t* enc) {
  e.step_count = 0; e.sb(6) = 0;
  reset_agent_status(e);
  e.sb(0) = NOPOS; e.sb(3) = NOPOS;
  reset_metrics(m, p);
  env_set_encoding(e, p, m, enc, true);
}

#if DCD_PITCH == 16
// The free cells of a level under construction as 15 row masks in registers (bit x of row[y]: cell (x,y) is None).
// `board_take` is place_uniform on that board: one draw, the k-th free cell in row-major order, which it marks as
// taken — the same draw and the same cell as place_uniform's two passes over the record, without the 30 row reads
// (the random reset places up to n_clutter + 2 objects back to back inside the step kernel, and the launch waits for it).
struct FreeBoard { uint16_t row[15]; int n; };
DCD_HD bool board_take(Env& e, const Params& p, FreeBoard& fb, int* px, int* py) {
  if (fb.n == 0) { e.err |= DCD_ERR_NO_FREE_CELL; return false; }
  int k = randint(e, p, fb.n);
  int ry = -1; uint32_t m = 0;
#pragma unroll
  for (int y = 0; y < 15; ++y) {
    const int c = __builtin_popcount((uint32_t)fb.row[y]);
    const bool here = ry < 0 && k < c;
    if (here) { ry = y; m = fb.row[y]; }
    else if (ry < 0) k -= c;
  }
  for (; k > 0; --k) m &= m - 1;                                         // drop the k lowest free cells of that row
  const int x = __builtin_ctz(m);
#pragma unroll
  for (int y = 0; y < 15; ++y) if (y == ry) fb.row[y] = (uint16_t)(fb.row[y] & ~(1u << x));
  fb.n -= 1;
  *px = x; *py = ry;
  return true;
}
#endif

// reset_random (adversarial.py:541-581)
DCD_HD void env_reset_random(Env& e, const Params& p, Metrics& m) {
  e.cached = false;                                         // writes the start pose below
  if (p.fixed_environment) e.set_ctr(0);
  e.step_count = 0; e.sb(6) = 0;
  reset_agent_status(e);
  e.sb(0) = NOPOS; e.sb(3) = NOPOS;
  reset_metrics(m, p);
  gen_grid(e, p);
  int px, py;
#if DCD_PITCH == 16
  FreeBoard fb;
  const int I = p.S - 2;
#pragma unroll
  for (int y = 0; y < 15; ++y) fb.row[y] = (uint16_t)((y >= 1 && y <= I) ? (((1u << I) - 1u) << 1) : 0u);   // the empty interior
  fb.n = I * I;
  if (board_take(e, p, fb, &px, &py)) { e.set_cell(px, py, C_GOAL); e.sb(3) = (uint8_t)px; e.sb(4) = (uint8_t)py; }
  e.sb(2) = (uint8_t)randint(e, p, 4);
  if (board_take(e, p, fb, &px, &py)) { place_agent_at_pos(e, px, py, false); e.sb(0) = (uint8_t)px; e.sb(1) = (uint8_t)py; }
  int nc;
  if (p.resample_n_clutter) { nc = randint(e, p, p.n_clutter); e.sb(7) = (uint8_t)(nc + 2); e.sb(5) |= 1; }
  else nc = p.n_clutter / 2;
  for (int k = 0; k < nc; ++k) {
    if (!board_take(e, p, fb, &px, &py)) break;                          // full grid: no draw is consumed from here on
    e.set_cell(px, py, C_WALL_GREY);
  }
#else
  if (place_uniform(e, p, C_GOAL, &px, &py)) { e.sb(3) = (uint8_t)px; e.sb(4) = (uint8_t)py; }
  e.sb(2) = (uint8_t)randint(e, p, 4);
  if (place_one_agent(e, p, false, &px, &py)) { e.sb(0) = (uint8_t)px; e.sb(1) = (uint8_t)py; }
  int nc;
  if (p.resample_n_clutter) { nc = randint(e, p, p.n_clutter); e.sb(7) = (uint8_t)(nc + 2); e.sb(5) |= 1; }
  else nc = p.n_clutter / 2;
  for (int k = 0; k < nc; ++k) place_uniform(e, p, C_WALL_GREY, &px, &py);
#endif
  compute_metrics(e, p, m);
  reset_agent(e);
}

// `list(set(values))` of small non-negative ints as CPython orders it (adversarial.py:327 iterates exactly that): the int is
// its own hash, 8 slots to start with, home slot hash & mask, then the next 9 slots while they lie inside the table, then
// i = 5 i + 1 + (perturb >>= 5); after an insertion with fill * 5 >= mask * 3 the table is rebuilt at the smallest power of
// two above 4 * used, old entries re-inserted in table order; iteration walks the table (Objects/setobject.c).  The oracle's
// restatement of the same is checked against the interpreter (tests/test_oracle_env.py); n <= 64 -> at most 128 slots.
// Cold path (the level editor): the local table lives in scratch.
DCD_HD int pyset_probe(const int16_t* table, int mask, int key, bool match) {
  uint32_t perturb = (uint32_t)key, i = (uint32_t)key & (uint32_t)mask;
  for (;;) {
    if (table[i] < 0 || (match && table[i] == key)) return (int)i;
    if (i + 9 <= (uint32_t)mask)
      for (uint32_t j = 1; j <= 9; ++j) if (table[i + j] < 0 || (match && table[i + j] == key)) return (int)(i + j);
    perturb >>= 5;
    i = (i * 5 + 1 + perturb) & (uint32_t)mask;
  }
}
DCD_HD int py_set_order(const uint8_t* values, int n, uint8_t* out) {
  int16_t table[128], old[128];
  int mask = 7, used = 0;
  for (int i = 0; i < 128; ++i) table[i] = -1;
  for (int k = 0; k < n; ++k) {
    const int key = values[k];
    const int slot = pyset_probe(table, mask, key, true);
    if (table[slot] == key) continue;
    table[slot] = (int16_t)key; ++used;
    if (used * 5 < mask * 3) continue;
    int newsize = 8;
    while (newsize <= used * 4) newsize <<= 1;
    const int oldmask = mask;
    for (int i = 0; i <= oldmask; ++i) old[i] = table[i];
    mask = newsize - 1;
    for (int i = 0; i <= mask; ++i) table[i] = -1;
    for (int i = 0; i <= oldmask; ++i) if (old[i] >= 0) table[pyset_probe(table, mask, old[i], false)] = old[i];
  }
  int m = 0;
  for (int i = 0; i <= mask; ++i) if (table[i] >= 0) out[m++] = (uint8_t)table[i];
  return m;
}

// mutate_level (adversarial.py:317-397)
DCD_HD void env_mutate_level(Env& e, const Params& p, Metrics& m, int num_edits) {
  const int I = p.S - 2, ntiles = I * I;
  uint8_t draws[64], locs[64], acts[64];
  if (num_edits > 64) num_edits = 64;
  for (int k = 0; k < num_edits; ++k) draws[k] = (uint8_t)randint(e, p, ntiles);      // np.random.randint(0, num_tiles, num_edits)
  const int nloc = py_set_order(draws, num_edits, locs);                                // list(set(...)): the set's own order
  for (int k = 0; k < nloc; ++k) acts[k] = (uint8_t)randint(e, p, p.n_editor_actions);
  uint16_t fm[13];                                  // free_mask rows (:332-334)
#pragma unroll
  for (int y = 0; y < 13; ++y) {
    uint16_t f = 0;
    if (y < I) for (int x = 0; x < I; ++x) f |= (uint16_t)((is_wall(e.cell(x + 1, y + 1)) ? 0 : 1) << x);
    fm[y] = f;
  }
  auto fm_set = [&](int x, int y, bool v) {         // interior coords
#pragma unroll
    for (int yy = 0; yy < 13; ++yy) if (yy == y) fm[yy] = v ? (uint16_t)(fm[yy] | (1u << x)) : (uint16_t)(fm[yy] & ~(1u << x));
  };
  if (e.sb(0) != NOPOS) fm_set(e.sb(0) - 1, e.sb(1) - 1, false);
  if (e.sb(3) != NOPOS) fm_set(e.sb(3) - 1, e.sb(4) - 1, false);
  for (int k = 0; k < nloc; ++k) {
    const int x = locs[k] % I + 1, y = locs[k] / I + 1;
    remove_wall(e, m, x, y);                         // _clean_loc (:296-306)
    uint8_t o = e.cell(x, y);
    if (o == C_GOAL) e.sb(3) = NOPOS; else if (is_agent(o)) e.sb(0) = NOPOS;
    e.set_cell(x, y, C_NONE);
    int a = acts[k];                                 // 0 '-', 1 '.', then 'g' (3 actions) or 'a','g' (4)
    const bool act_agent = (p.n_editor_actions == 4 && a == 2);
    const bool act_goal = (a >= 2 && !act_agent);
    if (a == 0) { e.set_cell(x, y, C_WALL_GREY); m.n_clutter_placed += 1; fm_set(x - 1, y - 1, false); }
    else if (a == 1) { fm_set(x - 1, y - 1, true); }
    else if (act_agent) {
      if (e.sb(0) != NOPOS) { e.set_cell(e.sb(0), e.sb(1), C_NONE); fm_set(e.sb(0) - 1, e.sb(1) - 1, true); }
      place_agent_at_pos(e, x, y, true);
      e.sb(0) = (uint8_t)x; e.sb(1) = (uint8_t)y; fm_set(x - 1, y - 1, false);
    } else if (act_goal) {
      if (e.sb(3) != NOPOS) { e.set_cell(e.sb(3), e.sb(4), C_NONE); fm_set(e.sb(3) - 1, e.sb(4) - 1, true); }
      e.set_cell(x, y, C_GOAL);
      e.sb(3) = (uint8_t)x; e.sb(4) = (uint8_t)y; fm_set(x - 1, y - 1, false);
    }
  }
  for (int pass = 0; pass < 2; ++pass) {             // goal, then agent (:370-388)
    if (pass == 0 ? (e.sb(3) != NOPOS) : (e.sb(0) != NOPOS)) continue;
    int n = 0;
#pragma unroll
    for (int y = 0; y < 13; ++y) n += __builtin_popcount((unsigned)fm[y]);
    if (n == 0) { e.err |= DCD_ERR_NO_FREE_CELL; continue; }
    int kk = randint(e, p, n), sx = -1, sy = -1;
    for (int y = 0; y < I && sx < 0; ++y) {
      uint16_t row = 0;
#pragma unroll
      for (int yy = 0; yy < 13; ++yy) if (yy == y) row = fm[yy];
      for (int x = 0; x < I; ++x) if ((row >> x) & 1) { if (kk-- == 0) { sx = x; sy = y; break; } }
    }
    fm_set(sx, sy, false);
    const int x = sx + 1, y = sy + 1;
    if (pass == 0) { e.set_cell(x, y, C_GOAL); e.sb(3) = (uint8_t)x; e.sb(4) = (uint8_t)y; }
    else { place_agent_at_pos(e, x, y, true); e.sb(0) = (uint8_t)x; e.sb(1) = (uint8_t)y; }
  }
  e.step_count = 0; e.sb(6) = 0;
  reset_metrics(m, p);
  compute_metrics(e, p, m);
  reset_agent(e);
  e.full_dirty = true;
}

// agent_is_done (multigrid.py:821-838).  The respawned agent is wiped by the auto reset_agent of
// the same step (parallel_wrappers.py:30-35); its position is observable ONLY through
// info['truncated_obs'] when the time limit fires on the same step.  When it is
// not (need_pos == false) the draw is still consumed — the counter stays in lock-step with the
// oracle — but the 225-cell scan is skipped and the agent is left unplaced.
// The two placements (a carried object dropped on a random free cell, the observable respawn) are large and almost
// never run: they are only REQUESTED here (pending_done) and executed by finish_agent_done at the end of
// step_one_agent, so that the path every reached goal takes stays a few instructions inside the hot code — with
// the placements inlined in the middle of it, skipping them cost two instruction-cache misses (~800 cycles) in the
// slowest workgroup of most launches.  Same operations in the same order (the draws keep their counters).
DCD_HD void agent_is_done(Env& e, const Params& p, bool need_pos) {
  e.set_cell(e.ax, e.ay, C_NONE);
  e.done = 1;
  if (DCD_RARE(e.carrying != C_NONE || need_pos)) { e.pending_done = need_pos ? 2 : 1; return; }
  e.ax = NOPOS; e.adir = 0; e.set_ctr(e.ctr() + 1);
}
DCD_HD void finish_agent_done(Env& e, const Params& p) {
  int px, py;
  if (e.carrying != C_NONE) { place_uniform(e, p, (uint8_t)e.carrying, &px, &py); e.carrying = C_NONE; }
  if (e.pending_done == 2) place_one_agent(e, p, true, &px, &py);
  else { e.ax = NOPOS; e.adir = 0; e.set_ctr(e.ctr() + 1); }
  e.pending_done = 0;
}

// MiniGridEnv._reward: 1 - 0.9 * (step_count / max_steps) in fp64.  The quotient by the launch-constant max_steps is
// formed from its reciprocal with one FMA correction (q0 = n*r; q = fma(fma(-q0, d, n), r, q0)); the engine checks at
// creation that this equals the IEEE quotient for EVERY step count 0..max_steps (Params::fast_div), otherwise the
// division itself runs.  The fp64 division sat in the divergent goal path of the slowest workgroup of a launch.
DCD_HD double step_quotient_fast(double n, double d, double r) {
  const double q0 = n * r;
  return __builtin_fma(__builtin_fma(-q0, d, n), r, q0);
}
DCD_HD double goal_reward(const Params& p, int step_count) {
  const double n = (double)step_count, d = (double)p.max_steps;
  const double q = p.fast_div ? step_quotient_fast(n, d, p.inv_max_steps) : n / d;
  return 1.0 - 0.9 * q;
}

// host side, at engine creation: enable the reciprocal form only if it reproduces the IEEE quotient for every step count
inline void init_reward_division(Params& p) {
  p.fast_div = 0; p.inv_max_steps = 0.0;
  if (p.max_steps <= 0) return;
  const double d = (double)p.max_steps, r = 1.0 / d;
  for (int n = 0; n <= p.max_steps; ++n)
    if (step_quotient_fast((double)n, d, r) != (double)n / d) return;
  p.inv_max_steps = r; p.fast_div = 1;
}

// step_one_agent (multigrid.py:866-941); returns the fp64 reward rounded to fp32 (obs_wrappers.py:179)
DCD_HD double step_one_agent(Env& e, const Params& p, int action, int step_count, bool need_respawn_pos) {
  const int d = e.adir;
  const int fx = e.ax + ((d == 0) - (d == 2)), fy = e.ay + ((d == 1) - (d == 3));   // DIR_TO_VEC
  const bool ok = fx >= 0 && fx < p.S && fy >= 0 && fy < p.S;
  if (DCD_RARE(!ok)) e.err |= DCD_ERR_OUT_OF_GRID;
  const uint8_t fwd = ok ? e.cell(fx, fy) : C_WALL_GREY;
  double reward = 0.0;
  // Three divergence classes instead of seven: turns, forward, and "everything else", whose object
  // interactions (pickup / drop / toggle) only do something next to a key/ball/box/door or while carrying.
  if (action <= DCD_ACT_RIGHT) {                                            // rotate_agent (:876-885)
    e.adir = (d + (action == DCD_ACT_LEFT ? 3 : 1)) & 3;
    e.set_cell(e.ax, e.ay, (uint8_t)(C_AGENT0 + e.adir));
  } else if (action == DCD_ACT_FORWARD) {
    if (DCD_RARE(!ok)) {
    } else if (fwd == C_GOAL || fwd == C_LAVA) {       // NOT hinted rare: some workgroup of nearly every launch takes it,
      agent_is_done(e, p, need_respawn_pos);           // and laid out of line it cost ~1000 cycles of instruction fetch
      if (fwd == C_GOAL) reward = goal_reward(p, step_count);               // MiniGridEnv._reward
    } else if (fwd == C_NONE || can_overlap(fwd)) {                         // move_agent :840-853
      e.set_cell(e.ax, e.ay, C_NONE);
      e.ax = fx; e.ay = fy;
      e.set_cell(fx, fy, (uint8_t)(C_AGENT0 + d));
    }
  } else if (DCD_RARE(action > DCD_ACT_DONE)) {
    e.err |= DCD_ERR_BAD_ACTION;
  } else if (DCD_RARE(ok && (e.carrying != C_NONE || fwd >= C_KEY0) && action != DCD_ACT_DONE)) {
    if (action == DCD_ACT_PICKUP) {                                          // :905-914
      if (can_pickup(fwd) && e.carrying == C_NONE) { e.carrying = fwd; e.set_cell(fx, fy, C_NONE); }
    } else if (action == DCD_ACT_DROP) {                                     // :917-924
      if (fwd == C_NONE && e.carrying != C_NONE) { e.set_cell(fx, fy, (uint8_t)e.carrying); e.carrying = C_NONE; }
    } else if (is_door(fwd)) {                                               // Door.toggle multigrid.py:113-123
      const int dd = fwd - C_DOOR0, col = dd / 3, st = dd % 3;
      if (st == 2) { if (e.carrying == C_KEY0 + col) e.set_cell(fx, fy, (uint8_t)(C_DOOR0 + col * 3)); }
      else e.set_cell(fx, fy, (uint8_t)(C_DOOR0 + col * 3 + (st == 0 ? 1 : 0)));
    } else if (fwd >= C_BOX0 && fwd < C_GOAL) {
      e.set_cell(fx, fy, C_NONE);                                            // Box.toggle: contents (None)
    }
  }
  if (DCD_RARE(e.pending_done)) finish_agent_done(e, p);
  return reward;
}

// gen_obs (multigrid.py:977-1055) via the closed-form view->world map of multigrid.py:1080-1093:
//   world = agent + f*(V-1-vy) + r*(vx - V/2), f = DIR_TO_VEC[dir], r = (-f.y, f.x)
// out: V*V*3 bytes indexed [vx][vy][c].  Out-of-grid cells render as Wall() (grey) (:330-334);
// the agent's own cell shows the carried object or empty (:1009-1013); other Agent objects seen in
// the view have their dir rotated with the view (rotate_left :308-314).
// one cell of that view as a packed code: (vx,vy) of the V x V window of an agent at (ax,ay) facing d
DCD_HD uint8_t view_cell_code(DCD_REC uint8_t* rec, int S, int V, int ax, int ay, int d, int carrying, int vx, int vy) {
  const int fx = (d == 0) - (d == 2), fy = (d == 1) - (d == 3), rx = -fy, ry = fx, h = V / 2;
  const int wx = ax + fx * (V - 1 - vy) + rx * (vx - h), wy = ay + fy * (V - 1 - vy) + ry * (vx - h);
  if (vx == h && vy == V - 1) return (uint8_t)carrying;
  if (wx < 0 || wx >= S || wy < 0 || wy >= S) return C_WALL_GREY;
  uint8_t c = rec[wy * ROW + wx];
  if (is_agent(c)) c = (uint8_t)(C_AGENT0 + (((c - C_AGENT0) - (d + 1)) & 3));
  return c;
}
// WorldObj.see_behind (gym-minigrid 1.0.1): Wall no, Door only when open, everything else (None, the
// multigrid Agent) yes.
DCD_HD bool see_behind(uint8_t c) { return !(is_wall(c) || (is_door(c) && ((c - C_DOOR0) % 3) != 0)); }
// Grid.process_vis (gym-minigrid 1.0.1) on bit boards, bit vx*V+vy: `transparent` marks the view cells
// one can see through, the result marks the visible ones.  Rows from the agent's (vy = V-1) upwards; per row a
// left-to-right then a right-to-left sweep; a visible see-through cell lights its neighbour in the sweep direction,
// the cell above itself and the cell above that neighbour.
DCD_HD uint64_t process_vis_mask(uint64_t transparent, int V) {
  uint64_t m = 1ull << ((V / 2) * V + V - 1);
  for (int vy = V - 1; vy >= 0; --vy) {
    for (int vx = 0; vx < V - 1; ++vx) {
      const uint64_t b = 1ull << (vx * V + vy);
      if (!(m & transparent & b)) continue;
      m |= b << V;                                             // (vx+1, vy)
      if (vy > 0) m |= (b << V) >> 1 | b >> 1;                 // (vx+1, vy-1), (vx, vy-1)
    }
    for (int vx = V - 1; vx >= 1; --vx) {
      const uint64_t b = 1ull << (vx * V + vy);
      if (!(m & transparent & b)) continue;
      m |= b >> V;                                             // (vx-1, vy)
      if (vy > 0) m |= (b >> V) >> 1 | b >> 1;
    }
  }
  return m;
}
DCD_HD void render_view(Env& e, const Params& p, DCD_REC uint8_t* out) {
  const int V = p.V, h = V / 2;
  uint64_t vis = ~0ull;
  if (!p.see_through_walls) {                                  // the agent's own cell holds the Agent object here: see-through
    uint64_t tr = 0;
    for (int vx = 0; vx < V; ++vx) for (int vy = 0; vy < V; ++vy)
      if ((vx == h && vy == V - 1) || see_behind(view_cell_code(e.r, p.S, V, e.ax, e.ay, e.adir, e.carrying, vx, vy)))
        tr |= 1ull << (vx * V + vy);
    vis = process_vis_mask(tr, V);
  }
  for (int vx = 0; vx < V; ++vx) for (int vy = 0; vy < V; ++vy) {
    uint32_t t = enc3(view_cell_code(e.r, p.S, V, e.ax, e.ay, e.adir, e.carrying, vx, vy));
    if (!((vis >> (vx * V + vy)) & 1)) t = 0;                   // Grid.encode(vis_mask): unseen cells stay (0,0,0)
    DCD_REC uint8_t* o = out + (vx * V + vy) * 3;
    o[0] = (uint8_t)t; o[1] = (uint8_t)(t >> 8); o[2] = (uint8_t)(t >> 16);
  }
}

struct StepResult { float reward; uint8_t done, info; float ep_ret; int32_t ep_len; bool have_term; };

// worker.step_env ∘ TimeLimit.step ∘ MultiGridEnv.step ∘ VecMonitor.step_wait [∘ runner last-step rule], in two halves
// so that the caller can render the pre-reset observation between them (the kernels do it cooperatively across a
// wave; a lane-serial render_view of 25 cells cost 30 us whenever a time limit fired):
//   env_step_pre   step_one_agent + done / TimeLimit detection            -> StepPre (have_term: obs wanted NOW)
//   env_step_post  auto-reset, VecMonitor, last-step rule                  -> StepResult
struct StepPre { double reward; bool done, have_term; uint8_t info; };

DCD_HD StepPre env_step_pre(Env& e, const Params& p, int64_t action64) {
  StepPre s;
  const int step_count = e.step_count + 1;                                 // multigrid.py:948
  e.step_count = step_count;
  const int action = (action64 < 0 || action64 > 6) ? 7 : (int)action64;
  // will TimeLimit attach truncated_obs this step?  (then the respawn position is observable)
  const bool will_term = p.max_episode_steps > 0 && e.elapsed + 1 >= p.max_episode_steps;
  s.reward = step_one_agent(e, p, action, step_count, will_term);
  bool done = e.done >= 1;                                               // competitive, n_agents == 1
  if (DCD_RARE(step_count >= p.max_steps)) done = true;                   // :972-973
  uint8_t info = 0; bool have_term = false;
  if (p.max_episode_steps > 0) {                                           // time_limit.py:24-33
    const int el = e.elapsed + 1;
    e.elapsed = el;
    if (DCD_RARE(el >= p.max_episode_steps)) {
      info |= DCD_INFO_TRUNC_KEY; if (!done) info |= DCD_INFO_TRUNCATED;
      have_term = true; done = true;
    }
  }
  s.done = done; s.info = info; s.have_term = have_term;
  return s;
}

DCD_HD StepResult env_step_post(Env& e, const Params& p, Metrics& m, bool& metrics_dirty, int flags, const StepPre& s) {
  StepResult r;
  bool done = s.done; uint8_t info = s.info;
  if (DCD_RARE(done)) {                                                    // parallel_wrappers.py:27-37
    if (flags & DCD_STEP_RESET_RANDOM) { env_reset_random(e, p, m); metrics_dirty = true; }
    reset_agent(e);
    e.elapsed = 0;                                                         // time_limit.py:39-49
  }
  // VecMonitor.step_wait (vec_monitor.py:60-85): eprets(f32) += rews(f64) is an fp64 add rounded to f32
  float epr = (float)((double)e.ep_ret + s.reward);
  int epl = e.ep_len + 1;
  r.ep_ret = 0.f; r.ep_len = 0;
  if (DCD_RARE(done)) { info |= DCD_INFO_EPISODE; r.ep_ret = epr; r.ep_len = epl; epr = 0.f; epl = 0; }
  e.ep_ret = epr; e.ep_len = epl;
  if (DCD_RARE(flags & DCD_STEP_LAST)) {                                   // adversarial_runner.py:521-530
    if (!done) info |= DCD_INFO_CLIFFHANGER | DCD_INFO_TRUNC_KEY | DCD_INFO_TRUNCATED;
    done = true;
  }
  r.reward = (float)s.reward; r.done = done; r.info = info;
  r.have_term = s.have_term && (info & DCD_INFO_EPISODE);                  // runner inserts only then (:539-549)
  return r;
}

// both halves with a lane-serial render between them (tests/hostsim, level kernels).
// term (V*V*3 bytes + 1 dir byte at term[V*V*3]) receives the pre-reset obs when have_term.
DCD_HD StepResult env_step(Env& e, const Params& p, Metrics& m, bool& metrics_dirty, int64_t action64,
                           int flags, DCD_REC uint8_t* term) {
  const StepPre s = env_step_pre(e, p, action64);
  if (DCD_RARE(s.have_term)) { render_view(e, p, term); term[p.V * p.V * 3] = (uint8_t)e.adir; }
  return env_step_post(e, p, m, metrics_dirty, flags, s);
}

}  // namespace DCD_NS
