// host_utils.cpp — host-side helpers of libdcd_hip.so that work on HOST memory (logging statistics only).
//
// dcd_action_complexity: RolloutStorage.get_action_complexity / get_batched_action_complexity
// (algos/storage.py:329-369).  Every shipped MultiGrid config sets log_action_complexity, and at 4096 envs the
// per-character Python loop (1M characters per rollout) cost 0.37 s per update.  Same definition as the
// `lempel_ziv_complexity` package the reference imports (storage.py:21): the number of distinct phrases of the
// LZ76 parsing of the episode's action string "a0 a1 a2 ..." (decimal, single spaces).
#include <stdint.h>
#include <string.h>
#include <stdio.h>
#include <stdlib.h>
#include <string>
#include <thread>
#include <vector>
#include <exception>
#include "../../include/dcd_hip.h"
#include "dcd_internal.h"

namespace {

struct Slot { uint64_t h; int32_t start, len; };

// number of phrases: i = 0; while i < n: extend j while s[i:j] is a known phrase; add s[i:j]; i = j
int lz76(const char* s, int n, std::vector<Slot>& table) {
  const size_t M = table.size();                      // power of two
  for (auto& t : table) t.len = 0;
  int phrases = 0, i = 0;
  while (i < n) {
    int j = i + 1;
    uint64_t h = 1469598103934665603ull;
    for (;;) {                                        // s[i:j], hash extended one character at a time (FNV-1a)
      if (j <= n) h = (h ^ (uint8_t)s[j - 1]) * 1099511628211ull;
      const int len = (j <= n ? j : n) - i;           // Python slicing clamps: s[i:n+1] == s[i:n]
      size_t q = (size_t)(h ^ (h >> 29)) & (M - 1);
      bool found = false;
      while (table[q].len) {
        if (table[q].h == h && table[q].len == len && memcmp(s + table[q].start, s + i, (size_t)len) == 0) { found = true; break; }
        q = (q + 1) & (M - 1);
      }
      if (found && j <= n) { ++j; continue; }
      if (!found) { table[q].h = h; table[q].start = i; table[q].len = len; ++phrases; }
      break;
    }
    i = j;
  }
  return phrases;
}

// envs [b0, b1): phrase counts summed per env (integers: exact in a double in any order)
void complexity_range(const int64_t* actions, const float* masks, int T, int B, int b0, int b1, double* tot_out, int64_t* n_out,
                      float* per_env) {
  std::vector<char> str((size_t)T * 21 + 1);
  std::vector<int> cuts(T + 3);
  std::vector<Slot> table(4096);
  double tot = 0; int64_t n = 0;
  for (int b = b0; b < b1; ++b) {
    int nc = 0;
    cuts[nc++] = 0;                                    // done_steps = [0] + nonzero(masks[:, b] == 0)   (:339, :357)
    for (int t = 0; t <= T; ++t) if (masks[(size_t)t * B + b] == 0.0f) cuts[nc++] = t;
    double sum = 0; int cnt = 0;
    for (int k = 0; k + 1 < nc; ++k) {
      int len = 0;
      for (int t = cuts[k]; t < cuts[k + 1] && t < T; ++t) {
        if (len) str[len++] = ' ';
        len += snprintf(&str[len], 21, "%lld", (long long)actions[(size_t)t * B + b]);
      }
      if ((size_t)len * 2 > table.size()) table.assign(table.size() * 4, Slot{0, 0, 0});
      sum += lz76(str.data(), len, table);
      ++cnt;
    }
    tot += sum; n += cnt;
    if (per_env) per_env[b] = cnt ? (float)(sum / cnt) : 0.f;
  }
  *tot_out = tot; *n_out = n;
}

}  // namespace

// Envs are independent: the batch is split over host threads (DCD_HOST_THREADS, default min(16, cores); 57 ms on one
// core for 4096 envs x 256 steps, once per student update in every shipped MultiGrid config).
extern "C" int dcd_action_complexity(const int64_t* actions, const float* masks, int32_t T, int32_t B, double* total,
                                     int64_t* n_episodes, float* per_env) {
  if (!actions || !masks || T <= 0 || B <= 0) return dcd::fail("dcd_action_complexity: NULL pointer or empty rollout");
  try {
    int nt = (int)std::thread::hardware_concurrency();
    if (const char* e = getenv("DCD_HOST_THREADS")) nt = atoi(e);
    nt = nt < 1 ? 1 : (nt > 16 ? 16 : nt);
    if (nt > B / 64) nt = B / 64 > 0 ? B / 64 : 1;     // at least 64 envs per thread
    std::vector<double> tots(nt, 0.0);
    std::vector<int64_t> ns(nt, 0);
    std::vector<std::string> errs(nt);
    auto work = [&](int i) {
      try { complexity_range(actions, masks, T, B, (int)((int64_t)B * i / nt), (int)((int64_t)B * (i + 1) / nt), &tots[i], &ns[i], per_env); }
      catch (const std::exception& e) { errs[i] = e.what(); if (errs[i].empty()) errs[i] = "exception"; }
    };
    std::vector<std::thread> pool;
    for (int i = 1; i < nt; ++i) pool.emplace_back(work, i);
    work(0);
    for (auto& th : pool) th.join();
    double tot = 0; int64_t n = 0;
    for (int i = 0; i < nt; ++i) {
      if (!errs[i].empty()) return dcd::fail("dcd_action_complexity: %s", errs[i].c_str());
      tot += tots[i]; n += ns[i];
    }
    if (total) *total = tot;
    if (n_episodes) *n_episodes = n;
    return 0;
  } catch (const std::exception& e) { return dcd::fail("dcd_action_complexity: %s", e.what()); }   // no C++ exception crosses the C ABI
}

// ---- LevelSampler: a batch of sequential replay draws (level_replay/level_sampler.py:664-750) ------------------
// The draws of one batch are sequential by definition (each one zeroes the chosen seed's staleness and ages the
// others), and each is O(N) numpy calls on the buffer: 4096 draws up front plus one per finished episode bound a
// replay update at 4096 envs per GPU (0.4 s on the host next to 0.25 s of GPU work).  This loop performs the SAME
// IEEE operations in the SAME order as the numpy statement in LevelSampler.sample_replay_levels — elementwise
// products / quotients, numpy's pairwise summation for np.sum, a running sum for cumsum, a right-sided binary search —
// for the common configuration (staleness transform 'power' with temperature 1).  The Python side keeps the numpy
// loop, checks this one against it on random inputs before first use, and takes over from the draw at which this
// loop meets a case it does not handle.
namespace {

// numpy/core/src/umath/loops_utils.h.src: pairwise summation of a contiguous double array
double np_pairwise_sum(const double* a, long n) {
  if (n < 8) {
    double res = 0.;
    for (long i = 0; i < n; ++i) res += a[i];
    return res;
  }
  if (n <= 128) {
    double r[8];
    for (int j = 0; j < 8; ++j) r[j] = a[j];
    long i;
    for (i = 8; i < n - (n % 8); i += 8)
      for (int j = 0; j < 8; ++j) r[j] += a[i + j];
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) res += a[i];
    return res;
  }
  long n2 = n / 2;
  n2 -= n2 % 8;
  return np_pairwise_sum(a, n2) + np_pairwise_sum(a + n2, n - n2);
}

}  // namespace

extern "C" double dcd_np_sum(const double* a, int64_t n) { return np_pairwise_sum(a, (long)n); }

// numpy's pairwise summation over elements PRODUCED ON THE FLY: gen(i0, m, buf) computes (and stores) elements
// [i0, i0+m) into buf — the same additions in the same order as np_pairwise_sum over the finished array, one pass less.
template <class Gen>
static inline double pairwise_gen(Gen& gen, double* a, long i0, long n) {
  if (n <= 128) {
    gen(i0, n, a + i0);
    return np_pairwise_sum(a + i0, n);                               // the block is in L1: the adds are all that is left
  }
  long n2 = n / 2;
  n2 -= n2 % 8;
  const double l = pairwise_gen(gen, a, i0, n2);
  return l + pairwise_gen(gen, a, i0 + n2, n - n2);
}

// returns the number of draws performed (== n unless an unhandled case stopped the loop early).
// target_clones: AVX2 / AVX-512 builds of the elementwise passes where the host has them (no FMA contraction either way).
// Three passes over the buffer per draw instead of eight: [age + mask + sum], [mix + sum], [normalise + running sum]; the
// last one is bound by the dependent additions of the cumulative sum (4 cycles each), which the result depends on.
extern "C" __attribute__((target_clones("avx512f", "avx2", "default")))
int64_t dcd_sampler_draws(int64_t N, int64_t n, const double* w_scores, double c, double* staleness, const double* unseen_keep,
                          const double* uniforms, int32_t update_staleness, int64_t* out_idx) {
  if (N <= 0 || n <= 0 || !w_scores || !staleness || !unseen_keep || !uniforms || !out_idx) return 0;
  std::vector<double> sw_, w_, cdf_;
  try { sw_.resize((size_t)N); w_.resize((size_t)N); cdf_.resize((size_t)N); } catch (const std::exception&) { return 0; }   // caller continues in numpy
  double* __restrict__ sw = sw_.data(); double* __restrict__ w = w_.data(); double* __restrict__ cdf = cdf_.data();
  double* __restrict__ st = staleness;
  // ---- fast path: integer staleness counts and a 0/1 mask (what LevelSampler holds) -------------------------------------
  // Then sw_i = staleness_i * keep_i are integers: np.sum(sw) is EXACT in any order, so z follows from two running integers
  // (A = sum a_i keep_i, K = sum keep_i: z = A + age K) instead of a pass; and "staleness += 1" is an offset (`age`) added
  // when an element is read, the chosen seed's base set to -(age + 1) — no pass writes the staleness array until the end.
  // Two sweeps per draw instead of three, 4 array streams instead of 7; every value that reaches the draw is the same double.
  bool ints = true;
  double A = 0., K = 0.;
  for (int64_t i = 0; i < N; ++i) {
    const double s = st[i], kp = unseen_keep[i];
    ints = ints && s >= 0. && s < 1e12 && s == (double)(int64_t)s && (kp == 0. || kp == 1.);
    A += s * kp; K += kp;
  }
  if (ints && (double)n < 1e12) {
    double age = 0.;
    auto finish = [&](int64_t done) { if (age != 0.) for (int64_t i = 0; i < N; ++i) st[i] = st[i] + age; return done; };
    for (int64_t k = 0; k < n; ++k) {
      const double z = A + age * K;
      if (!(z > 0)) return finish(k);
      auto gen = [&](long i0, long m, double* __restrict__ out) {
        for (long j = 0; j < m; ++j) {
          const double sw1 = (st[i0 + j] + age) * unseen_keep[i0 + j];
          const double q = sw1 / z; const double t = c * q; out[j] = w_scores[i0 + j] + t;
        }
      };
      const double total = pairwise_gen(gen, w, 0, (long)N);
      if (!(total > 1e-8 || total < -1e-8)) return finish(k);
      double run = 0., mn = w[0];
      const bool norm = total != 1.0;
      constexpr int64_t BK = 32;
      auto quot = [&](int64_t i0) {
        const int64_t m = N - i0 < BK ? N - i0 : BK;
        if (norm) for (int64_t j = 0; j < m; ++j) w[i0 + j] = w[i0 + j] / total;
        double bm = mn;
        for (int64_t j = 0; j < m; ++j) bm = w[i0 + j] < bm ? w[i0 + j] : bm;
        mn = bm;
      };
      quot(0);
      for (int64_t i0 = 0; i0 < N; i0 += BK) {
        if (i0 + BK < N) quot(i0 + BK);
        const int64_t m = N - i0 < BK ? N - i0 : BK;
        for (int64_t j = 0; j < m; ++j) { run += w[i0 + j]; cdf[i0 + j] = run; }
      }
      const double last = cdf[N - 1];
      if (!(last - 1.0 <= 1e-8 && 1.0 - last <= 1e-8) || mn < 0) return finish(k);
      const double u = uniforms[k];
      int64_t lo = 0, hi = N;
      while (lo < hi) { const int64_t mid = lo + (hi - lo) / 2; if (cdf[mid] / last <= u) lo = mid + 1; else hi = mid; }
      out_idx[k] = lo;
      if (update_staleness) {
        if (lo < N) { const double nb = -(age + 1.); A += (nb - st[lo]) * unseen_keep[lo]; st[lo] = nb; }
        age += 1.;
      }
    }
    return finish(n);
  }
  bool aging = false;                                                // staleness + 1 of the previous draw still to be applied
  auto flush = [&]() { if (aging) { for (int64_t i = 0; i < N; ++i) st[i] = st[i] + 1; aging = false; } };
  for (int64_t k = 0; k < n; ++k) {
    // pass 1: (staleness += 1 of the previous draw,) sw = staleness * unseen_keep, z = np.sum(sw), any staleness < 0
    double smin = 0.;
    const double inc = aging ? 1. : 0.;                              // x + 0. == x for every x >= 0 and for -1 + 1 (the chosen seed)
    auto gen1 = [&](long i0, long m, double* __restrict__ out) {
      double mn = smin;
      for (long j = 0; j < m; ++j) {
        const double s = st[i0 + j] + inc;
        st[i0 + j] = s;
        out[j] = s * unseen_keep[i0 + j];
        mn = s < mn ? s : mn;
      }
      smin = mn;
    };
    const double z = pairwise_gen(gen1, sw, 0, (long)N);
    aging = false;
    if (smin < 0) return k;
    if (!(z > 0)) return k;
    // pass 2: w = w_scores + c * (sw / z), total = np.sum(w)
    auto gen2 = [&](long i0, long m, double* __restrict__ out) {
      for (long j = 0; j < m; ++j) { const double q = sw[i0 + j] / z; const double t = c * q; out[j] = w_scores[i0 + j] + t; }
    };
    const double total = pairwise_gen(gen2, w, 0, (long)N);
    if (!(total > 1e-8 || total < -1e-8)) return k;                  // np.isclose(total, 0): the uniform fallback stays in Python
    // pass 3: w /= total (unless total == 1), cdf = cumsum(w): blocks of quotients (vector) ahead of the dependent additions
    double run = 0., mn = w[0];
    const bool norm = total != 1.0;
    constexpr int64_t BK = 32;
    auto quot = [&](int64_t i0) {                                     // block i0: quotients + running minimum (vector code)
      const int64_t m = N - i0 < BK ? N - i0 : BK;
      if (norm) for (int64_t j = 0; j < m; ++j) w[i0 + j] = w[i0 + j] / total;
      double bm = mn;
      for (int64_t j = 0; j < m; ++j) bm = w[i0 + j] < bm ? w[i0 + j] : bm;
      mn = bm;
    };
    quot(0);
    for (int64_t i0 = 0; i0 < N; i0 += BK) {                          // software pipeline: block i0+BK divides while block i0 adds
      if (i0 + BK < N) quot(i0 + BK);
      const int64_t m = N - i0 < BK ? N - i0 : BK;
      for (int64_t j = 0; j < m; ++j) { run += w[i0 + j]; cdf[i0 + j] = run; }
    }
    const double last = cdf[N - 1];
    if (!(last - 1.0 <= 1e-8 && 1.0 - last <= 1e-8) || mn < 0) return k;
    const double u = uniforms[k];
    int64_t lo = 0, hi = N;                                          // searchsorted(cdf / last, u, side='right')
    while (lo < hi) { const int64_t mid = lo + (hi - lo) / 2; if (cdf[mid] / last <= u) lo = mid + 1; else hi = mid; }
    out_idx[k] = lo;
    if (update_staleness) {                                          // staleness + 1, chosen seed -> 0: applied inside the next pass 1
      if (lo < N) st[lo] = -1.;
      aging = true;
    }
  }
  flush();
  return n;
}


// ---- per-kernel timing ------------------------------------------------------------------------------------------------------
#include <atomic>
#include <mutex>
namespace dcd {
namespace {
std::atomic<bool> g_prof_on{false};
std::mutex g_prof_mu;
std::vector<hipEvent_t> g_prof_ev;          // pairs
std::vector<int> g_prof_tag;                // tag of pair i
int g_prof_cap = 0;
}  // namespace
void prof_pair(int tag, hipEvent_t* start, hipEvent_t* stop) {
  if (!g_prof_on.load(std::memory_order_relaxed)) return;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  const int i = (int)g_prof_tag.size();
  if (!g_prof_on.load() || i >= g_prof_cap) return;
  g_prof_tag.push_back(tag);
  *start = g_prof_ev[2 * i]; *stop = g_prof_ev[2 * i + 1];
}
}  // namespace dcd

extern "C" {

int dcd_profile_begin(int32_t capacity) {
  if (capacity <= 0 || capacity > (1 << 20)) return dcd::fail("dcd_profile_begin: bad capacity %d", capacity);
  std::lock_guard<std::mutex> lk(dcd::g_prof_mu);
  if (dcd::g_prof_on.load()) return dcd::fail("dcd_profile_begin: a profile is already open (the state is process-global)");
  while ((int)dcd::g_prof_ev.size() < 2 * capacity) {
    hipEvent_t e;
    hipError_t err = hipEventCreate(&e);
    if (err != hipSuccess) return dcd::fail("dcd_profile_begin: hipEventCreate: %s", hipGetErrorString(err));
    dcd::g_prof_ev.push_back(e);
  }
  dcd::g_prof_tag.clear();
  dcd::g_prof_cap = capacity;
  dcd::g_prof_on.store(true);
  return 0;
}

int dcd_profile_end(int32_t n_tags, int32_t* n_launches, double* total_ms, double* max_ms, int32_t sample_cap, int32_t* sample_tags,
                    float* sample_ms, void* stream) {
  dcd::g_prof_on.store(false);
  hipError_t err = hipStreamSynchronize((hipStream_t)stream);
  if (err != hipSuccess) return dcd::fail("dcd_profile_end: hipStreamSynchronize: %s", hipGetErrorString(err));
  std::lock_guard<std::mutex> lk(dcd::g_prof_mu);
  for (int t = 0; t < n_tags; ++t) { if (n_launches) n_launches[t] = 0; if (total_ms) total_ms[t] = 0.0; if (max_ms) max_ms[t] = 0.0; }
  int32_t n_out = 0;                          // samples are COMPACTED: entry n_out is the n_out-th launch with a tag < n_tags
  for (size_t i = 0; i < dcd::g_prof_tag.size(); ++i) {
    const int t = dcd::g_prof_tag[i];
    if (t < 0 || t >= n_tags) continue;
    float ms = 0.f;
    err = hipEventElapsedTime(&ms, dcd::g_prof_ev[2 * i], dcd::g_prof_ev[2 * i + 1]);
    if (err != hipSuccess) return dcd::fail("dcd_profile_end: hipEventElapsedTime: %s", hipGetErrorString(err));
    if (n_launches) ++n_launches[t];
    if (total_ms) total_ms[t] += ms;
    if (max_ms && ms > max_ms[t]) max_ms[t] = ms;
    if (sample_tags && sample_ms && n_out < sample_cap) { sample_tags[n_out] = t; sample_ms[n_out] = ms; ++n_out; }
  }
  dcd::g_prof_tag.clear();
  return 0;
}

}  // extern "C"
