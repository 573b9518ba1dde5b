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
#include <vector>
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

}  // namespace

extern "C" int dcd_action_complexity(const int64_t* actions, const float* masks, int32_t T, int32_t B, double* total,
                                     int64_t* n_episodes, float* per_env) {
  if (!actions || !masks || T <= 0 || B <= 0) return dcd::fail("dcd_action_complexity: NULL pointer or empty rollout");
  std::vector<char> str((size_t)T * 21 + 1);
  std::vector<int> cuts(T + 3);
  std::vector<Slot> table(4096);
  double tot = 0; int64_t n = 0;
  for (int b = 0; b < B; ++b) {
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
  if (total) *total = tot;
  if (n_episodes) *n_episodes = n;
  return 0;
}
