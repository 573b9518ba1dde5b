// tests/hostsim/hostsim.cpp — runs the DEVICE env logic (dcd-isaac_amd/csrc/mg_env.h) on the CPU.
//
// Debugging / sanitizer aid for the GPU-less build container: the very same __host__ __device__
// functions the kernels in mg_kernels.hip inline are driven here one env at a time, so the
// packed-record logic can be diffed against oracle/mg_oracle.c (and run under -fsanitize) before
// a GPU box is spent.  It covers the per-lane logic only — LDS staging, coalesced emission and
// write-back live in the kernels and are covered by the `-m gpu` parity tests.
// TEST-ONLY: nothing under dcd-isaac_amd/ loads this library; the product has no CPU path.
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "../../dcd-isaac_amd/csrc/mg_env.h"

using namespace DCD_NS;

struct SimMg {
  DcdMgConfig cfg;
  Params p;
  std::vector<uint8_t> rec;
  std::vector<uint32_t> seed;
  std::vector<Metrics> met;
  uint32_t err_bits = 0;
  int first_env = -1;
};

static float div10(uint8_t k) { return (float)((double)k / 10.0); }
static void note(SimMg* h, int env, uint32_t bits) {
  if (!bits) return;
  if (h->first_env < 0 || env < h->first_env) h->first_env = env;
  h->err_bits |= bits;
}
static void emit_student_row(SimMg* h, Env& e, int row, const DcdStudentObs* o, const uint8_t* img_in) {
  if (!o) return;
  const int V = h->p.V, E = V * V * 3;
  uint8_t img[DCD_MAX_VIEW * DCD_MAX_VIEW * 3 + 1];
  if (img_in) memcpy(img, img_in, E + 1);
  else if (e.has_pos()) { render_view(e, h->p, img); img[E] = (uint8_t)e.adir; }
  else memset(img, 0, sizeof img);
  if (o->image_u8) memcpy(o->image_u8 + (size_t)row * E, img, E);
  if (o->dir_u8) o->dir_u8[row] = img[E];
  if (o->image_f32) {
    float* f = o->image_f32 + (size_t)row * E;
    for (int c = 0; c < 3; ++c) for (int k = 0; k < V * V; ++k) f[c * V * V + k] = div10(img[k * 3 + c]);
  }
  if (o->dir_f32) o->dir_f32[row] = (float)img[E];
}
static void emit_adv_row(SimMg* h, Env& e, int row, const DcdAdvObs* o, const float* z) {
  if (!o) return;
  const int S = h->p.S;
  for (int x = 0; x < S; ++x) for (int y = 0; y < S; ++y) {
    const uint32_t t = enc3(e.cell(x, y));
    for (int c = 0; c < 3; ++c) {
      const uint8_t v = (uint8_t)(t >> (8 * c));
      if (o->image_u8) o->image_u8[((size_t)row * S * S + x * S + y) * 3 + c] = v;
      if (o->image_f32) o->image_f32[(size_t)row * 3 * S * S + (c * S + x) * S + y] = div10(v);
    }
  }
  if (o->time_step) o->time_step[row] = (float)e.sb(6);
  if (o->random_z) memcpy(o->random_z + (size_t)row * h->p.random_z_dim, z, sizeof(float) * h->p.random_z_dim);
}

extern "C" {

SimMg* sim_mg_create(const DcdMgConfig* cfg) {
  if (!cfg || cfg->size < 3 || cfg->size > MAX_SIZE || !(cfg->agent_view_size & 1) || cfg->random_z_dim > 52) return nullptr;
  SimMg* h = new SimMg();
  h->cfg = *cfg;
  static const int NEDIT[3] = {2, 3, 4};
  Params& p = h->p;
  p.B = cfg->num_envs; p.S = cfg->size; p.V = cfg->agent_view_size; p.max_steps = cfg->max_steps;
  p.max_episode_steps = cfg->max_episode_steps; p.n_clutter = cfg->n_clutter; p.resample_n_clutter = cfg->resample_n_clutter;
  p.choose_goal_last = cfg->choose_goal_last; p.random_z_dim = cfg->random_z_dim; p.n_editor_actions = NEDIT[cfg->editor_actions];
  p.fixed_environment = cfg->fixed_environment; p.goal_noise = cfg->goal_noise; p.engine_seed = cfg->engine_seed;
  p.see_through_walls = cfg->see_through_walls != 0;
  init_reward_division(p);
  h->rec.assign((size_t)p.B * REC_BYTES, 0);
  h->seed.resize(p.B); h->met.resize(p.B);
  for (int i = 0; i < p.B; ++i) {
    h->seed[i] = (uint32_t)(cfg->env_index_base + i);
    Env e = make_env(&h->rec[(size_t)i * REC_BYTES], h->seed[i]);
    e.sb(7) = (uint8_t)(p.n_clutter + 2);
    env_reset(e, p, h->met[i], nullptr);
    e.store_dyn();
  }
  return h;
}
void sim_mg_destroy(SimMg* h) { delete h; }
void sim_mg_set_seed(SimMg* h, const int64_t* seeds) {
  for (int i = 0; i < h->p.B; ++i) {
    h->seed[i] = (uint32_t)seeds[i];
    Env e = make_env(&h->rec[(size_t)i * REC_BYTES], h->seed[i]);
    e.set_ctr(0);
  }
}
uint32_t sim_mg_poll_errors(SimMg* h, int32_t* first) {
  uint32_t b = h->err_bits; if (first) *first = h->first_env;
  h->err_bits = 0; h->first_env = -1; return b;
}
#define ENV(i) Env e = make_env(&h->rec[(size_t)(i) * REC_BYTES], h->seed[i]); Metrics& m = h->met[i]

void sim_mg_reset(SimMg* h, const DcdAdvObs* out) {
  float z[52];
  for (int i = 0; i < h->p.B; ++i) {
    ENV(i);
    env_reset(e, h->p, m, z);
    e.elapsed = 0; e.ep_ret = 0.f; e.ep_len = 0;
    e.store_dyn(); emit_adv_row(h, e, i, out, z); note(h, i, e.err);
  }
}
void sim_mg_step_adversary(SimMg* h, const int64_t* loc, const DcdAdvObs* out, uint8_t* done) {
  float z[52];
  for (int i = 0; i < h->p.B; ++i) {
    ENV(i);
    bool d = env_step_adversary(e, h->p, m, loc[i], z);
    if (done) done[i] = d;
    e.store_dyn(); emit_adv_row(h, e, i, out, z); note(h, i, e.err);
  }
}
void sim_mg_reset_agent(SimMg* h, const DcdStudentObs* out) {
  for (int i = 0; i < h->p.B; ++i) {
    ENV(i); (void)m;
    reset_agent(e);
    e.elapsed = 0; e.ep_ret = 0.f; e.ep_len = 0;
    e.store_dyn(); emit_student_row(h, e, i, out, nullptr); note(h, i, e.err);
  }
}
void sim_mg_reset_random(SimMg* h, const DcdStudentObs* out) {
  for (int i = 0; i < h->p.B; ++i) {
    ENV(i);
    env_reset_random(e, h->p, m);
    e.elapsed = 0; e.ep_ret = 0.f; e.ep_len = 0;
    e.store_dyn(); emit_student_row(h, e, i, out, nullptr); note(h, i, e.err);
  }
}
void sim_mg_reset_to_encoding(SimMg* h, const uint8_t* enc, const int32_t* idx, int32_t n, const DcdStudentObs* out) {
  const int S = h->p.S;
  for (int r = 0; r < n; ++r) {
    const int i = idx ? idx[r] : r;
    ENV(i);
    env_reset(e, h->p, m, nullptr);
    e.elapsed = 0;
    env_set_encoding(e, h->p, m, enc + (size_t)r * S * S * 3);
    e.store_dyn(); emit_student_row(h, e, r, out, nullptr); note(h, i, e.err);
  }
}
void sim_mg_reset_to_fixed_level(SimMg* h, const uint8_t* enc, const int32_t* idx, int32_t n, const DcdStudentObs* out) {
  const int S = h->p.S;
  for (int r = 0; r < n; ++r) {
    const int i = idx ? idx[r] : r;
    ENV(i);
    env_load_fixed_level(e, h->p, m, enc + (size_t)r * S * S * 3);
    e.elapsed = 0; e.ep_ret = 0.f; e.ep_len = 0;
    e.store_dyn(); emit_student_row(h, e, r, out, nullptr); note(h, i, e.err);
  }
}
void sim_mg_reset_to_actions(SimMg* h, const int64_t* actions, const int32_t* lens, int32_t max_len, const int32_t* idx, int32_t n,
                             const DcdStudentObs* out) {
  for (int r = 0; r < n; ++r) {
    const int i = idx ? idx[r] : r;
    ENV(i);
    env_reset(e, h->p, m, nullptr);
    e.elapsed = 0;
    if (h->p.resample_n_clutter) e.sb(7) = (uint8_t)lens[r];
    for (int k = 0; k < lens[r]; ++k)
      if (env_step_adversary(e, h->p, m, actions[(size_t)r * max_len + k], nullptr)) reset_agent(e);
    e.store_dyn(); emit_student_row(h, e, r, out, nullptr); note(h, i, e.err);
  }
}
void sim_mg_mutate_level(SimMg* h, int32_t num_edits, const DcdStudentObs* out) {
  for (int i = 0; i < h->p.B; ++i) {
    ENV(i);
    env_mutate_level(e, h->p, m, num_edits);
    e.store_dyn(); emit_student_row(h, e, i, out, nullptr); note(h, i, e.err);
  }
}
void sim_mg_step(SimMg* h, const int64_t* actions, int32_t flags, const DcdStepOut* out) {
  const int E = h->p.V * h->p.V * 3;
  for (int i = 0; i < h->p.B; ++i) {
    ENV(i);
    bool md = false;
    uint8_t term[DCD_MAX_VIEW * DCD_MAX_VIEW * 3 + 1];
    StepResult r = env_step(e, h->p, m, md, actions[i], flags, term);
    e.store_dyn();
    note(h, i, e.err);
    if (!out) continue;
    emit_student_row(h, e, i, &out->obs, nullptr);
    if (out->reward) out->reward[i] = r.reward;
    if (out->done) out->done[i] = r.done;
    if (out->info) out->info[i] = r.info;
    if (out->ep_ret) out->ep_ret[i] = r.ep_ret;
    if (out->ep_len) out->ep_len[i] = r.ep_len;
    if (r.have_term) { (void)E; emit_student_row(h, e, i, &out->term_obs, term); }
    if (out->masks) out->masks[i] = r.done ? 0.f : 1.f;
    if (out->bad_masks) out->bad_masks[i] = (r.info & DCD_INFO_TRUNC_KEY) ? 0.f : 1.f;
    if (out->cliffhanger_masks) out->cliffhanger_masks[i] = (r.info & DCD_INFO_CLIFFHANGER) ? 0.f : 1.f;
  }
}
void sim_mg_get_encodings(SimMg* h, uint8_t* out) {
  const int S = h->p.S;
  for (int i = 0; i < h->p.B; ++i) {
    ENV(i); (void)m;
    for (int x = 0; x < S; ++x) for (int y = 0; y < S; ++y) {
      const uint32_t t = enc3(e.cell(x, y));
      uint8_t* o = out + ((size_t)i * S * S + x * S + y) * 3;
      o[0] = (uint8_t)t; o[1] = (uint8_t)(t >> 8); o[2] = (uint8_t)(t >> 16);
    }
  }
}
void sim_mg_get_metrics(SimMg* h, int32_t* ncl, int32_t* pas, int32_t* spl, int32_t* dist) {
  for (int i = 0; i < h->p.B; ++i) {
    const Metrics& m = h->met[i];
    if (ncl) ncl[i] = m.n_clutter_placed;
    if (pas) pas[i] = m.passable;
    if (spl) spl[i] = m.shortest_path_length;
    if (dist) dist[i] = m.distance_to_goal;
  }
}
void sim_mg_get_state(SimMg* h, int32_t* out) {
  for (int i = 0; i < h->p.B; ++i) {
    ENV(i); (void)m;
    int32_t* o = out + (size_t)i * 16;
    o[0] = e.has_pos() ? e.ax : -1; o[1] = e.has_pos() ? e.ay : -1; o[2] = e.adir;
    o[3] = e.carrying ? (int32_t)enc3((uint8_t)e.carrying) : 0;
    o[4] = e.step_count; o[5] = e.elapsed; o[6] = e.ep_len;
    float f = e.ep_ret; memcpy(&o[7], &f, 4);
    o[8] = e.sb(0) != NOPOS ? e.sb(0) : -1; o[9] = e.sb(0) != NOPOS ? e.sb(1) : -1; o[10] = e.sb(2);
    o[11] = e.sb(3) != NOPOS ? e.sb(3) : -1; o[12] = e.sb(3) != NOPOS ? e.sb(4) : -1;
    o[13] = e.sb(6); o[14] = e.sb(7); o[15] = (int32_t)e.ctr();
  }
}

}  // extern "C"
