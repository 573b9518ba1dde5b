/*
 * dcd_hip.h — C ABI of the MI355X-native DCD rollout-and-regret engine.
 *
 * The reference (linjiw/dcd-isaac, pure Python) has no FFI seam for this path:
 * its boundary is the Python vec-env / storage / sampler API.  This header is
 * the C-ABI replacement that sits *under* that Python API: each
 * entry point names the reference function(s) it replaces (file:line into the
 * reference tree).  The ctypes stub that binds it is dcd-isaac_amd/dcd_hip/
 * __init__.py; INTEGRATION.md shows the reference-side call sites.
 *
 * Conventions
 *  - every data pointer is a caller-owned DEVICE pointer (hipMalloc / a
 *    PyTorch-ROCm tensor's data_ptr()); the library never frees or retains it
 *    beyond the call.  NULL output pointers mean "do not emit".
 *  - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream()
 *    .cuda_stream); every call is asynchronous on it.  No call synchronises
 *    except dcd_mg_poll_errors().
 *  - return value: 0 = ok, non-zero = error; dcd_last_error() returns a
 *    thread-local message.  Device-side semantic errors (bad teacher action,
 *    missing start position, malformed encoding: the reference's ValueError /
 *    KeyError / AssertionError sites) are latched per env in an error word that
 *    dcd_mg_poll_errors() reads back.
 *  - handles are not thread-safe: one caller thread per handle (the reference's
 *    learner is single-threaded, train.py:39).
 *  - integer outputs are bit-exact against oracle/ (the CPU restatement);
 *    fp32 scans follow the reference's evaluation order with FMA contraction
 *    off (see each function).
 */
#ifndef DCD_HIP_H
#define DCD_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DCD_ABI_VERSION 2

/* ---- cell encoding constants (gym-minigrid 1.0.1 OBJECT_TO_IDX etc.; SURVEY
 * Appendix A.2; used by envs/multigrid/multigrid.py:58,76-101) ------------- */
enum {
  DCD_OBJ_UNSEEN = 0, DCD_OBJ_EMPTY = 1, DCD_OBJ_WALL = 2, DCD_OBJ_FLOOR = 3,
  DCD_OBJ_DOOR = 4, DCD_OBJ_KEY = 5, DCD_OBJ_BALL = 6, DCD_OBJ_BOX = 7,
  DCD_OBJ_GOAL = 8, DCD_OBJ_LAVA = 9, DCD_OBJ_AGENT = 10
};
enum { DCD_ACT_LEFT = 0, DCD_ACT_RIGHT = 1, DCD_ACT_FORWARD = 2, DCD_ACT_PICKUP = 3,
       DCD_ACT_DROP = 4, DCD_ACT_TOGGLE = 5, DCD_ACT_DONE = 6 };

#define DCD_MAX_GRID 127  /* adversarial ids: 6, 10, 15 (the tuned 256-byte record); evaluation mazes up to 101 use wider records */
#define DCD_MAX_VIEW 7

/* per-env error bits latched on device (dcd_mg_poll_errors) */
enum {
  DCD_ERR_BAD_ACTION      = 1u << 0, /* multigrid.py:939 assert False 'unknown action' */
  DCD_ERR_BAD_ADV_ACTION  = 1u << 1, /* adversarial.py:465-466 ValueError */
  DCD_ERR_NO_START_POS    = 1u << 2, /* adversarial.py:248-249 ValueError */
  DCD_ERR_BAD_ENCODING    = 1u << 3, /* multigrid.py:73-105 KeyError / assert */
  DCD_ERR_NO_FREE_CELL    = 1u << 4, /* multigrid.py:597-599 RetriesExceeded / np.random.choice on empty */
  DCD_ERR_OUT_OF_GRID     = 1u << 5, /* gym_minigrid Grid.get assert (agent left the grid) */
  DCD_ERR_BAD_START_CELL  = 1u << 6  /* adversarial.py:259-261 ValueError */
};

/* flags for dcd_mg_step */
enum {
  DCD_STEP_RESET_RANDOM = 1 << 0, /* worker 'step_env_reset_random' (parallel_wrappers.py:31-33) */
  DCD_STEP_LAST         = 1 << 1  /* runner's last rollout step: force done, mark cliffhangers
                                      (adversarial_runner.py:521-530) */
};

/* bits of StepOut.info */
enum {
  DCD_INFO_EPISODE     = 1 << 0, /* info['episode'] present (vec_monitor.py:68-83) */
  DCD_INFO_TRUNC_KEY   = 1 << 1, /* 'truncated' key present (time_limit.py:28-31, runner :526) */
  DCD_INFO_TRUNCATED   = 1 << 2, /* info['truncated'] value */
  DCD_INFO_CLIFFHANGER = 1 << 3  /* info['cliffhanger'] (adversarial_runner.py:525) */
};

/* AdversarialEnv.__init__ (envs/multigrid/adversarial.py:67-149) + TimeLimit
 * (envs/wrappers/time_limit.py:15-22) + create_parallel_env seeds
 * (util/__init__.py:213-217). */
typedef struct DcdMgConfig {
  int32_t  num_envs;            /* B on this device */
  int32_t  size;                /* grid width == height, 3..127; teacher-side calls need size <= 15 */
  int32_t  agent_view_size;     /* V, odd, 3..7 (all registered ids: 5) */
  int32_t  max_steps;           /* MultiGridEnv.max_steps (multigrid.py:448,972) */
  int32_t  max_episode_steps;   /* TimeLimit; <=0: no wrapper */
  int32_t  n_clutter;
  int32_t  resample_n_clutter;
  int32_t  choose_goal_last;
  int32_t  see_through_walls;   /* 0: views masked by Grid.process_vis (the ...OpaqueWalls... ids, multigrid.py:1000-1002) */
  int32_t  random_z_dim;        /* <= 64 */
  int32_t  editor_actions;      /* 0 walls_none, 1 walls_none_goal, 2 walls_none_agent_goal */
  int32_t  fixed_environment;
  float    goal_noise;
  uint32_t engine_seed;         /* Philox key word 1 (args.seed) */
  int64_t  env_index_base;      /* global index of local env 0 (rank * B): default per-env seed */
  int32_t  replicas;            /* 0/1: one copy of the batch.  R > 1: R copies in one allocation (dcd_mg_replica,
                                   dcd_mg_step_pair): PAIRED's protagonist and antagonist play the same levels in
                                   independent rollouts (adversarial_runner.py:709-733) */
  int32_t  reserved;
} DcdMgConfig;
#define DCD_MAX_REPLICAS 4

typedef struct DcdMg DcdMg;     /* opaque: one per (device, env batch) */

/* Student observation of the whole batch (or of n selected envs, row r of each
 * output = r-th selected env).  gen_obs (multigrid.py:1017-1055) followed by
 * VecPreprocessImageWrapper._preprocess (obs_wrappers.py:88-115). */
typedef struct DcdStudentObs {
  uint8_t* image_u8;   /* [n,V,V,3]  raw encoding, indexed [vx][vy][c]        */
  uint8_t* dir_u8;     /* [n]                                                 */
  float*   image_f32;  /* [n,3,V,V]  = u8/10.0 (fp32 division), CHW           */
  float*   dir_f32;    /* [n,1]                                               */
} DcdStudentObs;

/* Teacher observation (adversarial.py:222-229,532-537 + obs_wrappers.py:183). */
typedef struct DcdAdvObs {
  uint8_t* image_u8;    /* [B,S,S,3] full-grid encoding [x][y][c]             */
  float*   image_f32;   /* [B,3,S,S] = u8/10.0, CHW                           */
  float*   time_step;   /* [B,1]                                              */
  float*   random_z;    /* [B,random_z_dim] U[0,1)                            */
} DcdAdvObs;

/* Everything worker.step_env + TimeLimit + VecMonitor + VecPreprocess return
 * for one step, plus the masks the runner derives (adversarial_runner.py:566-573). */
typedef struct DcdStepOut {
  DcdStudentObs obs;        /* post-auto-reset obs (parallel_wrappers.py:27-37)         */
  float*   reward;          /* [B]   fp32(fp64 reward)  (obs_wrappers.py:179)           */
  uint8_t* done;            /* [B]                                                      */
  uint8_t* info;            /* [B]   DCD_INFO_* bits                                    */
  float*   ep_ret;          /* [B]   info['episode']['r'] where DCD_INFO_EPISODE else 0 */
  int32_t* ep_len;          /* [B]   info['episode']['l']                               */
  DcdStudentObs term_obs;   /* info['truncated_obs'], rows written ONLY where the
                               runner would insert it (episode && truncated_obs:
                               adversarial_runner.py:546-549); other rows untouched   */
  float*   masks;           /* [B]   0 if done else 1                                   */
  float*   bad_masks;       /* [B]   0 if 'truncated' key present else 1                */
  float*   cliffhanger_masks;/*[B]   0 if 'cliffhanger' key present else 1              */
} DcdStepOut;

const char* dcd_last_error(void);
int  dcd_abi_version(void);

/* ---- lifetime ------------------------------------------------------------ */
/* envs.registration.make(id) × B  +  ParallelAdversarialVecEnv.__init__
 * (envs/registration.py:109-125, envs/wrappers/parallel_wrappers.py:103-145).
 * `device` is the HIP device ordinal.  Allocates B × 256-byte env records. */
int dcd_mg_create(const DcdMgConfig* cfg, int device, DcdMg** out);
int dcd_mg_destroy(DcdMg* h);
/* Replica k (1 <= k < cfg.replicas) of the batch behind `h` (the handle dcd_mg_create returned, which is replica 0):
 * a non-owning view that every other entry point accepts like a handle of its own; destroy it before `h`. */
int dcd_mg_replica(DcdMg* h, int32_t k, DcdMg** out);
/* dst := src (grids, start / goal cells, agent and episode state, Philox keys and draw counters, metrics).  In the
 * reference protagonist and antagonist step the SAME env objects one rollout after the other
 * (adversarial_runner.py:709-733); each rollout starts with reset_agent (:484), so on levels made of walls and a goal
 * (everything the teacher, reset_random and the editor can build) playing two copies side by side is the same game. */
int dcd_mg_clone_levels(DcdMg* dst, DcdMg* src, void* stream);
/* venv.set_seed(seeds) → MultiGridEnv.seed (parallel_wrappers.py:414-415,
 * multigrid.py:465-468).  seeds: device int64[B]; resets each env's draw counter. */
int dcd_mg_set_seed(DcdMg* h, const int64_t* seeds, void* stream);
/* Reads back (synchronises `stream`) the OR of all per-env error words and the
 * index of the first offending env; clears them.  first_env may be NULL. */
int dcd_mg_poll_errors(DcdMg* h, uint32_t* err_bits, int32_t* first_env, void* stream);

/* ---- teacher / level construction ---------------------------------------- */
/* AdversarialEnv.reset (adversarial.py:194-229) via venv.reset() */
int dcd_mg_reset(DcdMg* h, const DcdAdvObs* out, void* stream);
/* AdversarialEnv.step_adversary (adversarial.py:452-539); loc: int64[B];
 * done: uint8[B] (may be NULL). */
int dcd_mg_step_adversary(DcdMg* h, const int64_t* loc, const DcdAdvObs* out,
                          uint8_t* done, void* stream);
/* AdversarialEnv.reset_agent (adversarial.py:238-269) + TimeLimit/VecMonitor
 * resets (time_limit.py:47-49, vec_monitor.py:42-46) for the whole batch. */
int dcd_mg_reset_agent(DcdMg* h, const DcdStudentObs* out, void* stream);
/* AdversarialEnv.reset_random (adversarial.py:541-581) for the whole batch. */
int dcd_mg_reset_random(DcdMg* h, const DcdStudentObs* out, void* stream);
/* reset_to_level with a byte encoding (adversarial.py:271-294, multigrid.py:264-280).
 * enc: uint8[n,S,S,3]; idx: int32[n] env indices or NULL (n == B, identity). */
int dcd_mg_reset_to_encoding(DcdMg* h, const uint8_t* enc, const int32_t* idx, int32_t n,
                             const DcdStudentObs* out, void* stream);
/* MultiGridEnv.reset() of a FIXED level family — the zero-shot evaluation mazes (envs/multigrid/maze.py:76-94
 * as _gen_grid, multigrid.py:470-502; eval.py:250-329 drives them): the grid becomes the given encoding, the
 * agent starts on its cell facing the ENCODED direction (no random draw), episode / TimeLimit / VecMonitor
 * counters are zeroed.  Same argument conventions as dcd_mg_reset_to_encoding. */
int dcd_mg_reset_to_fixed_level(DcdMg* h, const uint8_t* enc, const int32_t* idx, int32_t n,
                                const DcdStudentObs* out, void* stream);
/* reset_to_level with an action string (adversarial.py:271-288): actions
 * int64[n,max_len] (row-major), lens int32[n]; replays reset() + step_adversary
 * per action (+ reset_agent when the design episode ends) in ONE launch. */
int dcd_mg_reset_to_actions(DcdMg* h, const int64_t* actions, const int32_t* lens,
                            int32_t max_len, const int32_t* idx, int32_t n,
                            const DcdStudentObs* out, void* stream);
/* The two resets above for a MASKED subset of the batch, levels taken from a device-resident TABLE (the runner's mirror of the
 * level store): level_idx int32[B]; env i is reset to level table[level_idx[i]] when level_idx[i] >= 0 and left untouched —
 * record, metrics, outputs — otherwise.  `out` rows are indexed by env (the caller passes the rollout buffers' next-step rows:
 * observations of re-sampled envs land in place).  Fixed launch shape whatever the number of resets: what a captured replay
 * rollout needs (adversarial_runner.py:551-558 with the host decisions replaced by dcd_replay_pop's).
 * table: uint8[rows,S,S,3] encodings, or int64[rows,max_len] teacher actions with table_lens int32[rows]. */
int dcd_mg_reset_to_encoding_rows(DcdMg* h, const uint8_t* table, const int32_t* level_idx, const DcdStudentObs* out, void* stream);
int dcd_mg_reset_to_actions_rows(DcdMg* h, const int64_t* table, const int32_t* table_lens, int32_t max_len, const int32_t* level_idx,
                                 const DcdStudentObs* out, void* stream);
/* Per-episode re-sampling of a replay rollout without the host (adversarial_runner.py:551-558; level_sampler.py:664-680).
 * The reference draws one replay level per finished env, in env order, between two steps; the draws are one sequential stream
 * of the sampler, speculated ahead by the host (LevelSampler.prefetch_next_update): queue_rows / queue_seeds int32[K] hold its
 * next K entries (table row and level seed).  In env order, the k-th env with DCD_INFO_EPISODE set in ep_info uint8[B] takes
 * entry *cursor + k: level_idx[env] = its table row, cur_seed[env] = its seed, solved[env] = 1 (optional); other envs get
 * level_idx -1.  *cursor advances by the number of finished envs; entries past K set *overflow (those envs get -1): the caller
 * must then discard the rollout and redo it with host draws. */
int dcd_replay_pop(const uint8_t* ep_info, int32_t B, const int32_t* queue_rows, const int32_t* queue_seeds, int32_t K, int32_t* cursor,
                   int32_t* overflow, int32_t* level_idx, int32_t* cur_seed, uint8_t* solved, void* stream);
/* AdversarialEnv.mutate_level (adversarial.py:317-397), whole batch. */
int dcd_mg_mutate_level(DcdMg* h, int32_t num_edits, const DcdStudentObs* out, void* stream);

/* ---- the student step ----------------------------------------------------- */
/* worker.step_env (parallel_wrappers.py:27-37) ∘ TimeLimit.step (time_limit.py:24-33)
 * ∘ MultiGridEnv.step / step_one_agent / gen_obs (multigrid.py:943-975,866-941,
 * 1017-1055) ∘ VecMonitor.step_wait (vec_monitor.py:60-85) ∘ VecPreprocessImageWrapper
 * .step_wait (obs_wrappers.py:168-181) [∘ last-step rule adversarial_runner.py:521-530
 * when DCD_STEP_LAST].  actions: int64[B]. */
int dcd_mg_step(DcdMg* h, const int64_t* actions, int32_t flags, const DcdStepOut* out,
                void* stream);
/* dcd_mg_step of two consecutive replicas in ONE launch (h1 = replica k+1 of h0's allocation): the two independent
 * agent_rollout calls of a PAIRED update (adversarial_runner.py:709-733) advance in lock-step, each with its own
 * action array and output set.  Identical results to the two dcd_mg_step calls. */
int dcd_mg_step_pair(DcdMg* h0, DcdMg* h1, const int64_t* actions0, const int64_t* actions1, int32_t flags,
                     const DcdStepOut* out0, const DcdStepOut* out1, void* stream);

/* Per-launch timing of the step kernel for the roofline line of bench.py: between begin and end
 * every dcd_mg_step launch on this handle is bracketed by HIP events attached to the dispatch
 * itself (hipExtLaunchKernelGGL start/stop events: the same begin/end timestamps rocprofv3
 * --kernel-trace reports).  end() synchronises the stream and returns the number of launches
 * and their summed duration in milliseconds.  capacity: max launches recorded. */
int dcd_mg_profile_begin(DcdMg* h, int32_t capacity);
int dcd_mg_profile_end(DcdMg* h, int32_t* n_launches, double* total_ms, void* stream);

/* Per-kernel timing of a whole hot-path pass (bench.py's `kernel_times`; no reference counterpart — the reference has no
 * kernels): between begin and end every launch of the pass kernels, from any entry point of this library, carries a HIP
 * start / stop event pair attached to the dispatch (the kernel's own begin -> end timestamps, as rocprofv3 --kernel-trace
 * reads them, with no tracer attached).  Tags: 0 mg_step_kernel (dcd_mg_step / dcd_mg_step_pair), 1 mg_level_kernel
 * (reset / step_adversary / reset_agent / ...), 2 rollout_tail_kernel or gae_kernel (dcd_rollout_tail_batch,
 * dcd_gae_returns*), 3 seg_score_lds_kernel / seg_score_kernel (dcd_episode_scores*), 4 the segment count / offset
 * kernels of the scoring when they run as their own launches, 5 return statistics / regret / value loss / clone.
 * end() synchronises `stream` and fills, per tag t < n_tags: launches, summed and maximum duration in ms.
 * Process-wide, for measurement runs: do not profile while other host threads launch. */
#define DCD_PROF_NTAGS 6
int dcd_profile_begin(int32_t capacity);
int dcd_profile_end(int32_t n_tags, int32_t* n_launches, double* total_ms, double* max_ms, int32_t sample_cap, int32_t* sample_tags,
                    float* sample_ms, void* stream);   /* sample_*: optional; the launches with a tag < n_tags in issue order,
                                                          COMPACTED (entry k = the k-th such launch: tag, duration in ms),
                                                          sum(n_launches) entries (at most sample_cap).  begin() fails while a
                                                          profile is open: the state is process-global. */

/* ---- state readback -------------------------------------------------------- */
/* venv.get_encodings (parallel_wrappers.py:420-421 → adversarial.py:162-164):
 * uint8[B,S,S,3]. */
int dcd_mg_get_encodings(DcdMg* h, uint8_t* out, void* stream);
/* get_num_blocks / get_passable / get_shortest_path_length / get_distance_to_goal
 * (parallel_wrappers.py:424-431,445-446; adversarial.py:184-192,407-447).
 * each int32[B]; passable is -1 (unset) / 0 / 1. */
int dcd_mg_get_metrics(DcdMg* h, int32_t* n_clutter_placed, int32_t* passable,
                       int32_t* shortest_path_length, int32_t* distance_to_goal, void* stream);
/* Test/diagnostic readback, int32[B,16]:
 *  0 agent_x (-1 none) 1 agent_y 2 agent_dir 3 carrying (type|color<<8|state<<16, 0 none)
 *  4 step_count 5 elapsed_steps 6 ep_len 7 ep_ret (fp32 bits) 8 start_x (-1 none)
 *  9 start_y 10 start_dir 11 goal_x (-1 none) 12 goal_y 13 adversary_step_count
 *  14 adversary_max_steps 15 rng draw counter */
int dcd_mg_get_state(DcdMg* h, int32_t* out, void* stream);

/* ---- RolloutStorage scans (algos/storage.py) ------------------------------- */
/* compute_gae_returns (storage.py:233-256).  rewards [T,B], value_preds [T+1,B]
 * (row T already holds next_value; pass truncated_value_preds when
 * use_proper_time_limits), masks [T+1,B], returns [T+1,B] (rows 0..T-1 written).
 * Evaluation order per element, every op rounded to fp32, no FMA:
 *   delta = (r + ((gamma*v[t+1])*m[t+1])) - v[t];
 *   gae   = delta + (((float)(gamma*lambda) * m[t+1]) * gae);   ret = gae + v[t]. */
int dcd_gae_returns(const float* rewards, const float* value_preds, const float* masks,
                    float* returns, int32_t T, int32_t B, double gamma, double gae_lambda,
                    void* stream);
/* The same scan for n <= DCD_MAX_SCAN_SETS independent rollouts of one shape in ONE launch (PAIRED: the protagonist's and
 * the antagonist's storages, adversarial_runner.py:604-605 after each of the two rollouts); pointer arrays in HOST memory. */
#define DCD_MAX_SCAN_SETS 2
int dcd_gae_returns_batch(int32_t n, const float* const* rewards, const float* const* value_preds, const float* const* masks,
                          float* const* returns, int32_t T, int32_t B, double gamma, double gae_lambda, void* stream);
/* compute_discounted_returns (storage.py:258-276): ret[T] = v[T];
 * ret[t] = ((ret[t+1]*gamma)*m[t+1]) + r[t]. */
int dcd_discounted_returns(const float* rewards, const float* value_preds, const float* masks,
                           float* returns, int32_t T, int32_t B, double gamma, void* stream);
/* get_batched_value_loss (storage.py:290-327): out[B] = clamp(mean_t f(R-V)^power, -1, 1).
 * mode 0 abs, 1 signed, 2 positive-only. */
int dcd_batched_value_loss(const float* returns, const float* value_preds, float* out,
                           int32_t T, int32_t B, int32_t mode, int32_t power, int32_t clipped,
                           void* stream);

/* ---- LevelSampler episode scoring (level_replay/level_sampler.py:486-578) ---- */
enum {
  DCD_SCORE_VALUE_L1 = 0,            /* :318-327 */
  DCD_SCORE_POSITIVE_VALUE_LOSS = 1, /* :340-349 */
  DCD_SCORE_SIGNED_VALUE_LOSS = 2,   /* :329-338  (== gae :307-316) */
  DCD_SCORE_POLICY_ENTROPY = 3,      /* :278-287.  The reference's expression `-exp(l) * l.sum(-1) / max_entropy`
                                        broadcasts [L,A] against [L]: defined only for episodes of L == 1 or L == A
                                        steps, where it is reproduced as written (entry (i,j) times the row sum of step
                                        j; mean / max over all L*A entries; golden: tests/golden/entropy.npz).  For any
                                        other L the reference raises; those episodes score the per-step entropy
                                        mean_t / max_t of sum_a(-p log p) / max_entropy. */
  DCD_SCORE_LEAST_CONFIDENCE = 4,    /* :289-296 */
  DCD_SCORE_MIN_MARGIN = 5,          /* :298-305 */
  DCD_SCORE_ONE_STEP_TD = 6,         /* :426-439 */
  DCD_SCORE_UNIFORM = 7,             /* :275-276 */
  DCD_SCORE_RETURN_SUM = 8           /* rewards[start:t].sum() only (grounded_* strategies :533) */
};
/* One record per (env, episode segment), emitted in the reference's iteration
 * order (actor-major, then time): level_sampler.py:497-578. */
typedef struct DcdEpisodeRec {
  int32_t actor;        /* env index on this device                         */
  int32_t seed;         /* level_seeds[start_t, actor]       (:507)          */
  int32_t start_t;      /* first step of the segment                        */
  int32_t num_steps;    /* len(episode_logits)               (:541,:572)     */
  float   mean_score;   /* score_function(...)[0]                            */
  float   max_score;    /* score_function(...)[1]                            */
  float   reward_sum;   /* rewards[start:t].sum()  (grounded value, :533)    */
  int32_t flags;        /* bit0 done (full episode) / 0 trailing partial;
                           bit1 cliffhanger[t] (score update skipped, :528)  */
} DcdEpisodeRec;
/* Segments every env's [0,T) at masks[t]==0 (t>=1), scores each segment.
 * returns/value_preds [T+1,B]; rewards [T,B]; masks, cliffhanger_masks [T+1,B];
 * level_seeds int32 [T,B]; action_log_dist [T,B,A] (may be NULL for value
 * strategies).  recs: capacity max_recs; n_recs: int32[1] total written;
 * workspace: int32[2*B + 64] scratch.  Order of recs == reference order. */
int dcd_episode_scores(const float* returns, const float* value_preds, const float* rewards,
                       const float* masks, const float* cliffhanger_masks,
                       const int32_t* level_seeds, const float* action_log_dist,
                       int32_t T, int32_t B, int32_t A, int32_t strategy, double gamma,
                       DcdEpisodeRec* recs, int32_t max_recs, int32_t* n_recs,
                       int32_t* workspace, void* stream);

/* dcd_episode_scores for n <= DCD_MAX_SCAN_SETS rollouts of one shape in one launch sequence (one record buffer each). */
typedef struct DcdEpisodeScoreArgs {
  const float* returns; const float* value_preds; const float* rewards; const float* masks; const float* cliffhanger_masks;
  const int32_t* level_seeds; const float* action_log_dist;
  DcdEpisodeRec* recs; int32_t max_recs; int32_t* n_recs; int32_t* workspace;
} DcdEpisodeScoreArgs;
int dcd_episode_scores_batch(int32_t n, const DcdEpisodeScoreArgs* sets, int32_t T, int32_t B, int32_t A, int32_t strategy,
                             double gamma, void* stream);

/* grounded_signed/positive_value_loss (level_sampler.py:351-424), second pass once the host
 * has resolved each record's grounded value g = max(grounded_values[seed_idx], reward_sum)
 * (:533-538): adv_t = g - V[t, actor] over the record's steps (dense: first step only),
 * optionally clamp(0); mean_out/max_out[k] for record k (total_steps/new_steps scaling is host
 * side).  grounded: float[n]. */
int dcd_grounded_scores(const DcdEpisodeRec* recs, int32_t n, const float* grounded,
                        const float* value_preds, int32_t B, int32_t positive_only, int32_t dense,
                        float* mean_out, float* max_out, void* stream);

/* ---- regret ---------------------------------------------------------------- */
/* _get_rollout_return_stats (adversarial_runner.py:257-271): per env mean and max
 * of the episode returns emitted during a rollout.  ep_ret [T,B] (the per-step
 * StepOut.ep_ret rows), info uint8 [T,B] (DCD_INFO_EPISODE marks valid entries).
 * mean is float(np.mean(float32 list)) → computed in fp32 pairwise-free
 * sequential fp64 accumulate, rounded to fp32 (tolerance 1 ulp documented). */
int dcd_rollout_return_stats(const float* ep_ret, const uint8_t* info, int32_t T, int32_t B,
                             float* mean_return, float* max_return, int32_t* n_episodes,
                             void* stream);
int dcd_rollout_return_stats_batch(int32_t n, const float* const* ep_ret, const uint8_t* const* info, int32_t T, int32_t B,
                                   float* const* mean_return, float* const* max_return, int32_t* const* n_episodes,
                                   void* stream);
/* Everything that follows a pair of rollouts, in TWO launches instead of four: GAE (dcd_gae_returns_batch), return statistics
 * (dcd_rollout_return_stats_batch) and — when `score` is given and its strategy takes the staged path — the episode segment
 * counts of dcd_episode_scores_batch side by side in one launch (they do not depend on each other and each is
 * latency-bound), then the episode scoring, which reads the returns just written.  score may be NULL (no level replay):
 * one launch.  Results are those of the three separate calls, bit for bit. */
int dcd_rollout_tail_batch(int32_t n, const float* const* rewards, const float* const* value_preds, const float* const* masks,
                           float* const* returns, const float* const* ep_ret, const uint8_t* const* info,
                           float* const* mean_return, float* const* max_return, int32_t* const* n_episodes,
                           const DcdEpisodeScoreArgs* score, int32_t T, int32_t B, int32_t A, int32_t strategy, double gamma,
                           double gae_lambda, void* stream);
/* _compute_env_return (adversarial_runner.py:637-674), mode 0 paired
 * max(antag_max - protag_mean, 0); 1 flexible_paired; 2 minimax (-protag_max). */
int dcd_env_return(const float* agent_mean, const float* agent_max, const float* adv_mean,
                   const float* adv_max, float* out, int32_t B, int32_t mode, void* stream);

/* ---- return normalisation (BASELINE configs 4-5 train with --normalize_returns) -------------------------------- */
/* VecNormalize(ob=False, ret=True).step_wait (envs/wrappers/vec_normalize.py:37-45) on the batch's rewards, between the
 * episode accounting (VecMonitor sees the raw rewards) and the preprocessing wrapper, as util/__init__.py:191-193 stacks
 * them: ret = ret*gamma + rewards; ret_rms.update(ret); rewards = clip(rewards / sqrt(ret_rms.var + epsilon), +-cliprew);
 * ret[done] = 0.  rewards float[B] in place; done uint8[B]; ret double[B] (the wrapper's running returns, zeroed by
 * the caller on reset*); rms double[3] = (mean, var, count) of baselines' RunningMeanStd (initially 0, 1, 1e-4). */
int dcd_return_normalize(float* rewards, const uint8_t* done, double* ret, double* rms, double gamma, double epsilon,
                         double cliprew, int32_t B, void* stream);

/* ---- Bézier track of a CarRacing level (BASELINE config 5) ------------------------------------------------------ */
/* envs/box2d/bezier.py:get_bezier_curve (:56-82) for a batch of levels, as CarRacingBezier._create_track_bezier calls it
 * (envs/box2d/car_racing_bezier.py:297-303: rad 0.2, edgy 0.2, numpoints 40): points double[B,max_points,2] (the first
 * n_points[b] rows of level b are its control points, 2 <= n <= 64), sorted counter-clockwise about their centroid, tangent
 * angles from the adjacent chords, one cubic segment per consecutive pair of the CLOSED polygon, numpoints samples each.
 * xy double[B, max_points*numpoints, 2]: the first n*numpoints rows of level b are its curve (x, y); control (optional)
 * double[B, max_points+1, 3]: the sorted, closed control points with their angles (the function's third result);
 * n_out (optional) int32[B] = n*numpoints.  fp64; transcendental functions differ from numpy's in the last bits. */
int dcd_bezier_track(const double* points, const int32_t* n_points, int32_t max_points, int32_t B, double rad, double edgy,
                     int32_t numpoints, double* xy, double* control, int32_t* n_out, void* stream);

/* ---- student policy step for rollouts (no grad) -------------------------------- */
/* Device pointers to the fp32 parameters of the student MultigridNetwork
 * (models/multigrid_models.py:45-88, util/make_agent.py:38-47): they stay PyTorch
 * parameters, the kernels read them in place on every call. */
typedef struct DcdStudentNet {
  const float* conv_w;     /* image_conv.0.weight  [F,3,3,3]                      */
  const float* conv_b;     /* image_conv.0.bias    [F]                            */
  const float* embed_w;    /* scalar_embed.weight  [E,D]   (E=scalar_fc, D=scalar_dim) */
  const float* embed_b;    /* scalar_embed.bias    [E]                            */
  const float* actor_w1;   /* actor.0.0.weight     [A1,H]  (+Tanh)                */
  const float* actor_b1;
  const float* actor_w2;   /* actor.1.linear.weight [NA,A1] (Categorical)          */
  const float* actor_b2;
  const float* critic_w1;  /* critic.0.0.weight    [C1,H]  (+Tanh)                */
  const float* critic_b1;
  const float* critic_w2;  /* critic.1.weight      [1,C1]                         */
  const float* critic_b2;
  int32_t V, F, E, D, H, A1, C1, NA;
} DcdStudentNet;
/* K = F*(V-2)^2 + E + H: width of the LSTM's stacked input row [features | h*mask]. */
int dcd_policy_input_width(const DcdStudentNet* net);
/* First half of MultigridNetwork._forward_base (multigrid_models.py:109-139) and of the masked
 * single-step LSTM (models/common.py:149-153): x[b] = [relu(conv(image[b])) flattened CHW |
 * scalar_embed(one_hot(direction[b])) | h[b]*masks[b]], x: float[B,x_pitch] (x_pitch >= K; columns [K,x_pitch) are
 * left untouched: zero them once to pad K for the GEMM).  image: float[B,3,V,V],
 * direction: float[B], hc: float[B,2H] (h | c: RolloutStorage.recurrent_hidden_states[step]),
 * masks: float[B].  The caller then forms gates = x @ [W_ih | W_hh]^T + (b_ih + b_hh) (one GEMM). */
int dcd_policy_features(const DcdStudentNet* net, const float* image, const float* direction, const float* hc,
                        const float* masks, float* x, int32_t x_pitch, int32_t B, void* stream);
/* dcd_policy_features reading the observation as the env wrote it: image_u8 uint8[B,V,V,3] (raw encoding, [vx][vy][c]),
 * dir_u8 uint8[B] — the /10 and the HWC->CHW transposition of VecPreprocessImageWrapper._preprocess
 * (obs_wrappers.py:88-115) are applied while the image is staged (exact: the fp32 nearest to k / 10.0). */
int dcd_policy_features_u8(const DcdStudentNet* net, const uint8_t* image_u8, const uint8_t* dir_u8, const float* hc,
                           const float* masks, float* x, int32_t x_pitch, int32_t B, void* stream);
/* VecPreprocessImageWrapper._preprocess (obs_wrappers.py:88-115) for N stored student observations:
 * image_f32[n,c,vx,vy] = image_u8[n,vx,vy,c] / 10.0, dir_f32[n] = dir_u8[n] (dir pointers may both be NULL; V up to the
 * grid size: the teacher's full-grid observations convert the same way).  Used where
 * fp32 observations are consumed in bulk (the PPO batch, the truncated-observation value pass). */
int dcd_obs_u8_to_f32(const uint8_t* image_u8, const uint8_t* dir_u8, float* image_f32, float* dir_f32, int64_t N, int32_t V,
                      void* stream);
/* The same row for a grid-sized observation (the teacher, multigrid_models.py:109-139, util/make_agent.py:22-36):
 * x[b] = [relu(conv(image[b])) flattened [F][(S-2)^2] | scalar_embed(one_hot(scalar[b])) | z[b] | h[b]*masks[b]].
 * Only conv_w, conv_b, embed_w, embed_b and V (= S <= 17), F (<= 128), E, D, H of `net` are read.  image:
 * float[B,3,S,S], scalar: float[B] (time_step), z: float[B,z_dim] (random_z; NULL with z_dim 0), x: float[B,x_pitch]. */
int dcd_policy_grid_features(const DcdStudentNet* net, const float* image, const float* scalar, const float* z,
                             int32_t z_dim, const float* hc, const float* masks, float* x, int32_t x_pitch, int32_t B,
                             void* stream);
/* dcd_policy_grid_features on the raw full-grid encoding uint8[B,S,S,3] ([x][y][c]: the teacher's rollout buffer in u8 mode). */
int dcd_policy_grid_features_u8(const DcdStudentNet* net, const uint8_t* image_u8, const float* scalar, const float* z,
                                int32_t z_dim, const float* hc, const float* masks, float* x, int32_t x_pitch, int32_t B,
                                void* stream);
/* Head weights re-laid out for the kernel below (stacked + transposed first layers, padded second layers):
 * packed: float[dcd_policy_packed_heads_floats()]; call again whenever the optimiser has stepped. */
int dcd_policy_packed_heads_floats(void);
int dcd_policy_pack_heads(const DcdStudentNet* net, float* packed, void* stream);
/* Rest of `act` (multigrid_models.py:141-167): LSTM cell on gates float[B,4H] (order i,f,g,o) with
 * c = hc_in[:,H:]*masks -> hc_out float[B,2H] (storage.recurrent_hidden_states[step+1]); actor / critic
 * heads; action_log_dist float[B,NA] = normalised logits (FixedCategorical.logits); action int64[B] sampled
 * by inverse CDF from one Philox4x32-10 uniform per env, key (seed,'ACT!'), counter (counter + *counter_base,
 * global env index) — or the first arg max when deterministic (counter_base: optional device uint32[1], read
 * at run time so that a replayed HIP graph draws fresh numbers); action_log_prob float[B]; value float[B].
 * u_out (optional, tests): the uniform used.  The reference samples with torch.multinomial on the global
 * torch RNG; any exact sampler of the same distribution is equivalent. */
int dcd_policy_cell_heads(const DcdStudentNet* net, const float* packed, const float* gates, const float* hc_in,
                          const float* masks,
                          float* hc_out, float* value, int64_t* action, float* action_log_prob,
                          float* action_log_dist, float* u_out, uint32_t seed, uint32_t counter,
                          const uint32_t* counter_base, int64_t env_index_base, int32_t deterministic, int32_t B, void* stream);

/* The teacher's rollout step around its GEMMs (`act` + the policy half of storage.insert for a recurrent grid-observing
 * network with up to 191 actions, one Linear + Tanh per head: multigrid_models.py:109-167, util/make_agent.py:22-36):
 *   dcd_policy_grid_rest_features   x[b] = [scalar_embed(one_hot(scalar[b])) | z[b] | h[b]*masks[b] | 1 | 0...], x float[B,x_pitch],
 *                                   x_pitch >= E + z_dim + H + 1: everything of the LSTM input but the conv features (whose
 *                                   projection is carried, dcd_grid_delta_rows); the 1 multiplies the bias row of the caller's
 *                                   stacked weight, so gates = xp_image + x @ W needs no bias operand;
 *   dcd_policy_wide_pack_heads      head weights re-laid out (float[dcd_policy_wide_packed_heads_floats(net)]; after every
 *                                   optimiser step);
 *   dcd_policy_wide_cell_heads      dcd_policy_cell_heads for NA < 192: a whole wave per env for the second layer, the
 *                                   log-softmax and the inverse-CDF draw (cumulative probabilities in index order by wave
 *                                   scans; same Philox keying as the students). */
int dcd_policy_grid_rest_features(const DcdStudentNet* net, const float* scalar, const float* z, int32_t z_dim, const float* hc,
                                  const float* masks, float* x, int32_t x_pitch, int32_t B, void* stream);
int dcd_policy_wide_packed_heads_floats(const DcdStudentNet* net);
int dcd_policy_wide_pack_heads(const DcdStudentNet* net, float* packed, void* stream);
int dcd_policy_wide_cell_heads(const DcdStudentNet* net, const float* packed, const float* gates, const float* hc_in,
                               const float* masks, float* hc_out, float* value, int64_t* action, float* action_log_prob,
                               float* action_log_dist, float* u_out, uint32_t seed, uint32_t counter,
                               const uint32_t* counter_base, int64_t env_index_base, int32_t deterministic, int32_t B,
                               void* stream);

/* Two students per launch (PAIRED: the protagonist's and the antagonist's `act` at the same step of their lock-step rollouts,
 * adversarial_runner.py:709-733 + 506-517): every pointer of dcd_policy_features_u8 / dcd_policy_cell_heads once per agent.
 * The two networks must have one geometry.  The caller forms both gates tensors between the two calls (one batched GEMM). */
typedef struct DcdPolicyAgent {
  const DcdStudentNet* net;
  const float* packed;                  /* dcd_policy_pack_heads of `net` */
  const uint8_t* image_u8;              /* features: observation bytes, state in, operand row out */
  const uint8_t* dir_u8;
  const float* hc_in;
  const float* masks;
  float* x;
  const float* gates;                   /* cell + heads */
  float* hc_out;
  float* value;
  int64_t* action;
  float* log_prob;
  float* log_dist;
  float* u_out;                         /* optional */
  const uint32_t* counter_base;         /* optional device word */
  int64_t env_index_base;
  uint32_t seed, counter;
  int32_t deterministic, reserved;
} DcdPolicyAgent;
int dcd_policy_features_u8_pair(const DcdPolicyAgent* a0, const DcdPolicyAgent* a1, int32_t x_pitch, int32_t B, void* stream);
int dcd_policy_cell_heads_pair(const DcdPolicyAgent* a0, const DcdPolicyAgent* a1, int32_t B, void* stream);

/* ---- masked LSTM sequence of the PPO update (models/common.py:157-207) ------------------ */
/* The reference cuts the flattened (T*N) sequence wherever ANY env has mask 0 and re-masks the state at each
 * cut — the same function as multiplying (h, c) by masks[t] before every step.  One step of that recurrence
 * is a GEMM (caller: gates = xp[t] + hm[t] @ W_hh^T) plus this pointwise kernel; the backward step likewise
 * (caller: dhm = dgates @ W_hh).  Gate order i,f,g,o (nn.LSTM).
 * fwd: gates float[N,4H] pre-activations (read only; the backward step recomputes the activations from them); c_prev float[N,H] = C[t-1]
 * (unmasked; multiplied by m_t float[N] here); c_out, h_out float[N,H]; hm_next float[N,H] = h_out * m_next
 * (the next step's GEMM operand; NULL with m_next at the last step). */
int dcd_lstm_seq_fwd_step(const float* gates, const float* c_prev, const float* m_t, const float* m_next, float* c_out,
                          float* h_out, float* hm_next, int32_t N, int32_t H, void* stream);
/* bwd of step t: act = the same PRE-activation gates the forward step read, c_t, c_prev (unmasked C[t-1]), m_t; dout = dL/dh_t from the heads (or
 * NULL); dhm_next = dgates[t+1] @ W_hh and dcf_next = dcf[t+1], both multiplied by m_next = masks[t+1] here
 * (all three NULL at t = T-1); dh_last / dc_last: grads of the returned final state (NULL if unused).
 * Writes dgates float[N,4H] (pre-activation grads: d xp[t]) and dcf float[N,H] = dL/dc_t * f_t. */
int dcd_lstm_seq_bwd_step(const float* act, const float* c_t, const float* c_prev, const float* m_t,
                          const float* dout, const float* dhm_next, const float* dcf_next, const float* m_next,
                          const float* dh_last, const float* dc_last, float* dgates, float* dcf, int32_t N,
                          int32_t H, void* stream);

/* ---- the student's PPO batch: features and heads as fused kernels (csrc/ppo_kernels.hip) --------------------------
 * `PPO.update` (algos/ppo.py:83-116) calls `evaluate_actions` (models/multigrid_models.py:169-186) on T*B rows per
 * minibatch.  Around the LSTM and the dense contractions (which stay GEMMs issued by the caller) the rest of that
 * forward / backward pass runs in these four kernels; the registered student only (5 x 5 view, Conv 3 -> 16,
 * embedding up to 8 x 8, 32-unit head layers, at most 8 actions), anything else is refused.
 *   features_fwd  x float[R,x_pitch]: [relu(conv(image)) flattened [16][3][3] | scalar_embed(one_hot(direction)) | 1 | 0..]
 *                 (multigrid_models.py:109-131; x_pitch >= 144 + E + 1: the 1 multiplies the bias column the caller
 *                 stacks onto W_ih).  obs_u8 = 0: image float[R,3,5,5] (CHW, bytes / 10), direction float[R];
 *                 obs_u8 = 1: image uint8[R,5,5,3], direction uint8[R] (the rollout buffers' raw bytes; / 10 and
 *                 HWC->CHW of obs_wrappers.py:88-115 applied while staging).
 *   features_bwd  from dx float[R,x_pitch] and the saved x (its own ReLU mask): partial float[n_wg, 512] with n_wg =
 *                 ceil(R / rows_per_wg) (rows_per_wg a multiple of 16), row layout [0,432) d conv.weight [f][c][ky][kx] |
 *                 [432,448) d conv.bias | 448 + 8 j + d: d scalar_embed.weight[j][d]; the caller adds the rows up
 *                 (d scalar_embed.bias[j] = sum over d).
 *   heads_fwd     pre_act float[R,64] = h @ [actor.0.0.weight ; critic.0.0.weight]^T (no bias) in, tanh(. + bias) out
 *                 (kept for the backward); actions int64[R]; logp_all float[R,8] (normalised logits, FixedCategorical,
 *                 models/distributions.py:18-56); value float[R]; logp_act float[R] = log_prob(action); entropy float[R]
 *                 (per row; evaluate_actions returns its mean).
 *   heads_bwd     upstream dvalue / dlogp / dentropy float[R] (any may be NULL = 0); act_dpre: tanh outputs in, d
 *                 pre-activations out (the caller forms dh = dpre @ W1 and dW1 = dpre^T h); partial float[n_wg, 384]
 *                 (rows_per_wg a multiple of 128): [0,256) d actor.1.linear.weight [k][i] (k < 8) | [256,264) its bias |
 *                 [264,296) d critic.1.weight | 296 its bias | [304,336) d actor.0.0.bias | [336,368) d critic.0.0.bias.
 * Sums run in a fixed order (no atomics): results do not depend on scheduling. */
int dcd_ppo_features_partial_floats(void);
int dcd_ppo_heads_partial_floats(void);
int dcd_ppo_features_fwd(const DcdStudentNet* net, const void* image, const void* direction, int32_t obs_u8, float* x, int32_t x_pitch,
                         int64_t R, void* stream);
int dcd_ppo_features_bwd(const DcdStudentNet* net, const void* image, const void* direction, int32_t obs_u8, const float* x, const float* dx,
                         int32_t x_pitch, int64_t R, int32_t rows_per_wg, float* partial, void* stream);
int dcd_ppo_heads_fwd(const DcdStudentNet* net, float* pre_act, const int64_t* actions, float* logp_all, float* value, float* logp_act,
                      float* entropy, int64_t R, void* stream);
int dcd_ppo_heads_bwd(const DcdStudentNet* net, float* act_dpre, const float* logp_all, const int64_t* actions, const float* dvalue,
                      const float* dlogp, const float* dentropy, int64_t R, int32_t rows_per_wg, float* partial, void* stream);

/* Scatter side of the teacher's incremental input projection (no reference counterpart: the reference multiplies the
 * dense 21 632-wide conv features every step, models/multigrid_models.py:122-138): out[r,:] += sum_k y[src[r,k],:],
 * src int32[R,K] (negative = no term), terms added in k order; y float[*,G], out float[R,G], G a multiple of 4,
 * y / out 16-byte aligned. */
int dcd_gather_sum_rows(const float* y, const int32_t* src, float* out, int32_t R, int32_t K, int32_t G, void* stream);
/* The gather side of the same projection, device-resident (nothing read back, fixed shapes: the teacher's rollout is a
 * HIP graph).  A constructive teacher step changes at most one cell of the observation (adversarial.py:466-530), so at most
 * nine conv positions per env see a different window.  For image_u8 / prev_u8 uint8[B,S,S,3] (prev_env_stride = S*S*3; 0:
 * prev_u8 is ONE image taken as every env's previous one) the call writes
 *   cell  int32[B]      the changed cell x*S+y, or -1;
 *   count int32[(S-2)^2] (env, position) pairs per conv position;
 *   src   int32[B,9]    row of pair (env, slot) in `rows`, -1 if none: pairs are binned by position, bins padded to R-row
 *                       tiles, rows inside a bin in env order (so no result depends on scheduling);
 *   rows  float[n_tiles*R, F]   relu(conv)(new window) - relu(conv)(old window) of every pair (other rows: not written);
 *   tile_pos int32[n_tiles]     the position whose weight block multiplies tile t (tiles past the last bin: any valid index);
 *   n_used int32[1]     number of tiles that hold rows (the bins' padded sizes / R);
 *   err   int32[1]      |= 1 if some env changed in more than one cell, |= 2 on tile overflow (cannot happen with
 *                       n_tiles >= dcd_grid_delta_tiles(S, B) = ceil((9 B + (R-1) (S-2)^2) / R), R = dcd_grid_delta_tile_rows()).
 * The caller then runs Y[t] = rows[t] @ Wp[tile_pos[t]] (dcd_grid_tile_gemm) and dcd_gather_sum_rows(Y, src, xp, B, 9, 4H).
 * Only V (= S), F, conv_w, conv_b of `net` are read. */
int dcd_grid_delta_tile_rows(void);
int dcd_grid_delta_tiles(int32_t S, int32_t B);
int dcd_grid_delta_rows(const DcdStudentNet* net, const uint8_t* image_u8, const uint8_t* prev_u8, int64_t prev_env_stride,
                        int32_t* cell, int32_t* count, int32_t* src, float* rows, int32_t* tile_pos, int32_t n_tiles,
                        int32_t* n_used, int32_t* err, int32_t B, void* stream);
/* Y[t] = rows[t] @ wp[tile_pos[t]] for t < *n_used (device word): rows float[n_tiles*R, F] (R = dcd_grid_delta_tile_rows() =
 * 128), wp float[(S-2)^2, F, G] the per-position blocks of the LSTM input weight (W_ih[:, f*P + p]^T), y float[n_tiles*R, G];
 * F a multiple of 4 up to 128, G a multiple of 128, all three 16-byte aligned.  fp32 MFMA (v_mfma_f32_32x32x2_f32): every
 * element is the k-ordered fmaf chain over the F features.  Tiles t >= *n_used are not computed (their y rows keep old values). */
int dcd_grid_tile_gemm(const float* rows, const float* wp, const int32_t* tile_pos, const int32_t* n_used, float* y,
                       int32_t n_tiles, int32_t F, int32_t G, void* stream);

/* ---- logging statistics (HOST pointers — the only entry point that is not a device call) ------------------ */
/* RolloutStorage.get_action_complexity / get_batched_action_complexity (algos/storage.py:329-369): LZ76 phrase
 * count (the `lempel_ziv_complexity` package, storage.py:21) of every episode's action string "a0 a1 ...";
 * episodes are cut at done_steps = [0] + {t : masks[t,b] == 0}.  actions int64[T,B], masks float[T+1,B] in HOST
 * memory; total = sum over episodes, n_episodes their number, per_env float[B] = per-env mean (may be NULL). */
int dcd_action_complexity(const int64_t* actions, const float* masks, int32_t T, int32_t B, double* total,
                          int64_t* n_episodes, float* per_env);

/* LevelSampler.sample_replay_levels (level_replay/level_sampler.py:664-750 as n sequential draws): the same IEEE
 * operations in the same order as the numpy statement, for staleness transform 'power' with temperature 1.  HOST
 * pointers: w_scores double[N] = (1 - c) * score weights, staleness double[N] (updated in place), unseen_keep
 * double[N] = 1 - unseen_seed_weights, uniforms double[n] (np.random.random_sample(n)), out_idx int64[n].  Returns the
 * number of draws performed; < n when a case the loop does not cover occurred at that draw (the caller continues in
 * numpy from there).  dcd_np_sum: numpy's pairwise summation (np.sum of a contiguous float64 array). */
int64_t dcd_sampler_draws(int64_t N, int64_t n, const double* w_scores, double c, double* staleness,
                          const double* unseen_keep, const double* uniforms, int32_t update_staleness, int64_t* out_idx);
double dcd_np_sum(const double* a, int64_t n);

/* ---- synthetic workload helpers (bench / tests only) ------------- */
/* actions[t,i] = mulhi(philox(key=(seed, env_index_base+i), ctr=(t0+t,0,0,0))[0], num_actions);
 * actions: int64[T,B] */
int dcd_random_actions(int64_t* actions, int32_t B, int32_t T, int64_t env_index_base, uint32_t seed,
                       uint32_t t0, int32_t num_actions, void* stream);
/* K fused random-action student steps in one launch (grid stays in LDS): the
 * random-agent rollout of models/multigrid_models.py:149-157 without a host round
 * trip.  Outputs of the LAST step only (any may be NULL); counters accumulate
 * total episodes/reward into stats int64[2] {episodes, sum of fp32 reward bits unused}. */
int dcd_mg_step_random(DcdMg* h, uint32_t action_seed, uint32_t t0, int32_t K, int32_t flags,
                       const DcdStepOut* out, int64_t* n_episodes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DCD_HIP_H */
