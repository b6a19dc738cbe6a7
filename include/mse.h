/*
 * mse.h -- C ABI of libmse_hip.so: the MI355X-native batched step() engine for the three
 * MARL-SortingEnv environments.
 *
 * The reference has no FFI: its boundary for this path is the Python Gymnasium protocol of
 * Env_1_Sorting / Env_2_Pressing / Env_3_Monolith (one env instance, NumPy on the host).  Each
 * entry point below names the reference interface it replaces (path:line in the reference
 * checkout).  N independent env instances live in one handle, one env per GPU lane, state as
 * struct-of-arrays planes in HBM that the library owns; every I/O buffer is CALLER-owned DEVICE
 * memory (contiguous, row-major, 16-byte aligned), e.g. a PyTorch-ROCm tensor's data_ptr().
 *
 * Conventions
 *   - every function returns 0 (MSE_OK) or a negative mse_status; nothing throws across the ABI;
 *     mse_last_error() gives the message of the calling thread's last failure;
 *   - all device work is enqueued on the caller's HIP stream (`stream` is a hipStream_t passed as
 *     void*, NULL = the default stream); no hidden synchronisation, graph-capture safe
 *     (mse_step / mse_rollout / mse_reset / mse_action_masks allocate nothing and never sync);
 *   - a handle is not thread-safe; different handles are independent;
 *   - there is NO CPU fallback: the library needs a gfx950 device and fails loudly without one.
 */
#ifndef MSE_H
#define MSE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MSE_VERSION 200 /* 0.2.0 */

typedef enum mse_status {
    MSE_OK = 0,
    MSE_ERR_INVALID_ARGUMENT = -1,
    MSE_ERR_UNSUPPORTED_CONFIG = -2,
    MSE_ERR_NO_DEVICE = -3,
    MSE_ERR_HIP = -4,
    MSE_ERR_NOT_RESET = -5,
    MSE_ERR_ALIGNMENT = -6
} mse_status;

/* env_1_sort.py:26 "sort", env_2_press.py:26 "press", env_monolith.py:28 "mono" */
typedef enum mse_env_kind { MSE_ENV_SORT = 1, MSE_ENV_PRESS = 2, MSE_ENV_MONO = 3 } mse_env_kind;

/* per-call step flags = the reference's step() keyword arguments
 * (env_monolith.py:109, env_2_press.py:88, env_1_sort.py:97) */
#define MSE_STEP_UNMASKED       1u /* use_action_masking=False: validate + sanitise the press action */
#define MSE_STEP_CHECK_OVERFLOW 2u /* check_overflow=True: overflow terminates with the -10 penalty  */
/* mse_rollout only: act with the reference's rule-based policy (mode='rule_based', env_monolith.py:166-184:
 * sorting_rules() env_super.py:469-482 + check_container_level() :689-720) instead of the random one */
#define MSE_ROLLOUT_RULE_BASED  4u
/* with MSE_STEP_UNMASKED on Env_3: test the press action's validity AFTER sort_material instead of at decode time -
 * what Env_3_Monolith.step(action=None, mode='random', use_action_masking=False) does (env_monolith.py:245-253:
 * sanitize_press_action runs in the "apply" section; an invalid action still skips press_action_rules) */
#define MSE_STEP_SANITIZE_LATE  8u
/* mse_model_actions only: leave the sorting / the pressing decision to an agent (no draw, that part is 0) */
#define MSE_MODEL_NO_SORT_DRAW  16u
#define MSE_MODEL_NO_PRESS_DRAW 32u

/* POD copy of the reference's config.yml plus the env constructor arguments
 * (env_super.py:25-137 reads the same keys; env_monolith.py:22-23 ctor). */
typedef struct mse_config {
    uint32_t struct_size;          /* = sizeof(mse_config); ABI check */
    int32_t  env_kind;             /* mse_env_kind */
    int32_t  max_steps;            /* ctor max_steps (<= 65535) */
    int32_t  auto_reset;           /* 1: a terminated env is reset inside the same step (VecEnv semantics) */
    int32_t  track_bales;          /* 1: keep the O(1) bale ledger summary (env_super.py:661-687) */
    int32_t  literal_choice;       /* 1: always evaluate Generator.choice(4,p) in literal fp64 (test switch);
                                      0: exact integer decision with literal fallback near ties */
    /* simulation (config.yml:4-9) */
    int32_t  input_batch_size;     /* <= 255.  If floor(ratio * batch) leaves units over (e.g. 90), the generator's
                                      random remainder + shuffle draws run on the device ("general generator mode",
                                      utils/input_generator.py:46-61) and the one-lane kernels serve the handle */
    int32_t  steps_per_pattern;    /* informational: reset() rebuilds the generator with its default 20
                                      (env_super.py:375), so 20 is what every episode uses */
    /* sorting_station (config.yml:12-18) */
    double   baseline_accuracy[4];
    double   boost;
    double   noise;                /* ctor noise_sorting (env_super.py:71) */
    int32_t  stage_capacity;
    /* pressing_station (config.yml:21-32) */
    int32_t  press_time[2];        /* <= 255 */
    int32_t  container_capacity;
    int32_t  bale_standard_size;   /* ctor balesize (env_super.py:87) */
    double   bale_remainder_threshold;
    double   quality_threshold[4];
    double   quality_threshold_r2[4]; /* python round(threshold, 2): purity of an EMPTY container
                                         (env_super.py:786-789) */
    /* rewards (config.yml:35-59) */
    double   purity_threshold_theta;
    double   tanh_temperature;
    double   overflow_penalty_catastrophic;
    double   overflow_penalty_severe;
    double   overflow_penalty_mild;
    double   bale_efficiency_factor;
    double   max_state_reward;
    double   overflow_termination_penalty;
    /* seasonal patterns 1 and 2, order A,B,C,D (utils/input_generator.py:17-20) */
    double   pattern_ratio[2][4];
    /* mse_rollout kernel choice: 0 = by size (while the batch fits the chip in one round of 256-env workgroups,
       n <= 256 x CUs - or fills most of a second one, 320 x CUs < n <= 512 x CUs -, the multi-role kernels:
       dynamics + observer waves, plus RNG waves when the config's draws per step fit the ring; otherwise one lane
       per env), 1 = dynamics + observer waves, 2 = one lane per env, 3 = dynamics + observer + RNG waves.
       mse_rollout_policy reads the same field: up to 256 envs x CUs (f16x3 products, no in-loop sorting policy) it
       runs actor + critic + RNG waves per 64 envs (0, when the draws fit the ring), actor + critic waves (1, or 0 when
       they do not) or one wave per 32 envs doing everything (2); above that size 64-env waves whatever the value.
       Results are identical. */
    int32_t  rollout_pipeline;
    int32_t  reserved0;
} mse_config;

typedef struct mse_env mse_env; /* opaque handle: N envs on one device */

/* library */
int         mse_version(void);
const char *mse_last_error(void);
const char *mse_status_string(int status);

/* Fills *cfg with config.yml's values and the env classes' ctor defaults
 * (max_steps=50, noise 0.05, balesize 200: env_monolith.py:22-23). */
int mse_config_default(mse_config *cfg);

/* Replaces Env_*.__init__ (env_monolith.py:22-36, env_super.py:25-137) for n_envs instances on
 * HIP device `device_id`.  Allocates the state planes; the envs hold no valid episode until
 * mse_reset() with seeds has run once (mse_step before that fails with MSE_ERR_NOT_RESET). */
int mse_create(mse_env **out, const mse_config *cfg, int64_t n_envs, int device_id);
/* Same, for one shard of a larger job: env i of this handle is global env index_offset + i.
 * The global index only feeds the random-policy stream, so a sharded run samples the same
 * actions as a single-handle run (env-index sharding). */
int mse_create_indexed(mse_env **out, const mse_config *cfg, int64_t n_envs, int device_id, int64_t index_offset);
int mse_destroy(mse_env *env);

int64_t mse_num_envs(const mse_env *env);
int     mse_obs_dim(const mse_env *env);      /* 13 / 16 / 29: observation_space (env_1_sort.py:71, env_2_press.py:62, env_monolith.py:72) */
int     mse_num_actions(const mse_env *env);  /* 2 / 11 / 22: action_space      (env_1_sort.py:72, env_2_press.py:64, env_monolith.py:79) */

/* Replaces Env_*.reset(seed=...) (env_super.py:365-420 and the variants' overrides
 * env_1_sort.py:81-85, env_2_press.py:73-75, env_monolith.py:91-93).
 *   seeds_dev  u64[N] or NULL.  Given: reset(seed=seeds[i]) - re-seeds the env's streams at
 *              seed+3/+4/+99 and the generator at seed, all on the device (NumPy SeedSequence +
 *              PCG64 restated).  NULL: reset(seed=None) - streams continue; the generator's
 *              pattern order follows the build's deterministic rule (DESIGN.md "unseeded reset").
 *   which_dev  u8[N] or NULL: only envs with which[i] != 0 are reset (NULL = all).
 *   obs_out    f32[N, D] or NULL;  mask_out u8[N, A] or NULL  (rows of envs not reset are
 *              rewritten with their current observation / mask). */
int mse_reset(mse_env *env, const uint64_t *seeds_dev, const uint8_t *which_dev,
              float *obs_out, uint8_t *mask_out, void *stream);

/* Replaces Env_*.step(action, use_action_masking, check_overflow)
 * (env_1_sort.py:97-154, env_2_press.py:88-165, env_monolith.py:109-284) for all N envs.
 *   action_dev     i32[N]: sort: mode 0|1; press: 0..10; mono: 0..21 (mode*11 + press action).
 *                  Out-of-range values are counted (mse_error_count) and treated as action 0.
 *   sort_mode_dev  i32[N] or NULL, Env_2 only: the sorting agent's decision
 *                  (env_2_press.py:101-104); NULL = the rule-based fallback sorting_rules()
 *                  (env_super.py:469-482).
 *   obs_out        f32[N, D]    next observation (after auto-reset: the reset observation)
 *   reward_out     f32[N]       or NULL
 *   reward64_out   f64[N]       or NULL (the reference returns a python float = f64)
 *   done_out       u8[N]        terminated flag (truncated is always False in the reference)
 *   mask_out       u8[N, A]     or NULL: action_masks() of the state the NEXT action sees
 *   terminal_obs_out f32[N, D]  or NULL: with auto_reset, the last observation of a finished
 *                  episode (rows of envs that did not finish are left untouched). */
int mse_step(mse_env *env, const int32_t *action_dev, const int32_t *sort_mode_dev, uint32_t flags,
             float *obs_out, float *reward_out, double *reward64_out, uint8_t *done_out,
             uint8_t *mask_out, float *terminal_obs_out, void *stream);

/* Replaces Env_*.action_masks() (env_monolith.py:81-85 -> env_super.py:887-898,
 * env_2_press.py:66-67 -> env_super.py:869-885, env_1_sort.py:74-76). mask_out u8[N, A]. */
int mse_action_masks(mse_env *env, uint8_t *mask_out, void *stream);

/* The observation Env_2_Pressing.step hands its sorting agent (env_2_press.py:95-104: get_sort_obs() after the
 * coming step's flow update, before the sensor setting), for every env: obs13_out f32[N, 13].  A preview - the
 * state is not changed; feed the agent's decisions to the next mse_step / mse_rollout as sort_mode_dev.
 * Defined for every env kind (the sorting view of the state one flow update ahead). */
int mse_sort_agent_obs(mse_env *env, float *obs13_out, void *stream);
/* The same preview for a pressing agent: get_press_obs() after the coming step's flow update, which is what
 * Env_3_Monolith.step(mode='model') hands its press_agent (env_monolith.py:198-210).  obs16_out f32[N, 16]. */
int mse_press_agent_obs(mse_env *env, float *obs16_out, void *stream);

/* K fused steps with the on-device random policy (the reference's mode='random', env_monolith.py:152-164, with a
 * counter-based policy stream instead of the global np.random): uniform over the set bits of action_masks(), or -
 * with MSE_STEP_UNMASKED - over the whole action space, the step sanitising what it gets (add MSE_STEP_SANITIZE_LATE
 * on Env_3 for the reference's exact mode='random' sequencing).  State stays in registers across the K steps.
 * Outputs are step-major: actions_out i32[K,N], obs_out f32[K,N,D], reward_out f32[K,N],
 * done_out u8[K,N], mask_out u8[K,N,A]; any of them may be NULL.  Needs auto_reset=1.
 *   sort_mode_dev i32[N] or NULL: Env_2's frozen per-env sorting decision (NULL = rule). */
int mse_rollout(mse_env *env, int32_t k_steps, uint64_t policy_seed, const int32_t *sort_mode_dev,
                uint32_t flags, int32_t *actions_out, float *obs_out, float *reward_out,
                uint8_t *done_out, uint8_t *mask_out, void *stream);

/* Samples one masked-uniform action per env from the CURRENT state (same policy stream as
 * mse_rollout); action_out i32[N]. */
int mse_sample_actions(mse_env *env, uint64_t policy_seed, int32_t *action_out, void *stream);

/* The reference's rule-based action per env for the CURRENT state (env_monolith.py:166-184). */
int mse_rule_actions(mse_env *env, int32_t *action_out, void *stream);

/* State export / import in a fixed record layout (tests, checkpoint/resume, dashboard trace):
 *   ints  i64[N, MSE_SNAP_INTS]  (column map: MSE_SNAP_* below)
 *   dbls  f64[N, 4]              accuracy_belt
 *   rng   u64[N, 30]             {state_hi,state_lo,inc_hi,inc_lo,has_uint32,uinteger} x
 *                                {rng (seed+99), rng_noise (seed+4), rng_pressing (seed+3), rng_sorting (seed+2),
 *                                 the input generator's private default_rng(seed) (utils/input_generator.py:28; it only
 *                                 advances in general generator mode, i.e. when floor(ratio * batch) leaves units over)}
 * Replaces reading/writing the attributes of Env_Super (env_super.py:52-137). */
#define MSE_SNAP_INTS 71
#define MSE_SNAP_RNG_WORDS 30
int mse_get_state(mse_env *env, int64_t *ints_out, double *dbls_out, uint64_t *rng_out, void *stream);
int mse_set_state(mse_env *env, const int64_t *ints_in, const double *dbls_in, const uint64_t *rng_in, void *stream);

/* Env_3_Monolith.step(action=None, mode='model') with no agents assigned (env_monolith.py:186-221): per env the
 * sorting decision rng_sorting.choice([0, 1]) (:195) and the press action rng_pressing.choice(flatnonzero(
 * press_action_masks())) (:214-217), or rng_pressing.choice(11) with MSE_STEP_UNMASKED (:219), NumPy-exact (buffered
 * Lemire draws on the env's own PCG64 streams, which advance).  action_out i32[N] = mode * 11 + press action; step
 * it with mse_step and flags WITHOUT MSE_STEP_UNMASKED: the reference applies a mode='model' action through
 * press_action_rules without sanitising it (env_monolith.py:254-257).  flags: MSE_STEP_UNMASKED,
 * MSE_MODEL_NO_SORT_DRAW, MSE_MODEL_NO_PRESS_DRAW (an assigned agent decides that part).  Env_3 handles only. */
int mse_model_actions(mse_env *env, uint32_t flags, int32_t *action_out, void *stream);

/* Opt-in trace of ONE env: what the reference appends per step to its Python ledgers - reward_data
 * (env_super.py:402-408, 928-946), press_actions_per_timestep (:631-637, 730-736; env_monolith.py:136;
 * env_2_press.py:131) and the press_bale calls behind bale_count (:661-687) - which the dashboard reads
 * (utils/plotting.py:32-48).  Between mse_trace_begin and mse_trace_end every mse_step appends one record
 * f64[MSE_TRACE_COLS] for env `env_index` to records_dev (caller-owned device memory, `capacity` records; a step
 * beyond the capacity fails with MSE_ERR_INVALID_ARGUMENT; mse_rollout is refused while a trace is attached).
 * The values are those of the state _log_step_data sees: after the step, before any auto-reset.  Unbounded
 * per-env ledgers are not kept for all N lanes (DESIGN.md "Not carried per lane"); this is the one-env form. */
#define MSE_TRACE_COLS       40
#define MSE_TRACE_ACTION      0 /* the flat action the step applied (Env_1: the sensor mode)                     */
#define MSE_TRACE_R_SORT      1 /* _log_step_data(r_sort, r_press): env_1_sort.py:141,151; env_2_press.py:152,162; */
#define MSE_TRACE_R_PRESS     2 /*                                  env_monolith.py:271,282                         */
#define MSE_TRACE_SETTING     3 /* sensor_current_setting                                                         */
#define MSE_TRACE_BELT        4 /* [4] current_material_belt (Belt_Occupancy, Belt_Proportions derive from it)    */
#define MSE_TRACE_CONT_TRUE   8 /* [4] container A..D true                                                        */
#define MSE_TRACE_CONT_FALSE 12 /* [4] container A..D false                                                       */
#define MSE_TRACE_CONT_E     16
#define MSE_TRACE_N_LOG      17 /* entries appended to press_actions_per_timestep this step (0..2)                */
#define MSE_TRACE_LOG        18 /* [2][2] {code, material}: code 0 no-op, 1|2 press started, 111|222 busy/invalid */
#define MSE_TRACE_N_BALE     22 /* press_bale calls this step (0..2)                                              */
#define MSE_TRACE_BALE       23 /* [2][3] {material 0..4, amount n, int(q*100)} in the reference's call order      */
#define MSE_TRACE_DONE       29
#define MSE_TRACE_STEP       30 /* current_step after the step                                                    */
#define MSE_TRACE_INTERNAL   31 /* Env_1: the press action sampled inside the env (env_1_sort.py:125)             */
#define MSE_TRACE_REWARD     32 /* the step's reward (f64)                                                        */
#define MSE_TRACE_ACC_BELT   33 /* [4] accuracy_belt after the step (the next step's accuracy_sorter)             */
#define MSE_TRACE_OVERFLOW   37 /* check_overflow terminated the episode: 1 + index of the first material A..E whose
                                   level exceeds the capacity (detect_overflow, env_super.py:900-905: the reference's
                                   info["overflow_material"]), else 0                                                */
int mse_trace_begin(mse_env *env, int64_t env_index, double *records_dev, int64_t capacity);
int mse_trace_end(mse_env *env, int64_t *n_records_out);

/* The step counter t of the on-device policy stream (word = f(policy seed, global env index, t); every mse_step
 * advances it by 1, every mse_rollout by k_steps).  It is host-side handle state, not part of the state record:
 * a checkpoint that must reproduce an on-device-policy rollout saves it with mse_get_state's record and restores
 * it after mse_set_state. */
int mse_get_policy_step(const mse_env *env, uint64_t *t_out);
int mse_set_policy_step(mse_env *env, uint64_t t);

/* Number of out-of-range actions (mse_step) and of unreachable values handed to mse_set_state (stage vectors that
 * are no generator output, accuracies outside [clip(baseline [+ boost] - noise), 1]) seen so far; synchronises
 * the device. */
int mse_error_count(mse_env *env, uint64_t *count_out);

/* Algorithmic HBM bytes per env-step used for the roofline figure. */
int mse_algorithmic_bytes_per_step(const mse_env *env);

/* Build constant: a `choice` draw whose 32-bit fraction view f satisfies f + 160 < window (32-bit wrap-around: f below
 * window - 160, or within 160 of 2^32) is decided by the literal fp64 cdf instead of the exact integer comparison
 * (DESIGN.md "choice"; the derivation of the margin is in csrc/mse_device.h).  162 in the shipped library; the
 * test-only build libmse_hip_widetie.so reports 0x08000000. */
uint32_t mse_tie_window(void);

/* ---- the policy on the caller's side of step() -----------------------------------------------
 * Batched forward of the reference's actor-critic MLP (src/training.py:115: net_arch=dict(pi=[32,32], vf=[32,32]),
 * tanh; MaskableActorCriticPolicy) with masked categorical sampling, on the f32 matrix cores.
 * weights_host: mse_policy_num_weights(D, A) floats, torch.nn.Linear layout ([out, in] row-major), in this order:
 *   pi_w1[32*D] pi_b1[32] pi_w2[32*32] pi_b2[32] act_w[A*32] act_b[A]   (mlp_extractor.policy_net.0/.2, action_net)
 *   vf_w1[32*D] vf_b1[32] vf_w2[32*32] vf_b2[32] val_w[32]   val_b[1]   (mlp_extractor.value_net.0/.2, value_net)
 * mse_policy_forward: obs_dev f32[N, D]; mask_dev u8[N, A] or NULL (invalid actions get logit -1e8, as
 * sb3_contrib's MaskableCategorical); deterministic != 0: argmax, else inverse-cdf sampling with the engine's
 * counter-based stream (seed, global env index = index_offset + i, t).  Outputs (each may be NULL):
 * action i32[N], log-probability f32[N], value f32[N], masked logits f32[N, A]. */
typedef struct mse_policy mse_policy;
int64_t mse_policy_num_weights(int obs_dim, int n_actions);
int mse_policy_create(mse_policy **out, int obs_dim, int n_actions, const float *weights_host, int device_id);
int mse_policy_destroy(mse_policy *policy);
/* Arithmetic of the matrix products: 1 = exact f32 (v_mfma_f32_32x32x2_f32, an fmaf chain); 2 = "f16x3": every f32
 * operand split into two f16 parts carrying 22 bits, three f16 MFMAs per product with f32 accumulation (logits within
 * ~1e-6 of the exact form, 5x the matrix rate; needs every folded weight below 65 504); 0 = f16x3 when the weights
 * allow it, else f32 (the default).  mse_policy_precision reports which one is in effect (1 | 2). */
int mse_policy_set_precision(mse_policy *policy, int mode);
int mse_policy_precision(const mse_policy *policy);
int mse_policy_forward(mse_policy *policy, int64_t n, int64_t index_offset, const float *obs_dev, const uint8_t *mask_dev,
                       uint64_t seed, uint64_t t, int deterministic, int32_t *action_out, float *logp_out,
                       float *value_out, float *logits_out, void *stream);

/* ---- SURVEY 8f ranks 1 + 2 fused: K steps of policy forward + env transition in ONE launch -------------------------
 * The consumer loop of the reference's training (SB3's collect_rollouts inside model.learn, src/training.py:191, with
 * the policy of src/training.py:115-131): per step k and env i
 *     observation / action mask / episode-start flag of the state the action is taken from   (obs_out f32[K,N,D],
 *                                                       mask_out u8[K,N,A], episode_start_out u8[K,N])
 *     action, log-probability, value = policy(observation, mask)   (sampled with the engine's stream at step counter
 *                                                       t + k, or argmax with deterministic != 0: actions_out i32[K,N],
 *                                                       logp_out f32[K,N], value_out f32[K,N])
 *     reward of the transition under that action, auto-reset on termination                  (reward_out f32[K,N])
 * i.e. the rows of SB3's MaskableRolloutBuffer, plus last_value_out f32[N] (the value of the state after the last
 * step, the bootstrap of compute_returns_and_advantage) and last_done_out u8[N].  Any output may be NULL.  The
 * observation never leaves the wave's registers between env_step and the policy's MFMA chain.  Bit-identical to
 * alternating mse_policy_forward and mse_step with the same seed and step counter.
 *   sort_policy   Env_2 only, or NULL: a second network (13 -> 2) standing where the reference expects the pre-trained
 *                 sorting agent (env_2_press.py:101-104: sort_agent.predict(get_sort_obs() after the step's flow update,
 *                 deterministic=True)): its actor is evaluated inside the loop, argmax (f16x3 form of both networks);
 *   sort_mode_dev i32[N] or NULL: else Env_2's per-env sorting decision (both NULL = sorting_rules()).
 *   flags: MSE_STEP_UNMASKED, MSE_STEP_CHECK_OVERFLOW.  Needs auto_reset=1; advances the policy step counter by K. */
int mse_rollout_policy(mse_env *env, mse_policy *policy, mse_policy *sort_policy, int32_t k_steps, uint64_t seed, int deterministic,
                       const int32_t *sort_mode_dev, uint32_t flags, float *obs_out, uint8_t *mask_out,
                       int32_t *actions_out, float *logp_out, float *value_out, float *reward_out,
                       uint8_t *episode_start_out, float *last_value_out, uint8_t *last_done_out, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* MSE_H */
