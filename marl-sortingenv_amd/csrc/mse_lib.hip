// mse_lib.hip -- kernels and the C ABI (include/mse.h) of libmse_hip.so.  gfx950 only.
//
// Build (see marl-sortingenv_amd/build.py):
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -std=c++17 -fPIC -shared -Iinclude \
//         marl-sortingenv_amd/csrc/mse_lib.hip -o marl-sortingenv_amd/libmse_hip.so
#include "mse_device.h"

#include "mse.h"

#include <cmath>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>

using namespace mse;

// ==========================================================================================
// output staging: rows computed one-per-lane are transposed through LDS so that global stores
// are 16 B per lane and contiguous across the workgroup (obs rows are 52/64/116 B long).
// ==========================================================================================
template <int D>
__device__ __forceinline__ void stage_rows_f32(float *lds, const float *row, float *gout, int n_valid, int tid)
{
#pragma unroll
    for (int j = 0; j < D; ++j) lds[tid * D + j] = row[j]; // odd D: conflict-free; D=16: 16-way on 4 B stores
    __syncthreads();
    const int total = n_valid * D;
    if ((reinterpret_cast<uintptr_t>(gout) & 15u) == 0) {
        const int nvec = total >> 2;
        const float4 *src = reinterpret_cast<const float4 *>(lds);
        float4 *dst = reinterpret_cast<float4 *>(gout);
        for (int v = tid; v < nvec; v += kBlock) dst[v] = src[v];
        for (int k = (nvec << 2) + tid; k < total; k += kBlock) gout[k] = lds[k];
    } else {
        for (int k = tid; k < total; k += kBlock) gout[k] = lds[k];
    }
    __syncthreads();
}

template <int A>
__device__ __forceinline__ void stage_rows_mask(uint8_t *lds, uint32_t bits, uint8_t *gout, int n_valid, int tid)
{
#pragma unroll
    for (int j = 0; j < A; ++j) lds[tid * A + j] = (uint8_t)((bits >> j) & 1u);
    __syncthreads();
    const int total = n_valid * A;
    if ((reinterpret_cast<uintptr_t>(gout) & 15u) == 0) {
        const int nvec = total >> 4;
        const uint4 *src = reinterpret_cast<const uint4 *>(lds);
        uint4 *dst = reinterpret_cast<uint4 *>(gout);
        for (int v = tid; v < nvec; v += kBlock) dst[v] = src[v];
        for (int k = (nvec << 4) + tid; k < total; k += kBlock) gout[k] = lds[k];
    } else {
        for (int k = tid; k < total; k += kBlock) gout[k] = lds[k];
    }
    __syncthreads();
}

template <int KIND>
struct alignas(16) StageLds {
    float obs[kBlock * Dims<KIND>::D];
    uint8_t mask[kBlock * Dims<KIND>::A + 16];
};

// auto-reset of a finished episode inside the step (reset(seed=None) semantics: streams continue)
template <int KIND>
__device__ __forceinline__ void auto_reset_env(Env &e, const Params &P, uint4 *__restrict__ planes, long long i,
                                               double purity[4])
{
    e.gen2 = unseeded_gen2(e);
    e.episode += 1u;
    reset_episode_state(e, P);
    if (P.track_bales) clear_bales(planes, P.n_pad, i);
#pragma unroll
    for (int m = 0; m < 4; ++m) purity[m] = P.thr_r2[m];
}

// ==========================================================================================
// kernels
// ==========================================================================================
template <int KIND, bool NOISE, bool LITERAL>
__global__ __launch_bounds__(kBlock) void k_step(Params P, uint4 *__restrict__ planes,
                                                 const int *__restrict__ action, const int *__restrict__ sort_mode,
                                                 uint32_t flags, float *__restrict__ obs_out,
                                                 float *__restrict__ reward_out, double *__restrict__ reward64_out,
                                                 uint8_t *__restrict__ done_out, uint8_t *__restrict__ mask_out,
                                                 float *__restrict__ terminal_obs_out,
                                                 unsigned long long *__restrict__ err_count)
{
    constexpr int D = Dims<KIND>::D, A = Dims<KIND>::A;
    __shared__ StageLds<KIND> lds;
    const int tid = threadIdx.x;
    const long long row0 = (long long)blockIdx.x * kBlock;
    const long long i = row0 + tid;
    const bool live = i < P.n;
    float o[D];
    uint32_t mbits = 0;
#pragma unroll
    for (int j = 0; j < D; ++j) o[j] = 0.0f;

    if (live) {
        Env e;
        load_env<KIND, NOISE>(e, planes, P.n_pad, i);
        int a = action[i];
        if (a < 0 || a >= A) {
            atomicAdd(err_count, 1ull);
            a = 0;
        }
        int sm = (KIND == 2 && sort_mode != nullptr) ? sort_mode[i] : -1;
        double purity[4];
        StepResult r = env_step<KIND, NOISE, LITERAL>(e, P, a, sm, flags, planes, i, purity);
        env_obs<KIND>(e, P, purity, o);
        if (r.done && P.auto_reset) {
            if (terminal_obs_out != nullptr) {
#pragma unroll
                for (int j = 0; j < D; ++j) terminal_obs_out[i * D + j] = o[j];
            }
            auto_reset_env<KIND>(e, P, planes, i, purity);
            env_obs<KIND>(e, P, purity, o);
        }
        mbits = action_mask_bits<KIND>(e, P);
        store_env<KIND, NOISE>(e, planes, P.n_pad, i, false);
        if (reward_out != nullptr) reward_out[i] = (float)r.reward;
        if (reward64_out != nullptr) reward64_out[i] = r.reward;
        if (done_out != nullptr) done_out[i] = (uint8_t)r.done;
    }
    long long rem = P.n - row0;
    const int n_valid = rem >= kBlock ? kBlock : (rem > 0 ? (int)rem : 0);
    if (obs_out != nullptr) stage_rows_f32<D>(lds.obs, o, obs_out + row0 * D, n_valid, tid);
    if (mask_out != nullptr) stage_rows_mask<A>(lds.mask, mbits, mask_out + row0 * A, n_valid, tid);
}

template <int KIND, bool NOISE, bool LITERAL>
__global__ __launch_bounds__(kBlock) void k_rollout(Params P, uint4 *__restrict__ planes, int k_steps,
                                                    uint64_t policy_seed, uint64_t policy_t0,
                                                    const int *__restrict__ sort_mode, uint32_t flags,
                                                    int *__restrict__ actions_out, float *__restrict__ obs_out,
                                                    float *__restrict__ reward_out, uint8_t *__restrict__ done_out,
                                                    uint8_t *__restrict__ mask_out)
{
    constexpr int D = Dims<KIND>::D, A = Dims<KIND>::A;
    __shared__ StageLds<KIND> lds;
    const int tid = threadIdx.x;
    const long long row0 = (long long)blockIdx.x * kBlock;
    const long long i = row0 + tid;
    const bool live = i < P.n;
    long long rem = P.n - row0;
    const int n_valid = rem >= kBlock ? kBlock : (rem > 0 ? (int)rem : 0);

    Env e;
    int sm = -1;
    if (live) {
        load_env<KIND, NOISE>(e, planes, P.n_pad, i);
        if (KIND == 2 && sort_mode != nullptr) sm = sort_mode[i];
    }
    float o[D];
#pragma unroll
    for (int j = 0; j < D; ++j) o[j] = 0.0f;

    for (int k = 0; k < k_steps; ++k) {
        uint32_t mbits = 0;
        const long long srow = (long long)k * P.n + row0;
        if (live) {
            // masked-uniform policy on the current state (env_monolith.py:152-158 with masking)
            uint32_t cur = action_mask_bits<KIND>(e, P);
            uint32_t cnt = (uint32_t)__popc(cur);
            uint32_t rr = policy_u32(policy_seed, (uint64_t)(P.index_offset + i), policy_t0 + (uint64_t)k);
            int a = select_kth_bit(cur, (int)(((uint64_t)rr * cnt) >> 32));
            double purity[4];
            StepResult r = env_step<KIND, NOISE, LITERAL>(e, P, a, sm, flags, planes, i, purity);
            if (r.done) auto_reset_env<KIND>(e, P, planes, i, purity);
            env_obs<KIND>(e, P, purity, o);
            mbits = action_mask_bits<KIND>(e, P);
            if (actions_out != nullptr) actions_out[(long long)k * P.n + i] = a;
            if (reward_out != nullptr) reward_out[(long long)k * P.n + i] = (float)r.reward;
            if (done_out != nullptr) done_out[(long long)k * P.n + i] = (uint8_t)r.done;
        }
        if (obs_out != nullptr) stage_rows_f32<D>(lds.obs, o, obs_out + srow * D, n_valid, tid);
        if (mask_out != nullptr) stage_rows_mask<A>(lds.mask, mbits, mask_out + srow * A, n_valid, tid);
    }
    if (live) store_env<KIND, NOISE>(e, planes, P.n_pad, i, false);
}

template <int KIND>
__global__ __launch_bounds__(kBlock) void k_reset(Params P, uint4 *__restrict__ planes,
                                                  const uint64_t *__restrict__ seeds,
                                                  const uint8_t *__restrict__ which, float *__restrict__ obs_out,
                                                  uint8_t *__restrict__ mask_out)
{
    constexpr int D = Dims<KIND>::D, A = Dims<KIND>::A;
    const long long i = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (i >= P.n) return;
    Env e;
    load_env<1, true>(e, planes, P.n_pad, i); // every plane, whatever the env kind
    const bool doit = which == nullptr || which[i] != 0;
    bool reseeded = false;
    if (doit) {
        if (seeds != nullptr) {
            // env_super.py:375-378: new SeasonalInputGenerator(seed) and set_seed(seed)
            const uint64_t seed = seeds[i];
            Pcg gen = pcg_seed(seed);
            uint64_t r = pcg_next64(gen);
            // permutation([1,2]) swaps iff the first buffered uint32 is even (utils/input_generator.py:28-30)
            e.gen2 = (((uint32_t)r) & 1u) == 0u ? 1 : 0;
            e.press = pcg_seed(seed + 3);
            e.press_has = 0;
            e.press_uint = 0;
            e.noise = pcg_seed(seed + 4);
            e.rng = pcg_seed(seed + 99);
            e.episode = 1u; // episodes are counted from the last seeded reset
            reseeded = true;
        } else {
            e.gen2 = unseeded_gen2(e);
            e.episode += 1u;
        }
        reset_episode_state(e, P);
        clear_bales(planes, P.n_pad, i);
        store_env<1, true>(e, planes, P.n_pad, i, reseeded);
    }
    if (obs_out != nullptr) {
        double purity[4];
        container_purity(e, P, purity);
        float o[D];
        env_obs<KIND>(e, P, purity, o);
#pragma unroll
        for (int j = 0; j < D; ++j) obs_out[i * D + j] = o[j];
    }
    if (mask_out != nullptr) {
        uint32_t bits = action_mask_bits<KIND>(e, P);
#pragma unroll
        for (int j = 0; j < A; ++j) mask_out[i * A + j] = (uint8_t)((bits >> j) & 1u);
    }
}

template <int KIND>
__global__ __launch_bounds__(kBlock) void k_masks(Params P, const uint4 *__restrict__ planes,
                                                  uint8_t *__restrict__ mask_out)
{
    constexpr int A = Dims<KIND>::A;
    const long long i = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (i >= P.n) return;
    Env e;
    load_env<KIND, false>(e, planes, P.n_pad, i);
    uint32_t bits = action_mask_bits<KIND>(e, P);
#pragma unroll
    for (int j = 0; j < A; ++j) mask_out[i * A + j] = (uint8_t)((bits >> j) & 1u);
}

template <int KIND>
__global__ __launch_bounds__(kBlock) void k_sample(Params P, const uint4 *__restrict__ planes, uint64_t policy_seed,
                                                   uint64_t policy_t, int *__restrict__ action_out)
{
    const long long i = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (i >= P.n) return;
    Env e;
    load_env<KIND, false>(e, planes, P.n_pad, i);
    uint32_t cur = action_mask_bits<KIND>(e, P);
    uint32_t cnt = (uint32_t)__popc(cur);
    uint32_t rr = policy_u32(policy_seed, (uint64_t)(P.index_offset + i), policy_t);
    action_out[i] = select_kth_bit(cur, (int)(((uint64_t)rr * cnt) >> 32));
}

// snapshot record <-> planes (column map: include/mse.h MSE_SNAP_*, shared with oracle/oracle.py SNAP)
__global__ __launch_bounds__(kBlock) void k_get_state(Params P, const uint4 *__restrict__ planes,
                                                      long long *__restrict__ I, double *__restrict__ Dd,
                                                      unsigned long long *__restrict__ R)
{
    const long long i = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (i >= P.n) return;
    Env e;
    load_env<1, true>(e, planes, P.n_pad, i);
    if (I != nullptr) {
        long long *r = I + i * MSE_SNAP_INTS;
        for (int c = 0; c < MSE_SNAP_INTS; ++c) r[c] = 0;
        for (int m = 0; m < 4; ++m) {
            r[0 + m] = e.in[m];
            r[4 + m] = e.belt[m];
            r[8 + m] = e.sort[m];
            r[12 + m] = e.ct[m];
            r[16 + m] = e.cf[m];
        }
        r[20] = e.ce;
        for (int p = 0; p < 2; ++p) {
            r[21 + p] = e.timer[p];
            r[23 + p] = e.pmat[p] == 0xFF ? -1 : e.pmat[p];
            r[25 + p] = e.pn[p];
            r[27 + p] = e.q100[p];
        }
        r[29] = e.mode;
        r[30] = e.lps;
        r[31] = e.lpa;
        r[32] = e.step;
        r[33] = e.gen2 ? 2 : 1;
        r[34] = e.gen_idx;
        r[35] = e.gen_cnt;
        for (int m = 0; m < 5; ++m) {
            uint4 c = planes[(long long)(PL_BALE0 + m) * P.n_pad + i];
            r[36 + m] = c.x;
            r[41 + m] = c.y;
            r[46 + m] = c.z;
            r[51 + m] = c.w;
        }
        r[56] = -1; // ledger codes are not kept per lane
        r[57] = -1;
        r[58] = 0;
        r[59] = -1;
        r[60] = e.episode;
    }
    if (Dd != nullptr)
        for (int m = 0; m < 4; ++m) Dd[i * 4 + m] = e.acc[m];
    if (R != nullptr) {
        unsigned long long *w = R + i * 18;
        w[0] = e.rng.s_hi; w[1] = e.rng.s_lo; w[2] = e.rng.i_hi; w[3] = e.rng.i_lo; w[4] = 0; w[5] = 0;
        w[6] = e.noise.s_hi; w[7] = e.noise.s_lo; w[8] = e.noise.i_hi; w[9] = e.noise.i_lo; w[10] = 0; w[11] = 0;
        w[12] = e.press.s_hi; w[13] = e.press.s_lo; w[14] = e.press.i_hi; w[15] = e.press.i_lo;
        w[16] = (unsigned long long)e.press_has;
        w[17] = e.press_uint;
    }
}

__global__ __launch_bounds__(kBlock) void k_set_state(Params P, uint4 *__restrict__ planes,
                                                      const long long *__restrict__ I, const double *__restrict__ Dd,
                                                      const unsigned long long *__restrict__ R)
{
    const long long i = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (i >= P.n) return;
    Env e;
    load_env<1, true>(e, planes, P.n_pad, i);
    if (I != nullptr) {
        const long long *r = I + i * MSE_SNAP_INTS;
        for (int m = 0; m < 4; ++m) {
            e.in[m] = (int)r[0 + m];
            e.belt[m] = (int)r[4 + m];
            e.sort[m] = (int)r[8 + m];
            e.ct[m] = (int)r[12 + m];
            e.cf[m] = (int)r[16 + m];
        }
        e.ce = (int)r[20];
        for (int p = 0; p < 2; ++p) {
            e.timer[p] = (int)r[21 + p];
            e.pmat[p] = r[23 + p] < 0 ? 0xFF : (int)r[23 + p];
            e.pn[p] = (int)r[25 + p];
            e.q100[p] = (int)r[27 + p];
        }
        e.mode = (int)r[29];
        e.lps = (int)r[30];
        e.lpa = (int)r[31];
        e.step = (int)r[32];
        e.gen2 = r[33] == 2 ? 1 : 0;
        e.gen_idx = (int)r[34];
        e.gen_cnt = (int)r[35];
        for (int m = 0; m < 5; ++m)
            planes[(long long)(PL_BALE0 + m) * P.n_pad + i] =
                make_uint4((uint32_t)r[36 + m], (uint32_t)r[41 + m], (uint32_t)r[46 + m], (uint32_t)r[51 + m]);
        e.episode = (uint32_t)r[60];
    }
    if (Dd != nullptr)
        for (int m = 0; m < 4; ++m) e.acc[m] = Dd[i * 4 + m];
    if (R != nullptr) {
        const unsigned long long *w = R + i * 18;
        e.rng.s_hi = w[0]; e.rng.s_lo = w[1]; e.rng.i_hi = w[2]; e.rng.i_lo = w[3];
        e.noise.s_hi = w[6]; e.noise.s_lo = w[7]; e.noise.i_hi = w[8]; e.noise.i_lo = w[9];
        e.press.s_hi = w[12]; e.press.s_lo = w[13]; e.press.i_hi = w[14]; e.press.i_lo = w[15];
        e.press_has = (int)w[16];
        e.press_uint = (uint32_t)w[17];
    }
    store_env<1, true>(e, planes, P.n_pad, i, true);
}

// ==========================================================================================
// host side of the C ABI
// ==========================================================================================
struct mse_env {
    Params P;
    mse_config cfg;
    uint4 *planes;
    unsigned long long *err_count;
    int device;
    bool seeded;
    bool noise_on;
    uint64_t policy_t;
};

static thread_local std::string g_last_error;

static int fail(int status, const std::string &msg)
{
    g_last_error = msg;
    return status;
}

#define MSE_HIP(call)                                                                                    \
    do {                                                                                                 \
        hipError_t _e = (call);                                                                          \
        if (_e != hipSuccess)                                                                            \
            return fail(MSE_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(_e));                 \
    } while (0)

static bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

static inline dim3 grid_of(const mse_env *h) { return dim3((unsigned)(h->P.n_pad / kBlock)); }

template <int KIND>
static void launch_step(mse_env *h, hipStream_t s, const int32_t *action, const int32_t *sort_mode, uint32_t flags,
                        float *obs, float *rew, double *rew64, uint8_t *done, uint8_t *mask, float *tobs)
{
    const bool lit = h->cfg.literal_choice != 0;
#define MSE_LAUNCH_STEP(NOISE, LIT)                                                                      \
    hipLaunchKernelGGL((k_step<KIND, NOISE, LIT>), grid_of(h), dim3(kBlock), 0, s, h->P, h->planes, action, \
                       sort_mode, flags, obs, rew, rew64, done, mask, tobs, h->err_count)
    if (h->noise_on) {
        if (lit) MSE_LAUNCH_STEP(true, true); else MSE_LAUNCH_STEP(true, false);
    } else {
        if (lit) MSE_LAUNCH_STEP(false, true); else MSE_LAUNCH_STEP(false, false);
    }
#undef MSE_LAUNCH_STEP
}

template <int KIND>
static void launch_rollout(mse_env *h, hipStream_t s, int k_steps, uint64_t policy_seed, const int32_t *sort_mode,
                           uint32_t flags, int32_t *actions, float *obs, float *rew, uint8_t *done, uint8_t *mask)
{
    const bool lit = h->cfg.literal_choice != 0;
#define MSE_LAUNCH_ROLLOUT(NOISE, LIT)                                                                   \
    hipLaunchKernelGGL((k_rollout<KIND, NOISE, LIT>), grid_of(h), dim3(kBlock), 0, s, h->P, h->planes, k_steps, \
                       policy_seed, h->policy_t, sort_mode, flags, actions, obs, rew, done, mask)
    if (h->noise_on) {
        if (lit) MSE_LAUNCH_ROLLOUT(true, true); else MSE_LAUNCH_ROLLOUT(true, false);
    } else {
        if (lit) MSE_LAUNCH_ROLLOUT(false, true); else MSE_LAUNCH_ROLLOUT(false, false);
    }
#undef MSE_LAUNCH_ROLLOUT
}

extern "C" {

int mse_version(void) { return MSE_VERSION; }

const char *mse_last_error(void) { return g_last_error.c_str(); }

const char *mse_status_string(int status)
{
    switch (status) {
    case MSE_OK: return "ok";
    case MSE_ERR_INVALID_ARGUMENT: return "invalid argument";
    case MSE_ERR_UNSUPPORTED_CONFIG: return "unsupported configuration";
    case MSE_ERR_NO_DEVICE: return "no HIP device";
    case MSE_ERR_HIP: return "HIP runtime error";
    case MSE_ERR_NOT_RESET: return "mse_reset with seeds has not run yet";
    case MSE_ERR_ALIGNMENT: return "buffer not 16-byte aligned";
    default: return "unknown status";
    }
}

int mse_config_default(mse_config *c)
{
    if (!c) return fail(MSE_ERR_INVALID_ARGUMENT, "cfg is NULL");
    std::memset(c, 0, sizeof(*c));
    c->struct_size = (uint32_t)sizeof(mse_config);
    c->env_kind = MSE_ENV_MONO;
    c->max_steps = 50;
    c->auto_reset = 1;
    c->track_bales = 1;
    c->literal_choice = 0;
    c->input_batch_size = 100;
    c->steps_per_pattern = 20;
    for (int m = 0; m < 4; ++m) {
        c->baseline_accuracy[m] = 0.75;
        c->quality_threshold[m] = 0.9;
        c->quality_threshold_r2[m] = 0.9;
    }
    c->boost = 0.5;
    c->noise = 0.05;
    c->stage_capacity = 100;
    c->press_time[0] = 12;
    c->press_time[1] = 15;
    c->container_capacity = 700;
    c->bale_standard_size = 200;
    c->bale_remainder_threshold = 0.5;
    c->purity_threshold_theta = 0.80;
    c->tanh_temperature = 0.5;
    c->overflow_penalty_catastrophic = -1.0;
    c->overflow_penalty_severe = -0.5;
    c->overflow_penalty_mild = -0.2;
    c->bale_efficiency_factor = 1.0;
    c->max_state_reward = 0.5;
    c->overflow_termination_penalty = -10.0;
    const double p1[4] = {0.40, 0.15, 0.35, 0.10}, p2[4] = {0.15, 0.40, 0.10, 0.35};
    std::memcpy(c->pattern_ratio[0], p1, sizeof(p1));
    std::memcpy(c->pattern_ratio[1], p2, sizeof(p2));
    return MSE_OK;
}

int mse_create(mse_env **out, const mse_config *cfg, int64_t n_envs, int device_id)
{
    return mse_create_indexed(out, cfg, n_envs, device_id, 0);
}

int mse_create_indexed(mse_env **out, const mse_config *cfg, int64_t n_envs, int device_id, int64_t index_offset)
{
    if (!out || !cfg) return fail(MSE_ERR_INVALID_ARGUMENT, "out/cfg is NULL");
    *out = nullptr;
    if (cfg->struct_size != sizeof(mse_config))
        return fail(MSE_ERR_INVALID_ARGUMENT, "mse_config.struct_size mismatch (ABI)");
    if (n_envs <= 0) return fail(MSE_ERR_INVALID_ARGUMENT, "n_envs must be positive");
    if (cfg->env_kind < MSE_ENV_SORT || cfg->env_kind > MSE_ENV_MONO)
        return fail(MSE_ERR_INVALID_ARGUMENT, "env_kind must be 1 (sort), 2 (press) or 3 (mono)");
    if (cfg->max_steps < 1 || cfg->max_steps > 65535)
        return fail(MSE_ERR_UNSUPPORTED_CONFIG, "max_steps must be in [1, 65535]");
    if (cfg->input_batch_size < 1 || cfg->input_batch_size > 255)
        return fail(MSE_ERR_UNSUPPORTED_CONFIG, "input_batch_size must be in [1, 255]");
    if (cfg->press_time[0] < 1 || cfg->press_time[0] > 255 || cfg->press_time[1] < 1 || cfg->press_time[1] > 255)
        return fail(MSE_ERR_UNSUPPORTED_CONFIG, "press_times must be in [1, 255]");
    if (cfg->bale_standard_size < 1 || cfg->container_capacity < 1 || cfg->stage_capacity < 1)
        return fail(MSE_ERR_UNSUPPORTED_CONFIG, "bale_standard_size / container_capacity / stage_capacity must be positive");
    if (!(cfg->noise >= 0.0)) return fail(MSE_ERR_UNSUPPORTED_CONFIG, "noise must be >= 0");

    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0)
        return fail(MSE_ERR_NO_DEVICE, "no HIP device visible: libmse_hip has no CPU path");
    if (device_id < 0 || device_id >= n_dev) return fail(MSE_ERR_NO_DEVICE, "device_id out of range");
    MSE_HIP(hipSetDevice(device_id));

    mse_env *h = new (std::nothrow) mse_env();
    if (!h) return fail(MSE_ERR_INVALID_ARGUMENT, "out of host memory");
    h->cfg = *cfg;
    h->device = device_id;
    h->seeded = false;
    h->policy_t = 0;
    h->noise_on = cfg->noise != 0.0;
    Params &P = h->P;
    std::memset(&P, 0, sizeof(P));
    P.n = n_envs;
    P.n_pad = (n_envs + kBlock - 1) / kBlock * kBlock;
    P.index_offset = index_offset;
    P.env_kind = cfg->env_kind;
    P.max_steps = cfg->max_steps;
    P.auto_reset = cfg->auto_reset ? 1 : 0;
    P.track_bales = cfg->track_bales ? 1 : 0;
    P.balesize = cfg->bale_standard_size;
    P.capacity = cfg->container_capacity;
    P.stage_capacity = cfg->stage_capacity;
    P.batch = cfg->input_batch_size;
    P.press_time[0] = cfg->press_time[0];
    P.press_time[1] = cfg->press_time[1];
    for (int k = 0; k < 2; ++k) {
        int sum = 0;
        for (int m = 0; m < 4; ++m) {
            // utils/input_generator.py:47: int(np.floor(ratio * batchsize))
            P.pat[k][m] = (int)std::floor(cfg->pattern_ratio[k][m] * (double)cfg->input_batch_size);
            sum += P.pat[k][m];
        }
        if (sum != cfg->input_batch_size) {
            delete h;
            return fail(MSE_ERR_UNSUPPORTED_CONFIG,
                        "input_batch_size leaves a floor() remainder for a seasonal pattern: the generator's "
                        "random remainder draws (utils/input_generator.py:50-55) are not on the restated path");
        }
    }
    for (int m = 0; m < 4; ++m) {
        P.base_acc[m] = cfg->baseline_accuracy[m];
        P.thr[m] = cfg->quality_threshold[m];
        P.thr_r2[m] = cfg->quality_threshold_r2[m];
    }
    P.boost = cfg->boost;
    P.noise = cfg->noise;
    P.theta = cfg->purity_threshold_theta;
    P.temperature = cfg->tanh_temperature;
    P.pen_cat = cfg->overflow_penalty_catastrophic;
    P.pen_sev = cfg->overflow_penalty_severe;
    P.pen_mild = cfg->overflow_penalty_mild;
    P.bef = cfg->bale_efficiency_factor;
    P.max_state_reward = cfg->max_state_reward;
    P.overflow_pen = cfg->overflow_termination_penalty;
    P.rem_thr = cfg->bale_remainder_threshold;

    size_t bytes = (size_t)PL_COUNT * (size_t)P.n_pad * sizeof(uint4);
    hipError_t e1 = hipMalloc(reinterpret_cast<void **>(&h->planes), bytes);
    if (e1 != hipSuccess) {
        delete h;
        return fail(MSE_ERR_HIP, std::string("hipMalloc(state planes): ") + hipGetErrorString(e1));
    }
    hipError_t e2 = hipMalloc(reinterpret_cast<void **>(&h->err_count), sizeof(unsigned long long));
    if (e2 != hipSuccess) {
        (void)hipFree(h->planes);
        delete h;
        return fail(MSE_ERR_HIP, std::string("hipMalloc(err_count): ") + hipGetErrorString(e2));
    }
    MSE_HIP(hipMemset(h->planes, 0, bytes));
    MSE_HIP(hipMemset(h->err_count, 0, sizeof(unsigned long long)));
    *out = h;
    return MSE_OK;
}

int mse_destroy(mse_env *h)
{
    if (!h) return MSE_OK;
    (void)hipSetDevice(h->device);
    (void)hipFree(h->planes);
    (void)hipFree(h->err_count);
    delete h;
    return MSE_OK;
}

int64_t mse_num_envs(const mse_env *h) { return h ? h->P.n : 0; }
int mse_obs_dim(const mse_env *h) { return !h ? 0 : (h->P.env_kind == 1 ? 13 : (h->P.env_kind == 2 ? 16 : 29)); }
int mse_num_actions(const mse_env *h) { return !h ? 0 : (h->P.env_kind == 1 ? 2 : (h->P.env_kind == 2 ? 11 : 22)); }

// minimal state read + state write + outputs + action, per env-step
int mse_algorithmic_bytes_per_step(const mse_env *h)
{
    if (!h) return 0;
    int noise_extra = h->noise_on ? 48 : 0; // rng_noise 32 R + 16 W
    switch (h->P.env_kind) {
    case MSE_ENV_SORT: return 363 + noise_extra;
    case MSE_ENV_PRESS: return 332 + noise_extra;
    default: return 391 + noise_extra;
    }
}


#define MSE_CHECK_LAUNCH()                                                                               \
    do {                                                                                                 \
        hipError_t _e = hipGetLastError();                                                               \
        if (_e != hipSuccess) return fail(MSE_ERR_HIP, std::string("kernel launch: ") + hipGetErrorString(_e)); \
    } while (0)

int mse_reset(mse_env *h, const uint64_t *seeds, const uint8_t *which, float *obs_out, uint8_t *mask_out, void *stream)
{
    if (!h) return fail(MSE_ERR_INVALID_ARGUMENT, "env is NULL");
    if (!seeds && !h->seeded)
        return fail(MSE_ERR_NOT_RESET, "the first mse_reset must carry seeds (the streams are not seeded yet)");
    if (seeds && which && !h->seeded)
        return fail(MSE_ERR_NOT_RESET, "the first mse_reset must seed every env (which_dev must be NULL)");
    hipStream_t s = static_cast<hipStream_t>(stream);
    switch (h->P.env_kind) {
    case 1: hipLaunchKernelGGL(k_reset<1>, grid_of(h), dim3(kBlock), 0, s, h->P, h->planes, seeds, which, obs_out, mask_out); break;
    case 2: hipLaunchKernelGGL(k_reset<2>, grid_of(h), dim3(kBlock), 0, s, h->P, h->planes, seeds, which, obs_out, mask_out); break;
    default: hipLaunchKernelGGL(k_reset<3>, grid_of(h), dim3(kBlock), 0, s, h->P, h->planes, seeds, which, obs_out, mask_out); break;
    }
    MSE_CHECK_LAUNCH();
    if (seeds) h->seeded = true;
    return MSE_OK;
}


int mse_step(mse_env *h, const int32_t *action, const int32_t *sort_mode, uint32_t flags, float *obs_out,
             float *reward_out, double *reward64_out, uint8_t *done_out, uint8_t *mask_out, float *terminal_obs_out,
             void *stream)
{
    if (!h) return fail(MSE_ERR_INVALID_ARGUMENT, "env is NULL");
    if (!h->seeded) return fail(MSE_ERR_NOT_RESET, "mse_step before mse_reset(seeds)");
    if (!action) return fail(MSE_ERR_INVALID_ARGUMENT, "action_dev is NULL");
    if (flags & ~(MSE_STEP_UNMASKED | MSE_STEP_CHECK_OVERFLOW)) return fail(MSE_ERR_INVALID_ARGUMENT, "unknown step flag");
    if ((obs_out && !aligned16(obs_out)) || (mask_out && !aligned16(mask_out)))
        return fail(MSE_ERR_ALIGNMENT, "obs_out / mask_out must be 16-byte aligned");
    hipStream_t s = static_cast<hipStream_t>(stream);
    switch (h->P.env_kind) {
    case 1: launch_step<1>(h, s, action, sort_mode, flags, obs_out, reward_out, reward64_out, done_out, mask_out, terminal_obs_out); break;
    case 2: launch_step<2>(h, s, action, sort_mode, flags, obs_out, reward_out, reward64_out, done_out, mask_out, terminal_obs_out); break;
    default: launch_step<3>(h, s, action, sort_mode, flags, obs_out, reward_out, reward64_out, done_out, mask_out, terminal_obs_out); break;
    }
    MSE_CHECK_LAUNCH();
    h->policy_t += 1;
    return MSE_OK;
}


int mse_rollout(mse_env *h, int32_t k_steps, uint64_t policy_seed, const int32_t *sort_mode, uint32_t flags,
                int32_t *actions_out, float *obs_out, float *reward_out, uint8_t *done_out, uint8_t *mask_out,
                void *stream)
{
    if (!h) return fail(MSE_ERR_INVALID_ARGUMENT, "env is NULL");
    if (!h->seeded) return fail(MSE_ERR_NOT_RESET, "mse_rollout before mse_reset(seeds)");
    if (k_steps < 1) return fail(MSE_ERR_INVALID_ARGUMENT, "k_steps must be >= 1");
    if (!h->P.auto_reset) return fail(MSE_ERR_INVALID_ARGUMENT, "mse_rollout needs auto_reset=1");
    if (flags & ~(MSE_STEP_UNMASKED | MSE_STEP_CHECK_OVERFLOW)) return fail(MSE_ERR_INVALID_ARGUMENT, "unknown step flag");
    hipStream_t s = static_cast<hipStream_t>(stream);
    switch (h->P.env_kind) {
    case 1: launch_rollout<1>(h, s, k_steps, policy_seed, sort_mode, flags, actions_out, obs_out, reward_out, done_out, mask_out); break;
    case 2: launch_rollout<2>(h, s, k_steps, policy_seed, sort_mode, flags, actions_out, obs_out, reward_out, done_out, mask_out); break;
    default: launch_rollout<3>(h, s, k_steps, policy_seed, sort_mode, flags, actions_out, obs_out, reward_out, done_out, mask_out); break;
    }
    MSE_CHECK_LAUNCH();
    h->policy_t += (uint64_t)k_steps;
    return MSE_OK;
}

int mse_sample_actions(mse_env *h, uint64_t policy_seed, int32_t *action_out, void *stream)
{
    if (!h || !action_out) return fail(MSE_ERR_INVALID_ARGUMENT, "env/action_out is NULL");
    if (!h->seeded) return fail(MSE_ERR_NOT_RESET, "mse_sample_actions before mse_reset(seeds)");
    hipStream_t s = static_cast<hipStream_t>(stream);
    switch (h->P.env_kind) {
    case 1: hipLaunchKernelGGL(k_sample<1>, grid_of(h), dim3(kBlock), 0, s, h->P, h->planes, policy_seed, h->policy_t, action_out); break;
    case 2: hipLaunchKernelGGL(k_sample<2>, grid_of(h), dim3(kBlock), 0, s, h->P, h->planes, policy_seed, h->policy_t, action_out); break;
    default: hipLaunchKernelGGL(k_sample<3>, grid_of(h), dim3(kBlock), 0, s, h->P, h->planes, policy_seed, h->policy_t, action_out); break;
    }
    MSE_CHECK_LAUNCH();
    return MSE_OK;
}

int mse_action_masks(mse_env *h, uint8_t *mask_out, void *stream)
{
    if (!h || !mask_out) return fail(MSE_ERR_INVALID_ARGUMENT, "env/mask_out is NULL");
    hipStream_t s = static_cast<hipStream_t>(stream);
    switch (h->P.env_kind) {
    case 1: hipLaunchKernelGGL(k_masks<1>, grid_of(h), dim3(kBlock), 0, s, h->P, h->planes, mask_out); break;
    case 2: hipLaunchKernelGGL(k_masks<2>, grid_of(h), dim3(kBlock), 0, s, h->P, h->planes, mask_out); break;
    default: hipLaunchKernelGGL(k_masks<3>, grid_of(h), dim3(kBlock), 0, s, h->P, h->planes, mask_out); break;
    }
    MSE_CHECK_LAUNCH();
    return MSE_OK;
}

int mse_get_state(mse_env *h, int64_t *ints_out, double *dbls_out, uint64_t *rng_out, void *stream)
{
    if (!h) return fail(MSE_ERR_INVALID_ARGUMENT, "env is NULL");
    hipLaunchKernelGGL(k_get_state, grid_of(h), dim3(kBlock), 0, static_cast<hipStream_t>(stream), h->P, h->planes,
                       reinterpret_cast<long long *>(ints_out), dbls_out, reinterpret_cast<unsigned long long *>(rng_out));
    MSE_CHECK_LAUNCH();
    return MSE_OK;
}

int mse_set_state(mse_env *h, const int64_t *ints_in, const double *dbls_in, const uint64_t *rng_in, void *stream)
{
    if (!h) return fail(MSE_ERR_INVALID_ARGUMENT, "env is NULL");
    hipLaunchKernelGGL(k_set_state, grid_of(h), dim3(kBlock), 0, static_cast<hipStream_t>(stream), h->P, h->planes,
                       reinterpret_cast<const long long *>(ints_in), dbls_in,
                       reinterpret_cast<const unsigned long long *>(rng_in));
    MSE_CHECK_LAUNCH();
    if (rng_in) h->seeded = true;
    return MSE_OK;
}

int mse_error_count(mse_env *h, uint64_t *count_out)
{
    if (!h || !count_out) return fail(MSE_ERR_INVALID_ARGUMENT, "env/count_out is NULL");
    unsigned long long v = 0;
    MSE_HIP(hipMemcpy(&v, h->err_count, sizeof(v), hipMemcpyDeviceToHost));
    *count_out = v;
    return MSE_OK;
}

} // extern "C"
