// mse_lib.hip -- kernels and the C ABI (include/mse.h) of libmse_hip.so.  gfx950 only.
//
// Build (see marl-sortingenv_amd/build.py):
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -std=c++17 -fPIC -shared -Iinclude \
//         marl-sortingenv_amd/csrc/mse_lib.hip -o marl-sortingenv_amd/libmse_hip.so
#include "mse_device.h"
#include "mse_policy_device.h"

#include "mse.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

using namespace mse;

// ==========================================================================================
// LDS: [obs staging f32 256 x D][mask staging u8 256 x A (+pad)][tables]   (all dynamic, 16-B aligned base)
// Rows computed one-per-lane are transposed through LDS so that global stores are 16 B per lane and
// contiguous across the workgroup (obs rows are 52/64/116 B long).  The barriers wait for LDS only:
// a __syncthreads() would also drain the global stores of the previous step (vmcnt(0)), which was 37 %
// of the wave time in the first profile (profiles/r01/v1_baseline_summary.json).
// ==========================================================================================
extern __shared__ uint4 mse_dyn_lds[];

template <int KIND>
struct LdsLayout {
    static constexpr int D = Dims<KIND>::D, A = Dims<KIND>::A;
    static constexpr int obs_bytes = kBlock * D * 4;
    static constexpr int mask_bytes = (kBlock * A + 15) / 16 * 16;
    static constexpr int bale_offset = obs_bytes + mask_bytes;    // [5][kBlock] uint4, rollout kernel only
    static constexpr int bale_bytes = 5 * kBlock * 16;
    static constexpr int table_offset_step = bale_offset;         // k_step keeps the ledger in global memory
    static constexpr int table_offset_rollout = bale_offset + bale_bytes;
};

// workgroup barrier that waits for this wave's LDS traffic only (global stores stay in flight)
__device__ __forceinline__ void lds_barrier_all()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// Table image -> LDS in 16-byte pieces, every load of a thread issued before its first LDS write (a word-by-word
// loop is one global round trip per iteration: 12 of them with 256 threads).  issue() / commit() are separate so a
// kernel can put other loads in flight between them.  `words` is a multiple of 4 (build_tables pads).
template <int THREADS, int U>
struct TableCopy {
    uint4 x[U];
    // loads are unconditional at a clamped index (a conditional load makes the compiler keep x[] in scratch memory:
    // load, wait, scratch store, scratch load, wait, LDS write - two memory round trips in the launch prologue)
    __device__ __forceinline__ void issue(const uint32_t *__restrict__ src, int words, int t)
    {
        const uint4 *s4 = reinterpret_cast<const uint4 *>(src);
        const int n4 = words >> 2;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int idx = u * THREADS + t;
            x[u] = s4[idx < n4 ? idx : n4 - 1];
        }
    }
    __device__ __forceinline__ void commit(uint32_t *dst, const uint32_t *__restrict__ src, int words, int t) const
    {
        uint4 *d4 = reinterpret_cast<uint4 *>(dst);
        const uint4 *s4 = reinterpret_cast<const uint4 *>(src);
        const int n4 = words >> 2;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int idx = u * THREADS + t;
            if (idx < n4) d4[idx] = x[u];
        }
        for (int idx = U * THREADS + t; idx < n4; idx += THREADS) d4[idx] = s4[idx]; // larger images than U pieces
    }
};

__device__ __forceinline__ void load_tables_to_lds(uint32_t *dst, const uint32_t *__restrict__ src, int words, int tid)
{
    TableCopy<kBlock, 4> tc;
    tc.issue(src, words, tid);
    tc.commit(dst, src, words, tid);
    __syncthreads();
}

// Streaming 16-byte store of an output piece: the rollout buffers are written once and read by someone else much
// later, while the state planes (9 MB at 65 536 envs) are read back by the very next launch - keep the outputs from
// pushing them out of L2.
typedef float nt_f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int nt_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store_stream(float4 *p, const float4 &v)
{
    nt_f32x4 x = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(x, reinterpret_cast<nt_f32x4 *>(p));
}
__device__ __forceinline__ void store_stream(uint4 *p, const uint4 &v)
{
    nt_u32x4 x = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(x, reinterpret_cast<nt_u32x4 *>(p));
}

// A row of A action-mask bytes (A even) from the mask bits, to a 2-byte aligned row: four bytes per dword by one
// 24-bit multiply (x * 0x204081 puts bit k of a nibble at bit 7 k + k; the mask keeps bits 0, 8, 16, 24), a trailing
// pair as 16 bits.  The compiler merges the stores (22 bytes: 16 + 4 + 2).  Used by the learned-policy kernel only: in the
// three-role kernel's observer the same 18 instructions measured 1-2 % slower end to end than the compiler's 45 bit operations
// for the pairwise form below (same-box A/B), so stage_and_store keeps that.
typedef uint32_t u32_align2 __attribute__((aligned(2)));
template <int A, class P16>
__device__ __forceinline__ void write_mask_row(P16 *row16, uint32_t mbits)
{
    static_assert(A % 2 == 0, "even rows only");
#pragma unroll
    for (int j = 0; j < A / 4; ++j) {
        const uint32_t four = __umul24((mbits >> (4 * j)) & 0xFu, 0x00204081u) & 0x01010101u;
        row16[2 * j] = (uint16_t)four;
        row16[2 * j + 1] = (uint16_t)(four >> 16);
    }
    if (A % 4 == 2) row16[A / 2 - 1] = (uint16_t)(__umul24((mbits >> (A - 2)) & 0x3u, 0x00204081u) & 0x0101u);
}

// Each WAVE stages the 64 rows of its own lanes and streams them out itself: the rows of one wave are a
// contiguous, 16-byte aligned run of the output (64 x 116 B for obs), and the LDS executes one wave's
// instructions in order, so no workgroup barrier is needed (the first profile showed 36 % of wave time
// parked at the per-step barriers).  gobs / gmask point at the WORKGROUP's first row.
template <int KIND>
__device__ __forceinline__ void stage_and_store(uint8_t *lds_base, int mask_offset, const float *o, uint32_t mbits,
                                                float *gobs, uint8_t *gmask, int n_valid_block, int tid)
{
    constexpr int D = Dims<KIND>::D, A = Dims<KIND>::A;
    const int wave = tid >> 6, lane = tid & 63;
    const int n_valid = min(max(n_valid_block - wave * 64, 0), 64);
    float *lobs = reinterpret_cast<float *>(lds_base) + wave * 64 * D;
    // mask_offset < 0: the wave's mask tile reuses its own obs tile (callers then stage obs and mask in two calls)
    uint8_t *lmask = mask_offset < 0 ? reinterpret_cast<uint8_t *>(lobs) : lds_base + mask_offset + wave * 64 * A;
    if (gobs != nullptr) {
#pragma unroll
        for (int j = 0; j < D; ++j) lobs[lane * D + j] = o[j];
    }
    if (gmask != nullptr) {
        if (A % 2 == 0) { // even row length: two mask bytes per 16-bit LDS store
            uint16_t *row = reinterpret_cast<uint16_t *>(lmask + lane * A);
#pragma unroll
            for (int j = 0; j < A / 2; ++j)
                row[j] = (uint16_t)(((mbits >> (2 * j)) & 1u) | (((mbits >> (2 * j + 1)) & 1u) << 8));
        } else {
#pragma unroll
            for (int j = 0; j < A; ++j) lmask[lane * A + j] = (uint8_t)((mbits >> j) & 1u);
        }
    }
    __builtin_amdgcn_wave_barrier();
    if (gobs != nullptr) {
        float *g = gobs + wave * 64 * D;
        const int total = n_valid * D;
        constexpr int NQ = 64 * D / 4;       // 16-byte pieces of a full tile (64 rows)
        constexpr int NFULL = NQ / 64;       // rounds in which every lane moves one piece
        constexpr int NTAIL = NQ - 64 * NFULL;
        if (n_valid == 64 && (reinterpret_cast<uintptr_t>(g) & 15u) == 0) {
            // full tile: all LDS reads first (unconditional), then the 16-byte stores
            const float4 *src = reinterpret_cast<const float4 *>(lobs);
            float4 *dst = reinterpret_cast<float4 *>(g);
            float4 buf[NFULL + 1];
#pragma unroll
            for (int j = 0; j <= NFULL; ++j) buf[j] = src[lane + 64 * j]; // the last one may run past the tile: unused
#pragma unroll
            for (int j = 0; j < NFULL; ++j) store_stream(dst + lane + 64 * j, buf[j]);
            if (NTAIL > 0 && lane < NTAIL) store_stream(dst + lane + 64 * NFULL, buf[NFULL]);
        } else {
            for (int q = lane; q < total; q += 64) g[q] = lobs[q];
        }
    }
    if (gmask != nullptr) {
        uint8_t *g = gmask + wave * 64 * A;
        const int total = n_valid * A;
        constexpr int NQ = 64 * A / 16; // whole 16-byte pieces of a full tile (64 * A is a multiple of 16 for even A)
        if (n_valid == 64 && (64 * A) % 16 == 0 && NQ <= 128 && (reinterpret_cast<uintptr_t>(g) & 15u) == 0) {
            const uint4 *src = reinterpret_cast<const uint4 *>(lmask);
            uint4 *dst = reinterpret_cast<uint4 *>(g);
            const uint4 b0 = src[lane];
            const uint4 b1 = src[lane + 64]; // may run past the tile: unused
            if (lane < NQ) store_stream(dst + lane, b0);
            if (NQ > 64 && lane + 64 < NQ) store_stream(dst + lane + 64, b1);
        } else {
            for (int q = lane; q < total; q += 64) g[q] = lmask[q];
        }
    }
    __builtin_amdgcn_wave_barrier(); // the tile is rewritten by this wave only, after its own reads
}

// The same staging for the one-lane rollout kernel with HALF the LDS: a wave streams its 64 observation rows in two
// passes of 32 rows through a 32-row tile (lanes 0-31 write theirs, all 64 lanes store; then lanes 32-63), the mask rows
// (64 x A <= 32 x D x 4 bytes) in one pass through the same tile.  LDS per workgroup decides how many workgroups a CU
// holds, and with one lane per env that is the number of waves per SIMD the transition's latencies hide under.
template <int KIND>
__device__ __forceinline__ void stage_and_store_halves(uint8_t *lds_base, const float *o, uint32_t mbits, float *gobs,
                                                       uint8_t *gmask, int n_valid_block, int tid)
{
    constexpr int D = Dims<KIND>::D, A = Dims<KIND>::A;
    static_assert(64 * A <= 32 * D * 4, "the mask tile must fit the half observation tile");
    const int wave = tid >> 6, lane = tid & 63;
    const int n_valid = min(max(n_valid_block - wave * 64, 0), 64);
    float *ltile = reinterpret_cast<float *>(lds_base) + wave * 32 * D;
    if (gobs != nullptr) {
        float *g = gobs + wave * 64 * D;
        constexpr int NQ = 32 * D / 4;                // 16-byte pieces of a half tile
        constexpr int NR = (NQ + 63) / 64;
        const bool fast = n_valid == 64 && (reinterpret_cast<uintptr_t>(g) & 15u) == 0 && (32 * D * 4) % 16 == 0;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            if ((lane >> 5) == half) {
#pragma unroll
                for (int j = 0; j < D; ++j) ltile[(lane & 31) * D + j] = o[j];
            }
            __builtin_amdgcn_wave_barrier();
            float *gh = g + half * 32 * D;
            if (fast) {
                const float4 *src = reinterpret_cast<const float4 *>(ltile);
                float4 buf[NR];
#pragma unroll
                for (int j = 0; j < NR; ++j) buf[j] = src[(lane + 64 * j) < NQ ? lane + 64 * j : 0];
#pragma unroll
                for (int j = 0; j < NR; ++j)
                    if (lane + 64 * j < NQ) store_stream(reinterpret_cast<float4 *>(gh) + lane + 64 * j, buf[j]);
            } else {
                const int rows = min(max(n_valid - half * 32, 0), 32);
                for (int q = lane; q < rows * D; q += 64) gh[q] = ltile[q];
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
    if (gmask != nullptr) {
        uint8_t *lmask = reinterpret_cast<uint8_t *>(ltile);
        uint8_t *g = gmask + wave * 64 * A;
        if (A % 2 == 0) {
            uint16_t *row = reinterpret_cast<uint16_t *>(lmask + lane * A);
#pragma unroll
            for (int j = 0; j < A / 2; ++j)
                row[j] = (uint16_t)(((mbits >> (2 * j)) & 1u) | (((mbits >> (2 * j + 1)) & 1u) << 8));
        } else {
#pragma unroll
            for (int j = 0; j < A; ++j) lmask[lane * A + j] = (uint8_t)((mbits >> j) & 1u);
        }
        __builtin_amdgcn_wave_barrier();
        constexpr int NQ = 64 * A / 16;
        if (n_valid == 64 && (64 * A) % 16 == 0 && NQ <= 128 && (reinterpret_cast<uintptr_t>(g) & 15u) == 0) {
            const uint4 *src = reinterpret_cast<const uint4 *>(lmask);
            const uint4 b0 = src[lane < NQ ? lane : 0];
            const uint4 b1 = src[(lane + 64) < NQ ? lane + 64 : 0];
            if (lane < NQ) store_stream(reinterpret_cast<uint4 *>(g) + lane, b0);
            if (NQ > 64 && lane + 64 < NQ) store_stream(reinterpret_cast<uint4 *>(g) + lane + 64, b1);
        } else {
            for (int q = lane; q < n_valid * A; q += 64) g[q] = lmask[q];
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// LDS of the one-lane rollout kernel: [half observation tiles, 32 rows per wave][bale ledger, 12-byte cells][tables up
// to the jump tables, which only the three-role kernel reads]: 38 KB for Env_3, four workgroups per CU
template <int KIND, bool NOISE = false>
struct RolloutLayout {
    static constexpr int D = Dims<KIND>::D;
    static constexpr int tile_bytes = (kBlock / 2 * D * 4 + 15) / 16 * 16;
    static constexpr int bale_offset = tile_bytes;
    static constexpr int bale_bytes = 5 * 3 * kBlock * 4; // BaleRefCompact: 12 bytes per cell
    // noise > 0: the increment of rng_noise (constant for the launch) waits here between two update_accuracy calls
    // instead of in four registers: at the 168-register cap of three workgroups per CU the Env_3 noise instantiation
    // kept two dwords of the step loop in scratch memory (20.0 G against 23.4 without noise at 262 144 envs)
    static constexpr int noise_offset = bale_offset + bale_bytes;
    static constexpr int noise_bytes = NOISE ? kBlock * 16 : 0;
    static constexpr int table_offset = noise_offset + noise_bytes;
};

// auto-reset of a finished episode inside the step (reset(seed=None) semantics: streams continue)
template <bool GEN = false, class BALES = BaleRef>
__device__ __forceinline__ void auto_reset_env(Env &e, const Params &P, const Tables &tb, const BALES &bales, int k[4])
{
    e.gen2 = unseeded_gen2(e);
    if (GEN) unseeded_generator(e);
    e.episode += 1u;
    reset_episode_state(e, tb.cst);
    if (P.track_bales) clear_bales(bales);
#pragma unroll
    for (int m = 0; m < 4; ++m) k[m] = 101; // empty containers
}

// action for the next step: the on-device masked-uniform policy (env_monolith.py:152-158 with masking) or,
// with MSE_ROLLOUT_RULE_BASED, the reference's rule-based policy
// pkey = mse_policy_key(policy seed, global env index): constant over a launch (mse_policy_stream.h)
template <int KIND, bool GEN = false>
__device__ __forceinline__ int policy_action(const Env &e, uint32_t cur, const Tables &tb, uint32_t flags,
                                             uint32_t pkey, uint64_t t)
{
    if (flags & MSE_ROLLOUT_RULE_BASED) {
        // the batch that will be on the belt: what update_environment moves from the input stage
        return rule_based_action<KIND>(e, Stage<GEN>::rule_mode(e.st_in, tb));
    }
    const uint32_t rr = mse_policy_word(pkey, t);
    // without masking the random mode draws from the whole action space (env_monolith.py:159-162) and the step
    // sanitises what it gets
    if (flags & MSE_STEP_UNMASKED) return (int)(((uint64_t)rr * (uint32_t)Dims<KIND>::A) >> 32);
    const uint32_t cnt = (uint32_t)__popc(cur); // cur = action_masks() of the current state
    return select_kth_bit(cur, (int)(((uint64_t)rr * cnt) >> 32));
}

// ==========================================================================================
// kernels
// ==========================================================================================
// TRACE: the launch also appends one record (include/mse.h MSE_TRACE_*) for env `trace_env` to trace_rec - the
// opt-in single-env trace behind the reference's dashboard ledgers (mse_trace_begin).
template <int KIND, bool NOISE, bool LITERAL, bool TRACE = false, bool GEN = false>
__global__ __launch_bounds__(kBlock) void k_step(Params P, uint4 *__restrict__ planes,
                                                 const uint32_t *__restrict__ table_image,
                                                 const int *__restrict__ action, const int *__restrict__ sort_mode,
                                                 uint32_t flags, float *__restrict__ obs_out,
                                                 float *__restrict__ reward_out, double *__restrict__ reward64_out,
                                                 uint8_t *__restrict__ done_out, uint8_t *__restrict__ mask_out,
                                                 float *__restrict__ terminal_obs_out,
                                                 unsigned long long *__restrict__ err_count,
                                                 double *__restrict__ trace_rec = nullptr, long long trace_env = -1)
{
    constexpr int D = Dims<KIND>::D, A = Dims<KIND>::A;
    uint8_t *lds = reinterpret_cast<uint8_t *>(mse_dyn_lds);
    uint32_t *ltab = reinterpret_cast<uint32_t *>(lds + LdsLayout<KIND>::table_offset_step);
    const int tid = threadIdx.x;
    const long long row0 = (long long)blockIdx.x * kBlock;
    const long long i = row0 + tid;
    const bool live = i < P.n;
    const BaleRef bales{planes + (long long)PL_BALE0 * P.n_pad + i, P.n_pad};
    float o[D];
    uint32_t mbits = 0;
#pragma unroll
    for (int j = 0; j < D; ++j) o[j] = 0.0f;

    // a single-step launch is mostly memory latency: issue the state and action loads first, fill the LDS
    // tables while they fly (the padded planes make the state load safe for every lane)
    Env e;
    load_env<KIND, NOISE>(e, planes, P, i);
    if (GEN) load_gen(e, planes, P, i);
    int a = live ? action[i] : 0;
    int sm = (KIND == 2 && sort_mode != nullptr && live) ? sort_mode[i] : -1;
    load_tables_to_lds(ltab, table_image, P.table_words, tid);
    const Tables tb = tables_at(ltab, P);

    if (live) {
        if (a < 0 || a >= A) {
            atomicAdd(err_count, 1ull);
            a = 0;
        }
        int k[4];
        Ledger lg;
        if (TRACE) {
            lg.n_log = lg.n_bale = 0;
            lg.internal = 0;
            for (int q = 0; q < 2; ++q) lg.code[q] = lg.mat[q] = lg.bmat[q] = lg.bn[q] = lg.bq[q] = -1;
        }
        StepResult r = env_step<KIND, NOISE, LITERAL, TRACE, GEN>(e, P, tb, a, sm, flags, bales, k, o, &lg);
        if (TRACE && i == trace_env) {
            // the state _log_step_data sees: after the step, before any auto-reset
            double *t = trace_rec;
            const uint32_t bw = stage_word(e.st_belt, P);
            t[MSE_TRACE_ACTION] = (double)a;
            t[MSE_TRACE_R_SORT] = r.r_sort;
            t[MSE_TRACE_R_PRESS] = r.r_press;
            t[MSE_TRACE_SETTING] = (double)e.mode;
            for (int m = 0; m < 4; ++m) {
                t[MSE_TRACE_BELT + m] = (double)((bw >> (8 * m)) & 0xFFu);
                t[MSE_TRACE_CONT_TRUE + m] = (double)e.ct[m];
                t[MSE_TRACE_CONT_FALSE + m] = (double)e.cf[m];
            }
            t[MSE_TRACE_CONT_E] = (double)e.ce;
            t[MSE_TRACE_N_LOG] = (double)lg.n_log;
            t[MSE_TRACE_N_BALE] = (double)lg.n_bale;
            for (int q = 0; q < 2; ++q) {
                t[MSE_TRACE_LOG + 2 * q] = (double)lg.code[q];
                t[MSE_TRACE_LOG + 2 * q + 1] = (double)lg.mat[q];
                t[MSE_TRACE_BALE + 3 * q] = (double)lg.bmat[q];
                t[MSE_TRACE_BALE + 3 * q + 1] = (double)lg.bn[q];
                t[MSE_TRACE_BALE + 3 * q + 2] = (double)lg.bq[q];
            }
            t[MSE_TRACE_DONE] = (double)r.done;
            t[MSE_TRACE_STEP] = (double)e.step;
            t[MSE_TRACE_INTERNAL] = (double)lg.internal;
            t[MSE_TRACE_REWARD] = r.reward;
            for (int m = 0; m < 4; ++m) t[MSE_TRACE_ACC_BELT + m] = e.acc[m];
            int ovf = 0; // env_super.py:900-905 detect_overflow: the first material above capacity
            if ((flags & MSE_STEP_CHECK_OVERFLOW) && r.done) {
                for (int m = 4; m >= 0; --m) {
                    const int lvl = m < 4 ? e.ct[m] + e.cf[m] : e.ce;
                    if (lvl > P.capacity) ovf = m + 1;
                }
            }
            t[MSE_TRACE_OVERFLOW] = (double)ovf;
        }
        if (r.done && P.auto_reset) {
            if (terminal_obs_out != nullptr) {
#pragma unroll
                for (int j = 0; j < D; ++j) terminal_obs_out[i * D + j] = o[j];
            }
            auto_reset_env<GEN>(e, P, tb, bales, k);
            env_obs<KIND, GEN>(e, P, tb, k, o);
        }
        mbits = action_mask_bits<KIND>(e, P);
        store_env<KIND, NOISE>(e, planes, P, i, false);
        if (GEN) store_gen(e, planes, P, i);
        if (reward_out != nullptr) reward_out[i] = (float)r.reward;
        if (reward64_out != nullptr) reward64_out[i] = r.reward;
        if (done_out != nullptr) done_out[i] = (uint8_t)r.done;
    }
    long long rem = P.n - row0;
    const int n_valid = rem >= kBlock ? kBlock : (rem > 0 ? (int)rem : 0);
    stage_and_store<KIND>(lds, LdsLayout<KIND>::obs_bytes, o, mbits, obs_out ? obs_out + row0 * D : nullptr,
                          mask_out ? mask_out + row0 * A : nullptr, n_valid, tid);
}

// Three workgroups per CU: three waves per SIMD, so at most 168 VGPRs (left alone the compiler spends 214 here: two).
// The LDS image (38 KB for Env_3) would let a fourth in, but a 128-register build spills 20 dwords inside the step loop
// and measured slower at every size (17.6 / 21.4 / 23.3 / 24.8 G against 20.3 / 23.4 / 23.3 / 26.8 at 131 072 / 196 608 /
// 262 144 / 1 048 576 envs).
template <int KIND, bool NOISE, bool LITERAL, bool GEN = false>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(3))) void k_rollout(Params P, uint4 *__restrict__ planes,
                                                    const uint32_t *__restrict__ table_image, int k_steps,
                                                    uint64_t policy_seed, uint64_t policy_t0,
                                                    const int *__restrict__ sort_mode, uint32_t flags,
                                                    int *__restrict__ actions_out, float *__restrict__ obs_out,
                                                    float *__restrict__ reward_out, uint8_t *__restrict__ done_out,
                                                    uint8_t *__restrict__ mask_out)
{
    constexpr int D = Dims<KIND>::D, A = Dims<KIND>::A;
    uint8_t *lds = reinterpret_cast<uint8_t *>(mse_dyn_lds);
    uint32_t *ltab = reinterpret_cast<uint32_t *>(lds + RolloutLayout<KIND, NOISE>::table_offset);
    uint32_t *lbale = reinterpret_cast<uint32_t *>(lds + RolloutLayout<KIND, NOISE>::bale_offset);
    uint4 *lnoise = reinterpret_cast<uint4 *>(lds + RolloutLayout<KIND, NOISE>::noise_offset);
    const int tid = threadIdx.x;
    const long long row0 = (long long)blockIdx.x * kBlock;
    const long long i = row0 + tid; // i < n_pad always: the planes are padded to whole workgroups
    const bool live = i < P.n;
    long long rem = P.n - row0;
    const int n_valid = rem >= kBlock ? kBlock : (rem > 0 ? (int)rem : 0);

    // the bale ledger of this workgroup's envs stays in LDS for the whole launch
    const BaleRefCompact bales{lbale + tid, kBlock};
    if (P.track_bales) {
#pragma unroll
        for (int m = 0; m < 5; ++m) bales.store(m, planes[(long long)(PL_BALE0 + m) * P.n_pad + i]);
    }
    load_tables_to_lds(ltab, table_image, GEN ? P.table_words : P.off_jump, tid); // off_jump is a multiple of 4 words
    const Tables tb = tables_at(ltab, P);

    Env e;
    int sm = -1;
    if (live) {
        load_env<KIND, NOISE>(e, planes, P, i);
        if (GEN) load_gen(e, planes, P, i);
        if (KIND == 2 && sort_mode != nullptr) sm = sort_mode[i];
    }
    // every load has landed before the step loop: inside it there are only stores, which nothing waits for
    __builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0)
    uint32_t cur_mask = live ? action_mask_bits<KIND>(e, P) : 1u;
    const uint32_t pkey = mse_policy_key(policy_seed, (uint64_t)(P.index_offset + i));
    if (NOISE && live) lnoise[tid] = pack_u64x2(e.noise.i_lo, e.noise.i_hi);

    for (int s = 0; s < k_steps; ++s) {
        float o[D]; // per step: nothing of the observation lives across the loop's back edge
#pragma unroll
        for (int j = 0; j < D; ++j) o[j] = 0.0f;
        uint32_t mbits = 0;
        const long long srow = (long long)s * P.n + row0;
        if (live) {
            if (NOISE) { // the parked increment comes back for this step's update_accuracy (RolloutLayout::noise_offset)
                const uint4 q = lnoise[tid];
                e.noise.i_lo = (uint64_t)q.x | ((uint64_t)q.y << 32);
                e.noise.i_hi = (uint64_t)q.z | ((uint64_t)q.w << 32);
            }
            int a = policy_action<KIND, GEN>(e, cur_mask, tb, flags, pkey, policy_t0 + (uint64_t)s);
            int k[4];
            StepResult r = env_step<KIND, NOISE, LITERAL, false, GEN, BaleRefCompact>(e, P, tb, a, sm, flags, bales, k, o);
            if (__builtin_expect(r.done != 0, 0)) { // every env of a batch finishes its episode on the same step: rare, wave-uniform
                auto_reset_env<GEN, BaleRefCompact>(e, P, tb, bales, k);
                env_obs<KIND, GEN>(e, P, tb, k, o);
            }
            mbits = action_mask_bits<KIND>(e, P);
            cur_mask = mbits; // what the next step's policy sees
            if (actions_out != nullptr) __builtin_nontemporal_store(a, &actions_out[(long long)s * P.n + i]);
            if (reward_out != nullptr) __builtin_nontemporal_store((float)r.reward, &reward_out[(long long)s * P.n + i]);
            if (done_out != nullptr) __builtin_nontemporal_store((uint8_t)r.done, &done_out[(long long)s * P.n + i]);
        }
        stage_and_store_halves<KIND>(lds, o, mbits, obs_out ? obs_out + srow * D : nullptr,
                                     mask_out ? mask_out + srow * A : nullptr, n_valid, tid);
    }
    if (live) {
        store_env<KIND, NOISE>(e, planes, P, i, false);
        if (GEN) store_gen(e, planes, P, i);
    }
    if (P.track_bales) {
#pragma unroll
        for (int m = 0; m < 5; ++m) planes[(long long)(PL_BALE0 + m) * P.n_pad + i] = bales.load(m);
    }
}

// ==========================================================================================
// Pipelined rollout: dynamics waves + observer waves  (DESIGN.md "Pipelined rollout")
//
// At 65 536 envs a one-lane-per-env grid is exactly one wave per SIMD, and a single wave issues one
// VALU instruction per ~5 cycles (2-cycle peak).  Here a workgroup of 512 threads serves 256 envs:
// waves 0-3 run env_dynamics (policy, flow, draws, presses) and hand a snapshot of each env through LDS
// to waves 4-7, which run env_observe (purities, rewards, observation, mask bytes) and stream the outputs.
// One s_barrier per step; the snapshot is double buffered, so the dynamics waves run at most one step
// ahead and never overwrite a buffer the observers still read:
//     D: [step s dynamics][write snap s%2][barrier B_s][step s+1 ...]
//     O:                                  [barrier B_s][read snap s%2][outputs of step s][barrier B_{s+1}]
// Every wave executes exactly k_steps barriers.
// ==========================================================================================
constexpr int kPoEnvs = 256;                 // envs per workgroup
constexpr int kPoThreads = 2 * kPoEnvs;      // waves 0-3 dynamics, waves 4-7 observers: a workgroup's waves go round the
                                             // CU's four SIMDs, so wave w and wave w+4 share one - each SIMD gets one
                                             // multiply-heavy dynamics wave and one observer wave
constexpr int kSnapWordsBase = 13;           // ct[4] cf[4] ce lpa packed action mask
constexpr int kSnapWordsNoise = kSnapWordsBase + 8; // + accuracy_belt (4 x f64)

template <int KIND, bool NOISE>
struct PoLayout {
    static constexpr int D = Dims<KIND>::D, A = Dims<KIND>::A;
    static constexpr int snap_words = NOISE ? kSnapWordsNoise : kSnapWordsBase;
    static constexpr int obs_bytes = kPoEnvs * D * 4;
    static constexpr int mask_bytes = (kPoEnvs * A + 15) / 16 * 16;
    static constexpr int snap_offset = obs_bytes + mask_bytes;
    static constexpr int snap_bytes = 2 * snap_words * kPoEnvs * 4;
    static constexpr int bale_offset = snap_offset + snap_bytes;
    static constexpr int bale_bytes = 5 * kPoEnvs * 16;
    static constexpr int table_offset = bale_offset + bale_bytes; // multiple of 16
};

template <int KIND, bool NOISE, bool LITERAL>
__global__ __launch_bounds__(kPoThreads) void k_rollout_po(Params P, uint4 *__restrict__ planes,
                                                       const uint32_t *__restrict__ table_image, int k_steps,
                                                       uint64_t policy_seed, uint64_t policy_t0,
                                                       const int *__restrict__ sort_mode, uint32_t flags,
                                                       int *__restrict__ actions_out, float *__restrict__ obs_out,
                                                       float *__restrict__ reward_out, uint8_t *__restrict__ done_out,
                                                       uint8_t *__restrict__ mask_out)
{
    using L = PoLayout<KIND, NOISE>;
    constexpr int D = L::D, A = L::A, SW = L::snap_words;
    uint8_t *lds = reinterpret_cast<uint8_t *>(mse_dyn_lds);
    uint32_t *ltab = reinterpret_cast<uint32_t *>(lds + L::table_offset);
    uint32_t *lsnap = reinterpret_cast<uint32_t *>(lds + L::snap_offset);
    uint4 *lbale = reinterpret_cast<uint4 *>(lds + L::bale_offset);
    const int tid = threadIdx.x;
    const bool observer = tid >= kPoEnvs;            // wave-uniform: waves 0-3 dynamics, waves 4-7 observers
    const int el = tid & (kPoEnvs - 1);              // env slot inside the workgroup (same for both roles)
    const long long row0 = (long long)blockIdx.x * kPoEnvs;
    const long long i = row0 + el;                   // < n_pad always (planes are padded to 256 envs)
    const bool live = i < P.n;

    {
        TableCopy<kPoThreads, 2> tc;
        tc.issue(table_image, P.table_words, tid);
        tc.commit(ltab, table_image, P.table_words, tid);
    }
    __syncthreads();
    const Tables tb = tables_at(ltab, P);

    if (!observer) {
        // ------------------------------------------------------------------ dynamics waves
        __builtin_amdgcn_s_setprio(3); // the critical path: win the issue arbitration over the observer wave
        const BaleRef bales{lbale + el, kPoEnvs};
        if (P.track_bales) {
#pragma unroll
            for (int m = 0; m < 5; ++m) lbale[m * kPoEnvs + el] = planes[(long long)(PL_BALE0 + m) * P.n_pad + i];
        }
        Env e;
        int sm = -1;
        if (live) {
            load_env<KIND, NOISE>(e, planes, P, i);
            if (KIND == 2 && sort_mode != nullptr) sm = sort_mode[i];
        }
        __builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0): no load is outstanding inside the step loop
        uint32_t cur_mask = live ? action_mask_bits<KIND>(e, P) : 1u;
        const uint32_t pkey = mse_policy_key(policy_seed, (uint64_t)(P.index_offset + i));
        for (int s = 0; s < k_steps; ++s) {
            // padding lanes (i >= n) hold no env: an all-zero PCG64 never leaves zero and would spin forever in
            // the Lemire rejection loop, so they only keep the barrier count
            if (live) {
                const int a = policy_action<KIND>(e, cur_mask, tb, flags, pkey, policy_t0 + (uint64_t)s);
                Snap sn;
                RngLocal rng{e.rng};
                env_dynamics<KIND, NOISE, LITERAL>(e, rng, P, tb, a, sm, flags, bales, sn);
                if (__builtin_expect(sn.done != 0, 0)) { // all envs of a batch finish together: rare, wave-uniform
                    int kdummy[4];
                    auto_reset_env(e, P, tb, bales, kdummy);
                }
                const uint32_t mbits = action_mask_bits<KIND>(e, P); // what the next action sees (after auto-reset)
                cur_mask = mbits;
                uint32_t *w = lsnap + (s & 1) * SW * kPoEnvs + el;
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    w[m * kPoEnvs] = (uint32_t)sn.ct[m];
                    w[(4 + m) * kPoEnvs] = (uint32_t)sn.cf[m];
                }
                w[8 * kPoEnvs] = (uint32_t)sn.ce;
                w[9 * kPoEnvs] = (uint32_t)sn.lpa;
                w[10 * kPoEnvs] = (uint32_t)sn.timer[0] | ((uint32_t)sn.timer[1] << 8) | ((uint32_t)sn.st_belt << 16) |
                                  ((uint32_t)sn.st_sort << 18) | ((uint32_t)sn.mode << 20) | ((uint32_t)sn.lps << 22) |
                                  ((uint32_t)sn.done << 23) | ((uint32_t)sn.overflowed << 24);
                w[11 * kPoEnvs] = (uint32_t)a;
                w[12 * kPoEnvs] = mbits;
                if (NOISE) {
#pragma unroll
                    for (int m = 0; m < 4; ++m) {
                        w[(13 + 2 * m) * kPoEnvs] = (uint32_t)__double2loint(sn.acc[m]);
                        w[(14 + 2 * m) * kPoEnvs] = (uint32_t)__double2hiint(sn.acc[m]);
                    }
                }
            }
            lds_barrier_all(); // B_s
        }
        if (live) store_env<KIND, NOISE>(e, planes, P, i, false);
        if (P.track_bales) {
#pragma unroll
            for (int m = 0; m < 5; ++m) planes[(long long)(PL_BALE0 + m) * P.n_pad + i] = lbale[m * kPoEnvs + el];
        }
    } else {
        // ------------------------------------------------------------------ observer waves
        long long rem = P.n - row0;
        const int n_valid_block = rem >= kPoEnvs ? kPoEnvs : (rem > 0 ? (int)rem : 0);
        float o[D];
#pragma unroll
        for (int j = 0; j < D; ++j) o[j] = 0.0f;
        for (int s = 0; s < k_steps; ++s) {
            lds_barrier_all(); // B_s: the snapshot of step s is complete
            const uint32_t *w = lsnap + (s & 1) * SW * kPoEnvs + el;
            Snap sn;
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                sn.ct[m] = (int)w[m * kPoEnvs];
                sn.cf[m] = (int)w[(4 + m) * kPoEnvs];
            }
            sn.ce = (int)w[8 * kPoEnvs];
            sn.lpa = (int)w[9 * kPoEnvs];
            const uint32_t pk = w[10 * kPoEnvs];
            const int a = (int)w[11 * kPoEnvs];
            const uint32_t mbits = w[12 * kPoEnvs];
            if (NOISE) {
#pragma unroll
                for (int m = 0; m < 4; ++m)
                    sn.acc[m] = __hiloint2double((int)w[(14 + 2 * m) * kPoEnvs], (int)w[(13 + 2 * m) * kPoEnvs]);
            }
            sn.timer[0] = (int)(pk & 0xFFu);
            sn.timer[1] = (int)((pk >> 8) & 0xFFu);
            sn.st_belt = (int)((pk >> 16) & 3u);
            sn.st_sort = (int)((pk >> 18) & 3u);
            sn.mode = (int)((pk >> 20) & 3u);
            sn.lps = (int)((pk >> 22) & 1u);
            sn.done = (int)((pk >> 23) & 1u);
            sn.overflowed = (int)((pk >> 24) & 1u);
            if (live) { // the snapshot slots of padding lanes are never written
                int k[4];
                StepResult r = env_observe<KIND, NOISE>(sn, P, tb, k, o);
                if (__builtin_expect(sn.done != 0, 0)) { // the step's observation is the one after the auto-reset
                    Snap rs;
                    snap_of_reset(rs, tb.cst);
                    int k2[4];
                    (void)env_observe<KIND, true>(rs, P, tb, k2, o);
                }
                if (actions_out != nullptr) __builtin_nontemporal_store(a, &actions_out[(long long)s * P.n + i]);
                if (reward_out != nullptr) __builtin_nontemporal_store((float)r.reward, &reward_out[(long long)s * P.n + i]);
                if (done_out != nullptr) __builtin_nontemporal_store((uint8_t)r.done, &done_out[(long long)s * P.n + i]);
            }
            const long long srow = (long long)s * P.n + row0;
            stage_and_store<KIND>(lds, L::obs_bytes, o, mbits, obs_out ? obs_out + srow * D : nullptr,
                                  mask_out ? mask_out + srow * A : nullptr, n_valid_block, tid - kPoEnvs);
        }
    }
}

// ==========================================================================================
// Ring rollout: dynamics waves + observer waves + RNG waves  (DESIGN.md "Ring rollout")
//
// In the pipelined kernel the dynamics wave is the critical path and 60 % of it is sort_material's draws,
// half of whose instructions are the PCG64 step itself.  The stream does not depend on the env's state, so a
// third set of waves produces it ahead of time: a 768-thread workgroup serves 256 envs with
//   waves 0-3  dynamics  (policy, flow, station splits, draw DECISIONS on ring outputs, presses)
//   waves 4-7  observers (as in k_rollout_po)
//   waves 8-11 RNG       (advance the env's PCG64 sequentially, write the upper 32 output bits to an LDS ring)
// and wave w, w+4, w+8 share a SIMD.  Flow control rides on the per-step barrier B_s: before B_s the dynamics
// lanes publish how many outputs they have consumed (r_s); after it the RNG lanes fill their ring up to
// r_s + 64.  A step consumes at most kRingMaxPerStep = 31 outputs (host-checked bound on the config), so the
// outputs of step s+1 (< r_s + 32) were all written before B_s (>= r_{s-1} + 64 >= r_s + 33), and what the RNG
// lanes overwrite after B_s (ring slots of outputs < r_s) has been consumed.  The env's stream state is only
// needed again at the end of the launch (start state jumped ahead by the consumed count) and for the 1e-7 draw
// that wants the literal cdf (full output recomputed by a jump).
// ==========================================================================================
constexpr int kRingThreads = 3 * kPoEnvs;

template <int KIND, bool NOISE>
struct RingLayout {
    static constexpr int D = Dims<KIND>::D, A = Dims<KIND>::A;
    static constexpr int snap_words = kSnapWordsBase + (NOISE ? 4 : 0); // accuracy_belt travels as 4 x f32
    static constexpr int ring_offset = 0;                               // 64 KiB aligned: RngRing::load masks the row in
    static constexpr int ring_bytes = kRingDepth * kPoEnvs * 4;
    static_assert(ring_bytes == 65536, "RngRing::load assumes 64 rows of 1 KiB");
    static constexpr int stage_offset = ring_bytes;
    static constexpr int stage_bytes = kPoEnvs * D * 4;                 // obs tile; the mask tile reuses it
    static constexpr int snap_offset = stage_offset + stage_bytes;
    static constexpr int snap_bytes = 2 * snap_words * kPoEnvs * 4;
    static constexpr int bale_offset = snap_offset + snap_bytes;
    static constexpr int bale_bytes = 5 * kPoEnvs * 16;
    static constexpr int pos_offset = bale_offset + bale_bytes;
    static constexpr int pos_bytes = kPoEnvs * 4;
    static constexpr int table_offset = pos_offset + pos_bytes; // multiple of 16
};

// The RNG role of k_rollout_ring: one lane per env runs that env's
// PCG64 stream ahead into the LDS ring.  Every wave of the workgroup meets at one barrier per step (lds_barrier_all): the
// first one (B_init) publishes the ring's priming, B_s the outputs step s + 1 may consume; lpos[el] is the consumer's
// count after step s.  two_halves: the consumer side's idle lanes have produced [worst, 2 worst) before B_init.
__device__ __forceinline__ void rng_server_role(const Params &P, uint4 *__restrict__ planes, const Tables &tb, long long i,
                                                bool live, int k_steps, uint32_t *lring, uint32_t *lpos, int el, bool two_halves,
                                                int barriers_before_init = 0)
{
    Pcg g;
    {
        const uint4 a = planes[PL_RNG_STATE * P.n_pad + i], b = planes[PL_RNG_INC * P.n_pad + i];
        g.s_lo = (uint64_t)a.x | ((uint64_t)a.y << 32);
        g.s_hi = (uint64_t)a.z | ((uint64_t)a.w << 32);
        g.i_lo = (uint64_t)b.x | ((uint64_t)b.y << 32);
        g.i_hi = (uint64_t)b.z | ((uint64_t)b.w << 32);
    }
    const uint32_t ring_lane_addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t *)(lring + el);
    uint64_t c2_lo, c2_hi; // (M + 1) inc: the increment of the double step s_{n+2} = M^2 s_n + (M + 1) inc
    mul128(0x4385DF649FCCF646ull, 0x2360ED051FC65DA4ull, g.i_lo, g.i_hi, c2_lo, c2_hi);
    // Production is demand-driven.  After B_s the lane knows r_s (outputs consumed through step s); the
    // outputs of step s+1 (< r_s + worst) are in place already, those of step s+2 (< r_s + 2 worst) must be
    // before B_{s+1}: that is `need`.  `cap` is how far a lane may run ahead: the ring's 64 slots, but
    // never past what the remaining steps of this launch can consume.  The wave keeps going while ANY lane
    // is below its need and every lane below its cap produces along, so a lane that drew a lot last step
    // does not cost the wave max(draws) iterations every step: the run-ahead slack (64 - 2 worst) smooths
    // the per-step spread (6 or 19 draws with the default config) towards the long-run mean.
    const uint32_t worst = (uint32_t)P.ring_worst;
    uint32_t w = 0, need = worst, cap = need; // step 0's outputs, in place before B_init
    // The ring is primed in two halves: this lane produces outputs [0, worst) and, in parallel on the same SIMD,
    // the env's OBSERVER lane - idle until the first snapshot - produces [worst, 2 worst) from a state it reaches
    // by one table-driven jump.  At B_init the ring then holds what B_0 asks for (r_{-1} + 2 worst), so step 0
    // does not wait for a second round of production; this lane jumps over the observer's half after B_init.
    // The deeper start also smooths the wave's per-step production for the rest of the launch.  Same-box T(K): the
    // observer's half ends ~0.7 us after this lane's, paid back from K = 16 on (K = 2: +0.75 us, 16-20: equal,
    // 32: -0.8 %, 64: -1.2 %), hence only for launches of kRingTwoHalvesFrom steps or more.
#ifdef MSE_TIMELINE
    Timeline tl;
    tl.start();
#endif
    for (int s = -1; s < k_steps; ++s) {
        // M = the largest shortfall in the wave (a 6-bit maximum found bit by bit with ballots, in SGPRs);
        // every lane then produces min(M, its room): a plain per-lane trip count for the loop below
        const uint32_t deficit = need > w ? need - w : 0u; // <= 2 worst <= 62
        uint32_t M = 0;
#pragma unroll
        for (int bit = 5; bit >= 0; --bit) {
            const uint32_t cand = M | (1u << bit);
            if (__builtin_amdgcn_ballot_w64(deficit >= cand) != 0ull) M = cand;
        }
        const uint32_t room = cap > w ? cap - w : 0u;
#ifdef MSE_TIMELINE
        if (blockIdx.x == 0) { // diagnostic: the wave's trip count and what its lanes actually produce (sections 3, 4)
            tl.acc[3] += M;
            unsigned long long produced = 0;
            for (int b = 0; b < 6; ++b) produced += ((unsigned long long)__builtin_popcountll(__builtin_amdgcn_ballot_w64((((M < room ? M : room) >> b) & 1u) != 0))) << b;
            tl.acc[4] += produced;
        }
#endif
#ifdef MSE_ABL_NORNG // (ablation timing builds only: the ring holds garbage, the stream is not advanced)
        w += M < room ? M : room;
#else
        {
            const uint32_t count = M < room ? M : room;
            ring_produce_pairs(g, w, count >> 1, ring_lane_addr, c2_lo, c2_hi); // two outputs per iteration ...
            ring_produce(g, w, count & 1u, ring_lane_addr);                     // ... and the odd one
        }
#endif
        MSE_TLB(tl, 0);
        if (s < 0) // (k_rollout_policy_roles: the workgroup's prologue barrier comes before B_init; lpos is set behind it)
            for (int b = 0; b < barriers_before_init; ++b) lds_barrier_all();
        lds_barrier_all(); // s == -1: B_init (first outputs are in place); else B_s
        MSE_TLB(tl, 1);
        if (s < 0 && two_halves) {
            pcg_affine(g, P.ring_fwd[0], P.ring_fwd[1], P.ring_fwd[2], P.ring_fwd[3]); // + worst
            w += worst;
        }
        const uint32_t steps_left = (uint32_t)(k_steps - 1 - s); // steps s+1 .. K-1
        const uint32_t r = lpos[el];                             // 0 at B_init
        const uint32_t ahead = worst * steps_left;
        need = r + worst * (steps_left < 2u ? steps_left : 2u);
        cap = r + (ahead < (uint32_t)kRingDepth ? ahead : (uint32_t)kRingDepth);
    }
    // hand the stream back: this lane stands d <= worst outputs past what the env consumed (nothing is
    // produced after the last barrier, and before it at most `worst` per remaining step)
    {
        uint32_t d = w - lpos[el]; // <= 64, the ring's depth
        if (__builtin_expect(d > 32u, 0)) { // only after steps that drew nothing (an episode's first two)
            pcg_step_back(g, 32u, tb.back);
            d -= 32u;
        }
        pcg_step_back(g, d, tb.back);
        if (live) planes[PL_RNG_STATE * P.n_pad + i] = pack_u64x2(g.s_lo, g.s_hi);
    }
#ifdef MSE_TIMELINE
    tl.flush(2);
#endif
}

constexpr int kRingTwoHalvesFrom = 16; // launches this long prime the ring in two halves (RNG + observer lanes)

#ifdef MSE_CLOCK_PROBE
__device__ unsigned long long g_clock_probe[3]; // shader cycles, 100 MHz ticks, launches (workgroup 0 of k_rollout_ring)
__device__ unsigned long long g_wg_probe[3 * 1024]; // per workgroup of the LAST launch: start, end (100 MHz ticks), XCC id
#endif
template <int KIND, bool NOISE>
__global__ __launch_bounds__(kRingThreads) void k_rollout_ring(Params P, uint4 *__restrict__ planes,
                                                               const uint32_t *__restrict__ table_image, int k_steps,
                                                               uint64_t policy_seed, uint64_t policy_t0,
                                                               const int *__restrict__ sort_mode, uint32_t flags,
                                                               int *__restrict__ actions_out,
                                                               float *__restrict__ obs_out,
                                                               float *__restrict__ reward_out,
                                                               uint8_t *__restrict__ done_out,
                                                               uint8_t *__restrict__ mask_out)
{
    using L = RingLayout<KIND, NOISE>;
    constexpr int D = L::D, A = L::A, SW = L::snap_words;
    uint8_t *lds = reinterpret_cast<uint8_t *>(mse_dyn_lds);
    uint32_t *ltab = reinterpret_cast<uint32_t *>(lds + L::table_offset);
    uint32_t *lsnap = reinterpret_cast<uint32_t *>(lds + L::snap_offset);
    uint4 *lbale = reinterpret_cast<uint4 *>(lds + L::bale_offset);
    uint32_t *lring = reinterpret_cast<uint32_t *>(lds + L::ring_offset);
    uint32_t *lpos = reinterpret_cast<uint32_t *>(lds + L::pos_offset);
    const int tid = threadIdx.x;
    const int role = tid / kPoEnvs;                  // wave-uniform: 0 dynamics, 1 observer, 2 RNG
    const int el = tid - role * kPoEnvs;             // env slot inside the workgroup (same for the three roles)
    const long long row0 = (long long)blockIdx.x * kPoEnvs;
    // Every lane steps an env: the lanes past the batch's end (the planes are padded to whole workgroups) mirror its
    // last env in all three roles and store nothing, so the step loops carry no per-lane "is there an env here" region.
    const bool live = row0 + el < P.n;
    const long long i = live ? row0 + el : P.n - 1;
#ifdef MSE_TIMELINE
    Timeline edge;
    edge.start();
#endif
#ifdef MSE_CLOCK_PROBE // diagnostic build only (tools/clock_probe.py): shader cycles and 100 MHz ticks of one wave's launch
    const unsigned long long probe_c0 = __builtin_readcyclecounter(), probe_r0 = __builtin_amdgcn_s_memrealtime();
#endif

    // Launch prologue, overlapped: the RNG waves start priming the ring at once (they never read a table before
    // the end of the launch); the dynamics waves issue their state loads and, with the observers, copy the table
    // image to LDS while those fly.  B_init is the first barrier: it publishes tables and ring together.
    const Tables tb = tables_at(ltab, P);
    Env e;
    EnvRaw raw;
    int sm = -1;
    if (role == 0) {
        load_env_raw<KIND, NOISE>(raw, planes, P, i);
        if (KIND == 2 && sort_mode != nullptr) sm = sort_mode[i];
    }
    if (role == 0) {
        TableCopy<2 * kPoEnvs, 2> tc;
        tc.issue(table_image, P.table_words, tid);
        tc.commit(ltab, table_image, P.table_words, tid);
        unpack_env<KIND, NOISE>(e, raw, P);
    } else if (role == 1) {
        // The observer lanes have nothing to observe before the first snapshot: they prime the second half of the
        // ring (see the RNG waves below) - outputs [worst, 2 worst) of their env's stream, reached by one jump - while
        // their share of the table image is in flight, and write that to LDS afterwards.
        uint4 ps = {}, pi = {};
        const bool two_halves = k_steps >= kRingTwoHalvesFrom;
        if (two_halves) {
            ps = planes[PL_RNG_STATE * P.n_pad + i];
            pi = planes[PL_RNG_INC * P.n_pad + i];
        }
        TableCopy<2 * kPoEnvs, 2> tc;
        tc.issue(table_image, P.table_words, tid);
        if (two_halves) {
            Pcg g;
            g.s_lo = (uint64_t)ps.x | ((uint64_t)ps.y << 32);
            g.s_hi = (uint64_t)ps.z | ((uint64_t)ps.w << 32);
            g.i_lo = (uint64_t)pi.x | ((uint64_t)pi.y << 32);
            g.i_hi = (uint64_t)pi.z | ((uint64_t)pi.w << 32);
            pcg_affine(g, P.ring_fwd[0], P.ring_fwd[1], P.ring_fwd[2], P.ring_fwd[3]);
            uint32_t w2 = (uint32_t)P.ring_worst;
            ring_produce(g, w2, (uint32_t)P.ring_worst,
                         (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t *)(lring + el));
        }
        tc.commit(ltab, table_image, P.table_words, tid);
    }
    MSE_TL(edge, 0); // state loads and the table image -> LDS

    if (role == 2) {
        rng_server_role(P, planes, tb, i, live, k_steps, lring, lpos, el, k_steps >= kRingTwoHalvesFrom);
    } else if (role == 0) {
        // ------------------------------------------------------------------ dynamics waves
        // the critical path of the pipeline: let their instructions win the SIMD's issue arbitration over the
        // observer and RNG wave that share it
        __builtin_amdgcn_s_setprio(3);
        const BaleRef bales{lbale + el, kPoEnvs};
        if (P.track_bales) {
#pragma unroll
            for (int m = 0; m < 5; ++m) lbale[m * kPoEnvs + el] = planes[(long long)(PL_BALE0 + m) * P.n_pad + i];
        }
        __builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0): no load is outstanding inside the step loop
        RngRing rng;
        rng.lane_addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t *)(lring + el);
        // (the ring starts at LDS address 0: RingLayout::ring_offset == 0 and ring_kernels_static_lds_free(), checked at mse_create)
        rng.jump_tab = tb.jump;
        rng.start = e.rng;
        rng.p10 = 0;
        rng.nxt = rng.nxt2 = 0;
#ifdef MSE_TIMELINE
        rng.tl = &e.tl;
#endif
        rng.f_min = 0xFFFFFFFFu;
        rng.f_max = 0u;
        uint32_t cur_mask = action_mask_bits<KIND>(e, P);
        const uint32_t pkey = mse_policy_key(policy_seed, (uint64_t)(P.index_offset + i));
        lpos[el] = 0;
        MSE_TL(edge, 1); // state load
        lds_barrier_all(); // B_init
        MSE_TL(edge, 2); // waiting for the ring to be primed
#ifdef MSE_TIMELINE
        e.tl.start();
#endif
        for (int s = 0; s < k_steps; ++s) {
            {
                const int a = policy_action<KIND>(e, cur_mask, tb, flags, pkey, policy_t0 + (uint64_t)s);
                MSE_TL(e.tl, 0);
                Snap sn;
                env_dynamics<KIND, NOISE, false>(e, rng, P, tb, a, sm, flags, bales, sn);
                // the snapshot goes to LDS before the auto-reset touches the env: the observer wants the pre-reset
                // counters, and written first they need no second set of registers beside the env's own
                uint32_t *w = lsnap + (s & 1) * SW * kPoEnvs + el;
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    w[m * kPoEnvs] = (uint32_t)sn.ct[m];
                    w[(4 + m) * kPoEnvs] = (uint32_t)sn.cf[m];
                }
                w[8 * kPoEnvs] = (uint32_t)sn.ce;
                w[9 * kPoEnvs] = (uint32_t)sn.lpa;
                w[10 * kPoEnvs] = (uint32_t)sn.timer[0] | ((uint32_t)sn.timer[1] << 8) | ((uint32_t)sn.st_belt << 16) |
                                  ((uint32_t)sn.st_sort << 18) | ((uint32_t)sn.mode << 20) | ((uint32_t)sn.lps << 22) |
                                  ((uint32_t)sn.done << 23) | ((uint32_t)sn.overflowed << 24);
                w[11 * kPoEnvs] = (uint32_t)a;
                if (NOISE) {
#pragma unroll
                    for (int m = 0; m < 4; ++m) w[(13 + m) * kPoEnvs] = __float_as_uint((float)sn.acc[m]);
                }
                if (__builtin_expect(sn.done != 0, 0)) { // all envs of a batch finish together: rare, wave-uniform
                    int kdummy[4];
                    auto_reset_env(e, P, tb, bales, kdummy);
                }
                const uint32_t mbits = action_mask_bits<KIND>(e, P); // what the next action sees (after auto-reset)
                cur_mask = mbits;
                w[12 * kPoEnvs] = mbits;
                lpos[el] = rng.pos(); // r_s for the RNG lane of this env
            }
            MSE_TLB(e.tl, 5);
            lds_barrier_all(); // B_s
            MSE_TLB(e.tl, 6);
        }
#ifdef MSE_TIMELINE
        e.tl.flush(0);
#endif
        MSE_TL(edge, 3); // the step loop
        if (live) {
            // the generator state is written by the RNG lane (it steps back to the consumed position)
            store_env<KIND, NOISE>(e, planes, P, i, false, /*write_rng_state=*/false);
        }
        if (P.track_bales && live) {
#pragma unroll
            for (int m = 0; m < 5; ++m) planes[(long long)(PL_BALE0 + m) * P.n_pad + i] = lbale[m * kPoEnvs + el];
        }
        MSE_TL(edge, 5); // state stores issued
#ifdef MSE_TIMELINE
        edge.flush(3);
#endif
    } else {
        // ------------------------------------------------------------------ observer waves
        long long rem = P.n - row0;
        const int n_valid_block = rem >= kPoEnvs ? kPoEnvs : (rem > 0 ? (int)rem : 0);
        float o[D];
#pragma unroll
        for (int j = 0; j < D; ++j) o[j] = 0.0f;
        lds_barrier_all(); // B_init
#ifdef MSE_TIMELINE
        Timeline tl;
        tl.start();
#endif
        for (int s = 0; s < k_steps; ++s) {
            lds_barrier_all(); // B_s: the snapshot of step s is complete
            MSE_TLB(tl, 0);
            const uint32_t *w = lsnap + (s & 1) * SW * kPoEnvs + el;
            Snap sn;
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                sn.ct[m] = (int)w[m * kPoEnvs];
                sn.cf[m] = (int)w[(4 + m) * kPoEnvs];
            }
            sn.ce = (int)w[8 * kPoEnvs];
            sn.lpa = (int)w[9 * kPoEnvs];
            const uint32_t pk = w[10 * kPoEnvs];
            const int a = (int)w[11 * kPoEnvs];
            const uint32_t mbits = w[12 * kPoEnvs];
            if (NOISE) {
#pragma unroll
                for (int m = 0; m < 4; ++m) sn.acc[m] = (double)__uint_as_float(w[(13 + m) * kPoEnvs]);
            }
            sn.timer[0] = (int)(pk & 0xFFu);
            sn.timer[1] = (int)((pk >> 8) & 0xFFu);
            sn.st_belt = (int)((pk >> 16) & 3u);
            sn.st_sort = (int)((pk >> 18) & 3u);
            sn.mode = (int)((pk >> 20) & 3u);
            sn.lps = (int)((pk >> 22) & 1u);
            sn.done = (int)((pk >> 23) & 1u);
            sn.overflowed = (int)((pk >> 24) & 1u);
            int k[4];
#ifdef MSE_ABL_NOOBS // (ablation timing builds only: rows of zeros are staged and stored)
            StepResult r{0.0, sn.done, 0.0, 0.0};
#else
            StepResult r = env_observe<KIND, NOISE>(sn, P, tb, k, o);
#endif
            if (__builtin_expect(sn.done != 0, 0)) { // the step's observation is the one after the auto-reset
                Snap rs;
                snap_of_reset(rs, tb.cst);
                int k2[4];
                (void)env_observe<KIND, true>(rs, P, tb, k2, o);
            }
            if (live) {
                if (actions_out != nullptr) __builtin_nontemporal_store(a, &actions_out[(long long)s * P.n + i]);
                if (reward_out != nullptr) __builtin_nontemporal_store((float)r.reward, &reward_out[(long long)s * P.n + i]);
                if (done_out != nullptr) __builtin_nontemporal_store((uint8_t)r.done, &done_out[(long long)s * P.n + i]);
            }
            MSE_TL(tl, 1);
            const long long srow = (long long)s * P.n + row0;
            {
            // the mask tile reuses the obs tile: each wave finishes streaming its obs rows before it writes mask rows
            stage_and_store<KIND>(lds + L::stage_offset, -1, o, mbits, obs_out ? obs_out + srow * D : nullptr, nullptr, n_valid_block,
                                  tid - kPoEnvs);
            stage_and_store<KIND>(lds + L::stage_offset, -1, o, mbits, nullptr, mask_out ? mask_out + srow * A : nullptr, n_valid_block,
                                  tid - kPoEnvs);
            }
            MSE_TLB(tl, 2);
        }
#ifdef MSE_TIMELINE
        tl.flush(1);
#endif
    }
#ifdef MSE_CLOCK_PROBE
    if (tid == kPoEnvs && blockIdx.x == 0) { // one observer lane: its wave is the last to finish
        atomicAdd(&g_clock_probe[0], __builtin_readcyclecounter() - probe_c0);
        atomicAdd(&g_clock_probe[1], __builtin_amdgcn_s_memrealtime() - probe_r0);
        atomicAdd(&g_clock_probe[2], 1ull);
    }
    if (tid == kPoEnvs && blockIdx.x < 1024) {
        g_wg_probe[3 * blockIdx.x] = probe_r0;
        g_wg_probe[3 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
        g_wg_probe[3 * blockIdx.x + 2] = (unsigned long long)__builtin_amdgcn_s_getreg(((4 - 1) << 11) | (0 << 6) | 20); // HW_REG_XCC_ID[3:0]
    }
#endif
}


// ==========================================================================================
// Learned-policy rollout: policy forward + env transition fused, K steps per launch  (DESIGN.md "Policy rollout")
//
// The consumer loop of the reference's training (SB3 collect_rollouts inside model.learn, src/training.py:191;
// policy of src/training.py:115-131) alternates policy(obs, mask) -> action and env.step(action).  Here one wave owns
// 64 envs for the whole launch and does both: the observation it computed stays in registers, is laid out as the
// MFMA B operand with one v_permlane32_swap per register pair (lanes 0-31 / 32-63 are the two 32-env tiles), goes
// through the actor-critic MLP on the f32 matrix cores (msep::policy_tile), and the sampled action feeds the same
// lane's env_dynamics.  Per step the wave writes a MaskableRolloutBuffer row: the observation and mask the action was
// taken from, the episode-start flag, action, log-probability, value and the step's reward.  No barrier in the step
// loop: the waves of a workgroup share only the read-only weight / table images in LDS, so with several waves per SIMD
// one wave's MFMA chain runs under another's dynamics.
// LDS: [weights][tables][per wave: obs tile (the mask tile reuses it) | bale ledger]
// ==========================================================================================
template <int KIND, int TILES>
struct PolLayout {
    static constexpr int D = Dims<KIND>::D, A = Dims<KIND>::A;
    static constexpr int ENVS = 32 * TILES;                          // envs per wave
    static constexpr int weight_bytes = msep::kLdsFloats * 4;
    static constexpr int tile_bytes = (ENVS * D * 4 + 15) / 16 * 16; // >= the ENVS x A mask tile
    static constexpr int bale_bytes = 5 * ENVS * 16;
    static constexpr int wave_bytes = tile_bytes + bale_bytes;
};

// A wave's ROWS observation rows (D floats each, one per lane) / mask rows through its private LDS tile to global
// memory as 16-byte pieces; ROWS = 32 | 64.  Same scheme as stage_and_store, for a wave-sized tile.
// The lane's row of a wave tile as an LDS address the compiler cannot see through: left alone it keeps the tile's
// constant offset out of the base register and, ds_write2's offsets being 8 bits, re-adds it before every store.
typedef __attribute__((address_space(3))) float lds_f32;
typedef __attribute__((address_space(3))) uint16_t lds_u16;
__device__ __forceinline__ uint32_t lds_row_address(const void *row)
{
    uint32_t a = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void *)row;
    asm volatile("" : "+v"(a));
    return a;
}

// a finished tile of ROWS observation rows (D floats each, row-major) in LDS to global memory as 16-byte pieces
template <int D, int ROWS>
__device__ __forceinline__ void wave_tile_to_global_f32(const float *ltile, float *g, int n_valid, int lane)
{
    constexpr int NQ = ROWS * D / 4; // 16-byte pieces of a full tile
    if (n_valid == ROWS && (reinterpret_cast<uintptr_t>(g) & 15u) == 0) {
        const float4 *src = reinterpret_cast<const float4 *>(ltile);
        float4 *dst = reinterpret_cast<float4 *>(g);
        constexpr int NR = (NQ + 63) / 64;
        float4 buf[NR];
#pragma unroll
        for (int j = 0; j < NR; ++j) buf[j] = src[(lane + 64 * j) < NQ ? lane + 64 * j : 0];
#pragma unroll
        for (int j = 0; j < NR; ++j)
            if (lane + 64 * j < NQ) store_stream(dst + lane + 64 * j, buf[j]);
    } else {
        for (int q = lane; q < n_valid * D; q += 64) g[q] = ltile[q];
    }
}
template <int D, int ROWS>
__device__ __forceinline__ void wave_store_rows_f32(float *ltile, uint32_t lrow, const float *o, float *g, int n_valid, int lane)
{
    if (lane < ROWS) {
        lds_f32 *row = (lds_f32 *)(uintptr_t)lrow; // = ltile + lane * D
#pragma unroll
        for (int j = 0; j < D; ++j) row[j] = o[j];
    }
    __builtin_amdgcn_wave_barrier();
    wave_tile_to_global_f32<D, ROWS>(ltile, g, n_valid, lane);
    __builtin_amdgcn_wave_barrier();
}
template <int A, int ROWS>
__device__ __forceinline__ void wave_store_rows_mask(uint8_t *ltile, uint32_t lrow, uint32_t mbits, uint8_t *g, int n_valid, int lane)
{
    if (lane < ROWS) {
        if (A % 2 == 0) { // even row length: two mask bytes per 16-bit LDS store (the compiler merges them further)
            if constexpr (A % 2 == 0) write_mask_row<A>((lds_u16 *)(uintptr_t)lrow, mbits); // lrow = ltile + lane * A
        } else {
#pragma unroll
            for (int j = 0; j < A; ++j) ltile[lane * A + j] = (uint8_t)((mbits >> j) & 1u);
        }
    }
    __builtin_amdgcn_wave_barrier();
    constexpr int NQ = ROWS * A / 16;
    if (n_valid == ROWS && (ROWS * A) % 16 == 0 && NQ <= 64 && (reinterpret_cast<uintptr_t>(g) & 15u) == 0) {
        const uint4 v = reinterpret_cast<const uint4 *>(ltile)[lane < NQ ? lane : 0];
        if (lane < NQ) store_stream(reinterpret_cast<uint4 *>(g) + lane, v);
    } else {
        for (int q = lane; q < n_valid * A; q += 64) g[q] = ltile[q];
    }
    __builtin_amdgcn_wave_barrier();
}

// TILES = 2: a wave owns 64 envs, one per lane, as two MFMA tiles (lanes 0-31 / 32-63).
// TILES = 1: a wave owns 32 envs in lanes 0-31 (lanes 32-63 only carry the other k-half of the MFMA operands): twice
//            the waves for the same batch - the shape for batches that would otherwise leave one wave per SIMD, where
//            a single wave issues one vector instruction per ~5 cycles and a second wave's come for free.
// SORTPOL (Env_2 only): a second network (13 -> 2, its actor evaluated deterministically) plays the pre-trained sorting
// agent of env_2_press.py:101-104 inside the loop: it sees get_sort_obs() of the coming step's flow update.
template <int KIND, bool NOISE, int TILES, bool F16X3, bool SORTPOL = false>
__global__ __launch_bounds__(512) void k_rollout_policy(Params P, uint4 *__restrict__ planes,
                                                        const uint32_t *__restrict__ table_image,
                                                        const float *__restrict__ weight_blob,
                                                        const float *__restrict__ sort_weight_blob, int k_steps,
                                                        uint64_t policy_seed, uint64_t policy_t0, int deterministic,
                                                        const int *__restrict__ sort_mode, uint32_t flags,
                                                        float *__restrict__ obs_out, uint8_t *__restrict__ mask_out,
                                                        int *__restrict__ actions_out, float *__restrict__ logp_out,
                                                        float *__restrict__ value_out, float *__restrict__ reward_out,
                                                        uint8_t *__restrict__ start_out,
                                                        float *__restrict__ last_value_out,
                                                        uint8_t *__restrict__ last_done_out)
{
    using L = PolLayout<KIND, TILES>;
    constexpr int D = L::D, A = L::A, NR = msep::regs_for_actions(A), ENVS = L::ENVS;
    uint8_t *lds = reinterpret_cast<uint8_t *>(mse_dyn_lds);
    float *lw = reinterpret_cast<float *>(lds);
    constexpr int kWeightBytes = L::weight_bytes * (SORTPOL ? 2 : 1); // [policy image][sorting policy image]
    uint32_t *ltab = reinterpret_cast<uint32_t *>(lds + kWeightBytes);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, n_waves = blockDim.x >> 6;
    const int table_bytes = P.table_words * 4;
    uint8_t *lwave = lds + kWeightBytes + table_bytes + wave * L::wave_bytes;
    uint4 *lbale = reinterpret_cast<uint4 *>(lwave + L::tile_bytes);
    const long long wave_row0 = ((long long)blockIdx.x * n_waves + wave) * ENVS;
    long long rem = P.n - wave_row0;
    const int n_valid = rem >= ENVS ? ENVS : (rem > 0 ? (int)rem : 0);
    const bool wave_active = n_valid > 0;
    // Every lane of an active wave steps an env, so that the step loop has no per-lane "is there an env here" region
    // (the compiler kept two copies of the env's ~90 registers around one and moved them back and forth every step):
    // TILES = 1: lanes 32-63 mirror lanes 0-31; lanes past the batch's end mirror its last env.  Mirrors compute what
    // their original computes and store nothing.
    const int src_lane = TILES == 1 ? (lane & 31) : lane;
    const bool live = wave_active && lane < ENVS && wave_row0 + lane < P.n;
    const long long i = wave_active ? (wave_row0 + src_lane < P.n ? wave_row0 + src_lane : P.n - 1) : 0;

    msep_copy_image(lw, weight_blob, F16X3, tid, blockDim.x);
    if (SORTPOL) msep_copy_image(lw + msep::kLdsFloats, sort_weight_blob, F16X3, tid, blockDim.x);
    for (int w = tid; w < P.table_words / 4; w += blockDim.x)
        reinterpret_cast<uint4 *>(ltab)[w] = reinterpret_cast<const uint4 *>(table_image)[w];
    const BaleRef bales{lbale + src_lane, ENVS};
    if (P.track_bales && wave_active) {
#pragma unroll
        for (int m = 0; m < 5; ++m) lbale[m * ENVS + src_lane] = planes[(long long)(PL_BALE0 + m) * P.n_pad + i];
    }
    Env e;
    int sm = -1;
    load_env<KIND, NOISE>(e, planes, P, i);
    if (KIND == 2 && sort_mode != nullptr) sm = sort_mode[i];
    __syncthreads();
    if (!wave_active) return;
    __builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0): nothing but stores inside the step loop
    const Tables tb = tables_at(ltab, P);
    msep::lds_f4 wl = (msep::lds_f4)(__attribute__((address_space(3))) float *)lw;

    // the observation and mask of the current state (what reset / the previous launch left)
    float o[32];
#pragma unroll
    for (int j = 0; j < 32; ++j) o[j] = 0.0f;
    int kcur[4]; // container purities of the current state (the sorting agent's view needs them)
    container_purity_k(e, kcur);
    env_obs<KIND>(e, P, tb, kcur, o);
    uint32_t mbits = action_mask_bits<KIND>(e, P);
    msep::lds_f4 wl_sort = (msep::lds_f4)(__attribute__((address_space(3))) float *)(lw + msep::kLdsFloats);
    // policy stream keys of the env(s) this lane serves as an MFMA column: tile t = envs 32 t .. 32 t + 31 of the wave
    const int h = lane >> 5, col = lane & 31;
    const uint32_t key0 = mse_policy_key(policy_seed, (uint64_t)(P.index_offset + wave_row0 + col));
    const uint32_t key1 = mse_policy_key(policy_seed, (uint64_t)(P.index_offset + wave_row0 + 32 + col));

    // action_masks() bits of env -> bit r: the action of accumulator register r of half h is legal
    auto legal_of = [&](uint32_t env_bits) -> uint32_t {
        const uint32_t t = env_bits >> (4 * h);
        return (t & 0xFu) | ((t >> 4) & 0xF0u) | ((t >> 8) & 0xF00u) | ((t >> 12) & 0xF000u);
    };
    // swap(o[2q], o[2q+1]) = {tile 0's k-step q operand, tile 1's}: lanes 32-63 of tile 0 get the odd entries of the
    // envs in lanes 0-31, lanes 0-31 of tile 1 the even entries of the envs in lanes 32-63
    auto operands_of = [&](const float *ob, float (*x)[16]) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(ob[2 * q]), __float_as_uint(ob[2 * q + 1]), false, false);
            x[0][q] = __uint_as_float(r[0]);
            x[1][q] = __uint_as_float(r[1]);
        }
    };

    const uint32_t lrow_obs = lds_row_address(reinterpret_cast<float *>(lwave) + src_lane * D);
    const uint32_t lrow_mask = lds_row_address(lwave + src_lane * A);
    int last_done = 0;
#ifdef MSE_TIMELINE
    Timeline ptl;
    ptl.start();
#endif
    for (int s = 0; s < k_steps; ++s) {
        __builtin_amdgcn_s_setprio(TILES == 2 ? 3 : 0); // policy phase (see below)
        const long long srow = (long long)s * P.n + wave_row0;
        // the row the action is taken from (MaskableRolloutBuffer: observations, action_masks, episode_starts)
        if (obs_out != nullptr) wave_store_rows_f32<D, ENVS>(reinterpret_cast<float *>(lwave), lrow_obs, o, obs_out + srow * D, n_valid, lane);
        if (mask_out != nullptr) wave_store_rows_mask<A, ENVS>(lwave, lrow_mask, mbits, mask_out + srow * A, n_valid, lane);
        if (live && start_out != nullptr)
            __builtin_nontemporal_store((uint8_t)(e.step == 0 ? 1 : 0), &start_out[(long long)s * P.n + i]);
        MSE_TLB(ptl, 0); // row stores
        // ---- policy
        float x[2][16];
        operands_of(o, x);
        // without masking (plain PPO on the unmasked env) the policy samples from the whole action space and the
        // step sanitises; the recorded mask row is action_masks() either way
        uint32_t mb0, mb1;
        msep::both_halves_u32((flags & MSE_STEP_UNMASKED) ? ((1u << A) - 1u) : mbits, mb0, mb1);
        const uint64_t t = policy_t0 + (uint64_t)s;
        const uint32_t legal[2] = {legal_of(mb0), legal_of(mb1)};
        const uint32_t words[2] = {mse_policy_word(key0, t), mse_policy_word(key1, t)};
        msep::TileOut p[2];
        msep::policy_tiles<NR, F16X3, TILES>(wl, lane, x, legal, deterministic != 0, words, nullptr, p);
        int a = p[0].action;
        float logp = p[0].logp, value = p[0].value;
        if (TILES == 2) { // lane l is env l: tile l >> 5, column l & 31 (results are valid in both halves)
            a = h ? p[1].action : a;
            logp = h ? p[1].logp : logp;
            value = h ? p[1].value : value;
        }
        MSE_TLB(ptl, 1); // policy forward
        // Phase priorities: the two waves of a SIMD fall into step with each other (both in the policy, both in the
        // dynamics) and then compete for the same pipe; letting one phase win the issue arbitration pulls them apart so
        // that one wave's MFMA chains run under the other's dynamics.  Which phase should win was measured on the box
        // (same-box A/B, 16 steps per launch): 64-env waves +8 % with the policy phase high, 32-env waves +8 % with the
        // dynamics phase high (the other choice +4 ... 6 % each; a static per-wave priority: nothing).
        __builtin_amdgcn_s_setprio(TILES == 2 ? 0 : 3);
        if (SORTPOL) {
            // the sorting agent's decision for the coming step: get_sort_obs() one flow update ahead (a copy steps
            // the flow; env_step will take the same step), through the second network's actor, argmax
            float so[32];
#pragma unroll
            for (int j = 0; j < 32; ++j) so[j] = 0.0f;
            {
                Env ec = e;
                update_environment<false>(ec, P);
                sort_obs<false>(ec, P, tb, kcur, so);
            }
            float sx[2][16];
            operands_of(so, sx);
            int sa[2];
            msep::actor_argmax2_tiles<F16X3, TILES>(wl_sort, lane, sx, sa);
            sm = (TILES == 2 && h) ? sa[1] : sa[0];
        }
        MSE_TLB(ptl, 2); // sorting agent
        // ---- the env transition under that action
        StepResult r = env_step<KIND, NOISE, false>(e, P, tb, a, sm, flags, bales, kcur, o);
        MSE_TLB(ptl, 3); // env transition
        if (__builtin_expect(r.done != 0, 0)) {
            auto_reset_env(e, P, tb, bales, kcur);
            env_obs<KIND>(e, P, tb, kcur, o);
        }
        mbits = action_mask_bits<KIND>(e, P);
        last_done = r.done;
        if (live) {
            const long long at = (long long)s * P.n + i;
            if (actions_out != nullptr) __builtin_nontemporal_store(a, &actions_out[at]);
            if (logp_out != nullptr) __builtin_nontemporal_store(logp, &logp_out[at]);
            if (value_out != nullptr) __builtin_nontemporal_store(value, &value_out[at]);
            if (reward_out != nullptr) __builtin_nontemporal_store((float)r.reward, &reward_out[at]);
        }
        MSE_TLB(ptl, 4); // auto-reset, mask, per-env stores
    }
#ifdef MSE_TIMELINE
    ptl.flush(0);
#endif
    // the bootstrap value of the state the rollout ends in: the critic alone
    if (last_value_out != nullptr) {
        float x[2][16], v[2];
        operands_of(o, x);
        msep::value_tiles<F16X3, TILES>(wl, lane, x, v);
        if (live) last_value_out[i] = (TILES == 2 && h) ? v[1] : v[0];
    }
    if (live && last_done_out != nullptr) last_done_out[i] = (uint8_t)last_done;
    if (live) store_env<KIND, NOISE>(e, planes, P, i, false);
    if (P.track_bales && live) {
#pragma unroll
        for (int m = 0; m < 5; ++m) planes[(long long)(PL_BALE0 + m) * P.n_pad + i] = lbale[m * ENVS + lane];
    }
}

// ---- the learned-policy rollout in roles (batches that leave a SIMD 64 envs: n <= 256 envs x CUs) --------------------
// k_rollout_policy at that size has to choose between one 64-env wave per SIMD, alone with its own latencies, and two
// 32-env waves that issue the env transition twice with half their lanes mirroring.  Here a SIMD's 64 envs are one full
// ACTOR wave - actor network, sampling, env transition: the serial chain of a step, nothing else - and beside it
//   a CRITIC wave that takes what the chain does not wait for: the value network, the observation / mask / episode-start
//     rows of the rollout buffer and, after the last step, the bootstrap value;
//   RING = true: an RNG wave that runs the envs' sort_material streams ahead into the LDS ring of k_rollout_ring (same
//     protocol, same device functions: rng_server_role on one side, env_dynamics on RngRing on the other), which takes
//     the 128-bit LCG steps - a long dependent chain - out of the actor wave.  Needs what the ring kernel needs (at
//     most kRingMaxPerStep draws per step, the ring at LDS address 0); else RING = false.
// Workgroup = four actor waves, four critic waves[, four RNG waves]; waves p, p + 4[, p + 8] serve the same 64 envs and
// meet on SIMD p.  The actor wave posts each step's observation rows and mask word into the pair's LDS tile before the
// step's barrier; the critic wave reads the MFMA operands and the rows' 16-byte pieces from it after the barrier, all at
// once, and then raises the tile's `taken` count, which the actor wave checks before it posts the next row (it never
// has to wait: the next post is a whole step away).  One barrier per step, shared with the ring's flow control.
// Same device functions as k_rollout_policy (actor_tiles and value_tiles are policy_tiles' two networks; env_step is
// env_dynamics + env_observe), so every buffer is bit-identical to it and to the two-launch collector.
template <int KIND, bool RING>
struct PolRolesLayout {
    static constexpr int D = Dims<KIND>::D, A = Dims<KIND>::A;
    static constexpr int kPairs = 4, kThreads = 64 * kPairs * (RING ? 3 : 2);
    static_assert(64 * kPairs == kPoEnvs, "the ring has one column per env of the workgroup");
    static constexpr int ring_bytes = RING ? kRingDepth * kPoEnvs * 4 : 0; // at LDS address 0 (RngRing::load)
    static constexpr int weight_offset = ring_bytes;
    static constexpr int weight_bytes = msep::kLdsFloats * 4;
    static constexpr int pos_offset = weight_offset + weight_bytes;
    static constexpr int pos_bytes = RING ? kPoEnvs * 4 : 0;
    static constexpr int taken_offset = pos_offset + pos_bytes;     // rows the critic wave has taken out of the tile: u32[kPairs]
    static constexpr int pair_offset = taken_offset + 16;
    static constexpr int tile_bytes = (64 * D * 4 + 15) / 16 * 16;
    static constexpr int mword_bytes = 64 * 4;                     // action-mask bits | episode-start flag << 31
    static constexpr int mask_bytes = (64 * A + 15) / 16 * 16;     // the critic wave's staging tile of mask rows
    static constexpr int bale_bytes = 5 * 64 * 16;
    static constexpr int rew_bytes = 3 * 64 * 4;                   // what env_reward needs of a step: u32[3][64]
    static constexpr int note_bytes = 3 * 64 * 4;                  // the step's bale bookings (BaleNote): u32[3][64]
    static constexpr int pair_bytes = tile_bytes + mword_bytes + mask_bytes + bale_bytes + rew_bytes + note_bytes;
    static constexpr int table_offset = pair_offset + kPairs * pair_bytes; // multiple of 16
};

#ifndef MSE_ROLES_PRIO
#define MSE_ROLES_PRIO 3
#endif
template <int KIND, bool NOISE, bool RING>
__global__ __launch_bounds__(RING ? 768 : 512) void k_rollout_policy_roles(Params P, uint4 *__restrict__ planes,
                                                                           const uint32_t *__restrict__ table_image,
                                                                           const float *__restrict__ weight_blob, int k_steps,
                                                                           uint64_t policy_seed, uint64_t policy_t0, int deterministic,
                                                                           const int *__restrict__ sort_mode, uint32_t flags,
                                                                           float *__restrict__ obs_out, uint8_t *__restrict__ mask_out,
                                                                           int *__restrict__ actions_out, float *__restrict__ logp_out,
                                                                           float *__restrict__ value_out, float *__restrict__ reward_out,
                                                                           uint8_t *__restrict__ start_out,
                                                                           float *__restrict__ last_value_out,
                                                                           uint8_t *__restrict__ last_done_out)
{
    using L = PolRolesLayout<KIND, RING>;
    constexpr int D = L::D, A = L::A, NR = msep::regs_for_actions(A);
    static_assert(A < 31, "the mask word keeps bit 31 for the episode-start flag");
    // Env_3's reward (a table look-up by the purities plus calculate_press_reward's fp64 division) leaves the actor wave's
    // chain for the critic wave: +4.4 % at 65 536 envs.  Env_2 (the division alone): +-0.5 %; Env_1 (one look-up): -1.3 %;
    // they keep it where it was.
    constexpr bool kRewardOnCritic = KIND == 3;
    // How the actor wave issues its two tiles: as a software pipeline (actor_tiles_pipelined) for Env_3, +0.4 % (+2.6 % with
    // noise); layer by layer for Env_1 / Env_2, where the pipeline cost 3 % / 7 % (same-box A/Bs, profiles/r03).
#ifdef MSE_ACTOR_PLAIN // (A/B builds)
    constexpr bool kPipelinedActor = false;
#else
    constexpr bool kPipelinedActor = KIND == 3;
#endif
    uint8_t *lds = reinterpret_cast<uint8_t *>(mse_dyn_lds);
    float *lw = reinterpret_cast<float *>(lds + L::weight_offset);
    uint32_t *ltab = reinterpret_cast<uint32_t *>(lds + L::table_offset);
    uint32_t *lring = reinterpret_cast<uint32_t *>(lds);
    uint32_t *lpos = reinterpret_cast<uint32_t *>(lds + L::pos_offset);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int pair = wave & (L::kPairs - 1), role = wave / L::kPairs; // wave-uniform: 0 actor, 1 critic, 2 RNG
    const int el = pair * 64 + lane;                                  // env slot inside the workgroup
    volatile uint32_t *ltaken = reinterpret_cast<volatile uint32_t *>(lds + L::taken_offset) + pair;
    uint8_t *lpair = lds + L::pair_offset + pair * L::pair_bytes;
    float *ltile = reinterpret_cast<float *>(lpair);                              // [64][D]
    uint32_t *lmword = reinterpret_cast<uint32_t *>(lpair + L::tile_bytes);       // [64]
    uint8_t *lmask = lpair + L::tile_bytes + L::mword_bytes;
    uint4 *lbale = reinterpret_cast<uint4 *>(lmask + L::mask_bytes);
    uint32_t *lrew = reinterpret_cast<uint32_t *>(lmask + L::mask_bytes + L::bale_bytes);
    uint32_t *lnote = lrew + L::rew_bytes / 4;
    const long long wave_row0 = ((long long)blockIdx.x * L::kPairs + pair) * 64;
    const long long rem = P.n - wave_row0;
    const int n_valid = rem >= 64 ? 64 : (rem > 0 ? (int)rem : 0);
    const bool pair_active = n_valid > 0;
    // every lane of an active pair steps an env: lanes past the batch's end mirror its last env and store nothing
    const bool live = pair_active && wave_row0 + lane < P.n;
    const long long i = pair_active ? (wave_row0 + lane < P.n ? wave_row0 + lane : P.n - 1) : 0;

    const Tables tb = tables_at(ltab, P);
    if (RING && role == 2) {
        // The RNG waves start priming the ring at once (they read no table before the end of the launch) while the other
        // waves bring the weight and table images and the env state in; rng_server_role meets them at the prologue's
        // barrier and then at the k_steps + 1 step barriers: B_init = the one before step 0, B_s = the one before step s + 1.
        if (pair_active) rng_server_role(P, planes, tb, i, live, k_steps, lring, lpos, el, false, /*barriers_before_init=*/1);
        else for (int s = -1; s <= k_steps; ++s) lds_barrier_all();
        return;
    }
    msep_copy_image(lw, weight_blob, true, tid, 128 * L::kPairs); // actor and critic waves: tid < 128 kPairs
    for (int w = tid; w < P.table_words / 4; w += 128 * L::kPairs)
        reinterpret_cast<uint4 *>(ltab)[w] = reinterpret_cast<const uint4 *>(table_image)[w];
    const BaleRef bales{lbale + lane, 64};
    Env e;
    int sm = -1;
    if (role == 1 && P.track_bales && pair_active) { // the bale ledger lives with the critic wave (BaleNote)
#pragma unroll
        for (int m = 0; m < 5; ++m) lbale[m * 64 + lane] = planes[(long long)(PL_BALE0 + m) * P.n_pad + i];
    }
    if (role == 0) {
        load_env<KIND, NOISE>(e, planes, P, i);
        if (KIND == 2 && sort_mode != nullptr) sm = sort_mode[i];
        if (lane == 0) *ltaken = 0u;
        if (RING) lpos[el] = 0u;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // nothing but stores inside the step loop
    lds_barrier_all();                                 // the images, lpos and `taken` are in place
    if (!pair_active) { // the step barriers count every wave of the workgroup
        for (int s = 0; s <= k_steps; ++s) lds_barrier_all();
        return;
    }
    msep::lds_f4 wl = (msep::lds_f4)(__attribute__((address_space(3))) float *)lw;
    const int h = lane >> 5, col = lane & 31;

    if (role == 0) {
        __builtin_amdgcn_s_setprio(MSE_ROLES_PRIO); // the chain of the step: ahead of the other waves whenever several can issue
        float o[32];
#pragma unroll
        for (int j = 0; j < 32; ++j) o[j] = 0.0f;
        int kcur[4];
        container_purity_k(e, kcur);
        env_obs<KIND>(e, P, tb, kcur, o);
        uint32_t mbits = action_mask_bits<KIND>(e, P);
        const uint32_t key0 = mse_policy_key(policy_seed, (uint64_t)(P.index_offset + wave_row0 + col));
        const uint32_t key1 = mse_policy_key(policy_seed, (uint64_t)(P.index_offset + wave_row0 + 32 + col));
        auto legal_of = [&](uint32_t env_bits) -> uint32_t { // as in k_rollout_policy
            const uint32_t t = env_bits >> (4 * h);
            return (t & 0xFu) | ((t >> 4) & 0xF0u) | ((t >> 8) & 0xF00u) | ((t >> 12) & 0xF000u);
        };
        RngRing ring;
        if (RING) { // as the dynamics role of k_rollout_ring
            ring.lane_addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t *)(lring + el);
            ring.jump_tab = tb.jump;
            ring.start = e.rng;
            ring.p10 = 0;
            ring.nxt = ring.nxt2 = 0;
            ring.f_min = 0xFFFFFFFFu;
            ring.f_max = 0u;
#ifdef MSE_TIMELINE
            ring.tl = &e.tl;
#endif
        }
        const uint32_t lrow = lds_row_address(ltile + lane * D);
        // `taken` is read a phase early (its LDS round trip rides under the observation's arithmetic) and again only if
        // the critic wave had not got to the tile by then, which a whole actor network + env transition makes unlikely
        uint32_t taken = 0;
        auto wait_taken = [&](int s) { // the critic wave has row s - 1 (and the reward words of step s - 2) in registers
            while (__builtin_amdgcn_readfirstlane(taken) < (uint32_t)s) {
                __builtin_amdgcn_s_sleep(1);
                taken = *ltaken;
            }
        };
        // The bale ledger is output-only state: the step notes its bookings (BaleNote) and the critic wave replays them
        uint32_t note_head = 0u, note_n[2] = {0u, 0u};
        const BaleNote note{&note_head, note_n};
        // (kRewardOnCritic) The step's reward is not on the chain: the critic wave evaluates it (env_reward) from three words - the amount
        // of a press started this step, the levels' sum, and {purity-hundredths sum, penalty classes, flags} - and
        // stores the rollout buffer's reward row.
        auto post_reward_words = [&](const Snap &sn, const int *k) {
            int lvl[5], s_sum = 0;
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                lvl[m] = sn.ct[m] + sn.cf[m];
                if (KIND != 2) s_sum += (k[m] == 101) ? P.k_thr[m] : k[m]; // (Env_2 has no sorting reward)
            }
            lvl[4] = sn.ce;
            uint32_t bits = (uint32_t)s_sum | (sn.overflowed ? 1u << 20 : 0u);
            if (KIND != 1) { // (Env_1 has no press reward)
                const PenaltyClass pc = classify_levels(lvl, P);
                bits |= (pc.any_cat ? 1u << 16 : 0u) | (pc.any_sev ? 1u << 17 : 0u) | (pc.any_mild ? 1u << 18 : 0u) |
                        (sn.lps ? 1u << 19 : 0u);
                lrew[lane] = (uint32_t)(sn.lps ? sn.lpa : 0);
                lrew[64 + lane] = (uint32_t)(lvl[0] + lvl[1] + lvl[2] + lvl[3] + lvl[4]);
            }
            lrew[128 + lane] = bits;
        };
        auto post_row = [&](int s) { // the state the coming action is taken from: row s of the rollout buffer
            wait_taken(s);
            lds_f32 *row = (lds_f32 *)(uintptr_t)lrow;
#pragma unroll
            for (int j = 0; j < D; ++j) row[j] = o[j];
            lmword[lane] = mbits | (e.step == 0 ? 0x80000000u : 0u);
        };
        int last_done = 0;
#ifdef MSE_TIMELINE
        e.tl.start();
#endif
        for (int s = 0; s < k_steps; ++s) {
            post_row(s);
            MSE_TLB(e.tl, 5); // observation, auto-reset, mask, per-env stores, the row's post
            lds_barrier_all();
            MSE_TLB(e.tl, 6); // barrier wait
            float x[2][16];
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(o[2 * q]), __float_as_uint(o[2 * q + 1]), false, false);
                x[0][q] = __uint_as_float(r[0]);
                x[1][q] = __uint_as_float(r[1]);
            }
            uint32_t mb0, mb1;
            msep::both_halves_u32((flags & MSE_STEP_UNMASKED) ? ((1u << A) - 1u) : mbits, mb0, mb1);
            const uint64_t t = policy_t0 + (uint64_t)s;
            const uint32_t legal[2] = {legal_of(mb0), legal_of(mb1)};
            const uint32_t words[2] = {mse_policy_word(key0, t), mse_policy_word(key1, t)};
            msep::TileOut p[2];
            if (kPipelinedActor) msep::actor_tiles_pipelined<NR>(wl, lane, x, legal, deterministic != 0, words, p);
            else msep::actor_tiles<NR, true, 2>(wl, lane, x, legal, deterministic != 0, words, p);
            const int a = h ? p[1].action : p[0].action; // lane l is env l: tile l >> 5, column l & 31
            const float logp = h ? p[1].logp : p[0].logp;
            MSE_TLB(e.tl, 0); // actor network and sampling
            Snap sn;
            note.begin_step();
            if (RING) {
                env_dynamics<KIND, NOISE, false, RngRing, false, false, BaleNote>(e, ring, P, tb, a, sm, flags, note, sn);
            } else { // env_step's two halves, with the lane's own generator
                RngLocal own{e.rng};
                env_dynamics<KIND, NOISE, false, RngLocal, false, false, BaleNote>(e, own, P, tb, a, sm, flags, note, sn);
            }
            taken = *ltaken;
            const StepResult r = env_observe<KIND, NOISE>(sn, P, tb, kcur, o); // (kRewardOnCritic: its reward arithmetic is dead code)
            if (kRewardOnCritic || P.track_bales) wait_taken(s + 1); // the words of step s - 1 are in the critic wave's registers
            if (kRewardOnCritic) post_reward_words(sn, kcur);         // (before the auto-reset overwrites the purities)
            if (__builtin_expect(r.done != 0, 0)) {
                auto_reset_env<false, BaleNote>(e, P, tb, note, kcur);
                env_obs<KIND>(e, P, tb, kcur, o);
            }
            if (P.track_bales) {
                lnote[lane] = note_head;
                lnote[64 + lane] = note_n[0];
                lnote[128 + lane] = note_n[1];
            }
            mbits = action_mask_bits<KIND>(e, P);
            last_done = r.done;
            if (RING) lpos[el] = ring.pos(); // what the env has consumed: the RNG lane reads it after the next barrier
            if (live) {
                const long long at = (long long)s * P.n + i;
                if (actions_out != nullptr) __builtin_nontemporal_store(a, &actions_out[at]);
                if (logp_out != nullptr) __builtin_nontemporal_store(logp, &logp_out[at]);
                if (!kRewardOnCritic && reward_out != nullptr) __builtin_nontemporal_store((float)r.reward, &reward_out[at]);
            }
        }
        post_row(k_steps); // the state the rollout ends in, for the bootstrap value
        lds_barrier_all();
#ifdef MSE_TIMELINE
        e.tl.flush(0);
#endif
        if (live && last_done_out != nullptr) last_done_out[i] = (uint8_t)last_done;
        // (RING: the generator state is written by the RNG lane, which steps back to the consumed position)
        if (live) store_env<KIND, NOISE>(e, planes, P, i, false, /*write_rng_state=*/!RING);
        return;
    }

    // ---- critic wave
    const uint32_t lrow_mask = lds_row_address(lmask + lane * A);
    constexpr int NQ = 64 * D / 4, NP = (NQ + 63) / 64; // 16-byte pieces of the tile, per lane
    // everything of a posted row set in one go: the layer-1 operands of this lane (register q of tile t = entry 2 q + h
    // of env 32 t + col), the rows as 16-byte pieces, the mask word; then the tile is the actor wave's again
    auto take_tile = [&](int s, float (*x)[16], float4 *piece, uint32_t &mw) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const float *row = ltile + (32 * t + col) * D;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                if (2 * q + 1 < D) x[t][q] = row[2 * q + h];
                else if (2 * q < D) x[t][q] = h ? 0.0f : row[2 * q];
                else x[t][q] = 0.0f;
            }
        }
        if (piece != nullptr) {
#pragma unroll
            for (int j = 0; j < NP; ++j) piece[j] = reinterpret_cast<const float4 *>(ltile)[(lane + 64 * j) < NQ ? lane + 64 * j : 0];
        }
        mw = lmword[lane];
        uint32_t rw[3] = {0u, 0u, 0u}, nw[3] = {0u, 0u, 0u};
        if (kRewardOnCritic && s > 0) { // the reward words of step s - 1, posted before this barrier
#pragma unroll
            for (int w = 0; w < 3; ++w) rw[w] = lrew[w * 64 + lane];
        }
        if (P.track_bales && s > 0) { // ... and its bale bookings
#pragma unroll
            for (int w = 0; w < 3; ++w) nw[w] = lnote[w * 64 + lane];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // the reads have returned
        if (lane == 0) *ltaken = (uint32_t)(s + 1);
        if (P.track_bales && s > 0) BaleNote::replay(nw[0], nw[1], nw[2], bales, P, tb.press_time);
        if (kRewardOnCritic && s > 0 && live && reward_out != nullptr) {
            const PenaltyClass pc{(rw[2] & (1u << 16)) != 0, (rw[2] & (1u << 17)) != 0, (rw[2] & (1u << 18)) != 0};
            const double rew = env_reward<KIND>((int)(rw[2] & 0xFFFFu), (rw[2] & (1u << 19)) != 0, (int)rw[0], (int)rw[1], pc,
                                                (rw[2] & (1u << 20)) != 0, P, tb);
            __builtin_nontemporal_store((float)rew, &reward_out[(long long)(s - 1) * P.n + i]);
        }
    };
#ifdef MSE_TIMELINE
    Timeline ctl;
    ctl.start();
#endif
    for (int s = 0; s < k_steps; ++s) {
        lds_barrier_all();
        MSE_TLB(ctl, 0); // barrier wait
        float x[2][16];
        float4 piece[NP];
        uint32_t mw;
        take_tile(s, x, piece, mw);
        MSE_TLB(ctl, 1); // tile -> registers
        const long long srow = (long long)s * P.n + wave_row0;
        if (obs_out != nullptr) {
            float *g = obs_out + srow * D;
            if (n_valid == 64 && (reinterpret_cast<uintptr_t>(g) & 15u) == 0) { // whole pieces (a ragged batch's later rows start anywhere)
#pragma unroll
                for (int j = 0; j < NP; ++j)
                    if (lane + 64 * j < NQ) store_stream(reinterpret_cast<float4 *>(g) + lane + 64 * j, piece[j]);
            } else { // entry by entry out of the pieces
#pragma unroll
                for (int j = 0; j < NP; ++j) {
                    const int q0 = 4 * (lane + 64 * j);
                    const float v4[4] = {piece[j].x, piece[j].y, piece[j].z, piece[j].w};
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        if (q0 + c < n_valid * D) g[q0 + c] = v4[c];
                }
            }
        }
        if (mask_out != nullptr) wave_store_rows_mask<A, 64>(lmask, lrow_mask, mw & 0x7FFFFFFFu, mask_out + srow * A, n_valid, lane);
        if (live && start_out != nullptr) __builtin_nontemporal_store((uint8_t)(mw >> 31), &start_out[(long long)s * P.n + i]);
        MSE_TLB(ctl, 2); // row stores
        float v[2];
        msep::value_tiles<true, 2>(wl, lane, x, v);
        if (live && value_out != nullptr) __builtin_nontemporal_store(h ? v[1] : v[0], &value_out[(long long)s * P.n + i]);
        MSE_TLB(ctl, 3); // value network
    }
#ifdef MSE_TIMELINE
    ctl.flush(1);
#endif
    lds_barrier_all();
    {
        float x[2][16], v[2];
        uint32_t mw;
        take_tile(k_steps, x, nullptr, mw); // (and the last step's reward row and bale bookings)
        if (P.track_bales && live) {
#pragma unroll
            for (int m = 0; m < 5; ++m) planes[(long long)(PL_BALE0 + m) * P.n_pad + i] = lbale[m * 64 + lane];
        }
        if (last_value_out != nullptr) {
            msep::value_tiles<true, 2>(wl, lane, x, v);
            if (live) last_value_out[i] = h ? v[1] : v[0];
        }
    }
}

template <int KIND>
__global__ __launch_bounds__(kBlock) void k_reset(Params P, uint4 *__restrict__ planes,
                                                  const uint32_t *__restrict__ table_image,
                                                  const uint64_t *__restrict__ seeds,
                                                  const uint8_t *__restrict__ which, float *__restrict__ obs_out,
                                                  uint8_t *__restrict__ mask_out)
{
    constexpr int D = Dims<KIND>::D, A = Dims<KIND>::A;
    const long long i = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (i >= P.n) return;
    const Tables tb = tables_at(table_image, P); // cold kernel: tables straight from global memory
    Env e;
    load_env<1, true>(e, planes, P, i); // every plane, whatever the env kind
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
            e.gen = gen; // the generator's stream goes on from there (observable in general generator mode only):
            e.gen_uint = (uint32_t)(r >> 32); // the permutation took the low half, the high half is buffered
            e.gen_has = 1;
            e.press = pcg_seed(seed + 3);
            e.press_has = 0;
            e.press_uint = 0;
            e.noise = pcg_seed(seed + 4);
            e.rng = pcg_seed(seed + 99);
            e.episode = 1u; // episodes are counted from the last seeded reset
            reseeded = true;
            const Pcg srt = pcg_seed(seed + 2); // rng_sorting (env_super.py:171)
            planes[(long long)PL_SORTRNG_STATE * P.n_pad + i] = pack_u64x2(srt.s_lo, srt.s_hi);
            planes[(long long)PL_SORTRNG_INC * P.n_pad + i] = pack_u64x2(srt.i_lo, srt.i_hi);
            planes[(long long)PL_SORTRNG_AUX * P.n_pad + i] = make_uint4(0, 0, 0, 0);
        } else {
            e.gen2 = unseeded_gen2(e);
            unseeded_generator(e);
            e.episode += 1u;
        }
        reset_episode_state(e, tb.cst);
        clear_bales(BaleRef{planes + (long long)PL_BALE0 * P.n_pad + i, P.n_pad});
        store_env<1, true>(e, planes, P, i, reseeded);
        store_gen(e, planes, P, i);
    }
    if (obs_out != nullptr) {
        int k[4];
        container_purity_k(e, k);
        float o[D];
        if (P.gen_mode) env_obs<KIND, true>(e, P, tb, k, o);
        else env_obs<KIND, false>(e, P, tb, k, o);
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
    load_env<KIND, false>(e, planes, P, i);
    uint32_t bits = action_mask_bits<KIND>(e, P);
#pragma unroll
    for (int j = 0; j < A; ++j) mask_out[i * A + j] = (uint8_t)((bits >> j) & 1u);
}

// What Env_2_Pressing.step hands its sorting agent (env_2_press.py:95-104): get_sort_obs() after the coming
// step's flow update and before the sensor is set.  A preview of state the step will recompute: nothing is stored
// (the flow update is a function of the state; the occupancy draw it discards comes from a stream nobody observes).
__global__ __launch_bounds__(kBlock) void k_sort_agent_obs(Params P, const uint4 *__restrict__ planes,
                                                           const uint32_t *__restrict__ table_image,
                                                           float *__restrict__ obs_out)
{
    const long long i = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (i >= P.n) return;
    const Tables tb = tables_at(table_image, P); // cold kernel: tables straight from global memory
    Env e;
    load_env<2, false>(e, planes, P, i);
    int k[4];
    container_purity_k(e, k);
    float o[13];
    if (P.gen_mode) { // a preview draws from a copy of the generator's stream: the step will draw the same
        load_gen(e, planes, P, i);
        update_environment<true>(e, P);
        sort_obs<true>(e, P, tb, k, o);
    } else {
        update_environment<false>(e, P);
        sort_obs<false>(e, P, tb, k, o);
    }
#pragma unroll
    for (int j = 0; j < 13; ++j) obs_out[i * 13 + j] = o[j];
}

// The same preview for a pressing agent (env_monolith.py:198-210: get_press_obs() after the flow update)
__global__ __launch_bounds__(kBlock) void k_press_agent_obs(Params P, const uint4 *__restrict__ planes,
                                                            const uint32_t *__restrict__ table_image,
                                                            float *__restrict__ obs_out)
{
    const long long i = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (i >= P.n) return;
    const Tables tb = tables_at(table_image, P);
    Env e;
    load_env<2, false>(e, planes, P, i);
    float o[16];
    if (P.gen_mode) {
        load_gen(e, planes, P, i);
        update_environment<true>(e, P);
        press_obs<true>(e, P, tb, o);
    } else {
        update_environment<false>(e, P);
        press_obs<false>(e, P, tb, o);
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) obs_out[i * 16 + j] = o[j];
}

template <int KIND>
__global__ __launch_bounds__(kBlock) void k_sample(Params P, const uint4 *__restrict__ planes,
                                                   const uint32_t *__restrict__ table_image, uint32_t flags,
                                                   uint64_t policy_seed, uint64_t policy_t, int *__restrict__ action_out)
{
    const long long i = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (i >= P.n) return;
    const Tables tb = tables_at(table_image, P);
    Env e;
    load_env<KIND, false>(e, planes, P, i);
    const uint32_t key = mse_policy_key(policy_seed, (uint64_t)(P.index_offset + i));
    action_out[i] = P.gen_mode ? policy_action<KIND, true>(e, action_mask_bits<KIND>(e, P), tb, flags, key, policy_t)
                               : policy_action<KIND, false>(e, action_mask_bits<KIND>(e, P), tb, flags, key, policy_t);
}

// Env_3_Monolith.step(mode='model') with no agents assigned (env_monolith.py:186-221): the sorting decision is
// rng_sorting.choice([0, 1]), the press action rng_pressing.choice(flatnonzero(press_action_masks())) with masking
// and rng_pressing.choice(11) without.  Both streams advance; the action is then stepped with masked semantics
// (the reference applies it through press_action_rules without sanitising, env_monolith.py:254-257).
__global__ __launch_bounds__(kBlock) void k_model_actions(Params P, uint4 *__restrict__ planes, uint32_t flags,
                                                          int *__restrict__ action_out)
{
    const long long i = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (i >= P.n) return;
    Env e;
    load_env<1, false>(e, planes, P, i); // KIND 1: with the rng_pressing planes
    Rng32 srt, prs;
    {
        const uint4 a = planes[(long long)PL_SORTRNG_STATE * P.n_pad + i], b = planes[(long long)PL_SORTRNG_INC * P.n_pad + i];
        const uint4 x = planes[(long long)PL_SORTRNG_AUX * P.n_pad + i];
        srt.g.s_lo = (uint64_t)a.x | ((uint64_t)a.y << 32);
        srt.g.s_hi = (uint64_t)a.z | ((uint64_t)a.w << 32);
        srt.g.i_lo = (uint64_t)b.x | ((uint64_t)b.y << 32);
        srt.g.i_hi = (uint64_t)b.z | ((uint64_t)b.w << 32);
        srt.uinteger = x.x;
        srt.has = (int)x.y;
    }
    prs.g = e.press;
    prs.uinteger = e.press_uint;
    prs.has = e.press_has;
    const int sort_mode = (flags & MSE_MODEL_NO_SORT_DRAW) ? 0 : (int)srt.lemire(2u);
    int press_action;
    if (flags & MSE_MODEL_NO_PRESS_DRAW) {
        press_action = 0;
    } else if (flags & MSE_STEP_UNMASKED) {
        press_action = (int)prs.lemire(11u);
    } else {
        const uint32_t bits = press_mask_bits(e, P);
        press_action = select_kth_bit(bits, (int)prs.lemire((uint32_t)__popc(bits)));
    }
    action_out[i] = P.env_kind == 1 ? sort_mode : (P.env_kind == 2 ? press_action : sort_mode * 11 + press_action);
    planes[(long long)PL_SORTRNG_STATE * P.n_pad + i] = pack_u64x2(srt.g.s_lo, srt.g.s_hi);
    planes[(long long)PL_SORTRNG_AUX * P.n_pad + i] = make_uint4(srt.uinteger, (uint32_t)srt.has, 0, 0);
    planes[(long long)PL_PRESS_STATE * P.n_pad + i] = pack_u64x2(prs.g.s_lo, prs.g.s_hi);
    // rng_pressing's 32-bit buffer lives in PL_MISC2 {.w = uinteger, flag bit in .x}
    uint4 m2 = planes[(long long)PL_MISC2 * P.n_pad + i];
    m2.w = prs.uinteger;
    m2.x = (m2.x & ~(FL_PRESS_HAS_U32 << 24)) | ((prs.has ? FL_PRESS_HAS_U32 : 0u) << 24);
    planes[(long long)PL_MISC2 * P.n_pad + i] = m2;
}

// snapshot record <-> planes (column map: include/mse.h MSE_SNAP_*, shared with oracle/oracle.py SNAP)
__global__ __launch_bounds__(kBlock) void k_get_state(Params P, const uint4 *__restrict__ planes,
                                                      long long *__restrict__ I, double *__restrict__ Dd,
                                                      unsigned long long *__restrict__ R)
{
    const long long i = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (i >= P.n) return;
    Env e;
    load_env<1, true>(e, planes, P, i);
    if (I != nullptr) {
        long long *r = I + i * MSE_SNAP_INTS;
        for (int c = 0; c < MSE_SNAP_INTS; ++c) r[c] = 0;
        const uint32_t w_in = stage_word(e.st_in, P), w_belt = stage_word(e.st_belt, P), w_sort = stage_word(e.st_sort, P);
        for (int m = 0; m < 4; ++m) {
            r[0 + m] = (w_in >> (8 * m)) & 0xFFu;
            r[4 + m] = (w_belt >> (8 * m)) & 0xFFu;
            r[8 + m] = (w_sort >> (8 * m)) & 0xFFu;
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
        unsigned long long *w = R + i * MSE_SNAP_RNG_WORDS;
        w[0] = e.rng.s_hi; w[1] = e.rng.s_lo; w[2] = e.rng.i_hi; w[3] = e.rng.i_lo; w[4] = 0; w[5] = 0;
        w[6] = e.noise.s_hi; w[7] = e.noise.s_lo; w[8] = e.noise.i_hi; w[9] = e.noise.i_lo; w[10] = 0; w[11] = 0;
        w[12] = e.press.s_hi; w[13] = e.press.s_lo; w[14] = e.press.i_hi; w[15] = e.press.i_lo;
        w[16] = (unsigned long long)e.press_has;
        w[17] = e.press_uint;
        const uint4 ss = planes[(long long)PL_SORTRNG_STATE * P.n_pad + i], si = planes[(long long)PL_SORTRNG_INC * P.n_pad + i];
        const uint4 sa = planes[(long long)PL_SORTRNG_AUX * P.n_pad + i];
        w[18] = (uint64_t)ss.z | ((uint64_t)ss.w << 32);
        w[19] = (uint64_t)ss.x | ((uint64_t)ss.y << 32);
        w[20] = (uint64_t)si.z | ((uint64_t)si.w << 32);
        w[21] = (uint64_t)si.x | ((uint64_t)si.y << 32);
        w[22] = sa.y;
        w[23] = sa.x;
        const uint4 gs = planes[(long long)PL_GEN_STATE * P.n_pad + i], gi = planes[(long long)PL_GEN_INC * P.n_pad + i];
        const uint4 ga = planes[(long long)PL_GEN_AUX * P.n_pad + i];
        w[24] = (uint64_t)gs.z | ((uint64_t)gs.w << 32);
        w[25] = (uint64_t)gs.x | ((uint64_t)gs.y << 32);
        w[26] = (uint64_t)gi.z | ((uint64_t)gi.w << 32);
        w[27] = (uint64_t)gi.x | ((uint64_t)gi.y << 32);
        w[28] = ga.y;
        w[29] = ga.x;
    }
}

// Stage vectors must be one of {zeros, pattern 1, pattern 2} (the only values the generator emits);
// anything else is counted in err_count and read as pattern 2.
__global__ __launch_bounds__(kBlock) void k_set_state(Params P, uint4 *__restrict__ planes,
                                                      const long long *__restrict__ I, const double *__restrict__ Dd,
                                                      const unsigned long long *__restrict__ R,
                                                      unsigned long long *__restrict__ err_count)
{
    const long long i = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (i >= P.n) return;
    Env e;
    load_env<1, true>(e, planes, P, i);
    if (I != nullptr) {
        const long long *r = I + i * MSE_SNAP_INTS;
        uint32_t w[3] = {0, 0, 0};
        for (int m = 0; m < 4; ++m) {
            w[0] |= ((uint32_t)r[0 + m] & 0xFFu) << (8 * m);
            w[1] |= ((uint32_t)r[4 + m] & 0xFFu) << (8 * m);
            w[2] |= ((uint32_t)r[8 + m] & 0xFFu) << (8 * m);
            e.ct[m] = (int)r[12 + m];
            e.cf[m] = (int)r[16 + m];
        }
        for (int s = 0; s < 3 && !P.gen_mode; ++s)
            if (w[s] != P.pat_word[0] && w[s] != P.pat_word[1] && w[s] != P.pat_word[2]) atomicAdd(err_count, 1ull);
        e.st_in = stage_id(w[0], P);
        e.st_belt = stage_id(w[1], P);
        e.st_sort = stage_id(w[2], P);
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
    if (Dd != nullptr) {
        for (int m = 0; m < 4; ++m) {
            e.acc[m] = Dd[i * 4 + m];
            // no reachable state has an accuracy outside [clip(baseline [+ boost] - noise), 1]; the three-role rollout
            // kernel's flow control (Params::ring_worst draws per step at most) is sized for that range
            if (!(e.acc[m] >= P.acc_floor[m] && e.acc[m] <= 1.0)) atomicAdd(err_count, 1ull);
        }
    }
    if (R != nullptr) {
        const unsigned long long *w = R + i * MSE_SNAP_RNG_WORDS;
        planes[(long long)PL_SORTRNG_STATE * P.n_pad + i] = pack_u64x2(w[19], w[18]);
        planes[(long long)PL_SORTRNG_INC * P.n_pad + i] = pack_u64x2(w[21], w[20]);
        planes[(long long)PL_SORTRNG_AUX * P.n_pad + i] = make_uint4((uint32_t)w[23], (uint32_t)w[22], 0, 0);
        planes[(long long)PL_GEN_STATE * P.n_pad + i] = pack_u64x2(w[25], w[24]);
        planes[(long long)PL_GEN_INC * P.n_pad + i] = pack_u64x2(w[27], w[26]);
        planes[(long long)PL_GEN_AUX * P.n_pad + i] = make_uint4((uint32_t)w[29], (uint32_t)w[28], 0, 0);
        e.rng.s_hi = w[0]; e.rng.s_lo = w[1]; e.rng.i_hi = w[2]; e.rng.i_lo = w[3];
        e.noise.s_hi = w[6]; e.noise.s_lo = w[7]; e.noise.i_hi = w[8]; e.noise.i_lo = w[9];
        e.press.s_hi = w[12]; e.press.s_lo = w[13]; e.press.i_hi = w[14]; e.press.i_lo = w[15];
        e.press_has = (int)w[16];
        e.press_uint = (uint32_t)w[17];
    }
    store_env<1, true>(e, planes, P, i, true);
}

// ==========================================================================================
// host side of the C ABI
// ==========================================================================================
struct mse_env {
    Params P;
    mse_config cfg;
    uint4 *planes;
    uint32_t *tables;            // device image of the lookup tables (build_tables)
    unsigned long long *err_count;
    int device;
    bool seeded;
    bool noise_on;
    bool literal;                // evaluate every Generator.choice draw in literal fp64
    bool pipelined;              // mse_rollout uses the dynamics/observer kernel (k_rollout_po)
    bool ring;                   // ... with RNG waves feeding an LDS ring (k_rollout_ring)
    bool ring_ok;                // the config allows an LDS ring at all (draws per step, integer draw path, build)
    uint64_t policy_t;
    // opt-in trace of one env (mse_trace_begin): records_dev f64[capacity][MSE_TRACE_COLS], caller-owned
    double *trace_rec;
    int64_t trace_env, trace_capacity, trace_count;
    int cus;                     // compute units of the device
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


// ------------------------------------------------------------------------------------------
// Lookup tables and per-pattern constants.  Every entry is the reference's own fp64 expression
// evaluated on the host for each possible integer argument (this translation unit is compiled
// with -ffp-contract=off), so a device lookup returns exactly what the reference computes.
// ------------------------------------------------------------------------------------------
static inline double host_round2(double x) { return std::nearbyint(x * 100.0) / 100.0; } // round(np.float64, 2)
// Python's round(float, 2): correctly rounded on the exact binary value, ties to even (glibc's printf rounds the same
// way) - what the reference computes where the operand is a plain Python float read from config.yml
static inline double host_round2_py(double x)
{
    char buf[64];
    std::snprintf(buf, sizeof buf, "%.2f", x);
    return std::strtod(buf, nullptr);
}
static inline float host_clip_f(float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); }

static int build_tables(const mse_config &c, Params &P, std::vector<uint32_t> &image, std::string &why)
{
    const int cap = c.container_capacity, S = c.bale_standard_size;
    uint32_t pat_rec[3][kPatStride];
    for (int k = 0; k < 3; ++k) {
        uint32_t w = 0;
        int sum = 0;
        for (int m = 0; m < 4; ++m) {
            // utils/input_generator.py:47: int(np.floor(ratio * batchsize)); id 0 = the empty stage after reset
            int cnt = k == 0 ? 0 : (int)std::floor(c.pattern_ratio[k - 1][m] * (double)c.input_batch_size);
            w |= (uint32_t)cnt << (8 * m);
            sum += cnt;
        }
        // utils/input_generator.py:49-55: units the floor()s leave over go to random materials - general generator mode
        P.gen_rem[k] = k > 0 ? c.input_batch_size - sum : 0;
        if (P.gen_rem[k] != 0) P.gen_mode = 1;
        P.pat_word[k] = w;
        auto f32_bits = [](float v) {
            uint32_t u;
            std::memcpy(&u, &v, 4);
            return u;
        };
        pat_rec[k][0] = w;
        // env_super.py:456 input_occupancy = round(sum/100, 2); get_sort_obs casts to f32 and clips to [-1,1]
        pat_rec[k][1] = f32_bits(host_clip_f((float)((double)sum / 100.0), -1.0f, 1.0f));
        pat_rec[k][3] = 0;
        double pr[4];
        for (int m = 0; m < 4; ++m) {
            int cnt = (int)((w >> (8 * m)) & 0xFFu);
            pr[m] = sum > 0 ? (double)cnt / (double)sum : 0.0;                                   // env_super.py:199-210
            pat_rec[k][4 + m] = f32_bits(host_clip_f((float)pr[m], -1.0f, 1.0f));
            pat_rec[k][8 + m] = f32_bits(host_clip_f((float)((double)cnt / (double)c.stage_capacity), 0.0f, 1.0f)); // :351
        }
        pat_rec[k][2] = (pr[0] + pr[2] > pr[1] + pr[3]) ? 0u : 1u;                               // env_super.py:479-482
    }
    P.pat_word1 = P.pat_word[1];
    P.pat_word2 = P.pat_word[2];
    {
        const float occ = host_clip_f((float)((double)c.input_batch_size / 100.0), -1.0f, 1.0f); // env_super.py:456, :318
        std::memcpy(&P.occ_nonempty, &occ, 4);
    }
    if (!P.gen_mode && (P.pat_word[1] == P.pat_word[2] || P.pat_word[1] == 0 || P.pat_word[2] == 0)) {
        why = "the two seasonal patterns must give distinct, non-empty material counts";
        return MSE_ERR_UNSUPPORTED_CONFIG;
    }
    // fill_ratio thresholds of calculate_press_reward as integer levels (env_super.py:1020-1027)
    P.sev_negative = c.overflow_penalty_severe < 0.0 ? 1 : 0;
    P.mild_negative = c.overflow_penalty_mild < 0.0 ? 1 : 0;
    P.thr_sev = P.thr_mild = cap;
    for (int L = cap; L >= 0; --L) {
        double fill = (double)L / (double)cap;
        if (fill > 0.95) P.thr_sev = L - 1;
        if (fill > 0.90) P.thr_mild = L - 1;
    }
    double acc_rows[3][4]; // np.clip(acc + 0, 0, 1) for mode 0, mode 1, any other mode (env_super.py:499-509)
    for (int m = 0; m < 4; ++m) {
        P.k_thr[m] = (int)std::nearbyint(c.quality_threshold_r2[m] * 100.0);
        if (P.k_thr[m] < 0 || P.k_thr[m] > 100) {
            why = "quality thresholds must lie in [0, 1]";
            return MSE_ERR_UNSUPPORTED_CONFIG;
        }
        const double lo = c.baseline_accuracy[m], hi = c.baseline_accuracy[m] + c.boost;
        const double lo_c = lo < 0.0 ? 0.0 : (lo > 1.0 ? 1.0 : lo), hi_c = hi < 0.0 ? 0.0 : (hi > 1.0 ? 1.0 : hi);
        acc_rows[0][m] = (m == 0 || m == 2) ? hi_c : lo_c;
        acc_rows[1][m] = (m == 1 || m == 3) ? hi_c : lo_c;
        acc_rows[2][m] = lo_c;
    }
    const double peaks[4] = {0.0, 1.0 / 3.0, 2.0 / 3.0, 1.0}; // env_super.py:1065
    auto put_f32 = [&](float v) {
        uint32_t u;
        std::memcpy(&u, &v, 4);
        image.push_back(u);
    };
    auto put_f64 = [&](double v) {
        uint64_t u;
        std::memcpy(&u, &v, 8);
        image.push_back((uint32_t)u);
        image.push_back((uint32_t)(u >> 32));
    };
    image.clear();
    P.off_lvl = (int)image.size(); // env_super.py:339-344,359
    for (int L = 0; L <= cap; ++L) put_f32(host_clip_f((float)((double)L / (double)cap), 0.0f, 1.0f));
    P.off_pdiff = (int)image.size(); // env_super.py:212-227, 771-791, 325
    for (int m = 0; m < 4; ++m) {
        for (int k = 0; k <= 101; ++k) {
            // a container's purity is an np.float64 quotient (numpy's round); an EMPTY container's is the threshold, a
            // Python float, and so is its difference (Python's round): env_super.py:212-227, 786-789
            const double diff = k <= 100 ? host_round2((double)k / 100.0 - c.quality_threshold[m])
                                         : host_round2_py(c.quality_threshold_r2[m] - c.quality_threshold[m]);
            put_f32(host_clip_f((float)diff, -1.0f, 1.0f));
        }
    }
    P.off_timer0 = (int)image.size(); // env_super.py:354-356
    for (int t = 0; t <= c.press_time[0]; ++t) put_f32(host_clip_f((float)((double)t / (double)c.press_time[0]), 0.0f, 1.0f));
    P.off_timer1 = (int)image.size();
    for (int t = 0; t <= c.press_time[1]; ++t) put_f32(host_clip_f((float)((double)t / (double)c.press_time[1]), 0.0f, 1.0f));
    if (image.size() & 1u) image.push_back(0u); // 8-byte alignment of the f64 tables
    P.off_tanh = (int)image.size(); // env_super.py:963-1003 by the sum s of the four purity hundredths
    for (int s = 0; s <= 400; ++s) {
        long double total = (long double)s / 100.0L - 4.0L * (long double)c.purity_threshold_theta;
        double state_based = (double)((total / 4.0L) * 2.0L);
        put_f64(std::tanh(state_based / c.tanh_temperature));
    }
    P.off_eff = (int)image.size(); // env_super.py:1058-1062
    for (int d = 0; d <= S / 2; ++d) put_f64((1.0 - 4.0 * ((double)d / (double)S)) * c.bale_efficiency_factor);
    P.off_acc = (int)image.size();
    for (int r = 0; r < 3; ++r)
        for (int m = 0; m < 4; ++m) put_f64(acc_rows[r][m]);
    P.off_bonus = (int)image.size(); // env_super.py:1065-1069
    for (int b = 0; b < 4; ++b) put_f64(peaks[b] - c.bale_efficiency_factor);
    while (image.size() & 3u) image.push_back(0u); // 16-byte alignment of the per-stage records (read as float4)
    P.off_pat = (int)image.size();
    for (int k = 0; k < 3; ++k)
        for (int w = 0; w < kPatStride; ++w) image.push_back(pat_rec[k][w]);
    P.off_ptime = (int)image.size();
    image.push_back((uint32_t)c.press_time[0]);
    image.push_back((uint32_t)c.press_time[1]);
    for (int w = 0; w < 4; ++w) image.push_back(P.qi_down[w]); // bale_quality_int's mask (set by mse_create before this)
    P.off_cst = (int)image.size(); // even: every section so far has an even word count after off_tanh
    {
        double cst[CST_COUNT] = {};
        cst[CST_PEN_CAT] = c.overflow_penalty_catastrophic;
        cst[CST_PEN_SEV] = c.overflow_penalty_severe;
        cst[CST_PEN_MILD] = c.overflow_penalty_mild;
        cst[CST_MAX_STATE] = c.max_state_reward;
        cst[CST_OVERFLOW_PEN] = c.overflow_termination_penalty;
        cst[CST_REM_THR] = c.bale_remainder_threshold;
        cst[CST_BOOST] = c.boost;
        cst[CST_NOISE] = c.noise;
        for (int m = 0; m < 4; ++m) cst[CST_BASE_ACC0 + m] = c.baseline_accuracy[m];
        for (int k = 0; k < CST_COUNT; ++k) put_f64(cst[k]);
    }
    while (image.size() & 3u) image.push_back(0u); // the one-lane rollout kernel copies [0, off_jump) in 16-byte pieces
    P.off_jump = (int)image.size();
    {
        // LCG jump-ahead by 2^j steps: A = M^(2^j), G = 1 + M + ... + M^(2^j - 1)  (mod 2^128)
        typedef unsigned __int128 u128;
        u128 A = (((u128)0x2360ED051FC65DA4ull) << 64) | (u128)0x4385DF649FCCF645ull, G = 1;
        for (int j = 0; j < kJumpBits; ++j) {
            uint64_t w4[4] = {(uint64_t)A, (uint64_t)(A >> 64), (uint64_t)G, (uint64_t)(G >> 64)};
            for (int q = 0; q < 4; ++q) {
                image.push_back((uint32_t)w4[q]);
                image.push_back((uint32_t)(w4[q] >> 32));
            }
            G = G * (A + 1); // G_{2n} = G_n (A_n + 1)
            A = A * A;
        }
    }
    P.off_back = (int)image.size(); // even
    {
        // jump back by d steps (pcg_step_back): A_{-d} = M^{-d}, G_{-d} = -M^{-d} G_d
        typedef unsigned __int128 u128;
        const u128 M = (((u128)0x2360ED051FC65DA4ull) << 64) | (u128)0x4385DF649FCCF645ull;
        u128 Minv = M; // Newton: x <- x (2 - M x) doubles the correct low bits; M * M = 1 mod 8 gives 3 to start
        for (int it = 0; it < 7; ++it) Minv = Minv * (2 - M * Minv);
        u128 Ainv = 1, Gd = 0, Ad = 1; // d = 0
        for (int d = 0; d < kRingBackSteps; ++d) {
            const u128 Gneg = (u128)0 - Ainv * Gd;
            const uint64_t w4[4] = {(uint64_t)Ainv, (uint64_t)(Ainv >> 64), (uint64_t)Gneg, (uint64_t)(Gneg >> 64)};
            for (int q = 0; q < 4; ++q) {
                image.push_back((uint32_t)w4[q]);
                image.push_back((uint32_t)(w4[q] >> 32));
            }
            Gd = Gd + Ad; // G_{d+1} = G_d + M^d
            Ad = Ad * M;
            Ainv = Ainv * Minv;
        }
    }
    P.off_gprop = P.off_gfrac = 0;
    if (P.gen_mode) { // per-count tables: every batch holds input_batch_size units, so a share is a function of the count
        P.off_gprop = (int)image.size();
        for (int k = 0; k < 256; ++k)
            put_f32(host_clip_f((float)((double)k / (double)c.input_batch_size), -1.0f, 1.0f));
        P.off_gfrac = (int)image.size();
        for (int k = 0; k < 256; ++k) put_f32(host_clip_f((float)((double)k / (double)c.stage_capacity), 0.0f, 1.0f));
    }
    while (image.size() & 3u) image.push_back(0u); // copied to LDS in 16-byte pieces
    P.table_words = (int)image.size();
    if (P.table_words > 16384) {
        why = "container_capacity / bale_standard_size too large for the LDS-resident tables (64 KiB)";
        return MSE_ERR_UNSUPPORTED_CONFIG;
    }
    return MSE_OK;
}

template <int KIND>
static size_t lds_bytes_step(const mse_env *h)
{
    return (size_t)LdsLayout<KIND>::table_offset_step + (size_t)h->P.table_words * 4u;
}
template <int KIND>
static size_t lds_bytes_rollout(const mse_env *h)
{
    return (size_t)(h->noise_on ? RolloutLayout<KIND, true>::table_offset : RolloutLayout<KIND, false>::table_offset) +
           (size_t)(h->P.gen_mode ? h->P.table_words : h->P.off_jump) * 4u;
}

static inline dim3 grid_of(const mse_env *h) { return dim3((unsigned)(h->P.n_pad / kBlock)); }

template <int KIND>
static void launch_step(mse_env *h, hipStream_t s, const int32_t *action, const int32_t *sort_mode, uint32_t flags,
                        float *obs, float *rew, double *rew64, uint8_t *done, uint8_t *mask, float *tobs)
{
    const bool lit = h->literal;
    const size_t lds = lds_bytes_step<KIND>(h);
    if (h->trace_rec != nullptr) { // the traced variant: one more record for env trace_env
        double *rec = h->trace_rec + h->trace_count * MSE_TRACE_COLS;
#define MSE_LAUNCH_STEP_T(NOISE, LIT, GEN)                                                               \
    hipLaunchKernelGGL((k_step<KIND, NOISE, LIT, true, GEN>), grid_of(h), dim3(kBlock), lds, s, h->P, h->planes, h->tables, \
                       action, sort_mode, flags, obs, rew, rew64, done, mask, tobs, h->err_count, rec, (long long)h->trace_env)
#define MSE_LAUNCH_STEP_TG(NOISE, LIT) do { if (h->P.gen_mode) MSE_LAUNCH_STEP_T(NOISE, LIT, true); else MSE_LAUNCH_STEP_T(NOISE, LIT, false); } while (0)
        if (h->noise_on) {
            if (lit) MSE_LAUNCH_STEP_TG(true, true); else MSE_LAUNCH_STEP_TG(true, false);
        } else {
            if (lit) MSE_LAUNCH_STEP_TG(false, true); else MSE_LAUNCH_STEP_TG(false, false);
        }
#undef MSE_LAUNCH_STEP_TG
#undef MSE_LAUNCH_STEP_T
        h->trace_count += 1;
        return;
    }
#define MSE_LAUNCH_STEP(NOISE, LIT, GEN)                                                                 \
    hipLaunchKernelGGL((k_step<KIND, NOISE, LIT, false, GEN>), grid_of(h), dim3(kBlock), lds, s, h->P, h->planes, h->tables, \
                       action, sort_mode, flags, obs, rew, rew64, done, mask, tobs, h->err_count, (double *)nullptr, -1LL)
#define MSE_LAUNCH_STEP_G(NOISE, LIT) do { if (h->P.gen_mode) MSE_LAUNCH_STEP(NOISE, LIT, true); else MSE_LAUNCH_STEP(NOISE, LIT, false); } while (0)
    if (h->noise_on) {
        if (lit) MSE_LAUNCH_STEP_G(true, true); else MSE_LAUNCH_STEP_G(true, false);
    } else {
        if (lit) MSE_LAUNCH_STEP_G(false, true); else MSE_LAUNCH_STEP_G(false, false);
    }
#undef MSE_LAUNCH_STEP_G
#undef MSE_LAUNCH_STEP
}

// RngRing::load masks an output's row into the lane's LDS address, which needs the ring on a 64 KiB LDS boundary:
// ring_offset == 0 inside the dynamic LDS block, and the dynamic block itself at LDS address 0, i.e. no static
// __shared__ in these kernels.  A property of the build; checked once on the host (mse_create) instead of by a trap
// in the kernel.
static bool ring_kernels_static_lds_free()
{
    static int cached = -1;
    if (cached < 0) {
        const void *fns[] = {(const void *)k_rollout_ring<1, false>, (const void *)k_rollout_ring<1, true>,
                             (const void *)k_rollout_ring<2, false>, (const void *)k_rollout_ring<2, true>,
                             (const void *)k_rollout_ring<3, false>, (const void *)k_rollout_ring<3, true>,
                             (const void *)k_rollout_policy_roles<1, false, true>, (const void *)k_rollout_policy_roles<1, true, true>,
                             (const void *)k_rollout_policy_roles<2, false, true>, (const void *)k_rollout_policy_roles<2, true, true>,
                             (const void *)k_rollout_policy_roles<3, false, true>, (const void *)k_rollout_policy_roles<3, true, true>};
        cached = 1;
        for (const void *f : fns) {
            hipFuncAttributes a{};
            if (hipFuncGetAttributes(&a, f) != hipSuccess || a.sharedSizeBytes != 0) cached = 0;
        }
    }
    return cached == 1;
}

// Params::ring_fwd: s_{n+d} = M^d s_n + (1 + M + ... + M^{d-1}) inc for d = ring_worst, the distance between the two
// halves of the ring's priming (k_rollout_ring)
static void set_ring_forward_jump(Params &P)
{
    typedef unsigned __int128 u128;
    const u128 M = (((u128)0x2360ED051FC65DA4ull) << 64) | (u128)0x4385DF649FCCF645ull;
    u128 A = 1, G = 0;
    for (int d = 0; d < P.ring_worst; ++d) {
        G = G + A;
        A = A * M;
    }
    P.ring_fwd[0] = (uint64_t)A;
    P.ring_fwd[1] = (uint64_t)(A >> 64);
    P.ring_fwd[2] = (uint64_t)G;
    P.ring_fwd[3] = (uint64_t)(G >> 64);
}

template <int KIND>
static void launch_rollout(mse_env *h, hipStream_t s, int k_steps, uint64_t policy_seed, const int32_t *sort_mode,
                           uint32_t flags, int32_t *actions, float *obs, float *rew, uint8_t *done, uint8_t *mask)
{
    const bool lit = h->literal;
    if (h->ring) {
        const dim3 grid((unsigned)(h->P.n_pad / kPoEnvs));
        const size_t table_bytes = (size_t)h->P.table_words * 4u;
        const size_t lds_n = (size_t)RingLayout<KIND, true>::table_offset + table_bytes;
        const size_t lds_p = (size_t)RingLayout<KIND, false>::table_offset + table_bytes;
        if (h->noise_on)
            hipLaunchKernelGGL((k_rollout_ring<KIND, true>), grid, dim3(kRingThreads), lds_n, s, h->P, h->planes,
                               h->tables, k_steps, policy_seed, h->policy_t, sort_mode, flags, actions, obs, rew, done,
                               mask);
        else
            hipLaunchKernelGGL((k_rollout_ring<KIND, false>), grid, dim3(kRingThreads), lds_p, s, h->P, h->planes,
                               h->tables, k_steps, policy_seed, h->policy_t, sort_mode, flags, actions, obs, rew, done,
                               mask);
        return;
    }
    if (h->pipelined) {
        const dim3 grid((unsigned)(h->P.n_pad / kPoEnvs));
        const size_t table_bytes = (size_t)h->P.table_words * 4u;
        const size_t lds_noise = (size_t)PoLayout<KIND, true>::table_offset + table_bytes;
        const size_t lds_plain = (size_t)PoLayout<KIND, false>::table_offset + table_bytes;
#define MSE_LAUNCH_PO(NOISE, LIT)                                                                        \
    hipLaunchKernelGGL((k_rollout_po<KIND, NOISE, LIT>), grid, dim3(kPoThreads), (NOISE ? lds_noise : lds_plain), s, \
                       h->P, h->planes, h->tables, k_steps, policy_seed, h->policy_t, sort_mode, flags, actions, obs, \
                       rew, done, mask)
        if (h->noise_on) {
            if (lit) MSE_LAUNCH_PO(true, true); else MSE_LAUNCH_PO(true, false);
        } else {
            if (lit) MSE_LAUNCH_PO(false, true); else MSE_LAUNCH_PO(false, false);
        }
#undef MSE_LAUNCH_PO
        return;
    }
    const size_t lds = lds_bytes_rollout<KIND>(h);
#define MSE_LAUNCH_ROLLOUT(NOISE, LIT, GEN)                                                              \
    hipLaunchKernelGGL((k_rollout<KIND, NOISE, LIT, GEN>), grid_of(h), dim3(kBlock), lds, s, h->P, h->planes, h->tables, \
                       k_steps, policy_seed, h->policy_t, sort_mode, flags, actions, obs, rew, done, mask)
#define MSE_LAUNCH_ROLLOUT_G(NOISE, LIT) do { if (h->P.gen_mode) MSE_LAUNCH_ROLLOUT(NOISE, LIT, true); else MSE_LAUNCH_ROLLOUT(NOISE, LIT, false); } while (0)
    if (h->noise_on) {
        if (lit) MSE_LAUNCH_ROLLOUT_G(true, true); else MSE_LAUNCH_ROLLOUT_G(true, false);
    } else {
        if (lit) MSE_LAUNCH_ROLLOUT_G(false, true); else MSE_LAUNCH_ROLLOUT_G(false, false);
    }
#undef MSE_LAUNCH_ROLLOUT_G
#undef MSE_LAUNCH_ROLLOUT
}

template <int KIND>
static int launch_rollout_policy(mse_env *h, const mse_policy *pol, const mse_policy *sort_pol, hipStream_t s, int k_steps,
                                 uint64_t seed, int deterministic, const int32_t *sort_mode, uint32_t flags, float *obs, uint8_t *mask,
                                 int32_t *actions, float *logp, float *value, float *rew, uint8_t *start,
                                 float *last_value, uint8_t *last_done)
{
    // Shape: eight waves per workgroup, two per SIMD.  While 32-env waves leave every CU at most one workgroup's worth
    // (n <= 256 envs x CUs) a wave owns 32 envs - at that size 64-env waves would run one per SIMD, at a vector
    // instruction per ~5 cycles; beyond, 64 envs.  The exact-f32 form only exists in the 64-env shape.
    const int cus = h->cus; // queried once at create: hipGetDeviceProperties is far too slow for a launch path
    const bool f16 = pol->use_f16();
    // Batches that leave a SIMD 64 envs (n <= 256 envs x CUs), f16x3 form, no in-loop sorting policy: the two-role
    // kernel in roles (one workgroup of four actor / critic[ / RNG] wave sets per 256 envs).  rollout_pipeline = 2 ("one
    // lane per env, no roles") keeps the plain kernel and 1 ("two roles") the form without the RNG waves, which is how the
    // tests hold the three against each other.
    if (f16 && sort_pol == nullptr && h->P.n <= (long long)256 * cus && h->cfg.rollout_pipeline != 2) {
        const size_t table_bytes = (size_t)h->P.table_words * 4u;
        const size_t lds_ring = (size_t)PolRolesLayout<KIND, true>::table_offset + table_bytes;
        const size_t lds_pair = (size_t)PolRolesLayout<KIND, false>::table_offset + table_bytes;
        const bool with_ring = h->ring_ok && lds_ring <= (size_t)160 * 1024 && h->cfg.rollout_pipeline != 1;
        if (with_ring || lds_pair <= (size_t)160 * 1024) {
            const dim3 grid_r((unsigned)((h->P.n + kPoEnvs - 1) / kPoEnvs));
#define MSE_LAUNCH_RPR(NOISE, RING)                                                                                  \
    hipLaunchKernelGGL((k_rollout_policy_roles<KIND, NOISE, RING>), grid_r, dim3((unsigned)PolRolesLayout<KIND, RING>::kThreads), \
                       (RING ? lds_ring : lds_pair), s, h->P, h->planes, h->tables, pol->blob, k_steps, seed, h->policy_t,   \
                       deterministic, sort_mode, flags, obs, mask, actions, logp, value, rew, start, last_value, last_done)
            if (with_ring) {
                if (h->noise_on) MSE_LAUNCH_RPR(true, true); else MSE_LAUNCH_RPR(false, true);
            } else {
                if (h->noise_on) MSE_LAUNCH_RPR(true, false); else MSE_LAUNCH_RPR(false, false);
            }
#undef MSE_LAUNCH_RPR
            return MSE_OK;
        }
    }
    const int tiles = (f16 && h->P.n <= (long long)256 * cus) ? 1 : 2;
    // (the exact-f32 form below that size: four 64-env waves per workgroup, so that every CU gets one)
    const int n_waves = (!f16 && h->P.n <= (long long)256 * cus) ? 4 : 8;
    const long long envs_per_wg = 32LL * tiles * n_waves;
    const dim3 grid((unsigned)((h->P.n + envs_per_wg - 1) / envs_per_wg)), block((unsigned)(64 * n_waves));
    const size_t wave_bytes = tiles == 1 ? PolLayout<KIND, 1>::wave_bytes : PolLayout<KIND, 2>::wave_bytes;
    const size_t lds = (size_t)msep::kLdsFloats * 4u * (sort_pol ? 2u : 1u) + (size_t)h->P.table_words * 4u + (size_t)n_waves * wave_bytes;
    if (lds > (size_t)160 * 1024) return MSE_ERR_UNSUPPORTED_CONFIG;
    if (KIND == 2 && sort_pol != nullptr) { // Env_2 with its sorting agent in the loop (f16x3 form of both networks)
#define MSE_LAUNCH_RPS(NOISE, TILES)                                                                                 \
    hipLaunchKernelGGL((k_rollout_policy<2, NOISE, TILES, true, true>), grid, block, lds, s, h->P, h->planes, h->tables, \
                       pol->blob, sort_pol->blob, k_steps, seed, h->policy_t, deterministic, sort_mode, flags, obs, mask, \
                       actions, logp, value, rew, start, last_value, last_done)
        if (tiles == 1) {
            if (h->noise_on) MSE_LAUNCH_RPS(true, 1); else MSE_LAUNCH_RPS(false, 1);
        } else {
            if (h->noise_on) MSE_LAUNCH_RPS(true, 2); else MSE_LAUNCH_RPS(false, 2);
        }
#undef MSE_LAUNCH_RPS
        return MSE_OK;
    }
#define MSE_LAUNCH_RP(NOISE, TILES, F16)                                                                             \
    hipLaunchKernelGGL((k_rollout_policy<KIND, NOISE, TILES, F16>), grid, block, lds, s, h->P, h->planes, h->tables, \
                       pol->blob, (const float *)nullptr, k_steps, seed, h->policy_t, deterministic, sort_mode, flags, obs, \
                       mask, actions, logp, value, rew, start, last_value, last_done)
    if (!f16) {
        if (h->noise_on) MSE_LAUNCH_RP(true, 2, false); else MSE_LAUNCH_RP(false, 2, false);
    } else if (tiles == 1) {
        if (h->noise_on) MSE_LAUNCH_RP(true, 1, true); else MSE_LAUNCH_RP(false, 1, true);
    } else {
        if (h->noise_on) MSE_LAUNCH_RP(true, 2, true); else MSE_LAUNCH_RP(false, 2, true);
    }
#undef MSE_LAUNCH_RP
    return MSE_OK;
}

extern "C" {

int mse_version(void) { return MSE_VERSION; }
uint32_t mse_tie_window(void) { return MSE_TIE_WINDOW; }
#ifdef MSE_CLOCK_PROBE
// diagnostic build: read and clear {shader cycles, 100 MHz ticks, launches} of workgroup 0's launches
int mse_debug_clock(unsigned long long *out3)
{
    unsigned long long zero[3] = {};
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(out3, HIP_SYMBOL(g_clock_probe), sizeof(zero)) != hipSuccess) return -1;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_clock_probe), zero, sizeof(zero)) != hipSuccess) return -1;
    return 0;
}
int mse_debug_wg_probe(unsigned long long *out3072)
{
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    return hipMemcpyFromSymbol(out3072, HIP_SYMBOL(g_wg_probe), sizeof(unsigned long long) * 3 * 1024) == hipSuccess ? 0 : -1;
}
#endif
#ifdef MSE_TIMELINE
// diagnostic build: read and clear the per-role section cycle sums (role-major, 8 sections each)
int mse_debug_timeline(unsigned long long *out32)
{
    unsigned long long zero[32] = {};
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(out32, HIP_SYMBOL(mse::g_timeline), sizeof(zero)) != hipSuccess) return -1;
    if (hipMemcpyToSymbol(HIP_SYMBOL(mse::g_timeline), zero, sizeof(zero)) != hipSuccess) return -1;
    return 0;
}
#endif

const char *mse_last_error(void) { return g_last_error.c_str(); }
} // extern "C"
int mse_internal_fail(int status, const char *msg) { return fail(status, msg); } // for the library's other translation units
extern "C" {

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

    mse_env *h = new (std::nothrow) mse_env(); // value-initialised: device pointers start out null
    if (!h) return fail(MSE_ERR_INVALID_ARGUMENT, "out of host memory");
    h->cfg = *cfg;
    h->device = device_id;
    h->seeded = false;
    h->policy_t = 0;
    h->trace_rec = nullptr;
    h->trace_env = -1;
    h->trace_capacity = h->trace_count = 0;
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
    P.press_time[0] = P.press_time0 = cfg->press_time[0];
    P.press_time[1] = P.press_time1 = cfg->press_time[1];
    P.inv_balesize = 1.0f / (float)cfg->bale_standard_size;
    for (int w = 0; w < 4; ++w) P.qi_down[w] = 0;
    for (int q = 0; q <= 100; ++q) { // env_super.py:664-666 with the literal expressions
        const double qd = (double)q / 100.0;
        const int qi = (int)(qd * 100.0);
        if (qi != q) P.qi_down[q >> 5] |= 1u << (q & 31);
        if (qi != q && qi != q - 1) return fail(MSE_ERR_UNSUPPORTED_CONFIG, "int(q*100) is not q or q-1");
    }
    P.rem_thr_units = (int)std::floor((double)cfg->bale_standard_size * cfg->bale_remainder_threshold);
    P.max_state_reward = cfg->max_state_reward;
    std::vector<uint32_t> image;
    std::string why;
    int trc = build_tables(*cfg, P, image, why);
    if (trc != MSE_OK) {
        delete h;
        return fail(trc, why);
    }
    // the byte-packed integer draw needs every prefix sum below 128; larger batches draw in literal fp64
    h->literal = cfg->literal_choice != 0 || cfg->input_batch_size > 127;
    // rollout kernel: the multi-role kernels serve 256 envs per workgroup, one workgroup per CU (their LDS image is
    // the whole CU's), so they pay off exactly while the batch fits the chip in one round: n <= 256 x CUs (65 536 on
    // an MI355X).  Beyond that a one-lane-per-env grid already gives every SIMD several waves and wins (measured at
    // 131 072 envs: 16.5 G env-steps/s against 15.9).  0 = decide by size, 1 / 3 = always, 2 = never
    int cus = 256;
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device_id) == hipSuccess && prop.multiProcessorCount > 0)
            cus = prop.multiProcessorCount;
    }
    h->cus = cus;
    // ... and again while it fills most of a SECOND round: same-box at 64 steps per launch, three-role kernel against
    // one lane per env: 98 304 envs 17.5 G against 15.5, 131 072 envs 22.9 against 20.5 (two full rounds run at the
    // one-round rate; the one-lane grid has only two waves per SIMD there), 196 608 envs 23.1 against 23.8
    const int64_t n_wg = (n_envs + kPoEnvs - 1) / kPoEnvs;
    const bool by_size = n_wg <= cus || (n_wg > cus + cus / 4 && n_wg <= 2 * (int64_t)cus);
    h->pipelined = cfg->rollout_pipeline == 1 || cfg->rollout_pipeline == 3 || (cfg->rollout_pipeline == 0 && by_size);
    if (P.gen_mode) { // general generator mode: the multi-role kernels carry stage ids, not counts
        if (cfg->rollout_pipeline == 1 || cfg->rollout_pipeline == 3) {
            delete h;
            return fail(MSE_ERR_UNSUPPORTED_CONFIG, "rollout_pipeline 1 / 3 need a remainder-free input_batch_size (the "
                                                    "one-lane kernels serve the general generator)");
        }
        h->pipelined = false;
    }
    {
        // draws per step are bounded by the mis-sorted units of the two stations a mode leaves unboosted at the
        // lowest accuracy the noise allows; the ring kernel needs that bound <= kRingMaxPerStep
        // false_m = target - rint(target * acc) grows with target (<= the pattern's count) and falls with acc
        // (>= clip(baseline [+ boost] - noise)); modes other than 0 / 1 (no boost) exist only for Env_2's
        // externally supplied sorting decision
        int worst = 0;
        for (int m = 0; m < 4; ++m) {
            const double lo = cfg->baseline_accuracy[m] - cfg->noise, hi = cfg->baseline_accuracy[m] + cfg->boost - cfg->noise;
            const double lo_c = lo < 0.0 ? 0.0 : (lo > 1.0 ? 1.0 : lo), hi_c = hi < 0.0 ? 0.0 : (hi > 1.0 ? 1.0 : hi);
            P.acc_floor[m] = lo_c < hi_c ? lo_c : hi_c;
        }
        const int n_modes = cfg->env_kind == MSE_ENV_PRESS ? 3 : 2;
        for (int k = 1; k <= 2; ++k) {
            for (int mode = 0; mode < n_modes; ++mode) {
                int sum = 0;
                for (int m = 0; m < 4; ++m) {
                    const bool boosted = mode == 0 ? (m == 0 || m == 2) : (mode == 1 ? (m == 1 || m == 3) : false);
                    double acc = cfg->baseline_accuracy[m] + (boosted ? cfg->boost : 0.0) - cfg->noise;
                    acc = acc < 0.0 ? 0.0 : (acc > 1.0 ? 1.0 : acc);
                    const int cnt = (int)((P.pat_word[k] >> (8 * m)) & 0xFFu);
                    sum += cnt - (int)std::nearbyint((double)cnt * acc);
                }
                worst = sum > worst ? sum : worst;
            }
        }
        // ... and its LDS image (ring, obs tile, two snapshots, bale ledger, tables) within the CU's 160 KiB
        size_t ring_lds = (size_t)P.table_words * 4u;
        const bool nz = h->noise_on;
        if (cfg->env_kind == MSE_ENV_SORT) ring_lds += nz ? RingLayout<1, true>::table_offset : RingLayout<1, false>::table_offset;
        else if (cfg->env_kind == MSE_ENV_PRESS) ring_lds += nz ? RingLayout<2, true>::table_offset : RingLayout<2, false>::table_offset;
        else ring_lds += nz ? RingLayout<3, true>::table_offset : RingLayout<3, false>::table_offset;
        P.ring_worst = worst;
        set_ring_forward_jump(P);
        h->ring_ok = worst <= kRingMaxPerStep && !h->literal && !P.gen_mode && ring_kernels_static_lds_free();
        const bool fits = worst <= kRingMaxPerStep && !h->literal && ring_lds <= (size_t)160 * 1024 && ring_kernels_static_lds_free();
        // (the second-round rule above was measured with the three-role kernel only)
        if (cfg->rollout_pipeline == 0 && n_wg > cus && !fits) h->pipelined = false;
        h->ring = h->pipelined && fits && cfg->rollout_pipeline != 1;
        if (cfg->rollout_pipeline == 3 && !fits) {
            delete h; // nothing is allocated on the device yet
            return fail(MSE_ERR_UNSUPPORTED_CONFIG, "rollout_pipeline=3 (ring kernel) needs at most 31 draws per step, "
                                                    "the integer draw path and an LDS image within 160 KiB");
        }
    }

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
    hipError_t e3 = hipMalloc(reinterpret_cast<void **>(&h->tables), image.size() * sizeof(uint32_t));
    if (e3 != hipSuccess) {
        (void)hipFree(h->planes);
        (void)hipFree(h->err_count);
        delete h;
        return fail(MSE_ERR_HIP, std::string("hipMalloc(tables): ") + hipGetErrorString(e3));
    }
    hipError_t e4 = hipMemcpy(h->tables, image.data(), image.size() * sizeof(uint32_t), hipMemcpyHostToDevice);
    if (e4 == hipSuccess) e4 = hipMemset(h->planes, 0, bytes);
    if (e4 == hipSuccess) e4 = hipMemset(h->err_count, 0, sizeof(unsigned long long));
    if (e4 != hipSuccess) {
        (void)mse_destroy(h);
        return fail(MSE_ERR_HIP, std::string("initialising the state planes / tables: ") + hipGetErrorString(e4));
    }
    *out = h;
    return MSE_OK;
}

int mse_destroy(mse_env *h)
{
    if (!h) return MSE_OK;
    (void)hipSetDevice(h->device);
    (void)hipFree(h->planes);
    (void)hipFree(h->tables);
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
    case 1: hipLaunchKernelGGL(k_reset<1>, grid_of(h), dim3(kBlock), 0, s, h->P, h->planes, h->tables, seeds, which, obs_out, mask_out); break;
    case 2: hipLaunchKernelGGL(k_reset<2>, grid_of(h), dim3(kBlock), 0, s, h->P, h->planes, h->tables, seeds, which, obs_out, mask_out); break;
    default: hipLaunchKernelGGL(k_reset<3>, grid_of(h), dim3(kBlock), 0, s, h->P, h->planes, h->tables, seeds, which, obs_out, mask_out); break;
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
    if (flags & ~(MSE_STEP_UNMASKED | MSE_STEP_CHECK_OVERFLOW | MSE_STEP_SANITIZE_LATE))
        return fail(MSE_ERR_INVALID_ARGUMENT, "unknown step flag");
    if ((flags & MSE_STEP_SANITIZE_LATE) && (h->P.env_kind != MSE_ENV_MONO || !(flags & MSE_STEP_UNMASKED)))
        return fail(MSE_ERR_INVALID_ARGUMENT, "MSE_STEP_SANITIZE_LATE applies to Env_3 with MSE_STEP_UNMASKED only");
    if ((obs_out && !aligned16(obs_out)) || (mask_out && !aligned16(mask_out)))
        return fail(MSE_ERR_ALIGNMENT, "obs_out / mask_out must be 16-byte aligned");
    if (h->trace_rec != nullptr && h->trace_count >= h->trace_capacity)
        return fail(MSE_ERR_INVALID_ARGUMENT, "the trace buffer is full: mse_trace_end, or begin a trace with more capacity");
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
    if (h->trace_rec != nullptr)
        return fail(MSE_ERR_INVALID_ARGUMENT, "a trace is attached (mse_trace_begin): only mse_step records, end it first");
    if (flags & ~(MSE_STEP_UNMASKED | MSE_STEP_CHECK_OVERFLOW | MSE_ROLLOUT_RULE_BASED | MSE_STEP_SANITIZE_LATE))
        return fail(MSE_ERR_INVALID_ARGUMENT, "unknown rollout flag");
    if ((flags & MSE_STEP_SANITIZE_LATE) && (h->P.env_kind != MSE_ENV_MONO || !(flags & MSE_STEP_UNMASKED)))
        return fail(MSE_ERR_INVALID_ARGUMENT, "MSE_STEP_SANITIZE_LATE applies to Env_3 with MSE_STEP_UNMASKED only");
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

int mse_rollout_policy(mse_env *h, mse_policy *pol, mse_policy *sort_pol, int32_t k_steps, uint64_t seed, int deterministic,
                       const int32_t *sort_mode, uint32_t flags, float *obs_out, uint8_t *mask_out, int32_t *actions_out,
                       float *logp_out, float *value_out, float *reward_out, uint8_t *episode_start_out,
                       float *last_value_out, uint8_t *last_done_out, void *stream)
{
    if (!h || !pol) return fail(MSE_ERR_INVALID_ARGUMENT, "env/policy is NULL");
    if (!h->seeded) return fail(MSE_ERR_NOT_RESET, "mse_rollout_policy before mse_reset(seeds)");
    if (k_steps < 1) return fail(MSE_ERR_INVALID_ARGUMENT, "k_steps must be >= 1");
    if (!h->P.auto_reset) return fail(MSE_ERR_INVALID_ARGUMENT, "mse_rollout_policy needs auto_reset=1");
    if (h->trace_rec != nullptr)
        return fail(MSE_ERR_INVALID_ARGUMENT, "a trace is attached (mse_trace_begin): only mse_step records, end it first");
    if (flags & ~(MSE_STEP_UNMASKED | MSE_STEP_CHECK_OVERFLOW))
        return fail(MSE_ERR_INVALID_ARGUMENT, "unknown rollout flag");
    if (pol->d_in != mse_obs_dim(h) || pol->n_act != mse_num_actions(h))
        return fail(MSE_ERR_INVALID_ARGUMENT, "the policy's observation / action dimensions do not match the env kind");
    if (pol->device != h->device) return fail(MSE_ERR_INVALID_ARGUMENT, "policy and env live on different devices");
    if (sort_pol != nullptr) {
        if (h->P.env_kind != MSE_ENV_PRESS || sort_pol->d_in != 13 || sort_pol->n_act != 2)
            return fail(MSE_ERR_INVALID_ARGUMENT, "a sorting policy (13 -> 2) only applies to Env_2_Pressing");
        if (sort_mode != nullptr) return fail(MSE_ERR_INVALID_ARGUMENT, "give sort_mode_dev or a sorting policy, not both");
        if (!pol->use_f16() || !sort_pol->use_f16() || sort_pol->device != h->device)
            return fail(MSE_ERR_UNSUPPORTED_CONFIG, "the in-loop sorting policy exists in the f16x3 form of both networks, on the env's device");
    }
    if ((obs_out && !aligned16(obs_out)) || (mask_out && !aligned16(mask_out)))
        return fail(MSE_ERR_ALIGNMENT, "obs_out / mask_out must be 16-byte aligned");
    if (h->literal || h->P.gen_mode)
        return fail(MSE_ERR_UNSUPPORTED_CONFIG, "mse_rollout_policy serves the integer draw path with a remainder-free batch "
                                                "(literal_choice, input_batch_size > 127 or with a floor() remainder: "
                                                "alternate mse_policy_forward and mse_step)");
    hipStream_t s = static_cast<hipStream_t>(stream);
    int rc;
    switch (h->P.env_kind) {
    case 1: rc = launch_rollout_policy<1>(h, pol, sort_pol, s, k_steps, seed, deterministic, sort_mode, flags, obs_out, mask_out, actions_out, logp_out, value_out, reward_out, episode_start_out, last_value_out, last_done_out); break;
    case 2: rc = launch_rollout_policy<2>(h, pol, sort_pol, s, k_steps, seed, deterministic, sort_mode, flags, obs_out, mask_out, actions_out, logp_out, value_out, reward_out, episode_start_out, last_value_out, last_done_out); break;
    default: rc = launch_rollout_policy<3>(h, pol, sort_pol, s, k_steps, seed, deterministic, sort_mode, flags, obs_out, mask_out, actions_out, logp_out, value_out, reward_out, episode_start_out, last_value_out, last_done_out); break;
    }
    if (rc != MSE_OK) return fail(rc, "the policy rollout kernel's LDS image does not fit this config's tables");
    MSE_CHECK_LAUNCH();
    h->policy_t += (uint64_t)k_steps;
    return MSE_OK;
}

static int sample_actions_impl(mse_env *h, uint32_t pflags, uint64_t policy_seed, int32_t *action_out, void *stream)
{
    if (!h || !action_out) return fail(MSE_ERR_INVALID_ARGUMENT, "env/action_out is NULL");
    if (!h->seeded) return fail(MSE_ERR_NOT_RESET, "policy sampling before mse_reset(seeds)");
    hipStream_t s = static_cast<hipStream_t>(stream);
    switch (h->P.env_kind) {
    case 1: hipLaunchKernelGGL(k_sample<1>, grid_of(h), dim3(kBlock), 0, s, h->P, h->planes, h->tables, pflags, policy_seed, h->policy_t, action_out); break;
    case 2: hipLaunchKernelGGL(k_sample<2>, grid_of(h), dim3(kBlock), 0, s, h->P, h->planes, h->tables, pflags, policy_seed, h->policy_t, action_out); break;
    default: hipLaunchKernelGGL(k_sample<3>, grid_of(h), dim3(kBlock), 0, s, h->P, h->planes, h->tables, pflags, policy_seed, h->policy_t, action_out); break;
    }
    MSE_CHECK_LAUNCH();
    return MSE_OK;
}

int mse_sample_actions(mse_env *h, uint64_t policy_seed, int32_t *action_out, void *stream)
{
    return sample_actions_impl(h, 0u, policy_seed, action_out, stream);
}

int mse_rule_actions(mse_env *h, int32_t *action_out, void *stream)
{
    return sample_actions_impl(h, MSE_ROLLOUT_RULE_BASED, 0, action_out, stream);
}

int mse_model_actions(mse_env *h, uint32_t flags, int32_t *action_out, void *stream)
{
    if (!h || !action_out) return fail(MSE_ERR_INVALID_ARGUMENT, "env/action_out is NULL");
    if (!h->seeded) return fail(MSE_ERR_NOT_RESET, "mse_model_actions before mse_reset(seeds)");
    if (flags & ~(MSE_STEP_UNMASKED | MSE_MODEL_NO_SORT_DRAW | MSE_MODEL_NO_PRESS_DRAW))
        return fail(MSE_ERR_INVALID_ARGUMENT, "unknown flag");
    if (h->P.env_kind != MSE_ENV_MONO)
        return fail(MSE_ERR_INVALID_ARGUMENT, "mode='model' exists on Env_3_Monolith only (env_monolith.py:186)");
    hipLaunchKernelGGL(k_model_actions, grid_of(h), dim3(kBlock), 0, static_cast<hipStream_t>(stream), h->P, h->planes,
                       flags, action_out);
    MSE_CHECK_LAUNCH();
    return MSE_OK;
}

int mse_trace_begin(mse_env *h, int64_t env_index, double *records_dev, int64_t capacity)
{
    if (!h || !records_dev) return fail(MSE_ERR_INVALID_ARGUMENT, "env/records_dev is NULL");
    if (env_index < 0 || env_index >= h->P.n) return fail(MSE_ERR_INVALID_ARGUMENT, "env_index out of range");
    if (capacity < 1) return fail(MSE_ERR_INVALID_ARGUMENT, "capacity must be >= 1");
    h->trace_rec = records_dev;
    h->trace_env = env_index;
    h->trace_capacity = capacity;
    h->trace_count = 0;
    return MSE_OK;
}

int mse_trace_end(mse_env *h, int64_t *n_records_out)
{
    if (!h) return fail(MSE_ERR_INVALID_ARGUMENT, "env is NULL");
    if (n_records_out) *n_records_out = h->trace_rec ? h->trace_count : 0;
    h->trace_rec = nullptr;
    h->trace_env = -1;
    h->trace_capacity = h->trace_count = 0;
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

int mse_sort_agent_obs(mse_env *h, float *obs13_out, void *stream)
{
    if (!h || !obs13_out) return fail(MSE_ERR_INVALID_ARGUMENT, "env/obs13_out is NULL");
    if (!h->seeded) return fail(MSE_ERR_NOT_RESET, "mse_sort_agent_obs before the first seeded mse_reset");
    hipLaunchKernelGGL(k_sort_agent_obs, grid_of(h), dim3(kBlock), 0, static_cast<hipStream_t>(stream), h->P, h->planes,
                       h->tables, obs13_out);
    MSE_CHECK_LAUNCH();
    return MSE_OK;
}

int mse_press_agent_obs(mse_env *h, float *obs16_out, void *stream)
{
    if (!h || !obs16_out) return fail(MSE_ERR_INVALID_ARGUMENT, "env/obs16_out is NULL");
    if (!h->seeded) return fail(MSE_ERR_NOT_RESET, "mse_press_agent_obs before the first seeded mse_reset");
    hipLaunchKernelGGL(k_press_agent_obs, grid_of(h), dim3(kBlock), 0, static_cast<hipStream_t>(stream), h->P, h->planes,
                       h->tables, obs16_out);
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
                       reinterpret_cast<const unsigned long long *>(rng_in), h->err_count);
    MSE_CHECK_LAUNCH();
    if (rng_in) h->seeded = true;
    return MSE_OK;
}

int mse_get_policy_step(const mse_env *h, uint64_t *t_out)
{
    if (!h || !t_out) return fail(MSE_ERR_INVALID_ARGUMENT, "env/t_out is NULL");
    *t_out = h->policy_t;
    return MSE_OK;
}

int mse_set_policy_step(mse_env *h, uint64_t t)
{
    if (!h) return fail(MSE_ERR_INVALID_ARGUMENT, "env is NULL");
    h->policy_t = t;
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
