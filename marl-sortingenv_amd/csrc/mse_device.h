// mse_device.h -- device-side state layout and the per-env state transition (gfx950 / CDNA4).
//
// One env instance per lane.  State lives in HBM as struct-of-arrays PLANES of 16-byte cells:
// plane p, env i  ->  planes[p * n_pad + i]  (uint4), so one wave-instruction moves 64 x 16 B =
// 1 KiB fully coalesced.  See DESIGN.md "Data layout in HBM".
//
// The arithmetic follows the reference (citations = path:line in the reference checkout) and
// numpy 2.2.6's Generator/PCG64/SeedSequence.  All fp64 work is compiled with -ffp-contract=off.
//
// Instruction-count discipline (the kernel is VALU-issue bound at one wave per SIMD, see
// DESIGN.md "Kernel"): every quotient of small integers that the reference evaluates per step
// (levels/700, timers/12, belt shares, purity differences ...) is a lookup in tables that the host
// fills with the reference's literal fp64 expression (mse_lib.hip build_tables), staged in LDS;
// the stage vectors are carried as seasonal-pattern ids; the sort_material draw decides
// Generator.choice with integer compares on the PCG64 output and falls back to the literal fp64
// cdf only within a hair of a tie.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mse_policy_stream.h"

namespace mse {

constexpr int kBlock = 256;             // threads per workgroup = 4 wavefronts of 64
constexpr int kGeneratorPeriod = 20;    // SeasonalInputGenerator default steps_per_pattern
                                        // (utils/input_generator.py:15; reset() uses it: env_super.py:375)

enum Plane : int {
    PL_RNG_STATE = 0,  // u64 lo, u64 hi           rng  (seed+99)  state        R/W every step
    PL_RNG_INC,        // u64 lo, u64 hi           rng  increment                R only
    PL_ACC01,          // f64 acc_belt[0], [1]                                   R/W
    PL_ACC23,          // f64 acc_belt[2], [3]                                   R/W
    PL_CONT_TRUE,      // i32 x4  containers A..D (true)                         R/W
    PL_CONT_FALSE,     // i32 x4  containers A_False..D_False                    R/W
    PL_MISC0,          // i32 E, i32 n_1, i32 n_2, i32 last_press_amount         R/W
    PL_MISC1,          // u8x4 input, u8x4 belt, u8x4 sorting, {timer1,timer2,mat1,mat2}   R/W
    PL_MISC2,          // {q1,q2,mode,flags}, {gen_idx,gen_counter,step u16}, episode, pressing-rng uinteger
    PL_NOISE_STATE,    // rng_noise (seed+4), touched only when noise > 0
    PL_NOISE_INC,
    PL_PRESS_STATE,    // rng_pressing (seed+3), touched only by Env_1
    PL_PRESS_INC,
    PL_BALE0,          // bale ledger summary per material {count, sum, last_size, last_q}; cold
    PL_BALE1, PL_BALE2, PL_BALE3, PL_BALE4,
    PL_SORTRNG_STATE,  // rng_sorting (seed+2): only Env_3's mode='model' fallback draws from it (k_model_actions); cold
    PL_SORTRNG_INC,
    PL_SORTRNG_AUX,    // {uinteger, has_uint32, 0, 0} of rng_sorting's 32-bit buffer
    PL_GEN_STATE,      // the SeasonalInputGenerator's private default_rng(seed) (utils/input_generator.py:28): only a batch
    PL_GEN_INC,        // size whose floor() leaves a remainder makes its draws observable (Params::gen_mode); cold otherwise
    PL_GEN_AUX,        // {uinteger, has_uint32, 0, 0}
    PL_COUNT
};

// flag bits in PL_MISC2.x byte 3
constexpr uint32_t FL_LAST_PRESS_STARTED = 1u;
constexpr uint32_t FL_GEN_FIRST_IS_2 = 2u;
constexpr uint32_t FL_PRESS_HAS_U32 = 4u;

constexpr int kPdiffStride = 102; // purity hundredths 0..100, [101] = empty container
// per-stage-id record in the LDS table image (12 words, 16-byte aligned):
//   [0] packed counts  [1] belt_occupancy f32  [2] sorting_rules() mode  [3] pad
//   [4..7] belt proportions f32   [8..11] sorting[m]/stage_capacity f32
constexpr int kPatStride = 12;

// fp64 constants kept in the LDS table image (Tables::cst)
enum Cst : int { CST_PEN_CAT = 0, CST_PEN_SEV, CST_PEN_MILD, CST_MAX_STATE, CST_OVERFLOW_PEN, CST_REM_THR, CST_BOOST,
                 CST_NOISE, CST_BASE_ACC0, CST_BASE_ACC1, CST_BASE_ACC2, CST_BASE_ACC3, CST_COUNT };

struct Params {
    long long n;            // envs in this handle
    long long n_pad;        // plane stride (multiple of kBlock)
    long long index_offset; // global index of env 0 (sharded runs)
    int env_kind, max_steps, auto_reset, track_bales;
    int balesize, capacity, stage_capacity, batch;
    int press_time[2];
    int press_time0, press_time1; // the same as two scalars: selected per lane with v_cndmask, never indexed
    float inv_balesize;           // 1.0f / bale_standard_size (quotient estimate, fixed up exactly)
    double max_state_reward;      // used every step: stays a kernel argument (SGPR pair)
    // stage vectors are one of three words: id 0 = empty (after reset), 1 / 2 = seasonal pattern
    uint32_t pat_word[3];   // packed u8x4 counts A..D (load/store conversion)
    uint32_t pat_word1, pat_word2; // the same as two scalars: a per-lane choice between them is two v_cndmask on
                                   // SGPRs (indexing the array per lane would be a global load)
    int thr_sev, thr_mild;  // levels above these have fill_ratio > 0.95 / > 0.90 (literal fp64 scan on the host)
    int sev_negative, mild_negative; // overflow_penalty_severe / _mild < 0 (then that bracket returns early)
    int k_thr[4];           // hundredths of python round(quality_threshold, 2): purity of an empty container
    // offsets (in 4-byte words) into the table image; see build_tables
    int off_lvl, off_pdiff, off_timer0, off_timer1, off_tanh, off_eff, off_pat, off_acc, off_bonus, off_ptime, off_cst, off_jump, off_back, table_words;
    uint32_t qi_down[4]; // bit q set: int((q / 100.0) * 100.0) == q - 1  (press_bale's stored quality)
    int rem_thr_units;   // floor(bale_standard_size * bale_remainder_threshold)
    int ring_worst; // most sort_material draws one step can make with this config (k_rollout_ring flow control)
    // the LCG's jump FORWARD by ring_worst steps, s' = A s + G inc (A_lo, A_hi, G_lo, G_hi): the two halves of the
    // ring's priming are that far apart (k_rollout_ring); kernel arguments because the observer lanes need them
    // before the table image is in LDS
    uint64_t ring_fwd[4];
    // General generator mode (utils/input_generator.py:46-61 with a floor() remainder, e.g. input_batch_size 90): the stage
    // vectors are carried as their packed counts instead of pattern ids, the generator's private stream runs on the
    // device (remainder draws + the shuffle's draws), and only the one-lane kernels serve the handle.
    int gen_mode;
    int gen_rem[3];            // units left after the floor()s, per pattern key (index 1 | 2)
    uint32_t occ_nonempty;     // f32 bits of clip(float(round(batch / 100, 2))): occupancy of a stage that holds a batch
    int off_gprop, off_gfrac;  // tables by count: clip(float(k / batch)), clip(float(k / stage_capacity)), k = 0..255
    double acc_floor[4]; // lowest accuracy_belt[m] the config can produce: clip(baseline [+ boost] - noise).  ring_worst
                         // is derived from it, so mse_set_state counts anything below as an error (mse_error_count)
};

// ------------------------------------------------------------------------------------------
// numpy PCG64 (pcg64.h): 128-bit LCG, XSL-RR output of the NEW state
// ------------------------------------------------------------------------------------------
struct Pcg {
    uint64_t s_lo, s_hi, i_lo, i_hi;
};

// 32x32+64 multiply-add on v_mad_u64_u32 (quarter-rate op: the 128-bit LCG step is built from exactly
// six of them plus four v_mul_lo_u32; hipcc's own expansion of the 64-bit C expression used 14).
// The multiplier limb is a wave-uniform constant and goes on the scalar bus.
__device__ __forceinline__ uint64_t mad64(uint32_t a, uint32_t b, uint64_t c)
{
    uint64_t d;
    asm("v_mad_u64_u32 %0, vcc, %1, %2, %3" : "=v"(d) : "v"(a), "s"(b), "v"(c) : "vcc");
    return d;
}
__device__ __forceinline__ uint64_t mul64(uint32_t a, uint32_t b)
{
    uint64_t d;
    asm("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(d) : "v"(a), "s"(b) : "vcc");
    return d;
}
// the same with both factors per-lane values
__device__ __forceinline__ uint64_t mul64_vv(uint32_t a, uint32_t b)
{
    uint64_t d;
    asm("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(d) : "v"(a), "v"(b) : "vcc");
    return d;
}
__device__ __forceinline__ uint32_t mul_lo_s(uint32_t a, uint32_t b)
{
    uint32_t d;
    asm("v_mul_lo_u32 %0, %1, %2" : "=v"(d) : "v"(a), "s"(b));
    return d;
}

// state = state * 0x2360ED051FC65DA44385DF649FCCF645 + inc  (mod 2^128), on 32-bit limbs.
// Carries ride in the 64-bit accumulator of v_mad_u64_u32: limb sums that can exceed 32 bits are
// fed back as the (zero-extended) addend of the next multiply-add.
__device__ __forceinline__ void pcg_advance(Pcg &g)
{
    const uint32_t m0 = 0x9FCCF645u, m1 = 0x4385DF64u, m2 = 0x1FC65DA4u, m3 = 0x2360ED05u;
    const uint32_t s0 = (uint32_t)g.s_lo, s1 = (uint32_t)(g.s_lo >> 32);
    const uint32_t s2 = (uint32_t)g.s_hi, s3 = (uint32_t)(g.s_hi >> 32);
    const uint64_t p0 = mul64(s0, m0);                            // limb 0 | carry
    const uint64_t x = mad64(s0, m1, (uint64_t)(uint32_t)(p0 >> 32)); // <= (2^32-1)^2 + 2^32-1: no overflow
    const uint64_t y = mad64(s1, m0, (uint64_t)(uint32_t)x);          // limb 1 | carry
    uint64_t z = mad64(s0, m2, (uint64_t)(uint32_t)(x >> 32));        // limbs 2..3, everything mod 2^64
    z = mad64(s1, m1, z);
    z = mad64(s2, m0, z);
    z += (uint64_t)(uint32_t)(y >> 32);
    const uint32_t h1 = (uint32_t)(z >> 32) + mul_lo_s(s0, m3) + mul_lo_s(s1, m2) + mul_lo_s(s2, m1) + mul_lo_s(s3, m0);
    uint32_t o0, o1, o2, o3;
    asm("v_add_co_u32 %0, vcc, %4, %8\n\t"
        "v_addc_co_u32 %1, vcc, %5, %9, vcc\n\t"
        "v_addc_co_u32 %2, vcc, %6, %10, vcc\n\t"
        "v_addc_co_u32 %3, vcc, %7, %11, vcc"
        : "=&v"(o0), "=&v"(o1), "=&v"(o2), "=&v"(o3)
        : "v"((uint32_t)p0), "v"((uint32_t)y), "v"((uint32_t)z), "v"(h1), "v"((uint32_t)g.i_lo),
          "v"((uint32_t)(g.i_lo >> 32)), "v"((uint32_t)g.i_hi), "v"((uint32_t)(g.i_hi >> 32))
        : "vcc");
    g.s_lo = (uint64_t)o0 | ((uint64_t)o1 << 32);
    g.s_hi = (uint64_t)o2 | ((uint64_t)o3 << 32);
}

__device__ __forceinline__ uint64_t pcg_output(const Pcg &g)
{
    uint64_t x = g.s_hi ^ g.s_lo;
    unsigned rot = (unsigned)(g.s_hi >> 58);
    return (x >> rot) | (x << ((64u - rot) & 63u));
}

// upper 32 bits of the XSL-RR output only (all the integer draw decision needs)
__device__ __forceinline__ uint32_t pcg_output_hi32(const Pcg &g)
{
    uint32_t x0 = (uint32_t)g.s_lo ^ (uint32_t)g.s_hi;
    uint32_t x1 = (uint32_t)(g.s_lo >> 32) ^ (uint32_t)(g.s_hi >> 32);
    uint32_t rot = (uint32_t)(g.s_hi >> 58);
    bool swap = (rot & 32u) != 0u;
    uint32_t a = swap ? x1 : x0, b = swap ? x0 : x1;
    return __builtin_amdgcn_alignbit(a, b, rot & 31u); // ({a,b} >> rot)[31:0]
}

__device__ __forceinline__ uint64_t pcg_next64(Pcg &g)
{
    pcg_advance(g);
    return pcg_output(g);
}

// numpy distributions.h next_double
__device__ __forceinline__ double u64_to_unit_double(uint64_t r)
{
    return (double)(r >> 11) * (1.0 / 9007199254740992.0);
}

// numpy _seed_seq.pyx SeedSequence(entropy < 2**64).generate_state(4, uint64), pool size 4
__device__ inline void seed_sequence(uint64_t entropy, uint64_t out[4])
{
    const uint32_t MULT_A = 0x931e8875u, MULT_B = 0x58f38dedu, MIX_L = 0xca01f9ddu, MIX_R = 0x4973f715u;
    uint32_t w0 = (uint32_t)entropy, w1 = (uint32_t)(entropy >> 32);
    uint32_t pool[4];
    uint32_t hc = 0x43b0d7e5u;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        uint32_t v = i == 0 ? w0 : (i == 1 ? w1 : 0u); // a zero high word hashes like padding
        v ^= hc;
        hc *= MULT_A;
        v *= hc;
        v ^= v >> 16;
        pool[i] = v;
    }
#pragma unroll
    for (int src = 0; src < 4; ++src) {
#pragma unroll
        for (int dst = 0; dst < 4; ++dst) {
            if (src != dst) {
                uint32_t v = pool[src];
                v ^= hc;
                hc *= MULT_A;
                v *= hc;
                v ^= v >> 16;
                uint32_t r = MIX_L * pool[dst] - MIX_R * v;
                r ^= r >> 16;
                pool[dst] = r;
            }
        }
    }
    uint32_t hb = 0x8b51f9ddu;
    uint32_t st[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        uint32_t v = pool[i & 3];
        v ^= hb;
        hb *= MULT_B;
        v *= hb;
        v ^= v >> 16;
        st[i] = v;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) out[i] = (uint64_t)st[2 * i] | ((uint64_t)st[2 * i + 1] << 32);
}

// np.random.default_rng(seed): pcg64_set_seed + pcg_setseq_128_srandom_r
__device__ inline Pcg pcg_seed(uint64_t seed)
{
    uint64_t w[4];
    seed_sequence(seed, w);
    Pcg g;
    // inc = (initseq << 1) | 1, initseq = (w[2] << 64) | w[3]
    g.i_hi = (w[2] << 1) | (w[3] >> 63);
    g.i_lo = (w[3] << 1) | 1ull;
    g.s_lo = 0;
    g.s_hi = 0;
    pcg_advance(g);
    // state += initstate, initstate = (w[0] << 64) | w[1]
    uint64_t lo = g.s_lo + w[1];
    g.s_hi = g.s_hi + w[0] + (lo < g.s_lo ? 1ull : 0ull);
    g.s_lo = lo;
    pcg_advance(g);
    return g;
}

// splitmix64 finaliser: the build's unseeded-reset rule and the random policy stream
__device__ __forceinline__ uint64_t mix64(uint64_t z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// index of the k-th (0-based) set bit of `bits`
// (masks here have at most 22 bits.)  Two branch-free halvings first - at bit 11, then at bit 6 - so that the
// clear-lowest-bit loop, whose trip count a wave pays as the maximum over its lanes, runs at most 5 times.
__device__ __forceinline__ int select_kth_bit(uint32_t bits, int k)
{
    uint32_t lo = bits & 0x7FFu;
    int c = __popc(lo);
    bool up = k >= c;
    k = up ? k - c : k;
    bits = up ? bits >> 11 : lo;
    int base = up ? 11 : 0;
    lo = bits & 0x3Fu;
    c = __popc(lo);
    up = k >= c;
    k = up ? k - c : k;
    bits = up ? bits >> 6 : lo;
    base += up ? 6 : 0;
    for (int i = 0; i < k; ++i) bits &= bits - 1u;
    return base + __ffs((int)bits) - 1;
}

// ------------------------------------------------------------------------------------------
// per-env register image
// ------------------------------------------------------------------------------------------
// Diagnostic build only (-DMSE_TIMELINE, tools/timeline.py): s_memtime deltas per section and role, summed
// into a device symbol by lane 0 of every wave.  The shipped library compiles none of this.
#ifdef MSE_TIMELINE
__device__ unsigned long long g_timeline[4 * 8]; // [3]: launch edges seen by the dynamics wave
struct Timeline {
    unsigned long long last;
    unsigned long long acc[8];
    __device__ __forceinline__ void start()
    {
        for (int k = 0; k < 8; ++k) acc[k] = 0;
        last = __builtin_readcyclecounter();
    }
    __device__ __forceinline__ void mark(int k)
    {
        if (blockIdx.x != 0) return; // s_memtime from every wave of the chip serialises: sample one workgroup
        const unsigned long long now = __builtin_readcyclecounter();
        acc[k] += now - last;
        last = now;
    }
    __device__ __forceinline__ void flush(int role) const
    {
        if ((threadIdx.x & 63) == 0 && blockIdx.x == 0)
            for (int k = 0; k < 8; ++k) atomicAdd(&g_timeline[role * 8 + k], acc[k]);
    }
};
// MSE_TIMELINE = 1: every section; 2: only the barrier waits (two reads per step, little disturbance)
#if MSE_TIMELINE == 2
#define MSE_TL(tl, k) ((void)0)
#define MSE_TLB(tl, k) (tl).mark(k)
#else
#define MSE_TL(tl, k) (tl).mark(k)
#define MSE_TLB(tl, k) (tl).mark(k)
#endif
#else
#define MSE_TLB(tl, k) ((void)0)
#define MSE_TL(tl, k) ((void)0)
#endif

struct Env {
#ifdef MSE_TIMELINE
    Timeline tl;
#endif
    Pcg rng;            // seed+99: sort_material
    Pcg noise;          // seed+4 : update_accuracy (noise > 0)
    Pcg press;          // seed+3 : Env_1's internal press sampling
    uint32_t press_uint;
    int press_has;
    Pcg gen;            // the input generator's private stream (general generator mode only)
    uint32_t gen_uint;
    int gen_has;
    double acc[4];      // accuracy_belt
    int ct[4], cf[4], ce;
    int pn[2], lpa;
    int st_in, st_belt, st_sort; // stage ids: 0 empty, 1 / 2 seasonal pattern (general generator mode: the packed counts)
    int timer[2], pmat[2], q100[2];
    int mode, lps, gen2, gen_idx, gen_cnt, step;
    uint32_t episode;
};

__device__ __forceinline__ int stage_id(uint32_t word, const Params &P)
{
    if (P.gen_mode) return (int)word; // general generator: the register image is the packed counts themselves
    return word == 0u ? 0 : (word == P.pat_word[1] ? 1 : 2);
}
__device__ __forceinline__ uint32_t stage_word(int id, const Params &P)
{
    if (P.gen_mode) return (uint32_t)id;
    return id == 0 ? P.pat_word[0] : (id == 1 ? P.pat_word[1] : P.pat_word[2]);
}

// The state planes of one env as loaded, before unpacking: a kernel that has something to do while the loads
// fly (k_rollout_ring copies the table image) issues load_env_raw first and unpacks afterwards.
struct EnvRaw {
    uint4 a, b, c0, c1, t, f, m0, m1, m2, ns, nq, ps, pq;
};

template <int KIND, bool NOISE>
__device__ __forceinline__ void load_env_raw(EnvRaw &r, const uint4 *__restrict__ planes, const Params &P, long long i)
{
    const long long n_pad = P.n_pad;
    r.a = planes[PL_RNG_STATE * n_pad + i];
    r.b = planes[PL_RNG_INC * n_pad + i];
    r.c0 = planes[PL_ACC01 * n_pad + i];
    r.c1 = planes[PL_ACC23 * n_pad + i];
    r.t = planes[PL_CONT_TRUE * n_pad + i];
    r.f = planes[PL_CONT_FALSE * n_pad + i];
    r.m0 = planes[PL_MISC0 * n_pad + i];
    r.m1 = planes[PL_MISC1 * n_pad + i];
    r.m2 = planes[PL_MISC2 * n_pad + i];
    if (NOISE) {
        r.ns = planes[PL_NOISE_STATE * n_pad + i];
        r.nq = planes[PL_NOISE_INC * n_pad + i];
    }
    if (KIND == 1) {
        r.ps = planes[PL_PRESS_STATE * n_pad + i];
        r.pq = planes[PL_PRESS_INC * n_pad + i];
    }
}

template <int KIND, bool NOISE>
__device__ __forceinline__ void unpack_env(Env &e, const EnvRaw &r, const Params &P)
{
    const uint4 a = r.a, b = r.b, c0 = r.c0, c1 = r.c1, t = r.t, f = r.f, m0 = r.m0, m1 = r.m1, m2 = r.m2;
    e.rng.s_lo = (uint64_t)a.x | ((uint64_t)a.y << 32);
    e.rng.s_hi = (uint64_t)a.z | ((uint64_t)a.w << 32);
    e.rng.i_lo = (uint64_t)b.x | ((uint64_t)b.y << 32);
    e.rng.i_hi = (uint64_t)b.z | ((uint64_t)b.w << 32);
    e.acc[0] = __hiloint2double((int)c0.y, (int)c0.x);
    e.acc[1] = __hiloint2double((int)c0.w, (int)c0.z);
    e.acc[2] = __hiloint2double((int)c1.y, (int)c1.x);
    e.acc[3] = __hiloint2double((int)c1.w, (int)c1.z);
    e.ct[0] = (int)t.x; e.ct[1] = (int)t.y; e.ct[2] = (int)t.z; e.ct[3] = (int)t.w;
    e.cf[0] = (int)f.x; e.cf[1] = (int)f.y; e.cf[2] = (int)f.z; e.cf[3] = (int)f.w;
    e.ce = (int)m0.x; e.pn[0] = (int)m0.y; e.pn[1] = (int)m0.z; e.lpa = (int)m0.w;
    e.st_in = stage_id(m1.x, P);
    e.st_belt = stage_id(m1.y, P);
    e.st_sort = stage_id(m1.z, P);
    e.timer[0] = (int)(m1.w & 0xFFu);
    e.timer[1] = (int)((m1.w >> 8) & 0xFFu);
    e.pmat[0] = (int)((m1.w >> 16) & 0xFFu);
    e.pmat[1] = (int)(m1.w >> 24);
    e.q100[0] = (int)(m2.x & 0xFFu);
    e.q100[1] = (int)((m2.x >> 8) & 0xFFu);
    e.mode = (int)((m2.x >> 16) & 0xFFu);
    uint32_t fl = m2.x >> 24;
    e.lps = (fl & FL_LAST_PRESS_STARTED) ? 1 : 0;
    e.gen2 = (fl & FL_GEN_FIRST_IS_2) ? 1 : 0;
    e.press_has = (fl & FL_PRESS_HAS_U32) ? 1 : 0;
    e.gen_idx = (int)(m2.y & 0xFFu);
    e.gen_cnt = (int)((m2.y >> 8) & 0xFFu);
    e.step = (int)(m2.y >> 16);
    e.episode = m2.z;
    e.press_uint = m2.w;
    if (NOISE) {
        const uint4 s = r.ns, q = r.nq;
        e.noise.s_lo = (uint64_t)s.x | ((uint64_t)s.y << 32);
        e.noise.s_hi = (uint64_t)s.z | ((uint64_t)s.w << 32);
        e.noise.i_lo = (uint64_t)q.x | ((uint64_t)q.y << 32);
        e.noise.i_hi = (uint64_t)q.z | ((uint64_t)q.w << 32);
    }
    if (KIND == 1) {
        const uint4 s = r.ps, q = r.pq;
        e.press.s_lo = (uint64_t)s.x | ((uint64_t)s.y << 32);
        e.press.s_hi = (uint64_t)s.z | ((uint64_t)s.w << 32);
        e.press.i_lo = (uint64_t)q.x | ((uint64_t)q.y << 32);
        e.press.i_hi = (uint64_t)q.z | ((uint64_t)q.w << 32);
    }
}

template <int KIND, bool NOISE>
__device__ __forceinline__ void load_env(Env &e, const uint4 *__restrict__ planes, const Params &P, long long i)
{
    EnvRaw r;
    load_env_raw<KIND, NOISE>(r, planes, P, i);
    unpack_env<KIND, NOISE>(e, r, P);
}

__device__ __forceinline__ uint4 pack_u64x2(uint64_t lo, uint64_t hi)
{
    return make_uint4((uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)hi, (uint32_t)(hi >> 32));
}

__device__ __forceinline__ uint4 pack_f64x2(double a, double b)
{
    return make_uint4((uint32_t)__double2loint(a), (uint32_t)__double2hiint(a),
                      (uint32_t)__double2loint(b), (uint32_t)__double2hiint(b));
}

// write_inc: the increments only change on a seeded reset
template <int KIND, bool NOISE>
__device__ __forceinline__ void store_env(const Env &e, uint4 *__restrict__ planes, const Params &P, long long i,
                                          bool write_inc, bool write_rng_state = true)
{
    const long long n_pad = P.n_pad;
    if (write_rng_state) planes[PL_RNG_STATE * n_pad + i] = pack_u64x2(e.rng.s_lo, e.rng.s_hi);
    if (write_inc) planes[PL_RNG_INC * n_pad + i] = pack_u64x2(e.rng.i_lo, e.rng.i_hi);
    planes[PL_ACC01 * n_pad + i] = pack_f64x2(e.acc[0], e.acc[1]);
    planes[PL_ACC23 * n_pad + i] = pack_f64x2(e.acc[2], e.acc[3]);
    planes[PL_CONT_TRUE * n_pad + i] = make_uint4((uint32_t)e.ct[0], (uint32_t)e.ct[1], (uint32_t)e.ct[2], (uint32_t)e.ct[3]);
    planes[PL_CONT_FALSE * n_pad + i] = make_uint4((uint32_t)e.cf[0], (uint32_t)e.cf[1], (uint32_t)e.cf[2], (uint32_t)e.cf[3]);
    planes[PL_MISC0 * n_pad + i] = make_uint4((uint32_t)e.ce, (uint32_t)e.pn[0], (uint32_t)e.pn[1], (uint32_t)e.lpa);
    uint32_t tw = (uint32_t)e.timer[0] | ((uint32_t)e.timer[1] << 8) | ((uint32_t)(e.pmat[0] & 0xFF) << 16) |
                  ((uint32_t)(e.pmat[1] & 0xFF) << 24);
    planes[PL_MISC1 * n_pad + i] = make_uint4(stage_word(e.st_in, P), stage_word(e.st_belt, P), stage_word(e.st_sort, P), tw);
    uint32_t fl = (e.lps ? FL_LAST_PRESS_STARTED : 0u) | (e.gen2 ? FL_GEN_FIRST_IS_2 : 0u) |
                  (e.press_has ? FL_PRESS_HAS_U32 : 0u);
    uint32_t x = (uint32_t)e.q100[0] | ((uint32_t)e.q100[1] << 8) | ((uint32_t)e.mode << 16) | (fl << 24);
    uint32_t y = (uint32_t)e.gen_idx | ((uint32_t)e.gen_cnt << 8) | ((uint32_t)e.step << 16);
    planes[PL_MISC2 * n_pad + i] = make_uint4(x, y, e.episode, e.press_uint);
    if (NOISE) {
        planes[PL_NOISE_STATE * n_pad + i] = pack_u64x2(e.noise.s_lo, e.noise.s_hi);
        if (write_inc) planes[PL_NOISE_INC * n_pad + i] = pack_u64x2(e.noise.i_lo, e.noise.i_hi);
    }
    if (KIND == 1) {
        planes[PL_PRESS_STATE * n_pad + i] = pack_u64x2(e.press.s_lo, e.press.s_hi);
        if (write_inc) planes[PL_PRESS_INC * n_pad + i] = pack_u64x2(e.press.i_lo, e.press.i_hi);
    }
}

// ------------------------------------------------------------------------------------------
// pieces of the transition
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ int level_of(const Env &e, int m) { return m < 4 ? e.ct[m] + e.cf[m] : e.ce; }

// env_super.py:869-885 press_action_masks as an 11-bit word (bit a = action a valid)
__device__ __forceinline__ uint32_t press_mask_bits(const Env &e, const Params &P)
{
    uint32_t full = 0;
#pragma unroll
    for (int m = 0; m < 5; ++m) full |= (level_of(e, m) >= P.balesize ? 1u : 0u) << m;
    uint32_t bits = 1u;
    if (e.timer[0] == 0) bits |= full << 1;
    if (e.timer[1] == 0) bits |= full << 6;
    return bits;
}

// action_masks(): env_1_sort.py:74-76, env_2_press.py:66-67, env_super.py:887-898
template <int KIND>
__device__ __forceinline__ uint32_t action_mask_bits(const Env &e, const Params &P)
{
    if (KIND == 1) return 3u;
    uint32_t m = press_mask_bits(e, P);
    return KIND == 2 ? m : (m | (m << 11));
}

// The reference's rule-based policy (env_monolith.py:166-184): sorting_rules() on the belt as it will be after
// this step's flow (env_super.py:469-482) and check_container_level() (env_super.py:689-720): the first free
// press takes the fullest non-empty container (strict >, order A..E).  `rule_mode_next` = sorting_rules() of
// the batch now in the input stage (it is on the belt when the step decides).
template <int KIND>
__device__ __forceinline__ int rule_based_action(const Env &e, int rule_mode_next)
{
    if (KIND == 1) return rule_mode_next;
    int press = 0; // 1 | 2, 0 = none free
    if (e.timer[0] == 0) press = 1;
    else if (e.timer[1] == 0) press = 2;
    int best_idx = -1, best_level = 0;
#pragma unroll
    for (int m = 0; m < 5; ++m) {
        const int lvl = m < 4 ? e.ct[m] + e.cf[m] : e.ce;
        if (lvl > best_level) {
            best_level = lvl;
            best_idx = m;
        }
    }
    const int press_action = (press != 0 && best_idx >= 0) ? (press - 1) * 5 + best_idx + 1 : 0;
    return KIND == 2 ? press_action : rule_mode_next * 11 + press_action;
}

// press action a in 1..10 -> (press index 0|1, material 0..4)  (env_super.py:804-809)
__device__ __forceinline__ void decode_press_action(int a, int &press, int &mat)
{
    press = a > 5 ? 1 : 0;
    mat = a - 1 - 5 * press;
}

// env_super.py:811-836 validate_press_action
__device__ __forceinline__ bool press_action_valid(const Env &e, const Params &P, int a)
{
    if (a == 0) return true;
    int press, mat;
    decode_press_action(a, press, mat);
    int tm = press ? e.timer[1] : e.timer[0];
    if (tm > 0) return false;
    int lvl = e.ce;
#pragma unroll
    for (int m = 0; m < 4; ++m) lvl = (mat == m) ? e.ct[m] + e.cf[m] : lvl;
    return lvl >= P.balesize;
}

// numpy's Generator draws on 32-bit words: next_uint32 hands out the low half of a PCG64 output and buffers the high
// half (pcg64.h pcg64_next32); choice(n) / choice(arr) / integers without p are Lemire's bounded draw on it
// (distributions.c buffered_bounded_lemire_uint32), no draw when n == 1.
struct Rng32 {
    Pcg g;
    uint32_t uinteger;
    int has;
    __device__ __forceinline__ uint32_t next32()
    {
        if (has) {
            has = 0;
            return uinteger;
        }
        const uint64_t r = pcg_next64(g);
        has = 1;
        uinteger = (uint32_t)(r >> 32);
        return (uint32_t)r;
    }
    __device__ __forceinline__ uint32_t lemire(uint32_t n) // uniform in [0, n), n >= 1
    {
        if (n <= 1u) return 0u;
        uint64_t m = (uint64_t)next32() * n;
        uint32_t leftover = (uint32_t)m;
        if (leftover < n) {
            const uint32_t threshold = (0xFFFFFFFFu - (n - 1u)) % n;
            while (leftover < threshold) {
                m = (uint64_t)next32() * n;
                leftover = (uint32_t)m;
            }
        }
        return (uint32_t)(m >> 32);
    }
};

// SeasonalInputGenerator.generate_input (utils/input_generator.py:37-64) followed by env_super.py:433-461
// update_environment.
//   GEN = false: with a remainder-free batch the material counts are a function of the pattern alone, the stage
//                vectors travel as pattern ids and the generator's private stream is never observed.
//   GEN = true:  floor(ratio * batch) leaves units over; each goes to rng.choice(material_names) (:49-55), and the
//                batch list is then shuffled (:58-61: Generator.shuffle of a Python list = one random_interval(i) per
//                i = n-1 .. 1, a masked rejection draw on next_uint32) - the shuffle's result is never read, but its
//                draws move the stream the next step's remainder choices come from, so they are made here, one by one.
template <bool GEN>
__device__ __forceinline__ void update_environment(Env &e, const Params &P)
{
    e.st_sort = e.st_belt;
    e.st_belt = e.st_in;
    if (e.gen_cnt >= kGeneratorPeriod) {
        e.gen_idx ^= 1;
        e.gen_cnt = 0;
    }
    const int key = 1 + (e.gen_idx ^ e.gen2);
    if (!GEN) {
        e.st_in = key;
    } else {
        Rng32 g{e.gen, e.gen_uint, e.gen_has};
        uint32_t word = key == 1 ? P.pat_word1 : P.pat_word2; // the floor()ed counts
        const int rem = key == 1 ? P.gen_rem[1] : P.gen_rem[2];
        for (int r = 0; r < rem; ++r) word += 1u << (8u * g.lemire(4u));
        int i = P.batch - 1; // random_interval(i): value = next_uint32() & mask(i) until value <= i
        while (i > 0) {
            const uint32_t mask = 0xFFFFFFFFu >> __builtin_clz((unsigned)i);
            if ((g.next32() & mask) <= (uint32_t)i) --i;
        }
        e.gen = g.g;
        e.gen_uint = g.uinteger;
        e.gen_has = g.has;
        e.st_in = (int)word;
    }
    e.gen_cnt += 1;
}

// env_super.py:484-509 set_multisensor_mode + update_accuracy; acc_sorter gets the OLD accuracy_belt
template <bool NOISE>
__device__ __forceinline__ void update_accuracy(Env &e, const double *cst, const double *acc_table, int mode,
                                                double acc_sorter[4])
{
#pragma unroll
    for (int m = 0; m < 4; ++m) acc_sorter[m] = e.acc[m];
    e.mode = mode;
    if (NOISE) {
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            bool boosted = (m == 0 || m == 2) ? (mode == 0) : (mode == 1);
            double a = cst[CST_BASE_ACC0 + m];
            if (boosted) a = a + cst[CST_BOOST];
            // Generator.uniform(-n, n): low + (high-low)*random(), separately rounded
            const double noise = cst[CST_NOISE];
            double range = noise - (-noise);
            double u = u64_to_unit_double(pcg_next64(e.noise));
            double scaled = range * u;
            double v = a + ((-noise) + scaled);
            e.acc[m] = v < 0.0 ? 0.0 : (v > 1.0 ? 1.0 : v);
        }
    } else {
        // noise == 0: clip(acc + 0) by mode from the table (modes other than 0/1 boost nothing: row 2);
        // the reference still advances rng_noise here, unobservably
        const double *row = acc_table + 4 * (mode == 0 ? 0 : (mode == 1 ? 1 : 2));
#pragma unroll
        for (int m = 0; m < 4; ++m) e.acc[m] = row[m];
    }
}

// Generator.choice(4, p=leftover/total), literal fp64 evaluation (numpy _generator.pyx):
// cdf = cumsum(p); cdf /= cdf[-1]; idx = searchsorted(cdf, u, 'right').   C = byte prefix sums of leftover.
__device__ __forceinline__ int choice4_literal(uint32_t C, uint64_t r64)
{
    const uint32_t L = C - (C << 8); // bytes l_0..l_3 (prefix sums are monotone: no borrows)
    int l0 = (int)(L & 0xFFu), l1 = (int)((L >> 8) & 0xFFu), l2 = (int)((L >> 16) & 0xFFu), l3 = (int)(L >> 24);
    double T = (double)(C >> 24);
    double p0 = (double)l0 / T, p1 = (double)l1 / T, p2 = (double)l2 / T, p3 = (double)l3 / T;
    double c0 = p0;
    double c1 = c0 + p1;
    double c2 = c1 + p2;
    double c3 = c2 + p3;
    double n0 = c0 / c3, n1 = c1 / c3, n2 = c2 / c3, n3 = c3 / c3;
    double u = u64_to_unit_double(r64);
    return (n0 <= u ? 1 : 0) + (n1 <= u ? 1 : 0) + (n2 <= u ? 1 : 0) + (n3 <= u ? 1 : 0);
}

// ------------------------------------------------------------------------------------------
// where sort_material's PCG64 outputs come from
// ------------------------------------------------------------------------------------------
// 128-bit LCG jump-ahead: n steps of s' = M s + inc are s_n = A_n s + G_n inc with A_n = M^n and
// G_n = 1 + M + ... + M^(n-1); the host tabulates (A, G) for n = 2^j (build_tables) and a jump multiplies
// the set bits of n together.  Used only off the hot path (final stream state of a ring rollout, and the
// full 64-bit output of a draw that needs the literal cdf).
constexpr int kJumpBits = 24;
__device__ __forceinline__ void mul128(uint64_t a_lo, uint64_t a_hi, uint64_t b_lo, uint64_t b_hi, uint64_t &r_lo,
                                       uint64_t &r_hi)
{
    r_lo = a_lo * b_lo;
    r_hi = __umul64hi(a_lo, b_lo) + a_lo * b_hi + a_hi * b_lo;
}
__device__ __forceinline__ void pcg_jump(Pcg &g, uint32_t n, const uint64_t *jump_tab /* [kJumpBits][4] = A_lo A_hi G_lo G_hi */)
{
    for (int j = 0; n != 0 && j < kJumpBits; ++j, n >>= 1) {
        if (n & 1u) {
            const uint64_t *t = jump_tab + 4 * j;
            uint64_t x_lo, x_hi, y_lo, y_hi;
            mul128(t[0], t[1], g.s_lo, g.s_hi, x_lo, x_hi);
            mul128(t[2], t[3], g.i_lo, g.i_hi, y_lo, y_hi);
            const uint64_t lo = x_lo + y_lo;
            g.s_hi = x_hi + y_hi + (lo < x_lo ? 1ull : 0ull);
            g.s_lo = lo;
        }
    }
}

// s_{n-d} = A_{-d} s_n + G_{-d} inc  with  A_{-d} = M^{-d},  G_{-d} = -M^{-d} (1 + M + ... + M^{d-1})  (mod 2^128):
// one table entry per d = 0..32, so going back d <= 32 steps is two 128-bit multiplies whatever d is.  The RNG waves
// of k_rollout_ring end a launch at most 64 (usually < `worst`) outputs ahead of what the env consumed and hand the
// stream back this way.
constexpr int kRingBackSteps = 33;
// s' = A s + G inc (mod 2^128): any number of LCG steps, forward or back, in two 128-bit multiplies
__device__ __forceinline__ void pcg_affine(Pcg &g, uint64_t a_lo, uint64_t a_hi, uint64_t g_lo, uint64_t g_hi)
{
    uint64_t x_lo, x_hi, y_lo, y_hi;
    mul128(a_lo, a_hi, g.s_lo, g.s_hi, x_lo, x_hi);
    mul128(g_lo, g_hi, g.i_lo, g.i_hi, y_lo, y_hi);
    const uint64_t lo = x_lo + y_lo;
    g.s_hi = x_hi + y_hi + (lo < x_lo ? 1ull : 0ull);
    g.s_lo = lo;
}
__device__ __forceinline__ void pcg_step_back(Pcg &g, uint32_t d, const uint64_t *back_tab)
{
    const uint64_t *t = back_tab + 4 * d; // d <= 32
    pcg_affine(g, t[0], t[1], t[2], t[3]);
}

// the lane advances the env's own generator
struct RngLocal {
    Pcg &g;
    __device__ __forceinline__ void begin_step() {}
    __device__ __forceinline__ uint32_t next_hi32()
    {
        pcg_advance(g);
        return pcg_output_hi32(g);
    }
    __device__ __forceinline__ uint64_t last_full() const { return pcg_output(g); }
};

// Outputs produced ahead by an RNG wave into an LDS ring of the upper 32 bits (k_rollout_ring).
// Output #p (p = 0, 1, ...) is the output of the state p+1 steps after `start`.  Layout [64 outputs][256 env
// slots]: output #p of env slot el sits at row p & 63, column el (a wave's lanes hit 64 different banks
// whatever their positions).
constexpr int kRingDepth = 64;        // outputs per env
constexpr int kRingMaxPerStep = 31;   // the producer lags one barrier: two steps' draws must fit in the ring
typedef __attribute__((address_space(3))) const uint32_t *lds_u32_ptr;
struct RngRing {
    uint32_t lane_addr;        // LDS byte address of the env slot's column; the ring itself is 64 KiB aligned
    const uint64_t *jump_tab;
    Pcg start;                 // stream state when the launch began (never advanced)
    uint32_t p10;              // outputs consumed so far (= index of the next one), times 1024
    uint32_t nxt, nxt2;        // prefetched outputs #pos and #pos+1
    uint32_t f_min, f_max;     // extremes of the step's fraction views f (near-tie monitor, see draw_units_ring)
#ifdef MSE_TIMELINE
    Timeline *tl;
#endif
    __device__ __forceinline__ uint32_t pos() const { return p10 >> 10; }
    // row (pos & 63) of the column: one v_and_or_b32
    __device__ __forceinline__ uint32_t load(uint32_t q10) const
    {
        return *(lds_u32_ptr)(uintptr_t)((q10 & 0xFC00u) | lane_addr);
    }
    __device__ __forceinline__ void begin_step()
    {
        nxt = load(p10); // after the step's barrier: everything this step can consume is in the ring
        nxt2 = load(p10 + 1024u);
        f_min = 0xFFFFFFFFu;
        f_max = 0u;
    }
};

// a generator positioned by jump-ahead that counts what it hands out (literal redo of a ring step)
struct RngCounted {
    Pcg g;
    uint32_t count;
    __device__ __forceinline__ void begin_step() {}
    __device__ __forceinline__ uint32_t next_hi32()
    {
        pcg_advance(g);
        ++count;
        return pcg_output_hi32(g);
    }
    __device__ __forceinline__ uint64_t last_full() const { return pcg_output(g); }
};

// env_super.py:511-609 sort_material.
//
// leftover[4] is carried as the four byte-wise PREFIX SUMS C = {c_0, c_1, c_2, c_3 = T} in one register
// (every count <= batch <= 127 on this path, so bytes never carry or borrow).
//
// One draw (env_super.py:553-571) = one PCG64 output r.  With U = r >> 11 (u = U/2^53) numpy's
// `cdf_k <= u` equals `c_k <= floor(u*T)` unless u*T is within ~2^-41 of an integer (the cdf carries
// <= 2^-49 of rounding); v = floor(u*T) and a 32-bit view f of the fraction come from (r >> 32) * T.
// Byte k of D = (0x80 + v) - c_k keeps bit 7 iff c_k <= v; the chosen bin is the first k with c_k > v and
// removing one unit there lowers every prefix sum from k on by one: C += (flags >> 7) - 0x01010101.
// If f is within a margin of 0 or 2^32 the literal fp64 path decides instead (DESIGN.md "choice").
//
// How wide the margin must be.  r >> 11 = r_hi 2^21 + eps with r_hi = r >> 32 and eps < 2^21 the next 21 bits, so in
// units of 2^-32:  u T = r_hi T + delta,  delta = eps T / 2^21 in [0, T), T <= 127:  floor(u T) = v unless
// f + delta >= 2^32, which needs f >= 2^32 - 127.  The fp64 cdf is within 2^-49 of c_k / T (four divisions l_k / T, up to
// three additions, the division by cdf[-1]: a dozen roundings of values <= 1, each <= 2^-53), i.e. within 2^-10 of these
// units after the scaling by T < 2^7: it can only disagree with the exact comparison when f + delta < 2^-10 (so f = 0) or
// f + delta > 2^32 - 2^-10 (so f >= 2^32 - 128).
// A draw is "near a tie" when f + MSE_TIE_HI < MSE_TIE_WINDOW (32-bit wrap-around): f < MSE_TIE_WINDOW - MSE_TIE_HI
// (= 2 as shipped: f = 0 with a margin of one) or f >= 2^32 - MSE_TIE_HI (= 160: the 128 with a margin of 32), once in
// 2.6 x 10^7 draws.  Round 2 shipped 16 / 512 (once in 8 x 10^6): at 65 536 envs a 20-step launch met ~2 such draws, and
// the workgroup that redoes a step held the whole launch back by 6 us (tools/clock_probe.py: 9 % of a 20-step launch).
// libmse_hip_widetie.so (tests only) is the same source with MSE_TIE_WINDOW = 2^27, so that the literal branches run on
// a few percent of the draws and are held to the same golden vectors.
#ifndef MSE_TIE_HI
#define MSE_TIE_HI 160u
#endif
#ifndef MSE_TIE_WINDOW
#define MSE_TIE_WINDOW (MSE_TIE_HI + 2u)
#endif

template <bool LITERAL, class RNG>
__device__ __forceinline__ void draw_units(RNG &rng, uint32_t &C, int &rem)
{
    // env_super.py:557-559 breaks out when the pool is empty.  That cannot happen: a station starts its
    // draws with its own false units in the pool (T >= rem) and every draw lowers both T and rem by one.
    while (rem > 0) {
        const uint32_t T = C >> 24;
        const uint32_t r_hi = rng.next_hi32();
        uint32_t flags = 0; // bit 7 of byte k set iff bin k is passed over (c_k <= v)
        bool literal = LITERAL;
        if (!LITERAL) {
            const uint64_t prod = mul64_vv(r_hi, T);
            const uint32_t f = (uint32_t)prod;
            const uint32_t v = (uint32_t)(prod >> 32);
            flags = ((__umul24(v, 0x010101u) | 0x00808080u) - C) & 0x00808080u; // bytes 0..2 only; v < 128
            literal = (f + MSE_TIE_HI) < MSE_TIE_WINDOW;
        }
        if (__builtin_expect(literal, 0)) { // ~1e-7 per draw: keep it out of the loop's straight line
            const int sel = choice4_literal(C, rng.last_full());
            flags = sel == 0 ? 0u : (sel == 1 ? 0x00000080u : (sel == 2 ? 0x00008080u : 0x00808080u));
        }
        C += (flags >> 7) + 0xFEFEFEFFu; // - 0x01010101 on the bins from the chosen one on
        --rem;
    }
}

// The draw loop when the outputs come from the LDS ring.
//
// The dynamics wave is one wave on its SIMD that matters (the observer and RNG waves have slack), and a single
// wave issues one instruction per ~5 cycles at best, ~9 if it depends on the previous one: what counts is the
// number of instructions per draw and that no two neighbours depend on each other.  The structured loop the
// compiler builds from C++ (two exits, phi copies, exec bookkeeping) costs ~27 instructions per draw; this
// hand-placed one 14 (tools/ubench/ubench_draw.hip: 179 -> 101 cycles per draw).  Per draw:
//   * T is tracked beside C (each draw lowers both by one); the product r * T of the NEXT draw is formed one
//     draw ahead (its T is known), so the loop-carried chain is only  Cb -> sub -> shift -> and -> add3;
//   * D = (V | 0x808080) - C is carried with the bias folded in (Cb = C - 0x808080); the broadcast of v is a
//     byte permute;
//   * two ring outputs are in flight in two fixed registers (the body is unrolled twice; lanes that make an
//     odd number of draws swap the pair on the way out), prefetched from an address that one 16-bit add advances;
//   * the near-tie test is only *recorded* (min and max of the fraction view f, folded in once per two draws):
//     sort_material looks at them once per step and, about once in 10^6 steps, redoes the step literally.
// On entry and exit rng.nxt / rng.nxt2 hold outputs #pos and #pos+1 (read after the step's barrier).
// The two products live in v[124:127]: inline asm cannot name the halves of a 64-bit operand.
#define MSE_RING_DRAW(PIN_HI, POUT, OUSE, OLOAD, EXTRA)        \
    "v_perm_b32 %[x], 0, " PIN_HI ", %[sel]\n\t"              \
    "v_add_u16 %[a], 0x400, %[a]\n\t"                         \
    "v_add_u32 %[t], -1, %[t]\n\t"                            \
    "v_sub_u32 %[x], %[x], %[cb]\n\t"                         \
    "v_cmp_ne_u32_e64 %[cm], %[t], %[tend]\n\t"               \
    "v_lshrrev_b32 %[x], 7, %[x]\n\t"                         \
    "ds_read_b32 %[" OLOAD "], %[a]\n\t"                      \
    EXTRA                                                      \
    "s_waitcnt lgkmcnt(1)\n\t"                                \
    "v_and_b32 %[x], 0x10101, %[x]\n\t"                       \
    "v_mad_u64_u32 " POUT ", %[dm], %[" OUSE "], %[t], 0\n\t" \
    "v_add3_u32 %[cb], %[cb], %[x], %[k]\n\t"                 \
    "s_and_b64 exec, exec, %[cm]\n\t"

__device__ __forceinline__ void draw_units_ring(RngRing &rng, uint32_t &C, int &rem)
{
#ifdef MSE_TIMELINE
    rng.tl->mark(2); // everything of sort_material outside the draw loops
#endif
    uint32_t T = C >> 24;
    const uint32_t n_draws = (uint32_t)rem;  // rem <= T: a station's own false units are in the pool
    const uint32_t T_end = T - n_draws;
    uint32_t Cb = C - 0x00808080u;
    uint32_t x;
    // the ring row of the next read as an LDS address of its own: output #pos+1 now, stepped by 0x400 before every read -
    // a 16-bit add, whose wrap at 2^16 IS the ring's wrap (64 rows of 1 KiB from LDS address 0): one instruction per draw
    // where the position and its address took two; the position itself moves once, behind the loop
    uint32_t a = ((rng.p10 + 0x400u) & 0xFC00u) | rng.lane_addr;
    uint64_t sv, cm, dm;
    const uint32_t kSel = 0x0C000000u, kK = 0xFEFEFEFFu;
    asm volatile(
        "s_mov_b64 %[sv], exec\n\t"
        "v_cmp_ne_u32 vcc, 0, %[n]\n\t"
        "s_and_b64 exec, exec, vcc\n\t"
        "s_cbranch_execz 3f\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"                         /* nothing but this loop's reads is counted below */
        "v_mad_u64_u32 v[124:125], %[dm], %[o0], %[t], 0\n"
        "1:\n\t"
        MSE_RING_DRAW("v125", "v[126:127]", "o1", "o0", "")
        "s_cbranch_execz 2f\n\t"
        MSE_RING_DRAW("v127", "v[124:125]", "o0", "o1",
                      "v_min3_u32 %[mn], %[mn], v124, v126\n\tv_max3_u32 %[mx], %[mx], v124, v126\n\t")
        "s_cbranch_execnz 1b\n"
        "2:\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "s_mov_b64 exec, %[sv]\n\t"
        "v_and_b32 %[x], 1, %[n]\n\t"
        "v_cmp_ne_u32 vcc, 0, %[x]\n\t"                    /* an odd number of draws: the pair is swapped and   */
        "s_and_b64 exec, exec, vcc\n\t"                     /* the last f is not in the monitor yet             */
        "v_swap_b32 %[o0], %[o1]\n\t"
        "v_min_u32 %[mn], %[mn], v124\n\t"
        "v_max_u32 %[mx], %[mx], v124\n"
        "3:\n\t"
        "s_mov_b64 exec, %[sv]"
        : [x] "=&v"(x), [a] "+v"(a), [o0] "+v"(rng.nxt), [o1] "+v"(rng.nxt2), [cb] "+v"(Cb),
          [t] "+v"(T), [mn] "+v"(rng.f_min), [mx] "+v"(rng.f_max), [sv] "=&s"(sv), [cm] "=&s"(cm), [dm] "=&s"(dm)
        : [sel] "s"(kSel), [k] "s"(kK), [tend] "v"(T_end), [n] "v"(n_draws)
        : "vcc", "memory", "v124", "v125", "v126", "v127");
    rng.p10 += n_draws << 10;
    C = Cb + 0x00808080u;
    rem = 0;
#ifdef MSE_TIMELINE
    rng.tl->mark(7); // the draw loops
#endif
}

// The RNG lane's production loop (k_rollout_ring): `count` PCG64 steps, the upper 32 output bits of each stored
// at LDS row (w & 63) of the lane's ring column (lane_addr, 64 KiB-aligned ring).  The LCG step takes 19 instructions
// (round 2: 21): limb 1 is the low word of Y = s1 m0 + (s0 m1 + hi(s0 m0)) as ONE 64-bit multiply-add whose carry-out
// (the 65th bit, an SGPR mask) joins limb 3 in the add that was there anyway - the upper word of Y then carries into
// limbs 2..3 by one zero-extending move instead of two moves and a 64-bit add.  With the 7-instruction output, an
// address kept as a register that one 16-bit add advances (its wrap is the ring's) and three of loop control: 31 per output.  Products and carries live in v[112:121].
__device__ __forceinline__ void ring_produce(Pcg &g, uint32_t &w, uint32_t count, uint32_t lane_addr)
{
    uint32_t s0 = (uint32_t)g.s_lo, s1 = (uint32_t)(g.s_lo >> 32), s2 = (uint32_t)g.s_hi, s3 = (uint32_t)(g.s_hi >> 32);
    const uint32_t i0 = (uint32_t)g.i_lo, i1 = (uint32_t)(g.i_lo >> 32), i2 = (uint32_t)g.i_hi, i3 = (uint32_t)(g.i_hi >> 32);
    const uint32_t m0 = 0x9FCCF645u, m1 = 0x4385DF64u, m2 = 0x1FC65DA4u, m3 = 0x2360ED05u;
    uint32_t a = ((w << 10) & 0xFC00u) | lane_addr, left = count; // LDS address of output #w's ring row
    uint32_t t0, t1, t2, t3, x, y;
    uint64_t sv, cm, dm, cy;
    asm volatile(
        "s_mov_b64 %[sv], exec\n\t"
        "v_cmp_ne_u32 vcc, 0, %[left]\n\t"
        "s_and_b64 exec, exec, vcc\n\t"
        "s_cbranch_execz 3f\n\t"
        "v_mov_b32 v121, 0\n"
        "1:\n\t"
        "v_mad_u64_u32 v[112:113], %[dm], %[s0], %[m0], 0\n\t"
        "v_mul_lo_u32 %[t3], %[s3], %[m0]\n\t"
        "v_mul_lo_u32 %[t0], %[s0], %[m3]\n\t"
        "v_mov_b32 v120, v113\n\t"
        "v_mad_u64_u32 v[114:115], %[dm], %[s0], %[m1], v[120:121]\n\t"
        "v_mul_lo_u32 %[t1], %[s1], %[m2]\n\t"
        "v_mul_lo_u32 %[t2], %[s2], %[m1]\n\t"
        "v_mad_u64_u32 v[116:117], %[cy], %[s1], %[m0], v[114:115]\n\t"
        "v_add3_u32 %[t0], %[t0], %[t1], %[t2]\n\t"
        "v_mov_b32 v120, v117\n\t"
        "v_mad_u64_u32 v[118:119], %[dm], %[s0], %[m2], v[120:121]\n\t"
        "v_add_u32 %[t0], %[t0], %[t3]\n\t"
        "v_mad_u64_u32 v[118:119], %[dm], %[s1], %[m1], v[118:119]\n\t"
        "v_mad_u64_u32 v[118:119], %[dm], %[s2], %[m0], v[118:119]\n\t"
        "v_addc_co_u32_e64 %[t0], %[dm], v119, %[t0], %[cy]\n\t"
        "v_add_co_u32 %[s0], vcc, v112, %[i0]\n\t"
        "v_addc_co_u32 %[s1], vcc, v116, %[i1], vcc\n\t"
        "v_addc_co_u32 %[s2], vcc, v118, %[i2], vcc\n\t"
        "v_addc_co_u32 %[s3], vcc, %[t0], %[i3], vcc\n\t"
        "v_xor_b32 %[x], %[s0], %[s2]\n\t"
        "v_xor_b32 %[y], %[s1], %[s3]\n\t"
        "v_lshrrev_b32 %[t1], 26, %[s3]\n\t"
        "v_cmp_gt_i32 vcc, 0, %[s3]\n\t"
        "v_cndmask_b32 %[t2], %[x], %[y], vcc\n\t"
        "v_cndmask_b32 %[t3], %[y], %[x], vcc\n\t"
        "v_add_u32 %[left], -1, %[left]\n\t"
        "v_alignbit_b32 %[t2], %[t2], %[t3], %[t1]\n\t"
        "v_cmp_ne_u32_e64 %[cm], 0, %[left]\n\t"
        "ds_write_b32 %[a], %[t2]\n\t"
        "v_add_u16 %[a], 0x400, %[a]\n\t"                    /* next row; the 16-bit wrap is the ring's */
        "s_and_b64 exec, exec, %[cm]\n\t"
        "s_cbranch_execnz 1b\n"
        "3:\n\t"
        "s_mov_b64 exec, %[sv]"
        : [s0] "+v"(s0), [s1] "+v"(s1), [s2] "+v"(s2), [s3] "+v"(s3), [a] "+v"(a), [left] "+v"(left), [t0] "=&v"(t0),
          [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3), [x] "=&v"(x), [y] "=&v"(y), [sv] "=&s"(sv), [cm] "=&s"(cm),
          [dm] "=&s"(dm), [cy] "=&s"(cy)
        : [i0] "v"(i0), [i1] "v"(i1), [i2] "v"(i2), [i3] "v"(i3), [m0] "s"(m0), [m1] "s"(m1), [m2] "s"(m2), [m3] "s"(m3)
        : "vcc", "memory", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121");
    g.s_lo = (uint64_t)s0 | ((uint64_t)s1 << 32);
    g.s_hi = (uint64_t)s2 | ((uint64_t)s3 << 32);
    w += count;
}

// Two outputs per iteration from ONE state: s_{n+1} = M s_n + inc and s_{n+2} = M^2 s_n + (M + 1) inc are independent
// 128-bit multiplies of the same s_n by two constants, so the two LCG chains (and the two outputs' XSL-RR) interleave
// instruction by instruction: 65 instructions per two outputs against 68 (one loop-control block instead of two), half
// the dependent chain.  Same-box A/B: +1.2 % env-steps/s at 20 and at 64 steps per launch - what the three instructions
// are worth; the shorter chain by itself is worth nothing (the SIMD's vector unit, not this wave's latency, is the limit:
// DESIGN.md section 6).  `pairs` double steps; c2 = (M + 1) inc.
// Chain A (-> outputs #w, #w+2, ...) works in v[112:127] and leaves its state in v[90:93]; chain B in v[94:109].
__device__ __forceinline__ void ring_produce_pairs(Pcg &g, uint32_t &w, uint32_t pairs, uint32_t lane_addr, uint64_t c2_lo,
                                                   uint64_t c2_hi)
{
    uint32_t s0 = (uint32_t)g.s_lo, s1 = (uint32_t)(g.s_lo >> 32), s2 = (uint32_t)g.s_hi, s3 = (uint32_t)(g.s_hi >> 32);
    const uint32_t i0 = (uint32_t)g.i_lo, i1 = (uint32_t)(g.i_lo >> 32), i2 = (uint32_t)g.i_hi, i3 = (uint32_t)(g.i_hi >> 32);
    const uint32_t j0 = (uint32_t)c2_lo, j1 = (uint32_t)(c2_lo >> 32), j2 = (uint32_t)c2_hi, j3 = (uint32_t)(c2_hi >> 32);
    const uint32_t m0 = 0x9FCCF645u, m1 = 0x4385DF64u, m2 = 0x1FC65DA4u, m3 = 0x2360ED05u;
    const uint32_t n0 = 0x20E0AE99u, n1 = 0x529ED9EBu, n2 = 0xDF69743Cu, n3 = 0x17BCE35Bu; // M^2 mod 2^128
    uint32_t a = ((w << 10) & 0xFC00u) | lane_addr, left = pairs; // LDS address of output #w's ring row
    uint64_t sv, cm, dm, cy, cy2;
    asm volatile(
        "s_mov_b64 %[sv], exec\n\t"
        "v_cmp_ne_u32 vcc, 0, %[left]\n\t"
        "s_and_b64 exec, exec, vcc\n\t"
        "s_cbranch_execz 3f\n\t"
        "v_mov_b32 v121, 0\n\t"
        "v_mov_b32 v109, 0\n"
        "1:\n\t"
        "v_mad_u64_u32 v[112:113], %[dm], %[s0], %[m0], 0\n\t"
        "v_mad_u64_u32 v[100:101], %[dm], %[s0], %[n0], 0\n\t"
        "v_mul_lo_u32 v125, %[s3], %[m0]\n\t"
        "v_mul_lo_u32 v97, %[s3], %[n0]\n\t"
        "v_mul_lo_u32 v122, %[s0], %[m3]\n\t"
        "v_mul_lo_u32 v94, %[s0], %[n3]\n\t"
        "v_mov_b32 v120, v113\n\t"
        "v_mov_b32 v108, v101\n\t"
        "v_mad_u64_u32 v[114:115], %[dm], %[s0], %[m1], v[120:121]\n\t"
        "v_mad_u64_u32 v[102:103], %[dm], %[s0], %[n1], v[108:109]\n\t"
        "v_mul_lo_u32 v123, %[s1], %[m2]\n\t"
        "v_mul_lo_u32 v95, %[s1], %[n2]\n\t"
        "v_mul_lo_u32 v124, %[s2], %[m1]\n\t"
        "v_mul_lo_u32 v96, %[s2], %[n1]\n\t"
        "v_mad_u64_u32 v[116:117], %[cy], %[s1], %[m0], v[114:115]\n\t"
        "v_mad_u64_u32 v[104:105], %[cy2], %[s1], %[n0], v[102:103]\n\t"
        "v_add3_u32 v122, v122, v123, v124\n\t"
        "v_add3_u32 v94, v94, v95, v96\n\t"
        "v_mov_b32 v120, v117\n\t"
        "v_mov_b32 v108, v105\n\t"
        "v_mad_u64_u32 v[118:119], %[dm], %[s0], %[m2], v[120:121]\n\t"
        "v_mad_u64_u32 v[106:107], %[dm], %[s0], %[n2], v[108:109]\n\t"
        "v_add_u32 v122, v122, v125\n\t"
        "v_add_u32 v94, v94, v97\n\t"
        "v_mad_u64_u32 v[118:119], %[dm], %[s1], %[m1], v[118:119]\n\t"
        "v_mad_u64_u32 v[106:107], %[dm], %[s1], %[n1], v[106:107]\n\t"
        "v_mad_u64_u32 v[118:119], %[dm], %[s2], %[m0], v[118:119]\n\t"
        "v_mad_u64_u32 v[106:107], %[dm], %[s2], %[n0], v[106:107]\n\t"
        "v_addc_co_u32_e64 v122, %[dm], v119, v122, %[cy]\n\t"
        "v_add_co_u32 v90, vcc, v112, %[i0]\n\t"
        "v_addc_co_u32 v91, vcc, v116, %[i1], vcc\n\t"
        "v_addc_co_u32 v92, vcc, v118, %[i2], vcc\n\t"
        "v_addc_co_u32 v93, vcc, v122, %[i3], vcc\n\t"
        "v_addc_co_u32_e64 v94, %[dm], v107, v94, %[cy2]\n\t"
        "v_add_co_u32 %[s0], vcc, v100, %[j0]\n\t"
        "v_addc_co_u32 %[s1], vcc, v104, %[j1], vcc\n\t"
        "v_addc_co_u32 %[s2], vcc, v106, %[j2], vcc\n\t"
        "v_addc_co_u32 %[s3], vcc, v94, %[j3], vcc\n\t"
        "v_xor_b32 v126, v90, v92\n\t"
        "v_xor_b32 v98, %[s0], %[s2]\n\t"
        "v_xor_b32 v127, v91, v93\n\t"
        "v_xor_b32 v99, %[s1], %[s3]\n\t"
        "v_lshrrev_b32 v123, 26, v93\n\t"
        "v_lshrrev_b32 v95, 26, %[s3]\n\t"
        "v_cmp_gt_i32 vcc, 0, v93\n\t"
        "v_cmp_gt_i32_e64 %[cm], 0, %[s3]\n\t"
        "v_cndmask_b32 v124, v126, v127, vcc\n\t"
        "v_cndmask_b32_e64 v96, v98, v99, %[cm]\n\t"
        "v_cndmask_b32 v125, v127, v126, vcc\n\t"
        "v_cndmask_b32_e64 v97, v99, v98, %[cm]\n\t"
        "v_alignbit_b32 v124, v124, v125, v123\n\t"
        "v_alignbit_b32 v96, v96, v97, v95\n\t"
        "v_add_u16 v94, 0x400, %[a]\n\t"                     /* the second output's row: a 16-bit add, whose wrap is */
        "v_add_u32 %[left], -1, %[left]\n\t"                 /* the ring's (64 rows of 1 KiB from LDS address 0)    */
        "ds_write_b32 %[a], v124\n\t"
        "v_cmp_ne_u32_e64 %[cm], 0, %[left]\n\t"
        "ds_write_b32 v94, v96\n\t"
        "v_add_u16 %[a], 0x400, v94\n\t"
        "s_and_b64 exec, exec, %[cm]\n\t"
        "s_cbranch_execnz 1b\n"
        "3:\n\t"
        "s_mov_b64 exec, %[sv]"
        : [s0] "+v"(s0), [s1] "+v"(s1), [s2] "+v"(s2), [s3] "+v"(s3), [a] "+v"(a), [left] "+v"(left), [sv] "=&s"(sv),
          [cm] "=&s"(cm), [dm] "=&s"(dm), [cy] "=&s"(cy), [cy2] "=&s"(cy2)
        : [i0] "v"(i0), [i1] "v"(i1), [i2] "v"(i2), [i3] "v"(i3), [j0] "v"(j0), [j1] "v"(j1), [j2] "v"(j2), [j3] "v"(j3),
          [m0] "s"(m0), [m1] "s"(m1), [m2] "s"(m2), [m3] "s"(m3), [n0] "s"(n0), [n1] "s"(n1), [n2] "s"(n2), [n3] "s"(n3)
        : "vcc", "memory", "v90", "v91", "v92", "v93", "v94", "v95", "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103",
          "v104", "v105", "v106", "v107", "v108", "v109", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120",
          "v121", "v122", "v123", "v124", "v125", "v126", "v127");
    g.s_lo = (uint64_t)s0 | ((uint64_t)s1 << 32);
    g.s_hi = (uint64_t)s2 | ((uint64_t)s3 << 32);
    w += 2u * pairs;
}

// The same hand-placed loop for a lane that advances its own generator (one-lane-per-env kernels: k_step,
// k_rollout).  Per draw: the 128-bit LCG step on 32-bit limbs (6 v_mad_u64_u32 + 4 v_mul_lo_u32, the {carry, 0}
// addends through one scratch pair), the upper half of the XSL-RR output, and the decision of draw_units_ring;
// ~40 instructions against ~50 from the structured C++ loop.  Near ties are only recorded (f_min / f_max);
// sort_material redoes the step literally when one shows.  Products and carries live in v[112:123].
__device__ __forceinline__ void draw_units_local(Pcg &g, uint32_t &C, int &rem, uint32_t &f_min, uint32_t &f_max)
{
    uint32_t T = C >> 24;
    const uint32_t n_draws = (uint32_t)rem;
    const uint32_t T_end = T - n_draws;
    uint32_t Cb = C - 0x00808080u;
    uint32_t s0 = (uint32_t)g.s_lo, s1 = (uint32_t)(g.s_lo >> 32), s2 = (uint32_t)g.s_hi, s3 = (uint32_t)(g.s_hi >> 32);
    const uint32_t i0 = (uint32_t)g.i_lo, i1 = (uint32_t)(g.i_lo >> 32), i2 = (uint32_t)g.i_hi, i3 = (uint32_t)(g.i_hi >> 32);
    const uint32_t m0 = 0x9FCCF645u, m1 = 0x4385DF64u, m2 = 0x1FC65DA4u, m3 = 0x2360ED05u;
    const uint32_t kSel = 0x0C000000u, kK = 0xFEFEFEFFu;
    uint32_t t0, t1, t2, t3, x, y;
    uint64_t sv, cm, dm, cy;
    asm volatile(
        "s_mov_b64 %[sv], exec\n\t"
        "v_cmp_ne_u32 vcc, 0, %[n]\n\t"
        "s_and_b64 exec, exec, vcc\n\t"
        "s_cbranch_execz 3f\n\t"
        "v_mov_b32 v121, 0\n"
        "1:\n\t"
        /* state * M + inc  (mod 2^128) */
        "v_mad_u64_u32 v[112:113], %[dm], %[s0], %[m0], 0\n\t"
        "v_mul_lo_u32 %[t3], %[s3], %[m0]\n\t"
        "v_mul_lo_u32 %[t0], %[s0], %[m3]\n\t"
        "v_mov_b32 v120, v113\n\t"
        "v_mad_u64_u32 v[114:115], %[dm], %[s0], %[m1], v[120:121]\n\t"
        "v_mul_lo_u32 %[t1], %[s1], %[m2]\n\t"
        "v_mul_lo_u32 %[t2], %[s2], %[m1]\n\t"
        "v_mad_u64_u32 v[116:117], %[cy], %[s1], %[m0], v[114:115]\n\t"
        "v_add3_u32 %[t0], %[t0], %[t1], %[t2]\n\t"
        "v_mov_b32 v120, v117\n\t"
        "v_mad_u64_u32 v[118:119], %[dm], %[s0], %[m2], v[120:121]\n\t"
        "v_add_u32 %[t0], %[t0], %[t3]\n\t"
        "v_mad_u64_u32 v[118:119], %[dm], %[s1], %[m1], v[118:119]\n\t"
        "v_mad_u64_u32 v[118:119], %[dm], %[s2], %[m0], v[118:119]\n\t"
        "v_addc_co_u32_e64 %[t0], %[dm], v119, %[t0], %[cy]\n\t"
        "v_add_co_u32 %[s0], vcc, v112, %[i0]\n\t"
        "v_addc_co_u32 %[s1], vcc, v116, %[i1], vcc\n\t"
        "v_addc_co_u32 %[s2], vcc, v118, %[i2], vcc\n\t"
        "v_addc_co_u32 %[s3], vcc, %[t0], %[i3], vcc\n\t"
        /* upper 32 bits of rotr64(hi ^ lo, hi >> 58) */
        "v_xor_b32 %[x], %[s0], %[s2]\n\t"
        "v_xor_b32 %[y], %[s1], %[s3]\n\t"
        "v_lshrrev_b32 %[t1], 26, %[s3]\n\t"
        "v_cmp_gt_i32 vcc, 0, %[s3]\n\t"
        "v_cndmask_b32 %[t2], %[x], %[y], vcc\n\t"
        "v_cndmask_b32 %[t3], %[y], %[x], vcc\n\t"
        "v_alignbit_b32 %[t2], %[t2], %[t3], %[t1]\n\t"
        /* the draw: v = floor(u T), flags by byte compare, one unit leaves the chosen bin */
        "v_mad_u64_u32 v[122:123], %[dm], %[t2], %[t], 0\n\t"
        "v_add_u32 %[t], -1, %[t]\n\t"
        "v_perm_b32 %[x], 0, v123, %[sel]\n\t"
        "v_min_u32 %[mn], %[mn], v122\n\t"
        "v_sub_u32 %[x], %[x], %[cb]\n\t"
        "v_max_u32 %[mx], %[mx], v122\n\t"
        "v_lshrrev_b32 %[x], 7, %[x]\n\t"
        "v_cmp_ne_u32_e64 %[cm], %[t], %[tend]\n\t"
        "v_and_b32 %[x], 0x10101, %[x]\n\t"
        "v_add3_u32 %[cb], %[cb], %[x], %[k]\n\t"
        "s_and_b64 exec, exec, %[cm]\n\t"
        "s_cbranch_execnz 1b\n"
        "3:\n\t"
        "s_mov_b64 exec, %[sv]"
        : [s0] "+v"(s0), [s1] "+v"(s1), [s2] "+v"(s2), [s3] "+v"(s3), [cb] "+v"(Cb), [t] "+v"(T), [mn] "+v"(f_min),
          [mx] "+v"(f_max), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3), [x] "=&v"(x), [y] "=&v"(y),
          [sv] "=&s"(sv), [cm] "=&s"(cm), [dm] "=&s"(dm), [cy] "=&s"(cy)
        : [i0] "v"(i0), [i1] "v"(i1), [i2] "v"(i2), [i3] "v"(i3), [m0] "s"(m0), [m1] "s"(m1), [m2] "s"(m2), [m3] "s"(m3),
          [sel] "s"(kSel), [k] "s"(kK), [tend] "v"(T_end), [n] "v"(n_draws)
        : "vcc", "memory", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123");
    g.s_lo = (uint64_t)s0 | ((uint64_t)s1 << 32);
    g.s_hi = (uint64_t)s2 | ((uint64_t)s3 << 32);
    C = Cb + 0x00808080u;
    rem = 0;
}

// station I: true = rint(target * acc), false = target - true, leftover[I] = false (env_super.py:535-546)
template <int I>
__device__ __forceinline__ void station_split(Env &e, uint32_t &C, const double acc_sorter[4], int &rem)
{
    const uint32_t hi_b = (C >> (8 * I)) & 0xFFu;
    const uint32_t lo_b = I == 0 ? 0u : ((C >> (8 * (I > 0 ? I - 1 : 0))) & 0xFFu);
    const uint32_t target = hi_b - lo_b;
    // int(round(np.int64 * float)) : half to even (env_super.py:539)
    const uint32_t t = (uint32_t)(int)rint((double)target * acc_sorter[I]);
    rem = (int)(target - t);
    C -= t * (0x01010101u << (8 * I));
    e.ct[I] += (int)t;
    e.cf[I] += rem;
}

// Stations are walked in the reference's order.  Stations I and I+1 share one draw loop: a lane whose
// station I has nothing to redistribute (accuracy 1.0 after the boost) starts station I+1 at once, so a
// wave runs max(f_I, f_{I+1}) iterations instead of f_I + f_{I+1} when its lanes sit in different modes.
template <bool LITERAL, int I, class RNG>
__device__ __forceinline__ void station_pair(Env &e, RNG &rng, uint32_t &C, const double acc_sorter[4])
{
    int rem;
    station_split<I>(e, C, acc_sorter, rem);
    const bool early = rem == 0;
    if (early) station_split<I + 1>(e, C, acc_sorter, rem);
    draw_units<LITERAL>(rng, C, rem);
    if (!early) {
        station_split<I + 1>(e, C, acc_sorter, rem);
        draw_units<LITERAL>(rng, C, rem);
    }
}

template <bool LITERAL, class RNG>
__device__ __forceinline__ void sort_material(Env &e, RNG &rng, uint32_t sorting_word, const double acc_sorter[4])
{
    rng.begin_step();
    uint32_t C = sorting_word * 0x01010101u; // prefix sums of current_material_sorting
    station_pair<LITERAL, 0>(e, rng, C, acc_sorter);
    station_pair<LITERAL, 2>(e, rng, C, acc_sorter);
    e.ce += (int)(C >> 24); // env_super.py:579,597
}

template <int I>
__device__ __forceinline__ void station_pair_ring(Env &e, RngRing &rng, uint32_t &C, const double acc_sorter[4])
{
    int rem;
    station_split<I>(e, C, acc_sorter, rem);
    const bool early = rem == 0;
    if (early) station_split<I + 1>(e, C, acc_sorter, rem);
    draw_units_ring(rng, C, rem);
    if (!early) {
        station_split<I + 1>(e, C, acc_sorter, rem);
        draw_units_ring(rng, C, rem);
    }
}

// sort_material on the ring source (overload chosen for RngRing).
template <bool LITERAL>
__device__ __forceinline__ void sort_material(Env &e, RngRing &rng, uint32_t sorting_word, const double acc_sorter[4])
{
    static_assert(!LITERAL, "the ring carries only the upper 32 output bits");
    const uint32_t p10_0 = rng.p10;
    const int ct0 = e.ct[0], ct1 = e.ct[1], ct2 = e.ct[2], ct3 = e.ct[3];
    const int cf0 = e.cf[0], cf1 = e.cf[1], cf2 = e.cf[2], cf3 = e.cf[3];
    const int ce0 = e.ce;
    rng.begin_step();
    uint32_t C = sorting_word * 0x01010101u;
    station_pair_ring<0>(e, rng, C, acc_sorter);
    station_pair_ring<2>(e, rng, C, acc_sorter);
    e.ce += (int)(C >> 24);
    // near a tie: f < window - hi or f >= 2^32 - hi  (f + hi < window, as in draw_units)
#ifdef MSE_ABL_NOTIE // (ablation timing builds only: near ties are never looked at - results can differ)
    if (false) {
#else
    if (__builtin_expect(rng.f_min < MSE_TIE_WINDOW - MSE_TIE_HI || rng.f_max >= 0u - MSE_TIE_HI, 0)) {
#endif
        // some draw of this step sat within the margin of a cdf boundary: take the step again from the saved counters
        // on the full 64-bit outputs of a generator positioned by jump-ahead, every draw decided as draw_units<false>
        // decides it - the exact integer comparison, and the literal fp64 cdf for the draw(s) inside the margin (the
        // whole workgroup waits for this lane: every fp64 cdf that need not be evaluated is 8 divisions saved)
        e.ct[0] = ct0; e.ct[1] = ct1; e.ct[2] = ct2; e.ct[3] = ct3;
        e.cf[0] = cf0; e.cf[1] = cf1; e.cf[2] = cf2; e.cf[3] = cf3;
        e.ce = ce0;
        RngCounted exact;
        exact.g = rng.start;
        exact.count = 0;
        pcg_jump(exact.g, p10_0 >> 10, rng.jump_tab);
        sort_material<false>(e, exact, sorting_word, acc_sorter);
        rng.p10 = p10_0 + (exact.count << 10);
    }
}

template <int I>
__device__ __forceinline__ void station_pair_local(Env &e, Pcg &g, uint32_t &C, const double acc_sorter[4],
                                                   uint32_t &f_min, uint32_t &f_max)
{
    int rem;
    station_split<I>(e, C, acc_sorter, rem);
    const bool early = rem == 0;
    if (early) station_split<I + 1>(e, C, acc_sorter, rem);
    draw_units_local(g, C, rem, f_min, f_max);
    if (!early) {
        station_split<I + 1>(e, C, acc_sorter, rem);
        draw_units_local(g, C, rem, f_min, f_max);
    }
}

// sort_material for a lane with its own generator and the integer decision (overload chosen for RngLocal,
// LITERAL = false): hand-placed loops, then - if any draw of the step sat in the near-tie window - the step is
// taken again from the saved generator and counters with every decision made by the literal fp64 cdf.
__device__ __forceinline__ void sort_material_local_fast(Env &e, Pcg &g, uint32_t sorting_word, const double acc_sorter[4])
{
    const Pcg g0 = g;
    const int ct0 = e.ct[0], ct1 = e.ct[1], ct2 = e.ct[2], ct3 = e.ct[3];
    const int cf0 = e.cf[0], cf1 = e.cf[1], cf2 = e.cf[2], cf3 = e.cf[3];
    const int ce0 = e.ce;
    uint32_t f_min = 0xFFFFFFFFu, f_max = 0u;
    uint32_t C = sorting_word * 0x01010101u;
    station_pair_local<0>(e, g, C, acc_sorter, f_min, f_max);
    station_pair_local<2>(e, g, C, acc_sorter, f_min, f_max);
    e.ce += (int)(C >> 24);
    if (__builtin_expect(f_min < MSE_TIE_WINDOW - MSE_TIE_HI || f_max >= 0u - MSE_TIE_HI, 0)) {
        e.ct[0] = ct0; e.ct[1] = ct1; e.ct[2] = ct2; e.ct[3] = ct3;
        e.cf[0] = cf0; e.cf[1] = cf1; e.cf[2] = cf2; e.cf[3] = cf3;
        e.ce = ce0;
        RngCounted exact; // the structured loop: integer decisions, the literal cdf for the draw(s) inside the margin
        exact.g = g0;
        exact.count = 0;
        sort_material<false>(e, exact, sorting_word, acc_sorter);
        g = exact.g;
    }
}

template <bool LITERAL>
__device__ __forceinline__ void sort_material(Env &e, RngLocal &rng, uint32_t sorting_word, const double acc_sorter[4])
{
    if (LITERAL) sort_material<true, RngLocal>(e, rng, sorting_word, acc_sorter);
    else sort_material_local_fast(e, rng.g, sorting_word, acc_sorter);
}

// buffered uint32 of PCG64 (pcg64.h pcg64_next32)
__device__ __forceinline__ uint32_t press_next32(Env &e)
{
    if (e.press_has) {
        e.press_has = 0;
        return e.press_uint;
    }
    uint64_t r = pcg_next64(e.press);
    e.press_has = 1;
    e.press_uint = (uint32_t)(r >> 32);
    return (uint32_t)r;
}

// env_super.py:291-300 sample_masked_press_action: rng_pressing.choice(flatnonzero(mask))
// = valid[buffered Lemire(len-1)], no draw when one action is valid
__device__ __forceinline__ int sample_masked_press_action(Env &e, const Params &P)
{
    uint32_t bits = press_mask_bits(e, P);
    uint32_t n = (uint32_t)__popc(bits);
    if (n <= 1u) return 0;
    uint32_t rng = n - 1u;
    uint64_t m = (uint64_t)press_next32(e) * n;
    uint32_t leftover = (uint32_t)m;
    if (leftover < n) {
        uint32_t threshold = (0xFFFFFFFFu - rng) % n;
        while (leftover < threshold) {
            m = (uint64_t)press_next32(e) * n;
            leftover = (uint32_t)m;
        }
    }
    return select_kth_bit(bits, (int)(m >> 32));
}

// round(true/total, 2) in hundredths = rint(fl(fl(true/total) * 100)), the way numpy rounds
// (env_super.py:754, 785-789).  The literal fp64 form is also the cheapest one here: at one wave per
// SIMD a kernel pays ~5 cycles per instruction whatever it is, and the division sequence is ~14
// instructions; an integer formulation (fp32 reciprocal, remainder fix-up, exact-tie table) was
// measured 6 % slower end to end.
__device__ __forceinline__ int purity_hundredths(int tru, int total)
{
    return (int)rint(((double)tru / (double)total) * 100.0);
}

// amount // S and amount % S for 0 <= amount < 2^24 without a division: fp32 estimate, exact fix-up
__device__ __forceinline__ void divmod_small(int amount, int S, float inv_S, int &q, int &r)
{
    q = (int)((float)amount * inv_S);
    r = amount - q * S;
    if (r < 0) {
        q -= 1;
        r += S;
    } else if (r >= S) {
        q += 1;
        r -= S;
    }
}

// int(q * 100) of press_bale (env_super.py:664): q100 minus a bit of qi_down (the 101 cases, found by the host with the
// literal expression).  The four mask words follow the press times in the table image (ptab = Tables::press_time): a
// lane selects one by its own q100, and a per-lane selection among kernel arguments becomes ONE global load at a
// selected address - inside the step loop, with an s_waitcnt vmcnt(0) behind it that also waits for every output store
// the wave has in flight (that is what the listing of round 2's kernels showed: 0.4-0.6 us per step on the critical path).
__device__ __forceinline__ uint32_t bale_quality_int(const int *ptab, int q100)
{
    const uint32_t word = (uint32_t)ptab[2 + ((uint32_t)q100 >> 5)];
    return (uint32_t)q100 - ((word >> ((uint32_t)q100 & 31u)) & 1u);
}

// What one step appends to the reference's per-env Python ledgers (opt-in trace of ONE env, mse_trace_begin):
// press_actions_per_timestep entries (env_super.py:631-637,730-736; env_monolith.py:136; env_2_press.py:131) and
// the press_bale calls (env_super.py:661-687).  Only the TRACE instantiation of k_step fills it.
struct Ledger {
    int n_log, code[2], mat[2];        // code 0 no-op, 1|2 press started, 111|222 busy / invalid; mat 0..4 or -1
    int n_bale, bmat[2], bn[2], bq[2]; // finished presses in the reference's order: material, amount, int(q*100)
    int internal;                      // Env_1: the press action the env sampled itself (env_1_sort.py:125)
};

// env_super.py:661-687 press_bale on the O(1) ledger summary {count, sum, last_size, last_q}.
// The reference's three floating-point expressions here are functions of small integers and are evaluated as
// such (a finishing press is a per-lane rarity but a per-wave certainty, so this path runs almost every step):
//   int(q * 100) with q = round(x, 2) = q100 / 100.0   ->  q100 minus a bit of P.qi_down (the 101 cases, found by the
//                                                          host with the literal expression: 0.29 -> 28, 0.57 -> 56, ...)
//   floor(n / S)                                       ->  integer division (exact: the quotient of two int32 as a
//                                                          double is within 2^-52 of the true value, never across an integer)
//   rem > S * bale_remainder_threshold                 ->  rem > P.rem_thr_units = floor(S * thr), same test on integers
template <class BALES>
__device__ __forceinline__ void press_bale(const BALES &bales, int m, const Params &P, const int *ptab, int n, int q100)
{
    uint4 c = bales.load(m);
    const uint32_t qi = bale_quality_int(ptab, q100);
    const uint32_t S = (uint32_t)P.balesize;
    int full_i, rem_i;
    if (__builtin_expect(n < (1 << 24), 1)) divmod_small(n, P.balesize, P.inv_balesize, full_i, rem_i);
    else {
        full_i = (int)floor((double)n / (double)S);
        rem_i = n - full_i * (int)S;
    }
    const uint32_t full = (uint32_t)full_i, rem = (uint32_t)rem_i;
    if (full > 0) {
        c.x += full;
        c.y += full * S;
        c.z = S;
        c.w = qi;
    }
    if (rem > 0) {
        if ((int)rem > P.rem_thr_units) {
            c.x += 1;
            c.y += rem;
            c.z = rem;
            c.w = qi;
        } else if (c.x > 0) {
            c.y += rem;
            c.z += rem;
        } else {
            c.x = 1;
            c.y = rem;
            c.z = rem;
            c.w = qi;
        }
    }
    bales.store(m, c);
}

// where the bale ledger of this lane lives: global planes (single-step kernel) or an LDS copy
// (rollout kernel: no global load may sit in the step loop, its vmcnt wait would drain the output stores)
struct BaleRef {
    uint4 *base;        // cell (m) at base[m * stride]
    long long stride;
    __device__ __forceinline__ uint4 load(int m) const { return base[(long long)m * stride]; }
    __device__ __forceinline__ void store(int m, const uint4 &c) const { base[(long long)m * stride] = c; }
};
// The same ledger in 12 bytes per cell, for kernels whose occupancy hangs on their LDS: words {count, sum,
// last_size | last_q << 24}, word w of material m of env slot el at base[(3 m + w) * stride + el].  A bale's size stays
// below 2^24 (an episode adds at most batch x 65 535 units to a container) and q is a percentage.
struct BaleRefCompact {
    uint32_t *base;     // already offset by the env slot
    int stride;
    __device__ __forceinline__ uint4 load(int m) const
    {
        const uint32_t w2 = base[(3 * m + 2) * stride];
        return make_uint4(base[(3 * m) * stride], base[(3 * m + 1) * stride], w2 & 0xFFFFFFu, w2 >> 24);
    }
    __device__ __forceinline__ void store(int m, const uint4 &c) const
    {
        base[(3 * m) * stride] = c.x;
        base[(3 * m + 1) * stride] = c.y;
        base[(3 * m + 2) * stride] = (c.z & 0xFFFFFFu) | (c.w << 24);
    }
};

// The ledger kept by ANOTHER wave (k_rollout_policy_roles: the critic wave): the lane that steps the env only notes
// what a step books - at most two press_bale calls, in the reference's order, and the auto-reset's clear after them -
// and hands the note over; the other wave replays it on the real ledger (replay()).  q100 < 128 (bale_quality_int).
struct BaleNote {
    uint32_t *head; // calls (2 bits) | clear << 2 | m_0 << 3 | m_1 << 6 | q100_0 << 9 | q100_1 << 16
    uint32_t *n;    // n[0], n[1]: the amounts
    __device__ __forceinline__ void begin_step() const
    {
        *head = 0u;
        n[0] = n[1] = 0u;
    }
    template <class BALES>
    static __device__ __forceinline__ void replay(uint32_t head, uint32_t n0, uint32_t n1, const BALES &ledger, const Params &P,
                                                  const int *ptab);
};
template <>
__device__ __forceinline__ void press_bale<BaleNote>(const BaleNote &note, int m, const Params &P, const int *ptab, int n, int q100)
{
    const uint32_t h = *note.head, second = h & 3u; // 0 | 1 calls so far
    *note.head = (h + 1u) | ((uint32_t)m << (second ? 6 : 3)) | ((uint32_t)q100 << (second ? 16 : 9));
    note.n[0] = second ? note.n[0] : (uint32_t)n;
    note.n[1] = second ? (uint32_t)n : note.n[1];
}
template <class BALES>
__device__ __forceinline__ void BaleNote::replay(uint32_t head, uint32_t n0, uint32_t n1, const BALES &ledger, const Params &P,
                                                 const int *ptab)
{
    const uint32_t calls = head & 3u;
    if (calls > 0u) press_bale(ledger, (int)((head >> 3) & 7u), P, ptab, (int)n0, (int)((head >> 9) & 127u));
    if (__builtin_expect(calls > 1u, 0)) press_bale(ledger, (int)((head >> 6) & 7u), P, ptab, (int)n1, (int)((head >> 16) & 127u));
    if (__builtin_expect((head & 4u) != 0u, 0)) {
#pragma unroll
        for (int m = 0; m < 5; ++m) ledger.store(m, make_uint4(0, 0, 0, 0));
    }
}

// env_super.py:626-640 press_action_rules = check_press_status (:642-659) then use_press (:722-769)
template <bool TRACE = false, class BALES = BaleRef>
__device__ __forceinline__ void press_action_rules(Env &e, const Params &P, const int *ptab, int press_action, const BALES &bales,
                                                   Ledger *lg = nullptr)
{
    // check_press_status: both timers tick; a press that reaches 0 books its bale.  A finishing press is rare per
    // lane but near-certain per wave, so the ledger update is written once (for press 0 if it finished, else press
    // 1) and the case of both finishing in the same step takes a second, almost never executed, copy.
    const bool fin0 = e.timer[0] == 1, fin1 = e.timer[1] == 1;
    e.timer[0] -= e.timer[0] > 0 ? 1 : 0;
    e.timer[1] -= e.timer[1] > 0 ? 1 : 0;
    if (fin0 || fin1) {
        const int first = fin0 ? 0 : 1;
        if (TRACE) {
            for (int p = 0; p < 2; ++p) {
                if (p == 0 ? fin0 : fin1) {
                    lg->bmat[lg->n_bale] = e.pmat[p];
                    lg->bn[lg->n_bale] = e.pn[p];
                    lg->bq[lg->n_bale] = (int)bale_quality_int(ptab, e.q100[p]);
                    lg->n_bale += 1;
                }
            }
        }
        if (P.track_bales)
            press_bale(bales, first ? e.pmat[1] : e.pmat[0], P, ptab, first ? e.pn[1] : e.pn[0], first ? e.q100[1] : e.q100[0]);
        if (__builtin_expect(fin0 && fin1, 0)) { // press 0 was booked above (reference order: press 1, then 2)
            if (P.track_bales) press_bale(bales, e.pmat[1], P, ptab, e.pn[1], e.q100[1]);
        }
        if (fin0) {
            e.pmat[0] = 0xFF;
            e.pn[0] = 0;
            e.q100[0] = 0;
        }
        if (fin1) {
            e.pmat[1] = 0xFF;
            e.pn[1] = 0;
            e.q100[1] = 0;
        }
    }
    if (press_action == 0) {
        if (TRACE) { // (0, None): env_super.py:629-637
            lg->code[lg->n_log] = 0;
            lg->mat[lg->n_log] = -1;
            lg->n_log += 1;
        }
        return;
    }
    int p, mat;
    decode_press_action(press_action, p, mat);
    const int tm = p ? e.timer[1] : e.timer[0];
    if (TRACE) { // busy: (111 | 222, material) env_super.py:725-733; else the action tuple :736
        lg->code[lg->n_log] = tm > 0 ? (p ? 222 : 111) : p + 1;
        lg->mat[lg->n_log] = mat;
        lg->n_log += 1;
    }
    if (tm > 0) return; // busy: no state change (env_super.py:725-733)
    int total = e.ce, tru = 0; // container E: quality 0 (env_super.py:755-757)
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        if (mat == m) {
            total = e.ct[m] + e.cf[m];
            tru = e.ct[m];
            e.ct[m] = 0;
            e.cf[m] = 0;
        }
    }
    if (mat == 4) e.ce = 0;
    e.lps = 1;
    e.lpa = total;
    int q = 0;
    if (tru > 0) q = purity_hundredths(tru, total); // round(x, 2)
    const int pt0 = P.press_time0, pt1 = P.press_time1;
    const int pt = p ? pt1 : pt0;
    if (p) {
        e.timer[1] = pt;
        e.pmat[1] = mat;
        e.pn[1] = total;
        e.q100[1] = q;
    } else {
        e.timer[0] = pt;
        e.pmat[0] = mat;
        e.pn[0] = total;
        e.q100[0] = q;
    }
}

// env_super.py:771-791 get_container_purity per material, in hundredths; [101] marks an empty
// container (its purity is the quality threshold)
__device__ __forceinline__ void container_purity_k(const Env &e, int k[4])
{
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const int total = e.ct[m] + e.cf[m];
        int v = 101;
        if (total > 0) v = purity_hundredths(e.ct[m], total);
        k[m] = v;
    }
}

// LDS-resident tables (image built on the host with the reference's literal expressions).  Everything a
// lane selects by a per-lane index lives here: an indexed read of kernel arguments would be a global
// load, and any vmcnt wait inside the step loop also waits for the previous step's output stores.
struct Tables {
    const float *lvl;    // [capacity+1]  clip(float(L / capacity))                  env_super.py:339-344
    const float *pdiff;  // [4][102]      clip(float(round(k/100 - thr, 2)))         env_super.py:212-227
    const float *timer0; // [press_time+1] clip(float(t / press_time))               env_super.py:354-356
    const float *timer1;
    const double *tanh_s; // [401]        sorting reward by sum of purity hundredths env_super.py:963-1003
    const double *eff;    // [S/2+1]      (1 - 4*(dist/S)) * bale_efficiency_factor  env_super.py:1058-1062
    const uint32_t *pat;  // [3][kPatStride] per stage id, see kPatStride
    const double *acc;    // [3][4]       clip(baseline + boost by mode 0 | 1 | none) env_super.py:499-509
    const double *bonus;  // [4]          target_peaks[min(num_bales,3)] - bale_efficiency_factor  :1065-1069
    const int *press_time; // [2] press times, then [4] the qi_down mask words (bale_quality_int)
    // rarely used fp64 constants live here too: as kernel arguments they were re-fetched with s_load inside
    // the step loop (SGPR pressure), each fetch a scalar-cache round trip
    const double *cst;     // [CST_COUNT], see enum Cst
    const uint64_t *jump;  // [kJumpBits][4] LCG jump-ahead (pcg_jump)
    const uint64_t *back;  // [kRingBackSteps][4] LCG jump BACK by d = 0..31 steps (pcg_step_back)
    const float *gprop;    // [256] general generator mode: clip(float(k / batch))          env_super.py:199-210
    const float *gfrac;    // [256]                         clip(float(k / stage_capacity)) env_super.py:351
};

__device__ __forceinline__ Tables tables_at(const uint32_t *base, const Params &P)
{
    Tables t;
    t.lvl = reinterpret_cast<const float *>(base + P.off_lvl);
    t.pdiff = reinterpret_cast<const float *>(base + P.off_pdiff);
    t.timer0 = reinterpret_cast<const float *>(base + P.off_timer0);
    t.timer1 = reinterpret_cast<const float *>(base + P.off_timer1);
    t.tanh_s = reinterpret_cast<const double *>(base + P.off_tanh);
    t.eff = reinterpret_cast<const double *>(base + P.off_eff);
    t.pat = base + P.off_pat;
    t.acc = reinterpret_cast<const double *>(base + P.off_acc);
    t.bonus = reinterpret_cast<const double *>(base + P.off_bonus);
    t.press_time = reinterpret_cast<const int *>(base + P.off_ptime);
    t.cst = reinterpret_cast<const double *>(base + P.off_cst);
    t.jump = reinterpret_cast<const uint64_t *>(base + P.off_jump);
    t.back = reinterpret_cast<const uint64_t *>(base + P.off_back);
    t.gprop = reinterpret_cast<const float *>(base + P.off_gprop);
    t.gfrac = reinterpret_cast<const float *>(base + P.off_gfrac);
    return t;
}

__device__ __forceinline__ float clip_f(float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); }

// What a step reads off a stage vector.  GEN = false: the vector is one of three words, carried as its id, and every
// derived quantity is a per-id record of the table image.  GEN = true: carried as its packed counts (u8 x 4); shares
// and fractions come from per-count tables (the batch total is constant), sorting_rules() is evaluated literally.
template <bool GEN>
struct Stage {
    static __device__ __forceinline__ uint32_t word(int st, const Params &P)
    {
        if (GEN) return (uint32_t)st;
        const uint32_t pw1 = P.pat_word1, pw2 = P.pat_word2;
        return st == 1 ? pw1 : (st == 2 ? pw2 : 0u);
    }
    static __device__ __forceinline__ float occupancy(int st, const Params &P, const Tables &tb)
    {
        if (GEN) return st != 0 ? __uint_as_float(P.occ_nonempty) : 0.0f;
        return __uint_as_float(tb.pat[st * kPatStride + 1]);
    }
    static __device__ __forceinline__ float4 proportions(int st, const Tables &tb) // env_super.py:199-210
    {
        if (GEN) {
            const uint32_t w = (uint32_t)st;
            if (w == 0u) return make_float4(0.f, 0.f, 0.f, 0.f);
            return make_float4(tb.gprop[w & 0xFFu], tb.gprop[(w >> 8) & 0xFFu], tb.gprop[(w >> 16) & 0xFFu], tb.gprop[w >> 24]);
        }
        return *reinterpret_cast<const float4 *>(tb.pat + st * kPatStride + 4);
    }
    static __device__ __forceinline__ float4 fractions(int st, const Tables &tb) // env_super.py:351
    {
        if (GEN) {
            const uint32_t w = (uint32_t)st;
            return make_float4(tb.gfrac[w & 0xFFu], tb.gfrac[(w >> 8) & 0xFFu], tb.gfrac[(w >> 16) & 0xFFu], tb.gfrac[w >> 24]);
        }
        return *reinterpret_cast<const float4 *>(tb.pat + st * kPatStride + 8);
    }
    static __device__ __forceinline__ int rule_mode(int st, const Tables &tb) // env_super.py:469-482 sorting_rules()
    {
        if (GEN) {
            const uint32_t w = (uint32_t)st;
            const int c0 = (int)(w & 0xFFu), c1 = (int)((w >> 8) & 0xFFu), c2 = (int)((w >> 16) & 0xFFu), c3 = (int)(w >> 24);
            const int total = c0 + c1 + c2 + c3;
            if (total == 0) return 1;
            const double t = (double)total;
            const double p0 = (double)c0 / t, p1 = (double)c1 / t, p2 = (double)c2 / t, p3 = (double)c3 / t;
            return (p0 + p2 > p1 + p3) ? 0 : 1;
        }
        return (int)tb.pat[st * kPatStride + 2];
    }
};

// env_super.py:306-325 get_sort_obs -> o[0..12]
template <bool GEN = false>
__device__ __forceinline__ void sort_obs(const Env &e, const Params &P, const Tables &tb, const int k[4], float *o)
{
    o[0] = Stage<GEN>::occupancy(e.st_belt, P, tb); // belt_occupancy = the input occupancy of the batch now on the belt
    const float4 prop = Stage<GEN>::proportions(e.st_belt, tb);
    o[1] = prop.x;
    o[2] = prop.y;
    o[3] = prop.z;
    o[4] = prop.w;
#pragma unroll
    for (int m = 0; m < 4; ++m) o[5 + m] = clip_f((float)e.acc[m], -1.0f, 1.0f);
#pragma unroll
    for (int m = 0; m < 4; ++m) o[9 + m] = tb.pdiff[m * kPdiffStride + k[m]];
}

// env_super.py:327-359 get_press_obs -> o[0..15]
template <bool GEN = false>
__device__ __forceinline__ void press_obs(const Env &e, const Params &P, const Tables &tb, float *o)
{
#pragma unroll
    for (int m = 0; m < 5; ++m) {
        int lvl = level_of(e, m);
        float v = tb.lvl[lvl < P.capacity ? lvl : P.capacity];
        o[m] = v;
        o[5 + m] = v;
    }
    const float4 frac = Stage<GEN>::fractions(e.st_sort, tb);
    o[10] = frac.x;
    o[11] = frac.y;
    o[12] = frac.z;
    o[13] = frac.w;
    o[14] = tb.timer0[e.timer[0]];
    o[15] = tb.timer1[e.timer[1]];
}

template <int KIND>
struct Dims;
template <>
struct Dims<1> {
    static constexpr int D = 13, A = 2;
};
template <>
struct Dims<2> {
    static constexpr int D = 16, A = 11;
};
template <>
struct Dims<3> {
    static constexpr int D = 29, A = 22;
};

template <int KIND, bool GEN = false>
__device__ __forceinline__ void env_obs(const Env &e, const Params &P, const Tables &tb, const int k[4], float *o)
{
    if (KIND == 1) {
        sort_obs<GEN>(e, P, tb, k, o);
    } else if (KIND == 2) {
        press_obs<GEN>(e, P, tb, o);
    } else {
        sort_obs<GEN>(e, P, tb, k, o);
        press_obs<GEN>(e, P, tb, o + 13);
    }
}

// The build's own rule for reset(seed=None) (the reference draws OS entropy there,
// env_super.py:375): pattern order from a hash of the env's stream identity and episode count.
__device__ __forceinline__ int unseeded_gen2(const Env &e)
{
    uint64_t h = mix64(e.rng.i_lo ^ ((uint64_t)e.episode * 0x9E3779B97F4A7C15ull));
    return (int)(h & 1ull); // 1 -> first pattern key is 2
}

// env_super.py:365-420 reset (state part; streams and episode count are handled by the caller)
__device__ __forceinline__ void reset_episode_state(Env &e, const double *cst)
{
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        e.ct[m] = 0;
        e.cf[m] = 0;
        e.acc[m] = cst[CST_BASE_ACC0 + m];
    }
    e.st_in = e.st_belt = e.st_sort = 0;
    e.ce = 0;
    e.pn[0] = e.pn[1] = 0;
    e.lpa = 0;
    e.lps = 0;
    e.timer[0] = e.timer[1] = 0;
    e.pmat[0] = e.pmat[1] = 0xFF;
    e.q100[0] = e.q100[1] = 0;
    e.mode = 0;
    e.gen_idx = 0;
    e.gen_cnt = 0;
    e.step = 0;
}

template <class BALES>
__device__ __forceinline__ void clear_bales(const BALES &bales)
{
#pragma unroll
    for (int m = 0; m < 5; ++m) bales.store(m, make_uint4(0, 0, 0, 0));
}
template <>
__device__ __forceinline__ void clear_bales<BaleNote>(const BaleNote &note)
{
    *note.head |= 4u;
}


// What the reward / observation side needs of an env after its dynamics ran (before any auto-reset).
// One env transition = env_dynamics (state-critical: flow, accuracy, sort_material, presses, flag
// bookkeeping) followed by env_observe (pure function of the snapshot: purities, rewards, observation).
// The single-role kernels run both in one lane; the pipelined rollout kernel runs them in different
// waves and passes the snapshot through LDS.
struct Snap {
    int ct[4], cf[4], ce;
    int timer[2];
    int st_belt, st_sort;
    int lps, lpa;    // _last_press_started / _amount as calculate_press_reward finds them
    int mode;        // 0 | 1 | 2 (no boost): selects accuracy_belt when noise == 0
    double acc[4];   // accuracy_belt (noise > 0)
    int done;        // terminated after this step
    int overflowed;  // check_overflow fired: reward = overflow_termination_penalty
};

struct PenaltyClass {
    bool any_cat, any_sev, any_mild;
};

// fill_ratio brackets of calculate_press_reward (env_super.py:1015-1030) on integer levels
__device__ __forceinline__ PenaltyClass classify_levels(const int lvl[5], const Params &P)
{
    PenaltyClass c{false, false, false};
#pragma unroll
    for (int m = 0; m < 5; ++m) {
        const bool cat = lvl[m] > P.capacity;                  // fill_ratio > 1.0
        const bool sev = !cat && lvl[m] > P.thr_sev;           // elif fill_ratio > 0.95
        const bool mild = !cat && !sev && lvl[m] > P.thr_mild; // elif fill_ratio > 0.90
        c.any_cat |= cat;
        c.any_sev |= sev;
        c.any_mild |= mild;
    }
    return c;
}

// env_1_sort.py:97-154 / env_2_press.py:88-165 / env_monolith.py:109-284 up to (not including) the reward
// and observation, plus the state side effects of calculate_press_reward and the step counter.
template <int KIND, bool NOISE, bool LITERAL, class RNG, bool TRACE = false, bool GEN = false, class BALES = BaleRef>
__device__ __forceinline__ void env_dynamics(Env &e, RNG &rng, const Params &P, const Tables &tb, int action,
                                             int sort_mode_in, uint32_t flags, const BALES &bales, Snap &sn,
                                             Ledger *lg = nullptr)
{
    const bool unmasked = (flags & 1u) != 0;
    const bool check_overflow = (flags & 2u) != 0;
    const bool late = (flags & 8u) != 0; // MSE_STEP_SANITIZE_LATE: Env_3 mode='random' without masking

    // input_action_rules draws rng_input.integers(60,81) and discards it (env_super.py:911-922, :433);
    // that stream is never observed, so it is not carried.
    update_environment<GEN>(e, P);
    const uint32_t sorting_word = Stage<GEN>::word(e.st_sort, P); // current_material_sorting

    int sort_mode, press_action = 0;
    bool run_press_rules = true;
    if (KIND == 1) {
        sort_mode = action;
    } else if (KIND == 2) {
        // the sorting agent's decision, else sorting_rules() on the post-flow belt (env_super.py:469-482)
        sort_mode = sort_mode_in >= 0 ? sort_mode_in : Stage<GEN>::rule_mode(e.st_belt, tb);
        press_action = action;
    } else {
        sort_mode = action >= 11 ? 1 : 0;
        press_action = action - 11 * sort_mode;
        // env_monolith.py:132-138: validated at decode time (pre-sort levels); an invalid action
        // skips press_action_rules altogether, so the timers do not tick this step
        if (unmasked && !late && !press_action_valid(e, P, press_action)) {
            run_press_rules = false;
            if (TRACE) { // invalid_info (111 | 222, name): env_monolith.py:132-136
                lg->code[lg->n_log] = press_action > 5 ? 222 : 111;
                lg->mat[lg->n_log] = (press_action - 1) % 5;
                lg->n_log += 1;
            }
        }
    }

    double acc_sorter[4];
    update_accuracy<NOISE>(e, tb.cst, tb.acc, sort_mode, acc_sorter);
    MSE_TL(e.tl, 1);
    sort_material<LITERAL>(e, rng, sorting_word, acc_sorter);
    MSE_TL(e.tl, 2);

    if (KIND == 1) {
        press_action = sample_masked_press_action(e, P); // env_1_sort.py:125-126 (mask before the tick)
        if (TRACE) lg->internal = press_action;
    } else if (KIND == 2) {
        // env_2_press.py:125-138: validated against post-sort levels; the timers still tick
        if (unmasked && !press_action_valid(e, P, press_action)) {
            if (TRACE) { // env_2_press.py:127-131: the invalid entry, then press_action_rules((None, None)) logs a no-op
                lg->code[lg->n_log] = press_action > 5 ? 222 : 111;
                lg->mat[lg->n_log] = (press_action - 1) % 5;
                lg->n_log += 1;
            }
            press_action = 0;
        }
    } else if (KIND == 3 && unmasked && late) {
        // env_monolith.py:245-253 (mode='random' without masking): sanitised after the sort; invalid -> no tick either
        if (!press_action_valid(e, P, press_action)) {
            run_press_rules = false;
            if (TRACE) {
                lg->code[lg->n_log] = press_action > 5 ? 222 : 111;
                lg->mat[lg->n_log] = (press_action - 1) % 5;
                lg->n_log += 1;
            }
        }
    }
#ifndef MSE_ABL_NOPRESS // (ablation timing builds only: tools/ablate.sh)
    if (run_press_rules) press_action_rules<TRACE, BALES>(e, P, tb.press_time, press_action, bales, lg);
#endif
    MSE_TL(e.tl, 3);

    // snapshot for the observer
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        sn.ct[m] = e.ct[m];
        sn.cf[m] = e.cf[m];
        sn.acc[m] = e.acc[m];
    }
    sn.ce = e.ce;
    sn.timer[0] = e.timer[0];
    sn.timer[1] = e.timer[1];
    sn.st_belt = e.st_belt;
    sn.st_sort = e.st_sort;
    sn.lps = e.lps;
    sn.lpa = e.lpa;
    sn.mode = e.mode == 0 ? 0 : (e.mode == 1 ? 1 : 2);

    // state side of the rewards: overflow termination (env_super.py:900-905) and the clearing of
    // _last_press_* by calculate_press_reward unless it returned a penalty first (env_super.py:1022-1030,1073-1075)
    int lvl[5];
#pragma unroll
    for (int m = 0; m < 5; ++m) lvl[m] = level_of(e, m);
    // Both lower brackets carry a penalty in the reference's config (wave-uniform flags): a step is then penalised iff
    // SOME level is above the lowest threshold, i.e. iff the highest one is (thr_mild <= thr_sev <= capacity) - two
    // three-way maxima and two compares where the bracket-by-bracket form takes fifteen compares
    bool any_cat, penalised;
    if (P.sev_negative && P.mild_negative) {
        const int top = max(max(max(lvl[0], lvl[1]), max(lvl[2], lvl[3])), lvl[4]);
        any_cat = top > P.capacity;
        penalised = top > P.thr_mild;
    } else {
        const PenaltyClass pc = classify_levels(lvl, P);
        any_cat = pc.any_cat;
        penalised = pc.any_cat || (pc.any_sev && P.sev_negative) || (pc.any_mild && P.mild_negative);
    }
    const bool overflowed = check_overflow && any_cat; // a level above capacity
    if (KIND != 1 && !overflowed) {
        if (!penalised) {
            e.lps = 0;
            e.lpa = 0;
        }
    }
    e.step += 1;
    sn.overflowed = overflowed ? 1 : 0;
    sn.done = (overflowed || e.step >= P.max_steps) ? 1 : 0;
    MSE_TL(e.tl, 4);
}

struct StepResult {
    double reward;
    int done;
    double r_sort, r_press; // the two arguments of _log_step_data (env_super.py:928): read by the trace only
};

// Rewards (env_super.py:963-1080) and observation (env_super.py:306-359) of a snapshot; k[] returns the
// container purities in hundredths ([101] = empty).
//
// Table reads are issued in two batches and consumed behind fp64 divisions: at one wave per SIMD an
// LDS read that is waited for right after its issue costs the full round trip.
template <int KIND, bool NOISE, bool GEN = false>
__device__ __forceinline__ StepResult env_observe(const Snap &sn, const Params &P, const Tables &tb, int k[4], float *o)
{
    // ---- batch A: everything that needs only the integer state -----------------------------------
    int lvl[5];
#pragma unroll
    for (int m = 0; m < 4; ++m) lvl[m] = sn.ct[m] + sn.cf[m];
    lvl[4] = sn.ce;
    float lvl_f[5], timer_f[2], occ_f = 0.0f;
    float4 prop = make_float4(0.f, 0.f, 0.f, 0.f), frac = prop;
    if (KIND != 1) { // get_press_obs pieces (env_super.py:327-359)
#pragma unroll
        for (int m = 0; m < 5; ++m) lvl_f[m] = tb.lvl[lvl[m] < P.capacity ? lvl[m] : P.capacity];
        timer_f[0] = tb.timer0[sn.timer[0]];
        timer_f[1] = tb.timer1[sn.timer[1]];
        frac = Stage<GEN>::fractions(sn.st_sort, tb);
    }
    double acc[4];
    if (KIND != 2) { // get_sort_obs pieces (env_super.py:306-325)
        occ_f = Stage<GEN>::occupancy(sn.st_belt, P, tb);
        prop = Stage<GEN>::proportions(sn.st_belt, tb);
#pragma unroll
        for (int m = 0; m < 4; ++m) acc[m] = NOISE ? sn.acc[m] : tb.acc[4 * sn.mode + m];
    }
    // action part of calculate_press_reward (env_super.py:1052-1075), branch-free: without a started
    // press the amount is 0 and the looked-up value is discarded
    double eff_v = 0.0, bonus_v = 0.0;
    if (KIND != 1) {
        const int amount = sn.lps ? sn.lpa : 0;
        int nb, rem;
        divmod_small(amount, P.balesize, P.inv_balesize, nb, rem);
        const int dist = rem < P.balesize - rem ? rem : P.balesize - rem;
        eff_v = tb.eff[dist];
        bonus_v = tb.bonus[nb > 3 ? 3 : nb];
    }

    // ---- purity: four fp64 divisions (they also cover batch A's LDS latency) ----------------------
    // env_super.py:771-791; round(true/total, 2) in hundredths, [101] = empty container
#pragma unroll
    for (int m = 0; m < 4; ++m) k[m] = lvl[m] > 0 ? purity_hundredths(sn.ct[m], lvl[m]) : 101;

    // ---- batch B: reads keyed by the purities -------------------------------------------------------
    float pdiff_f[4] = {0.f, 0.f, 0.f, 0.f};
    double tanh_v = 0.0;
    if (KIND != 2) {
#pragma unroll
        for (int m = 0; m < 4; ++m) pdiff_f[m] = tb.pdiff[m * kPdiffStride + k[m]];
        int s = 0;
#pragma unroll
        for (int m = 0; m < 4; ++m) s += (k[m] == 101) ? P.k_thr[m] : k[m];
        // calculate_sorting_reward (env_super.py:963-1003) tabulated by the integer sum of the four purity
        // hundredths (the literal fp64 sum differs from the tabulated one by < 1e-15)
        tanh_v = tb.tanh_s[s];
    }

    // ---- calculate_press_reward (env_super.py:1006-1080); its division covers batch B --------------
    double rp = 0.0;
    if (KIND != 1) {
        const PenaltyClass pc = classify_levels(lvl, P);
        const int total_level = lvl[0] + lvl[1] + lvl[2] + lvl[3] + lvl[4];
        const double state_reward = ((double)total_level / (double)(5 * P.capacity)) * P.max_state_reward;
        bool penalised = pc.any_cat;
        double penalty = 0.0;
        if (__builtin_expect(pc.any_cat || pc.any_sev || pc.any_mild, 0)) { // rare: constants fetched only here
            double max_pen = 0.0;
            if (pc.any_sev) max_pen = fmin(max_pen, tb.cst[CST_PEN_SEV]);
            if (pc.any_mild) max_pen = fmin(max_pen, tb.cst[CST_PEN_MILD]);
            penalty = pc.any_cat ? tb.cst[CST_PEN_CAT] : max_pen;
            penalised = pc.any_cat || max_pen < 0.0;
        }
        if (penalised) {
            rp = penalty;
        } else {
            const double ar = sn.lps ? eff_v + bonus_v : 0.0;
            const double v = state_reward + ar;
            rp = v < -1.0 ? -1.0 : (v > 1.0 ? 1.0 : v);
        }
    }

    // ---- observation (env_super.py:306-359) ---------------------------------------------------------
    if (KIND != 2) {
        o[0] = occ_f;
        o[1] = prop.x;
        o[2] = prop.y;
        o[3] = prop.z;
        o[4] = prop.w;
#pragma unroll
        for (int m = 0; m < 4; ++m) o[5 + m] = clip_f((float)acc[m], -1.0f, 1.0f);
#pragma unroll
        for (int m = 0; m < 4; ++m) o[9 + m] = pdiff_f[m];
    }
    if (KIND != 1) {
        float *op = KIND == 2 ? o : o + 13;
#pragma unroll
        for (int m = 0; m < 5; ++m) {
            op[m] = lvl_f[m];
            op[5 + m] = lvl_f[m];
        }
        op[10] = frac.x;
        op[11] = frac.y;
        op[12] = frac.z;
        op[13] = frac.w;
        op[14] = timer_f[0];
        op[15] = timer_f[1];
    }

    StepResult r;
    r.done = sn.done;
    r.reward = KIND == 1 ? tanh_v : (KIND == 2 ? rp : tanh_v + rp);
    r.r_sort = KIND == 2 ? 0.0 : tanh_v; // env_1_sort.py:151, env_2_press.py:162, env_monolith.py:282
    r.r_press = KIND == 1 ? 0.0 : rp;
    // env_super.py:900-905 + the variants' early return: the overflow penalty replaces the rewards
    if (__builtin_expect(sn.overflowed != 0, 0)) {
        r.reward = tb.cst[CST_OVERFLOW_PEN];
        r.r_sort = KIND == 3 ? r.reward / 2 : 0.0; // env_monolith.py:271; env_1_sort.py:141, env_2_press.py:152
        r.r_press = KIND == 3 ? r.reward / 2 : r.reward;
    }
    return r;
}

// The reward of env_observe alone, from what is left of a snapshot once the observation is made: the sum of the purity
// hundredths as calculate_sorting_reward takes them (empty container -> its threshold), the amount of a press started
// this step (0 if none), the five levels' sum and penalty classes, the overflow flag.  The same expressions in the same
// order as env_observe (which stays the one place the single-lane kernels compute it): k_rollout_policy_roles hands these
// few words from its actor wave to its critic wave instead of evaluating them on the step's chain.
template <int KIND>
__device__ __forceinline__ double env_reward(int s_sum, bool lps, int amount, int total_level, const PenaltyClass &pc,
                                             bool overflowed, const Params &P, const Tables &tb)
{
    double eff_v = 0.0, bonus_v = 0.0;
    if (KIND != 1) {
        int nb, rem;
        divmod_small(amount, P.balesize, P.inv_balesize, nb, rem);
        const int dist = rem < P.balesize - rem ? rem : P.balesize - rem;
        eff_v = tb.eff[dist];
        bonus_v = tb.bonus[nb > 3 ? 3 : nb];
    }
    double tanh_v = 0.0;
    if (KIND != 2) tanh_v = tb.tanh_s[s_sum];
    double rp = 0.0;
    if (KIND != 1) {
        const double state_reward = ((double)total_level / (double)(5 * P.capacity)) * P.max_state_reward;
        bool penalised = pc.any_cat;
        double penalty = 0.0;
        if (__builtin_expect(pc.any_cat || pc.any_sev || pc.any_mild, 0)) {
            double max_pen = 0.0;
            if (pc.any_sev) max_pen = fmin(max_pen, tb.cst[CST_PEN_SEV]);
            if (pc.any_mild) max_pen = fmin(max_pen, tb.cst[CST_PEN_MILD]);
            penalty = pc.any_cat ? tb.cst[CST_PEN_CAT] : max_pen;
            penalised = pc.any_cat || max_pen < 0.0;
        }
        if (penalised) {
            rp = penalty;
        } else {
            const double ar = lps ? eff_v + bonus_v : 0.0;
            const double v = state_reward + ar;
            rp = v < -1.0 ? -1.0 : (v > 1.0 ? 1.0 : v);
        }
    }
    double reward = KIND == 1 ? tanh_v : (KIND == 2 ? rp : tanh_v + rp);
    if (__builtin_expect(overflowed, 0)) reward = tb.cst[CST_OVERFLOW_PEN];
    return reward;
}

// the snapshot of a freshly reset env (observation after an auto-reset)
__device__ __forceinline__ void snap_of_reset(Snap &sn, const double *cst)
{
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        sn.ct[m] = 0;
        sn.cf[m] = 0;
        sn.acc[m] = cst[CST_BASE_ACC0 + m];
    }
    sn.ce = 0;
    sn.timer[0] = sn.timer[1] = 0;
    sn.st_belt = sn.st_sort = 0;
    sn.lps = 0;
    sn.lpa = 0;
    sn.mode = 2;
    sn.done = 0;
    sn.overflowed = 0;
}

// one env transition in one lane
template <int KIND, bool NOISE, bool LITERAL, bool TRACE = false, bool GEN = false, class BALES = BaleRef>
__device__ __forceinline__ StepResult env_step(Env &e, const Params &P, const Tables &tb, int action, int sort_mode_in,
                                               uint32_t flags, const BALES &bales, int k[4], float *o,
                                               Ledger *lg = nullptr)
{
    Snap sn;
    RngLocal rng{e.rng};
    env_dynamics<KIND, NOISE, LITERAL, RngLocal, TRACE, GEN, BALES>(e, rng, P, tb, action, sort_mode_in, flags, bales, sn, lg);
    return env_observe<KIND, NOISE, GEN>(sn, P, tb, k, o);
}

// the generator's private stream <-> its planes (general generator mode)
__device__ __forceinline__ void load_gen(Env &e, const uint4 *__restrict__ planes, const Params &P, long long i)
{
    const uint4 a = planes[(long long)PL_GEN_STATE * P.n_pad + i], b = planes[(long long)PL_GEN_INC * P.n_pad + i];
    const uint4 x = planes[(long long)PL_GEN_AUX * P.n_pad + i];
    e.gen.s_lo = (uint64_t)a.x | ((uint64_t)a.y << 32);
    e.gen.s_hi = (uint64_t)a.z | ((uint64_t)a.w << 32);
    e.gen.i_lo = (uint64_t)b.x | ((uint64_t)b.y << 32);
    e.gen.i_hi = (uint64_t)b.z | ((uint64_t)b.w << 32);
    e.gen_uint = x.x;
    e.gen_has = (int)x.y;
}
__device__ __forceinline__ void store_gen(const Env &e, uint4 *__restrict__ planes, const Params &P, long long i)
{
    planes[(long long)PL_GEN_STATE * P.n_pad + i] = pack_u64x2(e.gen.s_lo, e.gen.s_hi);
    planes[(long long)PL_GEN_INC * P.n_pad + i] = pack_u64x2(e.gen.i_lo, e.gen.i_hi);
    planes[(long long)PL_GEN_AUX * P.n_pad + i] = make_uint4(e.gen_uint, (uint32_t)e.gen_has, 0, 0);
}

// The build's own rule for the generator a reset(seed=None) builds (the reference draws OS entropy, env_super.py:375):
// default_rng(h) with h the hash that also picks the pattern order (unseeded_gen2).
__device__ __forceinline__ void unseeded_generator(Env &e)
{
    const uint64_t h = mix64(e.rng.i_lo ^ ((uint64_t)e.episode * 0x9E3779B97F4A7C15ull));
    e.gen = pcg_seed(h);
    e.gen_uint = 0;
    e.gen_has = 0;
}

} // namespace mse
