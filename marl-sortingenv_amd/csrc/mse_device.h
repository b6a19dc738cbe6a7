// mse_device.h -- device-side state layout and the per-env state transition (gfx950 / CDNA4).
//
// One env instance per lane.  State lives in HBM as struct-of-arrays PLANES of 16-byte cells:
// plane p, env i  ->  planes[p * n_pad + i]  (uint4), so one wave-instruction moves 64 x 16 B =
// 1 KiB fully coalesced.  See DESIGN.md "Data layout in HBM".
//
// The arithmetic follows the reference (citations = path:line in the reference checkout) and
// numpy 2.2.6's Generator/PCG64/SeedSequence.  All fp64 work is compiled with -ffp-contract=off.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

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
    PL_COUNT
};

// flag bits in PL_MISC2.x byte 3
constexpr uint32_t FL_LAST_PRESS_STARTED = 1u;
constexpr uint32_t FL_GEN_FIRST_IS_2 = 2u;
constexpr uint32_t FL_PRESS_HAS_U32 = 4u;

struct Params {
    long long n;            // envs in this handle
    long long n_pad;        // plane stride (multiple of kBlock)
    long long index_offset; // global index of env 0 (sharded runs)
    int env_kind, max_steps, auto_reset, track_bales;
    int balesize, capacity, stage_capacity, batch;
    int press_time[2];
    int pat[2][4];          // floor(ratio * batch) per pattern, order A..D
    double base_acc[4], boost, noise;
    double thr[4], thr_r2[4];
    double theta, temperature;
    double pen_cat, pen_sev, pen_mild, bef, max_state_reward, overflow_pen, rem_thr;
};

// ------------------------------------------------------------------------------------------
// numpy PCG64 (pcg64.h): 128-bit LCG, XSL-RR output of the NEW state
// ------------------------------------------------------------------------------------------
constexpr uint64_t kMulLo = 0x4385DF649FCCF645ull;
constexpr uint64_t kMulHi = 0x2360ED051FC65DA4ull;

struct Pcg {
    uint64_t s_lo, s_hi, i_lo, i_hi;
};

__device__ __forceinline__ void pcg_advance(Pcg &g)
{
    uint64_t lo = g.s_lo * kMulLo;
    uint64_t hi = __umul64hi(g.s_lo, kMulLo) + g.s_lo * kMulHi + g.s_hi * kMulLo;
    uint64_t rlo = lo + g.i_lo;
    hi += g.i_hi + (rlo < lo ? 1ull : 0ull);
    g.s_lo = rlo;
    g.s_hi = hi;
}

__device__ __forceinline__ uint64_t pcg_output(const Pcg &g)
{
    uint64_t x = g.s_hi ^ g.s_lo;
    unsigned rot = (unsigned)(g.s_hi >> 58);
    return (x >> rot) | (x << ((64u - rot) & 63u));
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

__device__ __forceinline__ uint32_t policy_u32(uint64_t seed, uint64_t env_index, uint64_t t)
{
    uint64_t x = seed ^ (env_index * 0x9E3779B97F4A7C15ull) ^ (t * 0xD1B54A32D192ED03ull);
    return (uint32_t)(mix64(x) >> 32);
}

// index of the k-th (0-based) set bit of `bits`
__device__ __forceinline__ int select_kth_bit(uint32_t bits, int k)
{
    for (int i = 0; i < k; ++i) bits &= bits - 1u;
    return __ffs((int)bits) - 1;
}

// round(np.float64, 2): multiply, rint (half-even), divide
__device__ __forceinline__ double round2(double x) { return rint(x * 100.0) / 100.0; }

// ------------------------------------------------------------------------------------------
// per-env register image
// ------------------------------------------------------------------------------------------
struct Env {
    Pcg rng;            // seed+99: sort_material
    Pcg noise;          // seed+4 : update_accuracy (noise > 0)
    Pcg press;          // seed+3 : Env_1's internal press sampling
    uint32_t press_uint;
    int press_has;
    double acc[4];      // accuracy_belt
    int ct[4], cf[4], ce;
    int pn[2], lpa;
    int in[4], belt[4], sort[4];
    int timer[2], pmat[2], q100[2];
    int mode, lps, gen2, gen_idx, gen_cnt, step;
    uint32_t episode;
};

__device__ __forceinline__ void unpack4(uint32_t w, int v[4])
{
    v[0] = (int)(w & 0xFFu);
    v[1] = (int)((w >> 8) & 0xFFu);
    v[2] = (int)((w >> 16) & 0xFFu);
    v[3] = (int)(w >> 24);
}
__device__ __forceinline__ uint32_t pack4(const int v[4])
{
    return (uint32_t)v[0] | ((uint32_t)v[1] << 8) | ((uint32_t)v[2] << 16) | ((uint32_t)v[3] << 24);
}

template <int KIND, bool NOISE>
__device__ __forceinline__ void load_env(Env &e, const uint4 *__restrict__ planes, long long n_pad, long long i)
{
    uint4 a = planes[PL_RNG_STATE * n_pad + i];
    uint4 b = planes[PL_RNG_INC * n_pad + i];
    uint4 c0 = planes[PL_ACC01 * n_pad + i];
    uint4 c1 = planes[PL_ACC23 * n_pad + i];
    uint4 t = planes[PL_CONT_TRUE * n_pad + i];
    uint4 f = planes[PL_CONT_FALSE * n_pad + i];
    uint4 m0 = planes[PL_MISC0 * n_pad + i];
    uint4 m1 = planes[PL_MISC1 * n_pad + i];
    uint4 m2 = planes[PL_MISC2 * n_pad + i];
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
    unpack4(m1.x, e.in);
    unpack4(m1.y, e.belt);
    unpack4(m1.z, e.sort);
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
        uint4 s = planes[PL_NOISE_STATE * n_pad + i];
        uint4 q = planes[PL_NOISE_INC * n_pad + i];
        e.noise.s_lo = (uint64_t)s.x | ((uint64_t)s.y << 32);
        e.noise.s_hi = (uint64_t)s.z | ((uint64_t)s.w << 32);
        e.noise.i_lo = (uint64_t)q.x | ((uint64_t)q.y << 32);
        e.noise.i_hi = (uint64_t)q.z | ((uint64_t)q.w << 32);
    }
    if (KIND == 1) {
        uint4 s = planes[PL_PRESS_STATE * n_pad + i];
        uint4 q = planes[PL_PRESS_INC * n_pad + i];
        e.press.s_lo = (uint64_t)s.x | ((uint64_t)s.y << 32);
        e.press.s_hi = (uint64_t)s.z | ((uint64_t)s.w << 32);
        e.press.i_lo = (uint64_t)q.x | ((uint64_t)q.y << 32);
        e.press.i_hi = (uint64_t)q.z | ((uint64_t)q.w << 32);
    }
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
__device__ __forceinline__ void store_env(const Env &e, uint4 *__restrict__ planes, long long n_pad, long long i,
                                          bool write_inc)
{
    planes[PL_RNG_STATE * n_pad + i] = pack_u64x2(e.rng.s_lo, e.rng.s_hi);
    if (write_inc) planes[PL_RNG_INC * n_pad + i] = pack_u64x2(e.rng.i_lo, e.rng.i_hi);
    planes[PL_ACC01 * n_pad + i] = pack_f64x2(e.acc[0], e.acc[1]);
    planes[PL_ACC23 * n_pad + i] = pack_f64x2(e.acc[2], e.acc[3]);
    planes[PL_CONT_TRUE * n_pad + i] = make_uint4((uint32_t)e.ct[0], (uint32_t)e.ct[1], (uint32_t)e.ct[2], (uint32_t)e.ct[3]);
    planes[PL_CONT_FALSE * n_pad + i] = make_uint4((uint32_t)e.cf[0], (uint32_t)e.cf[1], (uint32_t)e.cf[2], (uint32_t)e.cf[3]);
    planes[PL_MISC0 * n_pad + i] = make_uint4((uint32_t)e.ce, (uint32_t)e.pn[0], (uint32_t)e.pn[1], (uint32_t)e.lpa);
    uint32_t tw = (uint32_t)e.timer[0] | ((uint32_t)e.timer[1] << 8) | ((uint32_t)(e.pmat[0] & 0xFF) << 16) |
                  ((uint32_t)(e.pmat[1] & 0xFF) << 24);
    planes[PL_MISC1 * n_pad + i] = make_uint4(pack4(e.in), pack4(e.belt), pack4(e.sort), tw);
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

// env_super.py:811-836 validate_press_action
__device__ __forceinline__ bool press_action_valid(const Env &e, const Params &P, int a)
{
    if (a == 0) return true;
    int press = a <= 5 ? 0 : 1;
    int mat = (a - 1) % 5;
    if (e.timer[press] > 0) return false;
    int lvl = 0;
#pragma unroll
    for (int m = 0; m < 5; ++m) lvl = (mat == m) ? level_of(e, m) : lvl;
    return lvl >= P.balesize;
}

// SeasonalInputGenerator.generate_input reduced to material counts (utils/input_generator.py:37-64)
// followed by env_super.py:433-461 update_environment
__device__ __forceinline__ void update_environment(Env &e, const Params &P)
{
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        e.sort[m] = e.belt[m];
        e.belt[m] = e.in[m];
    }
    if (e.gen_cnt >= kGeneratorPeriod) {
        e.gen_idx ^= 1;
        e.gen_cnt = 0;
    }
    int pat2 = e.gen_idx ^ e.gen2; // 1 -> pattern key 2
#pragma unroll
    for (int m = 0; m < 4; ++m) e.in[m] = pat2 ? P.pat[1][m] : P.pat[0][m];
    e.gen_cnt += 1;
}

// env_super.py:484-509 set_multisensor_mode + update_accuracy; acc_sorter gets the OLD accuracy_belt
template <bool NOISE>
__device__ __forceinline__ void update_accuracy(Env &e, const Params &P, int mode, double acc_sorter[4])
{
#pragma unroll
    for (int m = 0; m < 4; ++m) acc_sorter[m] = e.acc[m];
    e.mode = mode;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        double a = P.base_acc[m];
        bool boosted = (mode == 0 && (m == 0 || m == 2)) || (mode == 1 && (m == 1 || m == 3));
        if (boosted) a = a + P.boost;
        double nz;
        if (NOISE) {
            // Generator.uniform(-n, n): low + (high-low)*random(), separately rounded
            double range = P.noise - (-P.noise);
            double u = u64_to_unit_double(pcg_next64(e.noise));
            double scaled = range * u;
            nz = (-P.noise) + scaled;
        } else {
            nz = 0.0; // uniform(-0, 0) = -0 + 0*u = 0 (the stream still advances in the reference, unobservably)
        }
        double v = a + nz;
        e.acc[m] = v < 0.0 ? 0.0 : (v > 1.0 ? 1.0 : v);
    }
}

__device__ __forceinline__ int sel4(const int v[4], int k)
{
    return k == 0 ? v[0] : (k == 1 ? v[1] : (k == 2 ? v[2] : v[3]));
}
__device__ __forceinline__ double sel4d(const double v[4], int k)
{
    return k == 0 ? v[0] : (k == 1 ? v[1] : (k == 2 ? v[2] : v[3]));
}

// Generator.choice(4, p=leftover/total), literal fp64 evaluation (numpy _generator.pyx):
// cdf = cumsum(p); cdf /= cdf[-1]; idx = searchsorted(cdf, u, 'right')
__device__ __forceinline__ int choice4_literal(const int l[4], int total, uint64_t r64)
{
    double T = (double)total;
    double p0 = (double)l[0] / T, p1 = (double)l[1] / T, p2 = (double)l[2] / T, p3 = (double)l[3] / T;
    double c0 = p0;
    double c1 = c0 + p1;
    double c2 = c1 + p2;
    double c3 = c2 + p3;
    double n0 = c0 / c3, n1 = c1 / c3, n2 = c2 / c3, n3 = c3 / c3;
    double u = u64_to_unit_double(r64);
    return (n0 <= u ? 1 : 0) + (n1 <= u ? 1 : 0) + (n2 <= u ? 1 : 0) + (n3 <= u ? 1 : 0);
}

// Exact decision of the same draw without fp64: with U = r64 >> 11 (u = U / 2^53) the literal cdf
// compare `cdf_k <= u` equals `c_k <= floor(u*T)` (c_k = integer prefix sums of leftover) unless
// u*T lies within ~2^-41 of an integer, which is excluded with a wide margin by looking at the
// 32 fractional bits f of (r64 >> 32) * T; only then the literal path runs.  DESIGN.md "choice".
__device__ __forceinline__ int choice4(const int l[4], int total, uint64_t r64, bool literal_only)
{
    uint64_t prod = (uint64_t)(uint32_t)(r64 >> 32) * (uint64_t)(uint32_t)total;
    uint32_t f = (uint32_t)prod;
    int v = (int)(prod >> 32);
    bool safe = (f >= 16u) && (f < 0xFFFFFE00u);
    if (literal_only || !safe) return choice4_literal(l, total, r64);
    int c0 = l[0], c1 = c0 + l[1], c2 = c1 + l[2];
    return (c0 <= v ? 1 : 0) + (c1 <= v ? 1 : 0) + (c2 <= v ? 1 : 0);
}

// env_super.py:511-609 sort_material.  The four stations are walked as one flattened loop so a
// wave runs max-over-lanes(total draws) iterations instead of the sum of per-station maxima.
template <bool LITERAL>
__device__ __forceinline__ void sort_material(Env &e, const double acc_sorter[4])
{
    int l[4] = {e.sort[0], e.sort[1], e.sort[2], e.sort[3]};
    int tr[4] = {0, 0, 0, 0}, fa[4] = {0, 0, 0, 0};
    int st = 0, rem = 0;
    for (;;) {
        while (rem == 0 && st < 4) {
            int target = sel4(l, st);
            double prod = (double)target * sel4d(acc_sorter, st);
            int t = (int)rint(prod); // int(round(np.float64)) : half to even (env_super.py:539)
            int f = target - t;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (k == st) {
                    tr[k] = t;
                    fa[k] = f;
                    l[k] = f;
                }
            }
            rem = f;
            ++st;
        }
        if (rem <= 0) break; // st == 4 and nothing left to draw (a negative count draws nothing)
        int total = l[0] + l[1] + l[2] + l[3];
        if (total == 0) { // env_super.py:557-559
            rem = 0;
            continue;
        }
        uint64_t r = pcg_next64(e.rng);
        int sel = choice4(l, total, r, LITERAL);
#pragma unroll
        for (int k = 0; k < 4; ++k) l[k] -= (k == sel) ? 1 : 0;
        --rem;
    }
    e.ce += l[0] + l[1] + l[2] + l[3];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        e.ct[m] += tr[m];
        e.cf[m] += fa[m];
    }
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

struct BaleCell {
    uint32_t count, sum, last_size, last_q;
};

// env_super.py:661-687 press_bale on the O(1) ledger summary
__device__ __forceinline__ void press_bale(uint4 *__restrict__ planes, long long n_pad, long long i, const Params &P,
                                           int mat, int n, int q100)
{
    uint4 *cell = &planes[(long long)(PL_BALE0 + mat) * n_pad + i];
    uint4 c = *cell;
    double q = (double)q100 / 100.0;   // the stored quality round(x, 2)
    uint32_t qi = (uint32_t)(int)(q * 100.0); // int(q*100) truncates (0.29 -> 28)
    uint32_t S = (uint32_t)P.balesize;
    uint32_t full = (uint32_t)n / S, rem = (uint32_t)n % S;
    if (full > 0) {
        c.x += full;
        c.y += full * S;
        c.z = S;
        c.w = qi;
    }
    if (rem > 0) {
        if ((double)rem > (double)S * P.rem_thr) {
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
    *cell = c;
}

// env_super.py:626-640 press_action_rules = check_press_status (:642-659) then use_press (:722-769)
__device__ __forceinline__ void press_action_rules(Env &e, const Params &P, int press_action,
                                                   uint4 *__restrict__ planes, long long i)
{
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        if (e.timer[p] > 0) {
            e.timer[p] -= 1;
            if (e.timer[p] == 0) {
                if (P.track_bales) press_bale(planes, P.n_pad, i, P, e.pmat[p], e.pn[p], e.q100[p]);
                e.pmat[p] = 0xFF;
                e.pn[p] = 0;
                e.q100[p] = 0;
            }
        }
    }
    if (press_action == 0) return;
    int p = press_action <= 5 ? 0 : 1;
    int mat = (press_action - 1) % 5;
    if (e.timer[p] > 0) return; // busy: no state change (env_super.py:725-733)
    int total = e.ce, tru = 0; // container E: quality 0 (env_super.py:755-757)
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        if (mat == m) {
            total = e.ct[m] + e.cf[m];
            tru = e.ct[m];
        }
    }
    e.lps = 1;
    e.lpa = total;
    int q = 0;
    if (mat < 4 && total > 0) q = (int)rint(((double)tru / (double)total) * 100.0); // round(x,2) in hundredths
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        if (mat == m) {
            e.ct[m] = 0;
            e.cf[m] = 0;
        }
    }
    if (mat == 4) e.ce = 0;
    e.timer[p] = P.press_time[p];
    e.pmat[p] = mat;
    e.pn[p] = total;
    e.q100[p] = q;
}

// env_super.py:771-791 get_container_purity, per material, as round(., 2) doubles
__device__ __forceinline__ void container_purity(const Env &e, const Params &P, double purity[4])
{
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        int total = e.ct[m] + e.cf[m];
        purity[m] = total > 0 ? round2((double)e.ct[m] / (double)total) : P.thr_r2[m];
    }
}

// env_super.py:963-1003 calculate_sorting_reward
__device__ __forceinline__ double sorting_reward(const Params &P, const double purity[4])
{
    double total = 0.0;
#pragma unroll
    for (int m = 0; m < 4; ++m) total = total + (purity[m] - P.theta);
    double state_based = (total / 4.0) * 2.0;
    return tanh(state_based / P.temperature);
}

// env_super.py:1006-1080 calculate_press_reward
__device__ __forceinline__ double press_reward(Env &e, const Params &P)
{
    double cap = (double)P.capacity;
    double max_pen = 0.0;
    bool catastrophic = false;
    int total_level = 0;
#pragma unroll
    for (int m = 0; m < 5; ++m) {
        int lvl = level_of(e, m);
        total_level += lvl;
        double fill = (double)lvl / cap;
        if (fill > 1.0)
            catastrophic = true;
        else if (fill > 0.95)
            max_pen = fmin(max_pen, P.pen_sev);
        else if (fill > 0.90)
            max_pen = fmin(max_pen, P.pen_mild);
    }
    if (catastrophic) return P.pen_cat;
    if (max_pen < 0.0) return max_pen;
    double state_reward = ((double)total_level / (double)(5 * P.capacity)) * P.max_state_reward;
    double action_reward = 0.0;
    if (e.lps) {
        int S = P.balesize;
        int nb = e.lpa / S, rem = e.lpa % S;
        int dist = rem < S - rem ? rem : S - rem;
        double eff = (1.0 - 4.0 * ((double)dist / (double)S)) * P.bef;
        double peak = nb <= 0 ? 0.0 : (nb == 1 ? 1.0 / 3.0 : (nb == 2 ? 2.0 / 3.0 : 1.0));
        double bonus = peak - P.bef;
        action_reward = eff + bonus;
        e.lps = 0;
        e.lpa = 0;
    }
    double r = state_reward + action_reward;
    return r < -1.0 ? -1.0 : (r > 1.0 ? 1.0 : r);
}

__device__ __forceinline__ float clip_f(float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); }

// env_super.py:306-325 get_sort_obs -> o[0..12]
__device__ __forceinline__ void sort_obs(const Env &e, const Params &P, const double purity[4], float *o)
{
    int total = e.belt[0] + e.belt[1] + e.belt[2] + e.belt[3];
    o[0] = clip_f((float)((double)total / 100.0), -1.0f, 1.0f); // belt_occupancy = last input_occupancy
#pragma unroll
    for (int m = 0; m < 4; ++m)
        o[1 + m] = total > 0 ? clip_f((float)((double)e.belt[m] / (double)total), -1.0f, 1.0f) : 0.0f;
#pragma unroll
    for (int m = 0; m < 4; ++m) o[5 + m] = clip_f((float)e.acc[m], -1.0f, 1.0f);
#pragma unroll
    for (int m = 0; m < 4; ++m) o[9 + m] = clip_f((float)round2(purity[m] - P.thr[m]), -1.0f, 1.0f);
}

// env_super.py:327-359 get_press_obs -> o[0..15]
__device__ __forceinline__ void press_obs(const Env &e, const Params &P, float *o)
{
    double cap = (double)P.capacity;
#pragma unroll
    for (int m = 0; m < 5; ++m) {
        float v = clip_f((float)((double)level_of(e, m) / cap), 0.0f, 1.0f);
        o[m] = v;
        o[5 + m] = v;
    }
#pragma unroll
    for (int m = 0; m < 4; ++m) o[10 + m] = clip_f((float)((double)e.sort[m] / (double)P.stage_capacity), 0.0f, 1.0f);
#pragma unroll
    for (int p = 0; p < 2; ++p) o[14 + p] = clip_f((float)((double)e.timer[p] / (double)P.press_time[p]), 0.0f, 1.0f);
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

template <int KIND>
__device__ __forceinline__ void env_obs(const Env &e, const Params &P, const double purity[4], float *o)
{
    if (KIND == 1) {
        sort_obs(e, P, purity, o);
    } else if (KIND == 2) {
        press_obs(e, P, o);
    } else {
        sort_obs(e, P, purity, o);
        press_obs(e, P, o + 13);
    }
}

// The build's own rule for reset(seed=None) (the reference draws OS entropy there,
// env_super.py:375): pattern order from a hash of the env's stream identity and episode count.
__device__ __forceinline__ int unseeded_gen2(const Env &e)
{
    uint64_t h = mix64(e.rng.i_lo ^ ((uint64_t)e.episode * 0x9E3779B97F4A7C15ull));
    return (int)(h & 1ull); // 1 -> first pattern key is 2
}

// env_super.py:365-420 reset (state part; streams are handled by the caller)
__device__ __forceinline__ void reset_episode_state(Env &e, const Params &P)
{
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        e.ct[m] = 0;
        e.cf[m] = 0;
        e.in[m] = 0;
        e.belt[m] = 0;
        e.sort[m] = 0;
        e.acc[m] = P.base_acc[m];
    }
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

__device__ __forceinline__ void clear_bales(uint4 *__restrict__ planes, long long n_pad, long long i)
{
#pragma unroll
    for (int m = 0; m < 5; ++m) planes[(long long)(PL_BALE0 + m) * n_pad + i] = make_uint4(0, 0, 0, 0);
}

struct StepResult {
    double reward;
    int done;
};

// One env transition: env_1_sort.py:97-154 / env_2_press.py:88-165 / env_monolith.py:109-284.
// `purity` returns the post-step container purities for the observation.
template <int KIND, bool NOISE, bool LITERAL>
__device__ __forceinline__ StepResult env_step(Env &e, const Params &P, int action, int sort_mode_in, uint32_t flags,
                                               uint4 *__restrict__ planes, long long i, double purity[4])
{
    const bool unmasked = (flags & 1u) != 0;
    const bool check_overflow = (flags & 2u) != 0;

    // input_action_rules draws rng_input.integers(60,81) and discards it (env_super.py:911-922, :433);
    // that stream is never observed, so it is not carried.
    update_environment(e, P);

    int sort_mode, press_action = 0;
    bool run_press_rules = true;
    if (KIND == 1) {
        sort_mode = action;
    } else if (KIND == 2) {
        if (sort_mode_in >= 0) {
            sort_mode = sort_mode_in;
        } else { // env_super.py:469-482 sorting_rules on the post-flow belt
            int total = e.belt[0] + e.belt[1] + e.belt[2] + e.belt[3];
            double pr[4];
#pragma unroll
            for (int m = 0; m < 4; ++m) pr[m] = total > 0 ? (double)e.belt[m] / (double)total : 0.0;
            sort_mode = (pr[0] + pr[2] > pr[1] + pr[3]) ? 0 : 1;
        }
        press_action = action;
    } else {
        sort_mode = action / 11;
        press_action = action - 11 * sort_mode;
        // env_monolith.py:132-138: validated at decode time (pre-sort levels); an invalid action
        // skips press_action_rules altogether, so the timers do not tick this step
        if (unmasked && !press_action_valid(e, P, press_action)) run_press_rules = false;
    }

    double acc_sorter[4];
    update_accuracy<NOISE>(e, P, sort_mode, acc_sorter);
    sort_material<LITERAL>(e, acc_sorter);

    if (KIND == 1) {
        press_action = sample_masked_press_action(e, P); // env_1_sort.py:125-126 (mask before the tick)
    } else if (KIND == 2) {
        // env_2_press.py:125-138: validated against post-sort levels; the timers still tick
        if (unmasked && !press_action_valid(e, P, press_action)) press_action = 0;
    }
    if (run_press_rules) press_action_rules(e, P, press_action, planes, i);

    StepResult r;
    if (check_overflow) { // env_super.py:900-905 + the variants' early return
        bool over = false;
#pragma unroll
        for (int m = 0; m < 5; ++m) over = over || (level_of(e, m) > P.capacity);
        if (over) {
            container_purity(e, P, purity);
            e.step += 1;
            r.reward = P.overflow_pen;
            r.done = 1;
            return r;
        }
    }
    container_purity(e, P, purity);
    if (KIND == 1) {
        r.reward = sorting_reward(P, purity);
    } else if (KIND == 2) {
        r.reward = press_reward(e, P);
    } else {
        double rs = sorting_reward(P, purity);
        double rp = press_reward(e, P);
        r.reward = rs + rp;
    }
    e.step += 1;
    r.done = e.step >= P.max_steps ? 1 : 0;
    return r;
}

} // namespace mse
