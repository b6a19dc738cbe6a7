// mse_policy_device.h -- the actor-critic MLP of the reference's (Maskable)PPO policy on the f32 matrix cores, as one
// device function per 32-env tile, shared by the standalone forward (mse_policy.hip: k_policy_mlp) and the fused
// learned-policy rollout (mse_lib.hip: k_rollout_policy) so that both produce bit-identical numbers.
//
// Network (src/training.py:115: net_arch=dict(pi=[32, 32], vf=[32, 32]), tanh; MaskableActorCriticPolicy): two
// separate 2x32 tanh MLPs on the observation, a linear action head (32 -> A) and a linear value head (32 -> 1);
// invalid actions get logit -1e8 before the softmax (sb3_contrib MaskableCategorical).
//
// One wavefront evaluates a tile of 32 envs with v_mfma_f32_32x32x2_f32 (f32 in, f32 accumulate = an fmaf chain).
// Outputs (hidden units / actions) are the MFMA rows, envs the columns.  C/D layout: lane l = (half h = l >> 5,
// column j = l & 31), register r holds row row_of(r, h) = (r & 3) + 8 (r >> 2) + 4 h.  The next layer sums over those
// rows, so it takes accumulator register s as the B operand of k-step s - no lane movement, no LDS, no transpose -
// provided the weights (A operand) come in the matching permuted k order; the host packs them that way.
//
// tanh is folded into the weights: tanh(p) = 1 - 2 r with r = 1 / (2^(c p) + 1), c = 2 log2 e.  The host scales each
// hidden layer by c and rewrites the following layer for the input r instead of tanh (W' = -2 W, b' = b + W 1), so a
// hidden unit costs v_exp_f32, v_add_f32, v_rcp_f32 and nothing else.
//
// Two arithmetic forms of the matrix products, same layout and epilogues:
//   F16X3 = false  v_mfma_f32_32x32x2_f32: exact f32 products (an fmaf chain); 64 flop / clk / SIMD - the f32 matrix
//                  rate is the vector rate, and at 10 240 flop per env it bounds the rollout at ~15 G env-steps/s;
//   F16X3 = true   every f32 operand split as hi + lo with hi = f16(x) (round to zero) and lo = f16(x - hi): the pair
//                  carries 22 bits of x (MFMA keeps f16 subnormals - tools/ubench/mfma_f16_denorm.hip - so a tiny
//                  remainder is not lost), and a product is a_hi b_hi + a_hi b_lo + a_lo b_hi on
//                  v_mfma_f32_32x32x16_f16 (exact f16 x f16 products, f32 accumulate): three MFMAs of 16 k-steps
//                  instead of sixteen of one, dropping only the a_lo b_lo term (2^-22 relative).  Measured against
//                  the exact form: logits within ~1e-6 (tests/test_gpu_policy.py), an order below the 2e-5 the
//                  parity test allows against PyTorch fp32.  Needs |folded weight| < 65 504 (checked at create).
//
// Sampling: inverse cdf of the engine's counter-based stream over the softmax masses in REGISTER order - half 0's
// rows (0-3, 8-11, 16-19, ...) first, then half 1's (4-7, 12-15, 20-23, ...) - an exact categorical sample whose
// walk needs no lane movement beyond two half-swaps.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace msep {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kHidden = 32;
constexpr int kTile = 32; // envs per MFMA tile
// packed blob, in floats (pack_weights in mse_policy.hip); layers: 0 actor-1, 1 actor-2, 2 action head, 3 critic-1,
// 4 critic-2:
//   B   [5 layers][2 halves][16]           bias = accumulator init: register r of half h = row row_of(r, h)
//   WV  [2 halves][16], BV                 value head on the critic's second hidden layer
//   W   [5 layers][4 groups][64 lanes][4]  f32 A operands, k-step s = 4 group + slot
//   W16 [2: hi, lo][5 layers][2 chunks][64 lanes][8 halves]   f16 A operands, k-step s = 8 chunk + slot
// The LDS image a kernel keeps is the common head [0, kOffW) followed by ONE of the two operand forms (same size).
constexpr int kOffB = 0;
constexpr int kOffWV = kOffB + 5 * 32;
constexpr int kOffBV = kOffWV + 32;
constexpr int kOffW = (kOffBV + 1 + 3) / 4 * 4;
constexpr int kOperandFloats = 5 * 16 * 64;          // either form: 20 KB
constexpr int kOffW16 = kOffW + kOperandFloats;      // in the blob only
constexpr int kBlobFloats = kOffW16 + kOperandFloats;
constexpr int kLdsFloats = kOffW + kOperandFloats;   // 21 264 bytes

__host__ __device__ constexpr int row_of(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }
// accumulator registers that can hold an action row < A in either half (A = 22 -> 12, 11 -> 7, 2 -> 2)
__host__ __device__ constexpr int regs_for_actions(int A)
{
    int n = 0;
    for (int r = 0; r < 16; ++r)
        if (row_of(r, 0) < A || row_of(r, 1) < A) n = r + 1;
    return n;
}

typedef __attribute__((address_space(3))) const volatile f32x4 *lds_f4;

// values of the other half / both halves of a column: v_permlane32_swap exchanges the upper half of its first
// operand with the lower half of its second, so swap(v, v) = {half-0 value in every lane, half-1 value in every lane}
struct Halves {
    float lo, hi;
};
__device__ __forceinline__ Halves both_halves(float v)
{
    const uint32_t u = __float_as_uint(v);
    auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return Halves{__uint_as_float(r[0]), __uint_as_float(r[1])};
}
__device__ __forceinline__ void both_halves_u32(uint32_t v, uint32_t &lo, uint32_t &hi)
{
    auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
    lo = r[0];
    hi = r[1];
}

__device__ __forceinline__ float max3f(float a, float b, float c)
{
    float d;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}

// r = 1 / (2^z + 1): the hidden activation in its folded form (see the header comment)
__device__ __forceinline__ float sigmoid2(float z)
{
    return __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(z) + 1.0f);
}

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __fp16 pk16x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) const volatile u32x4 *lds_u4;

__device__ __forceinline__ uint32_t pack_rtz(float a, float b)
{
    return __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(a, b));
}

// in[16] f32 -> the two B operands (k-steps 0-7 and 8-15) of the hi and of the lo parts, 8 halves per operand.
// hi = f16(x) toward zero (one v_cvt_pkrtz per pair); lo = f16(x - hi) to nearest, the subtraction and the conversion
// in one v_fma_mix{lo,hi}_f16 each (f32 fma of the f16 hi half, -1 and x, result written as a half): 1.5 instructions
// per value.  |x - hi - lo| <= 2^-21 |x| (or half an f16 subnormal step, 3e-8).
__device__ __forceinline__ void split16(const float *in, u32x4 hi[2], u32x4 lo[2])
{
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const uint32_t hp = pack_rtz(in[2 * q], in[2 * q + 1]);
        uint32_t lp;
        asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]\n\t"
            "v_fma_mixhi_f16 %0, %1, -1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
            : "=&v"(lp)
            : "v"(hp), "v"(in[2 * q]), "v"(in[2 * q + 1]));
        hi[q >> 2][q & 3] = hp;
        lo[q >> 2][q & 3] = lp;
    }
}

// The B operands of one layer for T tiles.  The f32 form keeps the 16 inputs as they are; the f16x3 form splits them.
template <bool F16X3, int T>
struct Operands;
template <int T>
struct Operands<false, T> {
    float v[T][16];
    __device__ __forceinline__ void set(int t, const float *in)
    {
#pragma unroll
        for (int r = 0; r < 16; ++r) v[t][r] = in[r];
    }
};
template <int T>
struct Operands<true, T> {
    u32x4 hi[T][2], lo[T][2];
    __device__ __forceinline__ void set(int t, const float *in) { split16(in, hi[t], lo[t]); }
};

// One layer for T tiles: acc[t] = bias; acc[t] += W in[t].  The A operands (weights) are read from LDS once and serve
// every tile; the T accumulation chains are independent, so one tile's MFMAs run behind the other's.
// f32 form: 16 k-steps of v_mfma_f32_32x32x2_f32.  f16x3 form: per 16-k chunk  W_lo in_hi + W_hi in_lo + W_hi in_hi
// (small terms first) on v_mfma_f32_32x32x16_f16.
template <int T>
__device__ __forceinline__ void bias_init(lds_f4 wl, int h, int L, f32x16 acc[T])
{
    lds_f4 b = wl + (kOffB + (L * 2 + h) * 16) / 4; // the same 64 bytes for every lane of a half: LDS broadcast
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const f32x4 v = b[g];
#pragma unroll
        for (int t = 0; t < T; ++t) {
            acc[t][4 * g + 0] = v[0];
            acc[t][4 * g + 1] = v[1];
            acc[t][4 * g + 2] = v[2];
            acc[t][4 * g + 3] = v[3];
        }
    }
}
template <int T>
__device__ __forceinline__ void apply_layer(lds_f4 wl, int lane, int h, int L, const Operands<false, T> &op, f32x16 acc[T])
{
    bias_init<T>(wl, h, L, acc);
    lds_f4 w = wl + (kOffW + L * 16 * 64) / 4 + lane; // [group][lane]: 16 bytes per lane, conflict-free
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const f32x4 a = w[g * 64];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#pragma unroll
            for (int t = 0; t < T; ++t)
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q], op.v[t][4 * g + q], acc[t], 0, 0, 0);
        }
    }
}
template <int T>
__device__ __forceinline__ void apply_layer(lds_f4 wl, int lane, int h, int L, const Operands<true, T> &op, f32x16 acc[T])
{
    bias_init<T>(wl, h, L, acc);
    lds_u4 wh = (lds_u4)(wl + kOffW / 4) + (L * 2) * 64 + lane; // [hi | lo][layer][chunk][lane]: 16 bytes per lane
    lds_u4 wlo = wh + 5 * 2 * 64;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const f16x8 ah = __builtin_bit_cast(f16x8, (u32x4)wh[c * 64]);
        const f16x8 al = __builtin_bit_cast(f16x8, (u32x4)wlo[c * 64]);
#pragma unroll
        for (int t = 0; t < T; ++t)
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, __builtin_bit_cast(f16x8, op.hi[t][c]), acc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < T; ++t)
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, __builtin_bit_cast(f16x8, op.lo[t][c]), acc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < T; ++t)
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, __builtin_bit_cast(f16x8, op.hi[t][c]), acc[t], 0, 0, 0);
    }
}

struct TileOut {
    int action;   // valid in both halves of the column
    float logp, value;
};

// masked softmax + sample of one tile.  lg: the head's accumulator; legal: bit r set iff the action in accumulator
// register r (row row_of(r, h)) exists and may be taken; word: the env's 32-bit draw for this step.
template <int NR>
__device__ __forceinline__ void sample_tile(const f32x16 &lg, int h, uint32_t legal, bool deterministic, uint32_t word,
                                            float *lgm_out, TileOut &out)
{
    // masked logits of the NR registers that can hold an action: illegal -> -1e8 (sb3_contrib's HUGE_NEG)
    float lgm[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const uint32_t keep = (uint32_t)((int32_t)(legal << (31 - r)) >> 31); // all ones iff bit r
        lgm[r] = __uint_as_float((__float_as_uint(lg[r]) & keep) | (0xCCBEBC20u & ~keep)); // -1e8f
    }
    if (lgm_out != nullptr) {
#pragma unroll
        for (int r = 0; r < NR; ++r) lgm_out[r] = lgm[r];
    }
    float m = lgm[0];
#pragma unroll
    for (int r = 1; r + 1 < NR; r += 2) m = max3f(m, lgm[r], lgm[r + 1]);
    if ((NR & 1) == 0) m = fmaxf(m, lgm[NR - 1]);
    {
        const Halves hm = both_halves(m);
        m = fmaxf(hm.lo, hm.hi);
    }
    // softmax masses and their running sums in register order
    const float nm = -m * 1.44269504088896340736f;
    float c[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const float e = __builtin_amdgcn_exp2f(fmaf(lgm[r], 1.44269504088896340736f, nm)); // exp(lgm - m); illegal -> 0
        c[r] = r == 0 ? e : c[r - 1] + e;
    }
    const Halves S = both_halves(c[NR - 1]); // S.lo = mass of half 0's actions, S.hi = half 1's
    const float total = S.lo + S.hi;
    int idx;
    float la;
    bool take_hi;
    if (deterministic) { // wave-uniform
        // argmax, ties to the lowest action index: rows ascend with r inside a lane
        idx = 0;
        la = lgm[0];
#pragma unroll
        for (int r = 1; r < NR; ++r) {
            const bool better = lgm[r] > la;
            idx = better ? r : idx;
            la = better ? lgm[r] : la;
        }
        const Halves b = both_halves(la);
        uint32_t i_lo, i_hi;
        both_halves_u32((uint32_t)idx, i_lo, i_hi);
        take_hi = b.hi > b.lo || (b.hi == b.lo && row_of((int)i_hi, 1) < row_of((int)i_lo, 0));
    } else {
        const float target = (float)(word >> 8) * 5.9604644775390625e-8f * total; // u in [0, 1) times the mass
        take_hi = !(target < S.lo) && S.hi > 0.0f;
        float tl = h ? target - S.lo : target;
        tl = fminf(tl, c[NR - 1] * 0.99999988079071044921875f); // rounding may not carry the target past the mass
        idx = NR - 1;
        la = lgm[NR - 1];
#pragma unroll
        for (int r = NR - 2; r >= 0; --r) { // descending: the first register whose running sum exceeds the target
            const bool gt = c[r] > tl;
            idx = gt ? r : idx;
            la = gt ? lgm[r] : la;
        }
    }
    const int a_mine = (idx & 3) + 8 * (idx >> 2) + 4 * h;
    uint32_t a_lo, a_hi;
    both_halves_u32((uint32_t)a_mine, a_lo, a_hi);
    const Halves l2 = both_halves(la);
    out.action = (int)(take_hi ? a_hi : a_lo);
    const float la_f = take_hi ? l2.hi : l2.lo;
    out.logp = (la_f - m) - __builtin_amdgcn_logf(total) * 0.693147180559945309417f; // v_log_f32 = log2
}

// value head of one tile: the critic's second hidden layer (pre-activation c2) . wv + bv
__device__ __forceinline__ float value_tile(lds_f4 wl, int h, const f32x16 &c2)
{
    lds_f4 wv = wl + (kOffWV + h * 16) / 4;
    float v = 0.0f;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const f32x4 w4 = wv[g];
#pragma unroll
        for (int q = 0; q < 4; ++q) v = fmaf(sigmoid2(c2[4 * g + q]), w4[q], v);
    }
    const Halves hv = both_halves(v);
    const float bv = (*(wl + kOffBV / 4))[0];
    return (hv.lo + hv.hi) + bv;
}

// T tiles through the actor-critic network.  x[t][16]: layer-1 B operands of this lane for tile t (register s =
// observation entry 2 s + h of env j); legal[t], word[t]: see sample_tile; lgm_out (optional): [T][NR].
// The arithmetic of a tile does not depend on T or on the order below (every tile's operations are its own); the order
// only decides what overlaps: with one tile the critic's and the actor's layers alternate, with two the tiles do.
template <int NR, bool F16X3, int T>
__device__ __forceinline__ void policy_tiles(lds_f4 wl, int lane, const float (*x)[16], const uint32_t *legal, bool deterministic,
                                             const uint32_t *word, float *lgm_out, TileOut *out)
{
    const int h = lane >> 5;
    Operands<F16X3, T> xin; // both networks read the observation: split it once
#pragma unroll
    for (int t = 0; t < T; ++t) xin.set(t, x[t]);
    f32x16 lg[T];
    if (T == 1) {
        Operands<F16X3, T> opc, opa;
        float hc[16], ha[16];
        f32x16 c1[T], a1[T], c2[T], a2[T];
        apply_layer<T>(wl, lane, h, 3, xin, c1);
        apply_layer<T>(wl, lane, h, 0, xin, a1);
#pragma unroll
        for (int r = 0; r < 16; ++r) hc[r] = sigmoid2(c1[0][r]);
        opc.set(0, hc);
        apply_layer<T>(wl, lane, h, 4, opc, c2);
#pragma unroll
        for (int r = 0; r < 16; ++r) ha[r] = sigmoid2(a1[0][r]);
        opa.set(0, ha);
        apply_layer<T>(wl, lane, h, 1, opa, a2);
        out[0].value = value_tile(wl, h, c2[0]);
#pragma unroll
        for (int r = 0; r < 16; ++r) ha[r] = sigmoid2(a2[0][r]);
        opa.set(0, ha);
        apply_layer<T>(wl, lane, h, 2, opa, lg);
    } else {
        Operands<F16X3, T> op;
        float hid[16];
        f32x16 acc[T], acc2[T];
        apply_layer<T>(wl, lane, h, 3, xin, acc); // critic
#pragma unroll
        for (int t = 0; t < T; ++t) {
#pragma unroll
            for (int r = 0; r < 16; ++r) hid[r] = sigmoid2(acc[t][r]);
            op.set(t, hid);
        }
        apply_layer<T>(wl, lane, h, 4, op, acc2);
        apply_layer<T>(wl, lane, h, 0, xin, acc); // the actor's first layer runs behind the value heads
#pragma unroll
        for (int t = 0; t < T; ++t) out[t].value = value_tile(wl, h, acc2[t]);
#pragma unroll
        for (int t = 0; t < T; ++t) {
#pragma unroll
            for (int r = 0; r < 16; ++r) hid[r] = sigmoid2(acc[t][r]);
            op.set(t, hid);
        }
        apply_layer<T>(wl, lane, h, 1, op, acc2);
#pragma unroll
        for (int t = 0; t < T; ++t) {
#pragma unroll
            for (int r = 0; r < 16; ++r) hid[r] = sigmoid2(acc2[t][r]);
            op.set(t, hid);
        }
        apply_layer<T>(wl, lane, h, 2, op, lg);
    }
#pragma unroll
    for (int t = 0; t < T; ++t)
        sample_tile<NR>(lg[t], h, legal[t], deterministic, word[t], lgm_out == nullptr ? nullptr : lgm_out + t * NR, out[t]);
}

// The actor alone, argmax of a two-action head, for T tiles: what Env_2_Pressing.step asks its sorting agent
// (sort_agent.predict(sort_obs, deterministic=True), env_2_press.py:101-104).  Same layers, same bits as policy_tiles;
// rows 0 and 1 sit in registers 0 and 1 of half 0, ties go to action 0 as in sample_tile.
template <bool F16X3, int T>
__device__ __forceinline__ void actor_argmax2_tiles(lds_f4 wl, int lane, const float (*x)[16], int *action)
{
    const int h = lane >> 5;
    Operands<F16X3, T> xin, op;
#pragma unroll
    for (int t = 0; t < T; ++t) xin.set(t, x[t]);
    float hid[16];
    f32x16 acc[T], acc2[T];
    apply_layer<T>(wl, lane, h, 0, xin, acc);
#pragma unroll
    for (int t = 0; t < T; ++t) {
#pragma unroll
        for (int r = 0; r < 16; ++r) hid[r] = sigmoid2(acc[t][r]);
        op.set(t, hid);
    }
    apply_layer<T>(wl, lane, h, 1, op, acc2);
#pragma unroll
    for (int t = 0; t < T; ++t) {
#pragma unroll
        for (int r = 0; r < 16; ++r) hid[r] = sigmoid2(acc2[t][r]);
        op.set(t, hid);
    }
    apply_layer<T>(wl, lane, h, 2, op, acc);
#pragma unroll
    for (int t = 0; t < T; ++t) {
        uint32_t lo, hi;
        both_halves_u32(acc[t][1] > acc[t][0] ? 1u : 0u, lo, hi); // half 0 holds rows 0 and 1
        action[t] = (int)lo;
    }
}

// The actor alone with policy_tiles' masked sampling, for T tiles: the action half of the two-role policy rollout
// (mse_lib.hip: k_rollout_policy_roles), whose critic runs on a partner wave.  Same layers, same bits as policy_tiles;
// out[t].value is left alone.
template <int NR, bool F16X3, int T>
__device__ __forceinline__ void actor_tiles(lds_f4 wl, int lane, const float (*x)[16], const uint32_t *legal, bool deterministic,
                                            const uint32_t *word, TileOut *out)
{
    const int h = lane >> 5;
    Operands<F16X3, T> xin, op;
#pragma unroll
    for (int t = 0; t < T; ++t) xin.set(t, x[t]);
    float hid[16];
    f32x16 acc[T], acc2[T], lg[T];
    apply_layer<T>(wl, lane, h, 0, xin, acc);
#pragma unroll
    for (int t = 0; t < T; ++t) {
#pragma unroll
        for (int r = 0; r < 16; ++r) hid[r] = sigmoid2(acc[t][r]);
        op.set(t, hid);
    }
    apply_layer<T>(wl, lane, h, 1, op, acc2);
#pragma unroll
    for (int t = 0; t < T; ++t) {
#pragma unroll
        for (int r = 0; r < 16; ++r) hid[r] = sigmoid2(acc2[t][r]);
        op.set(t, hid);
    }
    apply_layer<T>(wl, lane, h, 2, op, lg);
#pragma unroll
    for (int t = 0; t < T; ++t) sample_tile<NR>(lg[t], h, legal[t], deterministic, word[t], nullptr, out[t]);
}

// actor_tiles for two tiles as a software pipeline: a tile's matrix products are issued between the other tile's
// activation arithmetic (the MFMA pipe and the vector unit run side by side, and a wave issues in order: twelve MFMAs in
// a row keep it off the vector unit for 12 x 32 cycles per layer), stage by stage
//   M1(t0) | M1(t1) + S1(t0) | M2(t0) + S1(t1) | M2(t1) + S2(t0) | M3(t0) + S2(t1) | M3(t1) + sample(t0) | sample(t1)
// with M = a layer's six MFMAs and S = 16 x (exp, add, rcp) + the f16 split.  sched_group_barrier asks the scheduler
// for one MFMA per `per` vector instructions inside a stage; the arithmetic of a tile is what actor_tiles does.
#define MSEP_INTERLEAVE_6(per)                                          \
    do {                                                                \
        _Pragma("unroll") for (int k_ = 0; k_ < 6; ++k_) {              \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);          \
            __builtin_amdgcn_sched_group_barrier(0x002, per, 0);        \
        }                                                               \
    } while (0)
template <int NR>
__device__ __forceinline__ void actor_tiles_pipelined(lds_f4 wl, int lane, const float (*x)[16], const uint32_t *legal,
                                                      bool deterministic, const uint32_t *word, TileOut *out)
{
    const int h = lane >> 5;
    Operands<true, 1> op0, op1;
    float hid[16];
    f32x16 a0[1], a1[1], b0[1], b1[1];
    auto activate = [&](const f32x16 &acc, Operands<true, 1> &op) {
#pragma unroll
        for (int r = 0; r < 16; ++r) hid[r] = sigmoid2(acc[r]);
        op.set(0, hid);
    };
    op0.set(0, x[0]);
    apply_layer<1>(wl, lane, h, 0, op0, a0);
    __builtin_amdgcn_sched_barrier(0);
    op1.set(0, x[1]);
    apply_layer<1>(wl, lane, h, 0, op1, a1);
    activate(a0[0], op0);
    MSEP_INTERLEAVE_6(12);
    __builtin_amdgcn_sched_barrier(0);
    apply_layer<1>(wl, lane, h, 1, op0, b0);
    activate(a1[0], op1);
    MSEP_INTERLEAVE_6(12);
    __builtin_amdgcn_sched_barrier(0);
    apply_layer<1>(wl, lane, h, 1, op1, b1);
    activate(b0[0], op0);
    MSEP_INTERLEAVE_6(12);
    __builtin_amdgcn_sched_barrier(0);
    apply_layer<1>(wl, lane, h, 2, op0, a0);
    activate(b1[0], op1);
    MSEP_INTERLEAVE_6(12);
    __builtin_amdgcn_sched_barrier(0);
    apply_layer<1>(wl, lane, h, 2, op1, a1);
    sample_tile<NR>(a0[0], h, legal[0], deterministic, word[0], nullptr, out[0]);
    MSEP_INTERLEAVE_6(12);
    __builtin_amdgcn_sched_barrier(0);
    sample_tile<NR>(a1[0], h, legal[1], deterministic, word[1], nullptr, out[1]);
}

// The critic alone for T tiles: the bootstrap value of the state a rollout ends in.  Same layers, same bits as
// policy_tiles (a network's operations do not depend on what runs beside them).
template <bool F16X3, int T>
__device__ __forceinline__ void value_tiles(lds_f4 wl, int lane, const float (*x)[16], float *value)
{
    const int h = lane >> 5;
    Operands<F16X3, T> xin, op;
#pragma unroll
    for (int t = 0; t < T; ++t) xin.set(t, x[t]);
    float hid[16];
    f32x16 acc[T], acc2[T];
    apply_layer<T>(wl, lane, h, 3, xin, acc);
#pragma unroll
    for (int t = 0; t < T; ++t) {
#pragma unroll
        for (int r = 0; r < 16; ++r) hid[r] = sigmoid2(acc[t][r]);
        op.set(t, hid);
    }
    apply_layer<T>(wl, lane, h, 4, op, acc2);
#pragma unroll
    for (int t = 0; t < T; ++t) value[t] = value_tile(wl, h, acc2[t]);
}

// one tile (the standalone forward)
template <int NR, bool F16X3>
__device__ __forceinline__ TileOut policy_tile(lds_f4 wl, int lane, const float *x, uint32_t legal, bool deterministic,
                                               uint32_t word, float *lgm_out)
{
    float xs[1][16];
#pragma unroll
    for (int r = 0; r < 16; ++r) xs[0][r] = x[r];
    TileOut out[1];
    policy_tiles<NR, F16X3, 1>(wl, lane, xs, &legal, deterministic, &word, lgm_out, out);
    return out[0];
}

} // namespace msep

// handle of a packed policy (include/mse.h mse_policy): shared by the two translation units of the library
struct mse_policy {
    int d_in, n_act, device;
    float *blob;      // device image, msep::kBlobFloats floats
    size_t blob_floats;
    int f16_ok;       // every folded weight fits f16's range: the f16x3 form may be used
    int precision;    // 0 auto (f16x3 when f16_ok), 1 exact f32, 2 f16x3
    bool use_f16() const { return precision == 2 || (precision == 0 && f16_ok); }
};

// LDS image of a policy: the common head, then the operand form the kernel uses (msep::kLdsFloats floats)
__device__ __forceinline__ void msep_copy_image(float *lds, const float *__restrict__ blob, bool f16, int tid, int nthreads)
{
    const float4 *src = reinterpret_cast<const float4 *>(blob);
    float4 *dst = reinterpret_cast<float4 *>(lds);
    const int head4 = msep::kOffW / 4, body4 = msep::kOperandFloats / 4;
    const int from4 = (f16 ? msep::kOffW16 : msep::kOffW) / 4;
    for (int w = tid; w < head4; w += nthreads) dst[w] = src[w];
    for (int w = tid; w < body4; w += nthreads) dst[head4 + w] = src[from4 + w];
}
