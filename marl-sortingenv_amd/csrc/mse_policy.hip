// Batched forward of the reference's actor-critic policy with masked categorical sampling, on the matrix cores.
//
// The policy SB3 builds for the reference (src/training.py:115: net_arch=dict(pi=[32, 32], vf=[32, 32]), default
// tanh activations, MaskableActorCriticPolicy): two separate 2x32 tanh MLPs on the flattened observation, a linear
// action head (32 -> A) and a linear value head (32 -> 1); invalid actions get logit -1e8 before the softmax
// (sb3_contrib MaskableCategorical).  This is SURVEY 8f rank 2: the step on the caller's side of env.step().
//
// One wavefront serves a tile of 32 envs with v_mfma_f32_32x32x2_f32 (f32 in, f32 accumulate: the result is an
// fmaf chain, only the summation order differs from a CPU reference).  Orientation: outputs (hidden units or
// actions) are the rows, envs are the columns.  C/D layout: lane l = (half h = l >> 5, env j = l & 31), register
// r holds row (r & 3) + 8 (r >> 2) + 4 h.  A product that sums over those rows can take the accumulator
// registers straight as its B operand, one register per k-step, provided the A operand (the weights) is loaded
// in the matching k order - so the three layers chain with no lane movement, no LDS and no transposes: the
// weights of every layer are loaded into registers once per wave in that permuted order.
// gfx950 only.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <string>
#include <vector>

#include "mse.h"
#include "mse_policy_stream.h"

int mse_internal_fail(int status, const char *msg); // mse_lib.hip: sets mse_last_error()

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int kHidden = 32;
constexpr int kTile = 32; // envs per MFMA tile
constexpr int kPackedFloats = 11 * 16 * 64 + 4; // pack_weights(): eleven [16][64] arrays, then val_b (padded to 16 B)

__device__ __forceinline__ int row_of(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// tanh(x) = 1 - 2 / (2^(2 x log2 e) + 1): v_exp_f32 + v_rcp_f32, absolute error ~2e-7
__device__ __forceinline__ float fast_tanh(float x)
{
    const float t = __builtin_amdgcn_exp2f(x * 2.88539008177792681472f);
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(t + 1.0f);
}

__device__ __forceinline__ float xhalf(float v) { return __shfl_xor(v, 32); } // the other half's value for this env
__device__ __forceinline__ int xhalf_i(int v) { return __shfl_xor(v, 32); }

struct PolicyArgs {
    long long n;
    long long index_offset;
    int d_in, n_act;
    int deterministic;
    unsigned long long seed, t;
};

// blob: the weights packed by pack_weights() below in MFMA operand order
__global__ __launch_bounds__(512) void k_policy_mlp(PolicyArgs P, const float *__restrict__ blob,
                                                    const float *__restrict__ obs, const uint8_t *__restrict__ mask,
                                                    int *__restrict__ action_out, float *__restrict__ logp_out,
                                                    float *__restrict__ value_out, float *__restrict__ logits_out)
{
    const int lane = threadIdx.x & 63, h = lane >> 5, j = lane & 31;
    // The packed weights (A operands and bias accumulators in register order, [array][register s][lane]) are
    // copied to LDS once per workgroup and read from there layer by layer: a wave that kept all eleven arrays in
    // registers needs 308 VGPRs (one wave per SIMD) and every wave would pull the 45 KB through L2 by itself -
    // that transfer, not the MFMAs, was the whole kernel time of the first version.
    extern __shared__ float wlds[];
    for (int w = threadIdx.x; w < kPackedFloats / 4; w += blockDim.x)
        reinterpret_cast<float4 *>(wlds)[w] = reinterpret_cast<const float4 *>(blob)[w];
    __syncthreads();
    // volatile: re-read per tile (one conflict-free ds_read_b32 per operand) instead of 176 registers held per wave,
    // so that two waves fit on a SIMD and one's MFMA chain runs under the other's tanh / softmax work
    typedef __attribute__((address_space(3))) const volatile float *wptr; // stays a ds_read with an immediate offset
    wptr wbase = (wptr)(__attribute__((address_space(3))) float *)wlds + lane;
    wptr a1 = wbase + (0 * 16) * 64, a2 = wbase + (1 * 16) * 64, a3 = wbase + (2 * 16) * 64;
    wptr v1 = wbase + (3 * 16) * 64, v2 = wbase + (4 * 16) * 64, wv = wbase + (5 * 16) * 64;
    wptr c1p = wbase + (6 * 16) * 64, c2p = wbase + (7 * 16) * 64, c3p = wbase + (8 * 16) * 64;
    wptr cv1p = wbase + (9 * 16) * 64, cv2p = wbase + (10 * 16) * 64;
    const float bv = wlds[11 * 16 * 64];
    const int D = P.d_in, A = P.n_act;
    const long long n_tiles = (P.n + kTile - 1) / kTile;
    const long long wave = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const long long n_waves = (long long)gridDim.x * (blockDim.x >> 6);
    for (long long tile = wave; tile < n_tiles; tile += n_waves) {
        const long long env = tile * kTile + j;
        const bool valid = env < P.n;
        const long long env_c = valid ? env : 0; // clamp: loads of padding lanes stay in bounds, results unused
        float x[16];
#pragma unroll
        for (int s = 0; s < 16; ++s) { // unconditional loads at clamped indices, then a select: no exec-mask regions
            const int k_in = 2 * s + h;
            const float v = obs[env_c * D + (k_in < D ? k_in : 0)];
            x[s] = k_in < D ? v : 0.0f;
        }
        // critic
        f32x16 vc;
#pragma unroll
        for (int r = 0; r < 16; ++r) vc[r] = cv1p[r * 64];
#pragma unroll
        for (int s = 0; s < 16; ++s) vc = __builtin_amdgcn_mfma_f32_32x32x2f32(v1[s * 64], x[s], vc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 16; ++r) vc[r] = fast_tanh(vc[r]);
        f32x16 vc2;
#pragma unroll
        for (int r = 0; r < 16; ++r) vc2[r] = cv2p[r * 64];
#pragma unroll
        for (int s = 0; s < 16; ++s) vc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(v2[s * 64], vc[s], vc2, 0, 0, 0);
        float val = 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) val = fmaf(fast_tanh(vc2[r]), wv[r * 64], val);
        val += xhalf(val);
        val += bv;

        __builtin_amdgcn_sched_barrier(0); // critic first and finished (one scalar left), then the actor: fewer live registers
        // actor
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = c1p[r * 64];
#pragma unroll
        for (int s = 0; s < 16; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[s * 64], x[s], acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = fast_tanh(acc[r]);
        f32x16 acc2;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc2[r] = c2p[r * 64];
#pragma unroll
        for (int s = 0; s < 16; ++s) acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a2[s * 64], acc[s], acc2, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 16; ++r) acc2[r] = fast_tanh(acc2[r]);
        f32x16 lg;
#pragma unroll
        for (int r = 0; r < 16; ++r) lg[r] = c3p[r * 64];
#pragma unroll
        for (int s = 0; s < 16; ++s) lg = __builtin_amdgcn_mfma_f32_32x32x2f32(a3[s * 64], acc2[s], lg, 0, 0, 0);
        // masked logits: this lane owns actions row_of(r, h) < A of env j.  Everything below is selects, no branches:
        // a per-register `if` turns into an exec-mask region each, and sixteen of them cost more than the MFMAs.
        float m = -3.0e38f;
        uint32_t allowed = 0xFFFFu; // bit r: the action in register r may be taken
        if (mask != nullptr) {      // wave-uniform
            allowed = 0;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int a = row_of(r, h);
                allowed |= (mask[env_c * A + (a < A ? a : 0)] != 0 ? 1u : 0u) << r;
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const bool own = row_of(r, h) < A;
            lg[r] = own ? (((allowed >> r) & 1u) ? lg[r] : -1.0e8f) : -3.0e38f; // sb3_contrib: HUGE_NEG = -1e8
            m = fmaxf(m, lg[r]);
        }
        if (logits_out != nullptr) { // wave-uniform
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (row_of(r, h) < A && valid) logits_out[env * A + row_of(r, h)] = lg[r];
        }
        m = fmaxf(m, xhalf(m));
        float e[16], gsum[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            gsum[q] = 0.0f;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int r = 4 * q + b;
                // exp(x) = 2^(x log2 e); x <= 0 here; masked (-1e8 - m) and not-owned entries underflow to exactly 0
                e[r] = __builtin_amdgcn_exp2f((lg[r] - m) * 1.44269504088896340736f);
                gsum[q] += e[r];
            }
        }
        // groups of four actions alternate between the halves: group g = 2 q + h holds actions 4 g .. 4 g + 3
        float other[4], total = 0.0f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            other[q] = xhalf(gsum[q]);
            total += h == 0 ? gsum[q] + other[q] : other[q] + gsum[q]; // the same association in both halves
        }
        int act = 99;
        if (P.deterministic) { // wave-uniform
            float best = -3.0e38f;
#pragma unroll
            for (int r = 0; r < 16; ++r) { // rows ascend with r inside a lane: strict > keeps the lowest index
                const bool better = lg[r] > best;
                act = better ? row_of(r, h) : act;
                best = better ? lg[r] : best;
            }
            const float ob = xhalf(best);
            const int oa = xhalf_i(act);
            const bool take = ob > best || (ob == best && oa < act);
            act = take ? oa : act;
        } else {
            const uint32_t word = mse_policy_word(mse_policy_key(P.seed, (uint64_t)(P.index_offset + env)), P.t);
            const float target = (float)(word >> 8) * 5.9604644775390625e-8f * total; // u in [0, 1) times the mass
            float cum = 0.0f; // inclusive cumulative mass in action order, walked group by group
            int last = -1;    // the last action of this lane with any mass (fallback when target rounds up to the total)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float g0 = h == 0 ? gsum[q] : other[q], g1 = h == 0 ? other[q] : gsum[q];
                float c = cum + (h == 0 ? 0.0f : g0); // mass before this lane's group 2 q + h
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const int r = 4 * q + b;
                    c += e[r];
                    const bool has = e[r] > 0.0f;
                    const bool hit = has && c > target && act == 99;
                    act = hit ? row_of(r, h) : act;
                    last = has ? row_of(r, h) : last;
                }
                cum += g0 + g1;
            }
            const int oa = xhalf_i(act), ol = xhalf_i(last);
            act = oa < act ? oa : act;
            last = ol > last ? ol : last;
            act = act == 99 ? last : act;
        }
        // log-probability of the chosen action
        float la = -3.0e38f;
#pragma unroll
        for (int r = 0; r < 16; ++r)
            if (row_of(r, h) == act) la = lg[r];
        la = fmaxf(la, xhalf(la));
        if (valid && h == 0) {
            if (action_out != nullptr) action_out[env] = act;
            if (logp_out != nullptr) logp_out[env] = la - m - __logf(total);
            if (value_out != nullptr) value_out[env] = val;
        }
    }
}

// torch.nn.Linear tensors (include/mse.h order) -> [array][register s][lane] in the order the kernel's registers
// want them: lane = (half h, row j); layer 1 sums over observation entries k = 2 s + h, the later layers over the
// previous accumulator's rows k = row_of(s, h); bias register s of a lane belongs to output row row_of(s, h).
inline int host_row_of(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

std::vector<float> pack_weights(const float *w, int D, int A)
{
    const int H = kHidden;
    const float *pi_w1 = w, *pi_b1 = pi_w1 + H * D, *pi_w2 = pi_b1 + H, *pi_b2 = pi_w2 + H * H;
    const float *act_w = pi_b2 + H, *act_b = act_w + A * H;
    const float *vf_w1 = act_b + A, *vf_b1 = vf_w1 + H * D, *vf_w2 = vf_b1 + H, *vf_b2 = vf_w2 + H * H;
    const float *val_w = vf_b2 + H, *val_b = val_w + H;
    std::vector<float> out(kPackedFloats, 0.0f);
    for (int s = 0; s < 16; ++s) {
        for (int lane = 0; lane < 64; ++lane) {
            const int h = lane >> 5, j = lane & 31, k_in = 2 * s + h, k_hid = host_row_of(s, h), row = host_row_of(s, h);
            auto at = [&](int arr) -> float & { return out[(size_t)(arr * 16 + s) * 64 + lane]; };
            at(0) = k_in < D ? pi_w1[j * D + k_in] : 0.0f;
            at(1) = pi_w2[j * H + k_hid];
            at(2) = j < A ? act_w[j * H + k_hid] : 0.0f;
            at(3) = k_in < D ? vf_w1[j * D + k_in] : 0.0f;
            at(4) = vf_w2[j * H + k_hid];
            at(5) = val_w[k_hid];
            at(6) = pi_b1[row];
            at(7) = pi_b2[row];
            at(8) = row < A ? act_b[row] : 0.0f;
            at(9) = vf_b1[row];
            at(10) = vf_b2[row];
        }
    }
    out[11 * 16 * 64] = val_b[0];
    return out;
}

} // namespace

struct mse_policy {
    int d_in, n_act, device;
    float *blob;
    size_t blob_floats;
};

extern "C" {

int64_t mse_policy_num_weights(int obs_dim, int n_actions)
{
    const int64_t H = kHidden, D = obs_dim, A = n_actions;
    return 2 * (H * D + H + H * H + H) + A * H + A + H + 1;
}

int mse_policy_create(mse_policy **out, int obs_dim, int n_actions, const float *weights_host, int device_id)
{
    if (out == nullptr || weights_host == nullptr) return mse_internal_fail(MSE_ERR_INVALID_ARGUMENT, "mse_policy_create: null argument");
    if (obs_dim < 1 || obs_dim > 32 || n_actions < 1 || n_actions > 32)
        return mse_internal_fail(MSE_ERR_UNSUPPORTED_CONFIG, "mse_policy_create: obs_dim and n_actions must be in 1..32");
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count == 0 || device_id < 0 || device_id >= count)
        return mse_internal_fail(MSE_ERR_NO_DEVICE, "mse_policy_create: no such HIP device");
    if (hipSetDevice(device_id) != hipSuccess) return mse_internal_fail(MSE_ERR_HIP, "hipSetDevice failed");
    const std::vector<float> packed = pack_weights(weights_host, obs_dim, n_actions);
    mse_policy *p = new mse_policy{obs_dim, n_actions, device_id, nullptr, packed.size()};
    if (hipMalloc(reinterpret_cast<void **>(&p->blob), p->blob_floats * sizeof(float)) != hipSuccess ||
        hipMemcpy(p->blob, packed.data(), p->blob_floats * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) {
        if (p->blob) (void)hipFree(p->blob);
        delete p;
        return mse_internal_fail(MSE_ERR_HIP, "mse_policy_create: device allocation or copy failed");
    }
    *out = p;
    return MSE_OK;
}

int mse_policy_destroy(mse_policy *p)
{
    if (p == nullptr) return MSE_OK;
    (void)hipFree(p->blob);
    delete p;
    return MSE_OK;
}

int mse_policy_forward(mse_policy *p, int64_t n, int64_t index_offset, const float *obs_dev, const uint8_t *mask_dev,
                       uint64_t seed, uint64_t t, int deterministic, int32_t *action_out, float *logp_out,
                       float *value_out, float *logits_out, void *stream)
{
    if (p == nullptr || obs_dev == nullptr || n < 0) return mse_internal_fail(MSE_ERR_INVALID_ARGUMENT, "mse_policy_forward: bad argument");
    if (n == 0) return MSE_OK;
    PolicyArgs a{(long long)n, (long long)index_offset, p->d_in, p->n_act, deterministic ? 1 : 0, seed, t};
    const long long tiles = (n + kTile - 1) / kTile;
    long long blocks = (tiles + 7) / 8; // 8 waves per block = two per SIMD
    if (blocks > 256) blocks = 256;     // one workgroup per CU, then grid-stride over the tiles
    hipLaunchKernelGGL(k_policy_mlp, dim3((unsigned)blocks), dim3(512), kPackedFloats * sizeof(float),
                       static_cast<hipStream_t>(stream), a, p->blob,
                       obs_dev, mask_dev, action_out, logp_out, value_out, logits_out);
    if (hipGetLastError() != hipSuccess) return mse_internal_fail(MSE_ERR_HIP, "mse_policy_forward: kernel launch failed");
    return MSE_OK;
}

} // extern "C"
