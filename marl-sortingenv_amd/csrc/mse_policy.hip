// Batched forward of the reference's actor-critic policy with masked categorical sampling, on the matrix cores:
// the standalone launch (mse_policy_forward).  The network, its MFMA layout and the sampling rule live in
// mse_policy_device.h (one device function per 32-env tile, shared with the fused learned-policy rollout kernel of
// mse_lib.hip); this file holds the kernel that feeds it from observation / mask tensors, the host-side packing of
// torch.nn.Linear weights into MFMA operand order, and the C ABI.  gfx950 only.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "mse.h"
#include "mse_policy_device.h"
#include "mse_policy_stream.h"

int mse_internal_fail(int status, const char *msg); // mse_lib.hip: sets mse_last_error()

namespace {

using namespace msep;

struct PolicyArgs {
    long long n;
    long long index_offset;
    int d_in, n_act;
    int deterministic;
    unsigned long long seed, t;
};

// blob: the weights packed by pack_weights() below.  One wave per tile of 32 envs, grid-stride over the tiles.
template <int NR, bool F16X3>
__global__ __launch_bounds__(512) void k_policy_mlp(PolicyArgs P, const float *__restrict__ blob,
                                                    const float *__restrict__ obs, const uint8_t *__restrict__ mask,
                                                    int *__restrict__ action_out, float *__restrict__ logp_out,
                                                    float *__restrict__ value_out, float *__restrict__ logits_out)
{
    const int lane = threadIdx.x & 63, h = lane >> 5, j = lane & 31;
    // the packed weights are copied to LDS once per workgroup and read from there layer by layer (16-byte reads):
    // held in registers they would cost a wave ~100 VGPRs and every wave its own trip through L2
    extern __shared__ float wlds[];
    msep_copy_image(wlds, blob, F16X3, threadIdx.x, blockDim.x);
    __syncthreads();
    lds_f4 wl = (lds_f4)(__attribute__((address_space(3))) float *)wlds;
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
        uint32_t legal = 0; // bit r: the action of accumulator register r exists and may be taken
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const int a = row_of(r, h);
            const bool ok = a < A && (mask == nullptr || mask[env_c * A + (a < A ? a : 0)] != 0);
            legal |= (ok ? 1u : 0u) << r;
        }
        const uint32_t word = mse_policy_word(mse_policy_key(P.seed, (uint64_t)(P.index_offset + env)), P.t);
        float lgm[NR];
        const TileOut o = policy_tile<NR, F16X3>(wl, lane, x, legal, P.deterministic != 0, word, lgm);
        if (logits_out != nullptr) { // wave-uniform
#pragma unroll
            for (int r = 0; r < NR; ++r)
                if (row_of(r, h) < A && valid) logits_out[env * A + row_of(r, h)] = lgm[r];
        }
        if (valid && h == 0) {
            if (action_out != nullptr) action_out[env] = o.action;
            if (logp_out != nullptr) logp_out[env] = o.logp;
            if (value_out != nullptr) value_out[env] = o.value;
        }
    }
}

// torch.nn.Linear tensors (include/mse.h order) -> the image of mse_policy_device.h: A operands in the k order the
// accumulator registers impose, biases in register order, tanh folded into the weights.  Folding in double:
//   hidden unit:  r = 1 / (2^z + 1)  with  z = c (W in + b),  c = 2 log2 e,  tanh = 1 - 2 r
//   a layer fed by r instead of tanh:  W tanh + b = (b + W 1) + (-2 W) r
std::vector<float> pack_weights(const float *w, int D, int A, bool &f16_ok)
{
    const int H = kHidden;
    const float *pi_w1 = w, *pi_b1 = pi_w1 + H * D, *pi_w2 = pi_b1 + H, *pi_b2 = pi_w2 + H * H;
    const float *act_w = pi_b2 + H, *act_b = act_w + A * H;
    const float *vf_w1 = act_b + A, *vf_b1 = vf_w1 + H * D, *vf_w2 = vf_b1 + H, *vf_b2 = vf_w2 + H * H;
    const float *val_w = vf_b2 + H, *val_b = val_w + H;
    const double c = 2.0 * 1.4426950408889634073599246810019;
    // folded dense forms: Wd[L][out][in] (32 x 32, zero padded), bd[L][out]
    std::vector<double> Wd(5 * 32 * 32, 0.0), bd(5 * 32, 0.0);
    auto W = [&](int L, int o, int i) -> double & { return Wd[(size_t)(L * 32 + o) * 32 + i]; };
    auto B = [&](int L, int o) -> double & { return bd[(size_t)L * 32 + o]; };
    for (int o = 0; o < H; ++o) {
        for (int i = 0; i < D; ++i) {
            W(0, o, i) = c * (double)pi_w1[o * D + i];
            W(3, o, i) = c * (double)vf_w1[o * D + i];
        }
        B(0, o) = c * (double)pi_b1[o];
        B(3, o) = c * (double)vf_b1[o];
        double s1 = 0.0, s4 = 0.0;
        for (int i = 0; i < H; ++i) {
            W(1, o, i) = -2.0 * c * (double)pi_w2[o * H + i];
            W(4, o, i) = -2.0 * c * (double)vf_w2[o * H + i];
            s1 += (double)pi_w2[o * H + i];
            s4 += (double)vf_w2[o * H + i];
        }
        B(1, o) = c * ((double)pi_b2[o] + s1);
        B(4, o) = c * ((double)vf_b2[o] + s4);
    }
    for (int o = 0; o < A; ++o) {
        double s2 = 0.0;
        for (int i = 0; i < H; ++i) {
            W(2, o, i) = -2.0 * (double)act_w[o * H + i];
            s2 += (double)act_w[o * H + i];
        }
        B(2, o) = (double)act_b[o] + s2;
    }
    std::vector<float> out(kBlobFloats, 0.0f);
    f16_ok = true;
    uint16_t *h16 = reinterpret_cast<uint16_t *>(out.data() + kOffW16); // [hi | lo][layer][chunk][lane][8]
    auto f32_of_half = [](uint16_t hb) -> float {
        const uint32_t sgn = (uint32_t)(hb & 0x8000u) << 16, ex = (hb >> 10) & 0x1Fu, man = hb & 0x3FFu;
        if (ex == 0) return (sgn ? -1.0f : 1.0f) * std::ldexp((float)man, -24);
        uint32_t u = sgn | ((ex + 112u) << 23) | (man << 13);
        float f;
        std::memcpy(&f, &u, 4);
        return f;
    };
    auto half_rtz = [](float v) -> uint16_t { // f32 -> f16, round toward zero, subnormals kept, |v| < 65520
        uint32_t u;
        std::memcpy(&u, &v, 4);
        const uint16_t sgn = (uint16_t)((u >> 16) & 0x8000u);
        const int ex = (int)((u >> 23) & 0xFFu) - 127;
        const uint32_t man = (u & 0x7FFFFFu) | 0x800000u;
        if (((u >> 23) & 0xFFu) == 0) return sgn;                        // f32 zero / subnormal
        if (ex >= -14) return (uint16_t)(sgn | ((uint32_t)(ex + 15) << 10) | ((man >> 13) & 0x3FFu));
        if (ex < -25) return sgn;
        return (uint16_t)(sgn | (man >> (13 + (-14 - ex))));             // subnormal: shift the mantissa out
    };
    auto half_rne = [&](float v) -> uint16_t { // to nearest: the truncated value or its successor, whichever is closer
        const uint16_t lo_b = half_rtz(v);
        const uint16_t hi_b = (uint16_t)(lo_b + 1); // next magnitude (same sign); fine below the largest finite half
        const float a = f32_of_half(lo_b), b = f32_of_half(hi_b);
        const float da = std::fabs(v - a), db = std::fabs(b - v);
        return (db < da || (db == da && (hi_b & 1u) == 0)) ? hi_b : lo_b;
    };
    for (int L = 0; L < 5; ++L) {
        const bool input_layer = L == 0 || L == 3;
        for (int s = 0; s < 16; ++s) {
            for (int lane = 0; lane < 64; ++lane) {
                const int hp = lane >> 5, i = lane & 31;
                const int k = input_layer ? 2 * s + hp : row_of(s, hp); // what this k-step's B operand holds in half hp
                const float wf = (float)W(L, i, k);
                out[(size_t)kOffW + ((size_t)(L * 4 + (s >> 2)) * 64 + lane) * 4 + (s & 3)] = wf;
                if (!(std::fabs(wf) < 65504.0f)) f16_ok = false;
                const uint16_t hb = f16_ok ? half_rtz(wf) : 0;
                const uint16_t lb = f16_ok ? half_rne(wf - f32_of_half(hb)) : 0;
                const size_t at = ((size_t)(L * 2 + (s >> 3)) * 64 + lane) * 8 + (s & 7);
                h16[at] = hb;
                h16[(size_t)5 * 2 * 64 * 8 + at] = lb;
            }
        }
        for (int hh = 0; hh < 2; ++hh)
            for (int r = 0; r < 16; ++r) out[(size_t)kOffB + (L * 2 + hh) * 16 + r] = (float)B(L, row_of(r, hh));
    }
    double sv = 0.0;
    for (int i = 0; i < H; ++i) sv += (double)val_w[i];
    for (int hh = 0; hh < 2; ++hh)
        for (int r = 0; r < 16; ++r) out[(size_t)kOffWV + hh * 16 + r] = (float)(-2.0 * (double)val_w[row_of(r, hh)]);
    out[kOffBV] = (float)((double)val_b[0] + sv);
    return out;
}

} // namespace

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
    bool f16_ok = false;
    const std::vector<float> packed = pack_weights(weights_host, obs_dim, n_actions, f16_ok);
    mse_policy *p = new mse_policy{obs_dim, n_actions, device_id, nullptr, packed.size(), f16_ok ? 1 : 0, 0};
    if (hipMalloc(reinterpret_cast<void **>(&p->blob), p->blob_floats * sizeof(float)) != hipSuccess ||
        hipMemcpy(p->blob, packed.data(), p->blob_floats * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) {
        if (p->blob) (void)hipFree(p->blob);
        delete p;
        return mse_internal_fail(MSE_ERR_HIP, "mse_policy_create: device allocation or copy failed");
    }
    *out = p;
    return MSE_OK;
}

int mse_policy_set_precision(mse_policy *p, int mode)
{
    if (p == nullptr || mode < 0 || mode > 2) return mse_internal_fail(MSE_ERR_INVALID_ARGUMENT, "mse_policy_set_precision: bad argument");
    if (mode == 2 && !p->f16_ok)
        return mse_internal_fail(MSE_ERR_UNSUPPORTED_CONFIG, "mse_policy_set_precision: a folded weight exceeds f16's range (65 504)");
    p->precision = mode;
    return MSE_OK;
}

int mse_policy_precision(const mse_policy *p) { return p == nullptr ? -1 : (p->use_f16() ? 2 : 1); }

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
    if (blocks > 512) blocks = 512;     // two workgroups per CU (21 KB of LDS each), then grid-stride over the tiles
    const int nr = regs_for_actions(p->n_act); // accumulator registers that can hold an action: 2 .. 16
#define MSE_LAUNCH_POLICY(NR, F16)                                                                                   \
    hipLaunchKernelGGL((k_policy_mlp<NR, F16>), dim3((unsigned)blocks), dim3(512), kLdsFloats * sizeof(float),       \
                       static_cast<hipStream_t>(stream), a, p->blob, obs_dev, mask_dev, action_out, logp_out,        \
                       value_out, logits_out)
    if (p->use_f16()) {
        if (nr <= 2) MSE_LAUNCH_POLICY(2, true);
        else if (nr <= 7) MSE_LAUNCH_POLICY(7, true);
        else if (nr <= 12) MSE_LAUNCH_POLICY(12, true);
        else MSE_LAUNCH_POLICY(16, true);
    } else {
        if (nr <= 2) MSE_LAUNCH_POLICY(2, false);
        else if (nr <= 7) MSE_LAUNCH_POLICY(7, false);
        else if (nr <= 12) MSE_LAUNCH_POLICY(12, false);
        else MSE_LAUNCH_POLICY(16, false);
    }
#undef MSE_LAUNCH_POLICY
    if (hipGetLastError() != hipSuccess) return mse_internal_fail(MSE_ERR_HIP, "mse_policy_forward: kernel launch failed");
    return MSE_OK;
}

} // extern "C"
