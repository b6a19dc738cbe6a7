// Does v_mfma_f32_32x32x16_f16 keep f16 subnormal inputs, or flush them?  (decides whether the f16x3 split of the policy
// MLP needs its low parts pre-scaled).  A = all rows 1.0 in slot k=0; B slot k=0 = a subnormal f16 (2^-20) per column.
// Expected C[i][j] = 2^-20 = 9.5367e-07 if subnormals are honoured, 0 if flushed.
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/mfma_f16_denorm.hip -o /tmp/mfma_f16_denorm && /tmp/mfma_f16_denorm
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k(float *out, float tiny)
{
    h8 a = {}, b = {};
    if (threadIdx.x < 32) {
        a[0] = (_Float16)1.0f;
        b[0] = (_Float16)tiny;
    }
    f32x16 c = {};
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    out[threadIdx.x] = c[0];
    out[64 + threadIdx.x] = (float)b[0];
}
int main()
{
    float *d, h[128];
    hipMalloc(&d, sizeof(h));
    for (float tiny : {9.5367431640625e-07f /* 2^-20: f16 subnormal */, 6.103515625e-05f /* 2^-14: smallest normal */}) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, tiny);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("input %.6e as f16 -> %.6e ; mfma result C[0][0] = %.6e  (%s)\n", tiny, h[64], h[0],
               h[0] == tiny ? "kept" : (h[0] == 0.0f ? "FLUSHED" : "other"));
    }
    return 0;
}
