// Instruction-rate micro-benchmarks for the ops the step kernel is made of (gfx950).
// One workgroup of 256 threads per CU; W waves per SIMD via grid size.  Each kernel runs REPS
// iterations of an unrolled body of 16 instructions, either as one dependent chain (latency) or as
// 8 independent chains (issue rate).  Cycles come from s_memtime stamped by lane 0 of each wave.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
#include <algorithm>

#define REPS 2048

#define BODY16(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)
#define BODY16D(OP) OP(0) OP(0) OP(0) OP(0) OP(0) OP(0) OP(0) OP(0) OP(0) OP(0) OP(0) OP(0) OP(0) OP(0) OP(0) OP(0)

template <int KIND, bool DEP>
__global__ __launch_bounds__(256) void k(uint64_t *out_cycles, uint32_t *sink, uint32_t seed)
{
    uint32_t a[8], b = seed | 1u;
    uint64_t c[8];
    double d[8], e = 1.0000001 + seed * 1e-9;
    for (int i = 0; i < 8; ++i) {
        a[i] = threadIdx.x * 2654435761u + i * 40503u + seed;
        c[i] = ((uint64_t)a[i] << 32) | (a[i] ^ 0x9E3779B9u);
        d[i] = 1.0 + (double)a[i] * 1e-12;
    }
    uint64_t t0 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_sched_barrier(0);
    for (int r = 0; r < REPS; ++r) {
        if (KIND == 0) { // v_mad_u64_u32
#define OP(i) asm volatile("v_mad_u64_u32 %0, s[10:11], %1, %2, %0" : "+v"(c[i]) : "v"(a[i]), "v"(b) : "s10", "s11");
            if (DEP) { BODY16D(OP) } else { BODY16(OP) }
#undef OP
        } else if (KIND == 1) { // v_mul_lo_u32
#define OP(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            if (DEP) { BODY16D(OP) } else { BODY16(OP) }
#undef OP
        } else if (KIND == 2) { // v_mul_hi_u32
#define OP(i) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            if (DEP) { BODY16D(OP) } else { BODY16(OP) }
#undef OP
        } else if (KIND == 3) { // v_mul_u32_u24
#define OP(i) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            if (DEP) { BODY16D(OP) } else { BODY16(OP) }
#undef OP
        } else if (KIND == 4) { // v_add_u32
#define OP(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            if (DEP) { BODY16D(OP) } else { BODY16(OP) }
#undef OP
        } else if (KIND == 5) { // v_fma_f64
#define OP(i) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(d[i]) : "v"(e));
            if (DEP) { BODY16D(OP) } else { BODY16(OP) }
#undef OP
        } else if (KIND == 6) { // v_mul_f64
#define OP(i) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"(e));
            if (DEP) { BODY16D(OP) } else { BODY16(OP) }
#undef OP
        } else if (KIND == 7) { // v_rcp_f64
#define OP(i) asm volatile("v_rcp_f64 %0, %0" : "+v"(d[i]));
            if (DEP) { BODY16D(OP) } else { BODY16(OP) }
#undef OP
        } else if (KIND == 8) { // v_lshl_add_u64
#define OP(i) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(c[i]) : "v"(c[(i + 1) & 7]));
            if (DEP) { BODY16D(OP) } else { BODY16(OP) }
#undef OP
        } else if (KIND == 9) { // v_cndmask_b32 + v_cmp pair
#define OP(i) asm volatile("v_cmp_lt_u32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b) : "vcc");
            if (DEP) { BODY16D(OP) } else { BODY16(OP) }
#undef OP
        } else if (KIND == 10) { // v_add_f64
#define OP(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(e));
            if (DEP) { BODY16D(OP) } else { BODY16(OP) }
#undef OP
        } else if (KIND == 11) { // v_mad_u32_u24
#define OP(i) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(a[i]) : "v"(b));
            if (DEP) { BODY16D(OP) } else { BODY16(OP) }
#undef OP
        } else if (KIND == 12) { // fp64 division as the compiler emits it
            for (int i = 0; i < (DEP ? 1 : 8); ++i) d[i] = e / d[i];
            if (DEP) for (int j = 0; j < 15; ++j) d[0] = e / d[0]; else for (int i = 0; i < 8; ++i) d[i] = e / d[i];
        } else if (KIND == 13) { // v_cvt_f64_u32 + v_cvt_u32_f64
#define OP(i) asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(d[i]) : "v"(a[i])); asm volatile("v_cvt_u32_f64 %0, %1" : "=v"(a[i]) : "v"(d[i]));
            if (DEP) { BODY16D(OP) } else { BODY16(OP) }
#undef OP
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    uint64_t t1 = __builtin_amdgcn_s_memtime();
    uint32_t acc = 0;
    for (int i = 0; i < 8; ++i) acc ^= a[i] ^ (uint32_t)c[i] ^ (uint32_t)(c[i] >> 32) ^ (uint32_t)__double2loint(d[i]);
    if (acc == 0x12345678u) sink[0] = acc;
    if ((threadIdx.x & 63) == 0) out_cycles[(blockIdx.x * 256 + threadIdx.x) / 64] = t1 - t0;
}

template <int KIND, bool DEP>
static void run(const char *name, int waves_per_simd)
{
    int blocks = 256 * waves_per_simd;
    uint64_t *dc;
    uint32_t *sink;
    hipMalloc(&dc, sizeof(uint64_t) * blocks * 4);
    hipMalloc(&sink, 16);
    hipLaunchKernelGGL((k<KIND, DEP>), dim3(blocks), dim3(256), 0, 0, dc, sink, 7u);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<KIND, DEP>), dim3(blocks), dim3(256), 0, 0, dc, sink, 9u);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<uint64_t> h(blocks * 4);
    hipMemcpy(h.data(), dc, sizeof(uint64_t) * blocks * 4, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    double med = (double)h[h.size() / 2];
    int per_iter = (KIND == 9 || KIND == 13) ? 32 : 16;
    double n_inst = (double)REPS * per_iter;
    // s_memtime ticks at 100 MHz on gfx9 (constant clock): convert with the wall time instead
    printf("%-28s %s W=%d  wall %.3f ms  => %.2f ns per wave-instr per wave; memtime ticks/instr %.3f\n", name,
           DEP ? "dep " : "indep", waves_per_simd, ms, ms * 1e6 / n_inst, med / n_inst);
    hipFree(dc);
    hipFree(sink);
}

#define RUNALL(K, NAME)                  \
    run<K, true>(NAME, 1);               \
    run<K, false>(NAME, 1);              \
    run<K, false>(NAME, 2);              \
    run<K, false>(NAME, 4);

int main()
{
    RUNALL(4, "v_add_u32")
    RUNALL(0, "v_mad_u64_u32")
    RUNALL(1, "v_mul_lo_u32")
    RUNALL(2, "v_mul_hi_u32")
    RUNALL(3, "v_mul_u32_u24")
    RUNALL(11, "v_mad_u32_u24")
    RUNALL(8, "v_lshl_add_u64")
    RUNALL(9, "v_cmp+v_cndmask (x2)")
    RUNALL(5, "v_fma_f64")
    RUNALL(6, "v_mul_f64")
    RUNALL(10, "v_add_f64")
    RUNALL(7, "v_rcp_f64")
    RUNALL(13, "cvt f64<->u32 (x2)")
    RUNALL(12, "f64 division (compiler seq)")
    return 0;
}
