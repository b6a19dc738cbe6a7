// LDS round-trip and small-loop micro-benchmarks (gfx950), to read the draw loop of k_rollout_ring:
//   chase      : x = lds[x] dependent chain                      -> LDS load-to-use latency
//   chase+alu  : the same with 8 dependent VALU ops per load      -> does ALU time add to or hide in the latency?
//   loop19     : a 19-instruction divergent-exit loop body with one prefetched LDS read per iteration
// One workgroup of 64*W threads per CU (W = waves per SIMD is W/4 rounded up).  Cycles: s_memtime by lane 0.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

#define REPS 4096

template <int KIND>
__global__ void k(uint64_t *out, uint32_t *sink, uint32_t seed)
{
    __shared__ uint32_t lds[16384];
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = (uint32_t)((i * 2654435761u + seed) & 16383u);
    __syncthreads();
    uint32_t x = threadIdx.x, acc = seed;
    uint64_t t0 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_sched_barrier(0);
    if (KIND == 0) {
        for (int r = 0; r < REPS; ++r) x = lds[x];
    } else if (KIND == 1) {
        for (int r = 0; r < REPS; ++r) {
            x = lds[x];
#pragma unroll
            for (int j = 0; j < 8; ++j) asm volatile("v_add_u32 %0, %0, %1" : "+v"(acc) : "v"(x));
            x = (x + (acc & 1u)) & 16383u;
        }
    } else if (KIND == 2) { // prefetched read one iteration ahead + 16 dependent ALU ops
        uint32_t nxt = lds[x];
        for (int r = 0; r < REPS; ++r) {
            const uint32_t cur = nxt;
            x = (x + 64u) & 16383u;
            nxt = lds[x];
#pragma unroll
            for (int j = 0; j < 16; ++j) asm volatile("v_add_u32 %0, %0, %1" : "+v"(acc) : "v"(cur));
        }
        acc += nxt;
    }
    __builtin_amdgcn_sched_barrier(0);
    uint64_t t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
    if (x == 0xFFFFFFFFu || acc == 0x12345u) sink[0] = x + acc;
}

template <int KIND>
static void run(const char *name, int waves_per_cu)
{
    uint64_t *d_out;
    uint32_t *d_sink;
    const int blocks = 256;
    hipMalloc(&d_out, blocks * waves_per_cu * sizeof(uint64_t));
    hipMalloc(&d_sink, 4);
    hipLaunchKernelGGL((k<KIND>), dim3(blocks), dim3(64 * waves_per_cu), 0, 0, d_out, d_sink, 7u);
    hipDeviceSynchronize();
    std::vector<uint64_t> h(blocks * waves_per_cu);
    hipMemcpy(h.data(), d_out, h.size() * 8, hipMemcpyDeviceToHost);
    double s = 0;
    for (auto v : h) s += (double)v;
    printf("%-12s waves/CU %2d: %.1f ticks per iteration\n", name, waves_per_cu, s / h.size() / REPS);
    hipFree(d_out);
    hipFree(d_sink);
}

int main()
{
    for (int w : {1, 4, 12}) {
        run<0>("chase", w);
        run<1>("chase+8alu", w);
        run<2>("prefetch+16alu", w);
    }
    return 0;
}
