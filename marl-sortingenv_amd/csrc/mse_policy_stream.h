// mse_policy_stream.h -- the counter-based stream of the on-device policies, shared by the step engine
// (mse_lib.hip: masked-uniform random policy) and the policy network (mse_policy.hip: categorical sampling).
//
// One 32-bit word per (policy seed, global env index, step counter t).  The env's KEY is a two-round hash of
// (seed, index) and is constant over a launch (hoisted out of the step loop); the word of step t is murmur3's
// 32-bit finaliser of (t * odd) XOR key.  The key enters by XOR, not by addition: with an additive key the
// streams of two envs are the same sequence shifted by a fixed number of steps (env g' = a delayed copy of env
// g for ever); with XOR two streams coincide at isolated (t, t') only, as any 2^32-state generator must.
// Sharded runs key by the GLOBAL env index, so they sample what a single-handle run samples.
#pragma once

#include <stdint.h>

#if defined(__HIPCC__)
#define MSE_HD __host__ __device__ __forceinline__
#else
#define MSE_HD static inline
#endif

MSE_HD uint32_t mse_fmix32(uint32_t h)
{
    h ^= h >> 16;
    h *= 0x85EBCA6Bu;
    h ^= h >> 13;
    h *= 0xC2B2AE35u;
    h ^= h >> 16;
    return h;
}

MSE_HD uint32_t mse_policy_key(uint64_t seed, uint64_t env_index)
{
    const uint32_t s = mse_fmix32((uint32_t)seed ^ mse_fmix32((uint32_t)(seed >> 32) + 0x9E3779B9u));
    const uint32_t g = (uint32_t)env_index * 0x9E3779B1u + (uint32_t)(env_index >> 32) * 0xC2B2AE3Du;
    return mse_fmix32(s + g);
}

MSE_HD uint32_t mse_policy_word(uint32_t key, uint64_t t)
{
    const uint32_t c = (uint32_t)t * 0x85EBCA77u + (uint32_t)(t >> 32) * 0x27D4EB2Fu;
    return mse_fmix32(c ^ key);
}
