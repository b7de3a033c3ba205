// The wave-level building block shared by the pair kernels: one j-tile resident in LDS, 8 wave-uniform
// i-rows on the scalar path, v_and_b32 (SGPR x VGPR) + v_bcnt_u32_b32 (accumulating) as the whole
// inner loop.
//
// What the compiler has to be told (checked in the .s, see DESIGN.md "inner loop"):
//   * the i-row pointer must be a `const T *__restrict__` KERNEL ARGUMENT (not a struct member) and the
//     address wave-uniform, or the 8 row fetches become vector global_loads instead of s_load_dwordx16;
//   * `acc += popcount(x)` chains are re-associated into v_bcnt(x, 0) + v_add3; the empty asm after each
//     step keeps the add attached to its popcount so it selects the accumulating v_bcnt_u32_b32;
//   * scalar loads return out of order, so any wait on them is lgkmcnt(0).  The two-stage pipeline below
//     therefore keeps exactly ONE generation of loads in flight: `touch` forces the wait for stage k
//     before stage k+1 is issued, and sched_barrier pins the issue ahead of the 256 VALU instructions
//     that cover its latency.
#pragma once

#include "ldx_common.h"

namespace ldx {

constexpr int kWaves = 16;              // waves per workgroup (1024 threads, one workgroup per CU)
constexpr int kThreads = kWaves * 64;

struct Acc {
    uint32_t v[kGroup][2];
};

struct Stage {
    uint4 a[kGroup];   // 8 i-rows x 128 haplotypes, SGPRs
    uint4 b0, b1;      // this lane's two j-rows, VGPRs
};

__device__ __forceinline__ void popacc(uint32_t &acc, uint32_t x)
{
    acc = __builtin_popcount(x) + acc;
    asm("" : "+v"(acc));
}

__device__ __forceinline__ void load_stage(Stage &s, const uint4 *__restrict__ ai, const uint4 *jt, uint32_t c,
                                           uint32_t lane)
{
    s.b0 = jt[c * kSlab + lane];
    s.b1 = jt[c * kSlab + 64u + lane];
#pragma unroll
    for (int r = 0; r < (int)kGroup; ++r) s.a[r] = ai[c * kSlab + r];   // 128 contiguous bytes -> 2 x s_load_dwordx16
    __builtin_amdgcn_sched_barrier(0);
}

__device__ __forceinline__ void touch(const Stage &s)
{
    asm volatile("" ::"s"(s.a[kGroup - 1].w), "v"(s.b1.w));
    __builtin_amdgcn_sched_barrier(0);
}

__device__ __forceinline__ void compute_stage(const Stage &s, Acc &acc)
{
#pragma unroll
    for (int r = 0; r < (int)kGroup; ++r) {
        const uint4 a = s.a[r];
        popacc(acc.v[r][0], a.x & s.b0.x);
        popacc(acc.v[r][1], a.x & s.b1.x);
        popacc(acc.v[r][0], a.y & s.b0.y);
        popacc(acc.v[r][1], a.y & s.b1.y);
        popacc(acc.v[r][0], a.z & s.b0.z);
        popacc(acc.v[r][1], a.z & s.b1.z);
        popacc(acc.v[r][0], a.w & s.b0.w);
        popacc(acc.v[r][1], a.w & s.b1.w);
    }
}

// n11 of 8 i-rows (scalar side) x 2 j-rows per lane over all chunks (calc_ld.py:32 for 1024 pairs).
// `ai` points at chunk 0 / first row of the group inside its slab (uint4 units: chunk stride = 128 rows),
// `jt` at the LDS tile.
__device__ __forceinline__ void count_unit(const uint4 *__restrict__ ai, const uint4 *jt, uint32_t nchunks,
                                           uint32_t lane, Acc &acc)
{
#pragma unroll
    for (int r = 0; r < (int)kGroup; ++r) acc.v[r][0] = acc.v[r][1] = 0;
    Stage s0, s1;
    load_stage(s0, ai, jt, 0, lane);
    uint32_t c = 0;
    for (; c + 2 <= nchunks; c += 2) {
        touch(s0);
        load_stage(s1, ai, jt, c + 1, lane);
        compute_stage(s0, acc);
        touch(s1);
        // the last prefetch re-reads a valid chunk and is discarded
        load_stage(s0, ai, jt, (c + 2 < nchunks) ? c + 2 : c, lane);
        compute_stage(s1, acc);
    }
    if (c < nchunks) compute_stage(s0, acc);
}

__device__ __forceinline__ void stage_tile(uint4 *jt, const uint4 *__restrict__ src, uint32_t n16)
{
    for (uint32_t k = threadIdx.x; k < n16; k += kThreads) jt[k] = src[k];
}

}  // namespace ldx
