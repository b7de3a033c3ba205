// ld_triangle on the matrix cores: n11 = G . G^T with {0,1} int8 operands (v_mfma_i32_32x32x32_i8),
// fused with the same fp64 epilogue as the popcount kernel.  calc_ld.py:32 for 8192 pairs per wave unit.
//
// Why: AND + BCNT run at 64 lanes/clk/CU (no packed form), i.e. 16 haplotype-pairs per lane-instruction;
// the int8 MFMA does 1024 MACs/clk/SIMD -- 4x the VALU ceiling -- so the count moves to the matrix pipe
// and the VALU is left with expanding bits to bytes and the epilogue.
//
// Structure (same persistent skeleton as triangle_kernel):
//   * the j-tile (128 SNP rows, all chunks) sits in LDS still BIT-PACKED (80 KiB at 5008 haplotypes);
//   * a wave's unit is 64 i-rows x 128 j-rows: 2 x 4 accumulator tiles of 32x32 (128 VGPRs);
//   * per 32-haplotype K-step a lane holds 16 bits of "its" row (row = lane % 32, k-half = lane / 32) for
//     2 A tiles (global loads: 32 consecutive rows of one chunk are 512 contiguous bytes) and 4 B tiles
//     (ds_read_b128, two lanes per address), expands each 16 bits to 16 bytes
//     (bfe, * 0x00204081, & 0x01010101) and issues 8 MFMAs;
//   * A and B are expanded by the same function from the same bit positions, so whatever order the
//     hardware gives the 16 k-slots of a lane, slot s of A meets slot s of B: the sum over k is the
//     AND-popcount.  Row/column placement follows the documented 32x32 C/D map
//     (col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)), checked bit-exactly against
//     the popcount kernel and the oracle in tests/.
#include "ldx_common.h"
#include "ldx_tile.h"

namespace ldx {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

constexpr int kMfmaWaves = 8;                 // 512 threads: two waves per SIMD at <= 256 VGPRs
constexpr int kMfmaThreads = kMfmaWaves * 64;
constexpr uint32_t kRows64 = 64;              // i-rows per wave unit

// 16 haplotype bits (bits 0..15 of `bits`) -> 16 bytes of 0/1
__device__ __forceinline__ v4i expand16(uint32_t bits)
{
    v4i r;
    r.x = (int)(((bits & 0xFu) * 0x00204081u) & 0x01010101u);
    r.y = (int)((((bits >> 4) & 0xFu) * 0x00204081u) & 0x01010101u);
    r.z = (int)((((bits >> 8) & 0xFu) * 0x00204081u) & 0x01010101u);
    r.w = (int)((((bits >> 12) & 0xFu) * 0x00204081u) & 0x01010101u);
    return r;
}

__device__ __forceinline__ uint32_t word_of(const uint4 &v, int w)
{
    return w == 0 ? v.x : (w == 1 ? v.y : (w == 2 ? v.z : v.w));
}

template <bool kRaw, bool kN11>
__global__ void __launch_bounds__(kMfmaThreads)
triangle_mfma_kernel(const uint4 *__restrict__ alt, const double *__restrict__ fa, const double *__restrict__ fr,
                     const double *__restrict__ q, uint32_t n_snps, uint32_t n_slabs, uint32_t nchunks, double n,
                     uint64_t u_begin, uint64_t u_end, ldx_ld32 *__restrict__ out, ldx_ld64 *__restrict__ raw,
                     uint32_t *__restrict__ n11)
{
    extern __shared__ uint4 lds[];
    uint4 *jt = lds;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t l32 = lane & 31u;
    const uint32_t half = lane >> 5;
    const uint32_t sh = half * 16u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);

    // 64-row units: tile t owns groups g64 in [2t, 2T); unit v <-> small units [8v, 8v+8)
    const uint64_t G64 = (uint64_t)n_slabs * 2u;
    auto base64 = [&](uint64_t t) { return t * G64 - t * (t - 1u); };
    const uint64_t v_begin = u_begin / 8u, v_end = (u_end + 7u) / 8u;   // units that intersect the range
    const uint64_t total = v_end - v_begin;
    const uint64_t b0 = v_begin + total * blockIdx.x / gridDim.x;
    const uint64_t b1 = v_begin + total * (blockIdx.x + 1) / gridDim.x;
    if (b0 >= b1) return;   // block-uniform

    uint32_t t;
    {
        uint32_t lo = 0, hi = n_slabs;   // largest t with base64(t) <= b0
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) / 2;
            if (base64(mid) <= b0) lo = mid; else hi = mid;
        }
        t = lo;
    }
    uint64_t v = b0;
    while (v < b1) {   // block-uniform trip count
        const uint64_t tb = base64(t), te = base64(t + 1u);
        const uint64_t seg_end = b1 < te ? b1 : te;
        const uint32_t seg_len = (uint32_t)(seg_end - v);
        __syncthreads();
        for (uint32_t k = threadIdx.x; k < nchunks * kSlab; k += kMfmaThreads)
            jt[k] = alt[(size_t)t * nchunks * kSlab + k];
        double fa2[4], fr2[4];
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
            fa2[tt] = fa[t * kSlab + 32u * tt + l32];
            fr2[tt] = fr[t * kSlab + 32u * tt + l32];
        }
        __syncthreads();

        for (uint32_t k = wave; k < seg_len; k += kMfmaWaves) {
            const uint64_t vv = v + k;
            const uint32_t g64 = (uint32_t)(vv - tb) + 2u * t;
            const uint32_t row0 = g64 * kRows64;
            // this lane's A rows: row0 + 32*m + l32 (both inside one slab: 64 | 128)
            const uint4 *ai = alt + ((size_t)(row0 / kSlab) * nchunks) * kSlab + (row0 % kSlab) + l32;
            v16i acc[2][4];
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int tt = 0; tt < 4; ++tt)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[m][tt][e] = 0;

            uint4 a_raw[2], b_raw[4];
#pragma unroll
            for (int m = 0; m < 2; ++m) a_raw[m] = ai[32 * m];
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) b_raw[tt] = jt[32 * tt + l32];
            for (uint32_t c = 0; c < nchunks; ++c) {
                uint4 a_nxt[2], b_nxt[4];
                const uint32_t cn = c + 1 < nchunks ? c + 1 : c;   // last prefetch re-reads a valid chunk
#pragma unroll
                for (int m = 0; m < 2; ++m) a_nxt[m] = ai[(size_t)cn * kSlab + 32 * m];
#pragma unroll
                for (int tt = 0; tt < 4; ++tt) b_nxt[tt] = jt[cn * kSlab + 32 * tt + l32];
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    v4i af[2], bf[4];
#pragma unroll
                    for (int m = 0; m < 2; ++m) af[m] = expand16(word_of(a_raw[m], w) >> sh);
#pragma unroll
                    for (int tt = 0; tt < 4; ++tt) bf[tt] = expand16(word_of(b_raw[tt], w) >> sh);
#pragma unroll
                    for (int m = 0; m < 2; ++m)
#pragma unroll
                        for (int tt = 0; tt < 4; ++tt)
                            acc[m][tt] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[m], bf[tt], acc[m][tt], 0, 0, 0);
                }
#pragma unroll
                for (int m = 0; m < 2; ++m) a_raw[m] = a_nxt[m];
#pragma unroll
                for (int tt = 0; tt < 4; ++tt) b_raw[tt] = b_nxt[tt];
            }

            // epilogue: acc[m][tt][e] is pair (i, j) with
            //   i = row0 + 32*m + (e & 3) + 8*(e >> 2) + 4*half,  j = 128*t + 32*tt + l32
#pragma unroll
            for (int m = 0; m < 2; ++m) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const uint32_t ri = 32u * m + (e & 3) + 8u * (e >> 2) + 4u * half;   // row inside the unit
                    const uint32_t i = row0 + ri;
                    const double fa1 = fa[i], fr1 = fr[i], q1 = q[i];
                    const uint64_t us = vv * 8u + ri / kGroup;   // the small unit this row belongs to
                    const bool in_range = us >= u_begin && us < u_end;
#pragma unroll
                    for (int tt = 0; tt < 4; ++tt) {
                        const uint32_t jl = 32u * tt + l32;
                        const uint32_t j = t * kSlab + jl;
                        const bool valid = (i > j) && (i < n_snps);
                        const uint32_t cnt = (uint32_t)acc[m][tt][e];
                        ldx_ld32 res = {0.0f, 0.0f};
                        ldx_ld64 rw = {0.0, 0.0};
                        if (valid) {
                            const LdRaw lr = ld_epilogue((double)cnt / n, fa1, fr1, q1, fa2[tt], fr2[tt]);
                            res = round_pair(lr);
                            rw.r_square = lr.rsq;
                            rw.d_prime = lr.dprime;
                        }
                        if (in_range) {
                            const size_t o = (size_t)(us - u_begin) * LDX_UNIT_PAIRS + (size_t)(ri % kGroup) * kSlab + jl;
                            out[o] = res;
                            if (kRaw) raw[o] = rw;
                            if (kN11) n11[o] = valid ? cnt : 0u;
                        }
                    }
                }
            }
        }
        v = seg_end;
        ++t;
    }
}

template <bool kRaw, bool kN11>
static int launch_mfma(const void *alt, const double *fa, const double *fr, const double *q, uint32_t n_snps,
                       uint32_t n_hap, uint64_t unit_begin, uint64_t unit_end, ldx_ld32 *out, ldx_ld64 *out_raw,
                       uint32_t *out_n11, hipStream_t s)
{
    const uint32_t nch = n_chunks(n_hap);
    const size_t lds = (size_t)nch * kSlab * 16u;
    LDX_HIP(hipFuncSetAttribute((const void *)triangle_mfma_kernel<kRaw, kN11>,
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int dev = 0, cus = 256;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess &&
        prop.multiProcessorCount > 0)
        cus = prop.multiProcessorCount;
    const uint64_t total = (unit_end + 7u) / 8u - unit_begin / 8u;
    uint64_t grid = (uint64_t)cus;
    const uint64_t max_grid = (total + kMfmaWaves - 1) / kMfmaWaves;
    if (grid > max_grid) grid = max_grid;
    if (grid < 1) grid = 1;
    triangle_mfma_kernel<kRaw, kN11><<<(uint32_t)grid, kMfmaThreads, lds, s>>>(
        (const uint4 *)alt, fa, fr, q, n_snps, n_slabs(n_snps), nch, (double)n_hap, unit_begin, unit_end, out, out_raw,
        out_n11);
    LDX_HIP(hipGetLastError());
    return LDX_OK;
}

int triangle_mfma(const void *alt, const double *fa, const double *fr, const double *q, uint32_t n_snps, uint32_t n_hap,
                  uint64_t unit_begin, uint64_t unit_end, ldx_ld32 *out, ldx_ld64 *out_raw, uint32_t *out_n11,
                  hipStream_t s)
{
    if (out_raw && out_n11)
        return launch_mfma<true, true>(alt, fa, fr, q, n_snps, n_hap, unit_begin, unit_end, out, out_raw, out_n11, s);
    if (out_raw)
        return launch_mfma<true, false>(alt, fa, fr, q, n_snps, n_hap, unit_begin, unit_end, out, out_raw, out_n11, s);
    if (out_n11)
        return launch_mfma<false, true>(alt, fa, fr, q, n_snps, n_hap, unit_begin, unit_end, out, out_raw, out_n11, s);
    return launch_mfma<false, false>(alt, fa, fr, q, n_snps, n_hap, unit_begin, unit_end, out, out_raw, out_n11, s);
}

}  // namespace ldx
