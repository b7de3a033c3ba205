// Packing of allele codes into the tiled 2-plane SNP x haplotype matrix, per-SNP counts and
// frequency vectors.  Replaces the per-pair genotype list assembly of
// ld_triangle.py:160-186 / ld_area.py:182-187,230-235 (each SNP is packed once) and the
// per-pair list scans of calc_ld.py:37-44 (allele counts are per-SNP constants).
#include "ldx_common.h"

namespace ldx {

// 16 codes (one uint4 of int8) -> 16 alt bits (code == 1) and 16 ref bits (code == 0).
// Per 4-byte word, SWAR: bit 7 of a byte of `le1` is set iff the code is 0 or 1 (exact for every byte value: the
// 7-bit add cannot carry out of its byte); the code's own bit 0, moved to bit 7, splits that into the ALT and REF
// flags.  The four flag bytes of a word (0x80 or 0) are gathered by ONE v_dot4_u32_u8 against per-byte weights
// 2^i (bits 0-3) or 2^(4+i) (bits 4-7): the sum is 128 x the 8-bit group of two words.
__device__ inline void codes16(uint4 v, uint32_t &alt, uint32_t &ref)
{
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
    uint32_t a2[2] = {0u, 0u}, r2[2] = {0u, 0u};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t m = w[k] & 0xFEFEFEFEu;
        const uint32_t t = ((m & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | m;   // bit 7: the code has a bit other than bit 0
        const uint32_t le1 = ~t & 0x80808080u;
        const uint32_t hi = w[k] << 7;                               // bit 7 of each byte: the code's bit 0
        const uint32_t wt = (k & 1) ? 0x80402010u : 0x08040201u;
        a2[k >> 1] = __builtin_amdgcn_udot4(le1 & hi, wt, a2[k >> 1], false);
        r2[k >> 1] = __builtin_amdgcn_udot4(le1 & ~hi, wt, r2[k >> 1], false);
    }
    alt = (a2[0] >> 7) | (a2[1] << 1);
    ref = (r2[0] >> 7) | (r2[1] << 1);
}

// One workgroup per (slab, pair of chunks): 128 SNP rows x 256 haplotypes = 32 KiB of codes.  A thread owns 32
// consecutive haplotypes of a row (two 16-byte loads; a wave covers 8 rows x 256 contiguous bytes) in four rows, all
// eight loads issued before the first use; the 32 ALT / 32 REF bits go to LDS as one word each, and after one
// barrier the workgroup writes the two chunk images of the slab -- 2 x 2 KiB per plane, contiguous in the tiled
// layout -- as one 16-byte store per thread and plane.  Every element of the planes is written (pad rows and pad
// haplotypes as zero bits), so the planes need no memset; the per-SNP counts are accumulated with integer atomics
// (one per row, plane and workgroup) into vectors the caller has zeroed.
__global__ void __launch_bounds__(256) pack_codes_kernel(const int8_t *__restrict__ codes, uint32_t n_snps,
                                                         uint32_t n_hap, size_t ld, uint4 *__restrict__ alt,
                                                         uint4 *__restrict__ ref, uint32_t *__restrict__ acnt,
                                                         uint32_t *__restrict__ rcnt, uint32_t nchunks)
{
    __shared__ uint32_t la[2][kSlab][4], lr[2][kSlab][4];   // [chunk of the pair][row][32-haplotype word]
    const uint32_t t = threadIdx.x, g = t & 7u, rbase = t >> 3;
    const uint32_t slab = blockIdx.y, c0 = blockIdx.x * 2u;
    const uint32_t h0 = c0 * 128u + g * 32u;
    uint4 v[4][2];
    bool fast[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const uint32_t row = slab * kSlab + it * 32u + rbase;
        const int8_t *src = codes + (size_t)(row < n_snps ? row : 0u) * ld + h0;
        fast[it] = row < n_snps && h0 + 32u <= n_hap && ((reinterpret_cast<uintptr_t>(src) & 15u) == 0);
        if (fast[it]) {
            v[it][0] = reinterpret_cast<const uint4 *>(src)[0];
            v[it][1] = reinterpret_cast<const uint4 *>(src)[1];
        }
    }
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const uint32_t rin = it * 32u + rbase, row = slab * kSlab + rin;
        uint32_t a = 0, r = 0;
        if (fast[it]) {
            uint32_t a0, r0, a1, r1;
            codes16(v[it][0], a0, r0);
            codes16(v[it][1], a1, r1);
            a = a0 | (a1 << 16);
            r = r0 | (r1 << 16);
        } else if (row < n_snps) {   // ragged end of the row or an unaligned row: byte by byte
            const int8_t *src = codes + (size_t)row * ld;
            for (uint32_t k = 0; k < 32u && h0 + k < n_hap; ++k) {
                const int8_t c = src[h0 + k];
                a |= (uint32_t)(c == 1) << k;
                r |= (uint32_t)(c == 0) << k;
            }
        }
        la[g >> 2][rin][g & 3u] = a;
        lr[g >> 2][rin][g & 3u] = r;
        uint32_t ca = __builtin_popcount(a), cr = __builtin_popcount(r);
#pragma unroll
        for (int off = 4; off > 0; off >>= 1) {   // the 8 lanes of a row are adjacent
            ca += __shfl_xor(ca, off);
            cr += __shfl_xor(cr, off);
        }
        if (g == 0 && row < n_snps) {
            if (ca) atomicAdd(&acnt[row], ca);
            if (rcnt && cr) atomicAdd(&rcnt[row], cr);
        }
    }
    block_sync();
    const uint32_t chunk = t >> 7, rin = t & 127u;
    if (c0 + chunk < nchunks) {
        const size_t idx = ((size_t)slab * nchunks + c0 + chunk) * kSlab + rin;
        alt[idx] = *reinterpret_cast<const uint4 *>(&la[chunk][rin][0]);
        if (ref) ref[idx] = *reinterpret_cast<const uint4 *>(&lr[chunk][rin][0]);
    }
}

// row-major bit plane (uint32 words) -> tiled plane; one wavefront per row, lane per 32-bit word
__global__ void __launch_bounds__(256) tile_plane_kernel(const uint32_t *__restrict__ rowmajor, uint32_t n_snps,
                                                         uint32_t n_hap, size_t ldw, uint32_t *__restrict__ tiled,
                                                         uint32_t *__restrict__ cnt, uint32_t nchunks)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t row = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (row >= n_snps) return;
    const uint32_t nwords = (n_hap + 31u) / 32u;
    const uint32_t slab = row / kSlab, rin = row % kSlab;
    uint32_t c = 0;
    for (uint32_t w = lane; w < nwords; w += 64u) {
        uint32_t v = rowmajor[(size_t)row * ldw + w];
        const uint32_t rem = n_hap - w * 32u;
        if (rem < 32u) v &= (1u << rem) - 1u;   // pad bits must be zero
        c += __builtin_popcount(v);
        tiled[(((size_t)slab * nchunks + (w >> 2)) * kSlab + rin) * 4u + (w & 3u)] = v;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off);
    if (lane == 0 && cnt) cnt[row] = c;
}

// fa = a/n, fr = r/n (calc_ld.py:41-44: one correctly rounded division each), q = fa*fr.
__global__ void snp_stats_kernel(const uint32_t *__restrict__ acnt, const uint32_t *__restrict__ rcnt,
                                 uint32_t n_snps, uint32_t n_pad, double n, double *__restrict__ fa,
                                 double *__restrict__ fr, double *__restrict__ q)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pad) return;
    double a = 0.0, r = 0.0;
    if (i < n_snps) {
        a = (double)acnt[i] / n;
        r = (double)rcnt[i] / n;
    }
    fa[i] = a;
    fr[i] = r;
    q[i] = a * r;
}

// round(a/n, 4) per SNP: var_i_alt_freq of calc_ld.py:96-97 and the query alt_freq of ld_area.py:188-189
__global__ void alt_freq4_kernel(const uint32_t *__restrict__ acnt, uint32_t n_snps, double n, double *__restrict__ out)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_snps) out[i] = round4_k((double)acnt[i] / n) / 1e4;
}

}  // namespace ldx

using namespace ldx;

extern "C" int ldx_pack_codes_dev(const int8_t *codes, uint32_t n_snps, uint32_t n_hap, size_t ld_codes,
                                  void *alt, void *ref, uint32_t *acnt, uint32_t *rcnt, void *stream)
{
    LDX_REQUIRE(codes && alt && acnt, "codes, alt and acnt must be non-null");
    LDX_REQUIRE((ref == nullptr) == (rcnt == nullptr), "ref and rcnt must be given together");
    LDX_REQUIRE(n_snps > 0 && n_hap > 0 && ld_codes >= n_hap, "bad shape");
    hipStream_t s = (hipStream_t)stream;
    const uint32_t npad = ldx_padded_snps(n_snps);
    LDX_HIP(hipMemsetAsync(acnt, 0, npad * sizeof(uint32_t), s));   // the planes are written in full by the kernel
    if (ref) LDX_HIP(hipMemsetAsync(rcnt, 0, npad * sizeof(uint32_t), s));
    const uint32_t nch = n_chunks(n_hap);
    pack_codes_kernel<<<dim3((nch + 1u) / 2u, n_slabs(n_snps)), 256, 0, s>>>(codes, n_snps, n_hap, ld_codes, (uint4 *)alt,
                                                                             (uint4 *)ref, acnt, rcnt, nch);
    LDX_HIP(hipGetLastError());
    return LDX_OK;
}

extern "C" int ldx_tile_plane_dev(const uint32_t *rowmajor, uint32_t n_snps, uint32_t n_hap, size_t ld_words,
                                  void *tiled, uint32_t *cnt, void *stream)
{
    LDX_REQUIRE(rowmajor && tiled, "null pointer");
    LDX_REQUIRE(n_snps > 0 && n_hap > 0 && ld_words * 32u >= n_hap, "bad shape");
    hipStream_t s = (hipStream_t)stream;
    LDX_HIP(hipMemsetAsync(tiled, 0, ldx_plane_bytes(n_snps, n_hap), s));
    if (cnt) LDX_HIP(hipMemsetAsync(cnt, 0, ldx_padded_snps(n_snps) * sizeof(uint32_t), s));
    tile_plane_kernel<<<(n_snps + 3u) / 4u, 256, 0, s>>>(rowmajor, n_snps, n_hap, ld_words, (uint32_t *)tiled, cnt,
                                                        n_chunks(n_hap));
    LDX_HIP(hipGetLastError());
    return LDX_OK;
}

extern "C" int ldx_snp_stats_dev(const uint32_t *acnt, const uint32_t *rcnt, uint32_t n_snps, uint32_t n_hap,
                                 double *fa, double *fr, double *q, void *stream)
{
    LDX_REQUIRE(acnt && rcnt && fa && fr && q, "null pointer");
    LDX_REQUIRE(n_snps > 0 && n_hap > 0, "bad shape");
    const uint32_t npad = ldx_padded_snps(n_snps);
    snp_stats_kernel<<<(npad + 255u) / 256u, 256, 0, (hipStream_t)stream>>>(acnt, rcnt, n_snps, npad, (double)n_hap,
                                                                           fa, fr, q);
    LDX_HIP(hipGetLastError());
    return LDX_OK;
}

extern "C" int ldx_alt_freq4_dev(const uint32_t *acnt, uint32_t n_snps, uint32_t n_hap, double *freq4, void *stream)
{
    LDX_REQUIRE(acnt && freq4, "null pointer");
    LDX_REQUIRE(n_snps > 0 && n_hap > 0, "bad shape");
    alt_freq4_kernel<<<(n_snps + 255u) / 256u, 256, 0, (hipStream_t)stream>>>(acnt, n_snps, (double)n_hap, freq4);
    LDX_HIP(hipGetLastError());
    return LDX_OK;
}
