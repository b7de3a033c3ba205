// ld_triangle on the matrix cores: n11 = G . G^T with {0,1} int8 operands (v_mfma_i32_32x32x32_i8),
// fused with the same fp64 epilogue as the popcount kernel.  calc_ld.py:32 for 8192 pairs per wave unit.
//
// Why: AND + BCNT run at 64 lanes/clk/CU (no packed form), i.e. 16 haplotype-pairs per lane-instruction;
// the int8 MFMA does 1024 MACs/clk/SIMD -- 4x the VALU ceiling -- so the count moves to the matrix pipe
// and the VALU is left with expanding bits to bytes and with the epilogue.
//
// Structure
//   * 256-thread workgroups (4 waves), two per CU (<= 256 VGPRs, 36 KiB LDS each): one wave per SIMD and
//     workgroup, so the partner on a SIMD belongs to the OTHER workgroup and drifts out of phase -- one
//     runs its VALU epilogue while the other feeds the matrix pipe.
//   * a wave's unit is 64 i-rows x 128 j-rows: 2 x 4 accumulator tiles of 32x32 (128 VGPRs); the four
//     waves of a workgroup take four consecutive units of the same j-tile.
//   * B side (the 128 j-rows, shared by the 4 waves): per 128-haplotype chunk the workgroup expands the
//     j-tile's bits to bytes ONCE (each lane: 64 bits -> 4 x ds_write_b128) into a double-buffered LDS
//     image [128 rows][144 B] (rows padded by 16 B: a 16-lane group of ds_read_b128 then hits 16 distinct
//     16-byte slots); one barrier per chunk.  Fragments are plain ds_read_b128.
//   * A side (a wave's own 64 rows): each lane loads the 16-byte chunk of "its" row (row = lane % 32 of
//     each 32-row tile; 32 consecutive rows of a chunk are 512 contiguous bytes), three chunks deep in
//     registers, and expands 16 bits per K-step in registers (bfe, * 0x00204081, & 0x01010101).
//   * A and B are expanded from the same bit positions by the same arithmetic, so whatever order the
//     hardware gives the 16 k-slots of a lane, slot s of A meets slot s of B: the sum over k is the
//     AND-popcount.  Row/column placement follows the documented 32x32 C/D map
//     (col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)); tests compare every cell with
//     the popcount kernel and the oracle.
#include <stdlib.h>

#include "ldx_common.h"
#include "ldx_tile.h"

namespace ldx {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

#ifndef LDX_MFMA_WAVES
#define LDX_MFMA_WAVES 4   // waves per workgroup: 4 (two workgroups per CU) or 8 (one per CU, B shared 8 ways)
#endif
constexpr int kMfmaWaves = LDX_MFMA_WAVES;
constexpr int kMfmaThreads = kMfmaWaves * 64;
constexpr uint32_t kRows64 = 64;              // i-rows per wave unit
constexpr uint32_t kBRow = 144;               // bytes per expanded j-row in LDS (128 + 16 pad)
constexpr uint32_t kBBuf = kSlab * kBRow;     // one buffer: 18 KiB

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

// tuning-only build variants (python ld_tools_amd/build.py --out libldx_x.so -DLDX_AB_...; results are wrong)
#ifdef LDX_AB_NOAEXP
#define EXPAND_A(x) v4i{(int)(x), 1, 1, 1}
#else
#define EXPAND_A(x) expand16(x)
#endif
#ifdef LDX_AB_NOBEXP
#define EXPAND_B(x) v4i{(int)(x), 1, 1, 1}
#else
#define EXPAND_B(x) expand16(x)
#endif

__device__ __forceinline__ uint32_t word_of(const uint4 &v, int w)
{
    return w == 0 ? v.x : (w == 1 ? v.y : (w == 2 ? v.z : v.w));
}

// this thread's share of the j-tile expansion for one chunk: row = tid / 2, 64 haplotypes (tid % 2)
__device__ __forceinline__ void expand_b_share(uint2 bits, unsigned char *buf, uint32_t tid)
{
    v4i *dst = reinterpret_cast<v4i *>(buf + (tid >> 1) * kBRow + (tid & 1u) * 64u);
    dst[0] = expand16(bits.x);
    dst[1] = expand16(bits.x >> 16);
    dst[2] = expand16(bits.y);
    dst[3] = expand16(bits.y >> 16);
}

template <bool kRaw, bool kN11>
__global__ void __launch_bounds__(kMfmaThreads, 2)
triangle_mfma_kernel(const uint4 *__restrict__ alt, const double *__restrict__ fa, const double *__restrict__ fr,
                     const double *__restrict__ q, uint32_t n_snps, uint32_t n_slabs, uint32_t nchunks, double n,
                     double rn, uint64_t u_begin, uint64_t u_end, ldx_ld32 *__restrict__ out, ldx_ld64 *__restrict__ raw,
                     uint32_t *__restrict__ n11, int ablate_arg)
{
    // experiment 64: workgroups of the first half of the grid run the K loop only, the others the epilogue
    // only (do the two phases overlap when they sit on the same SIMD?); 128: only the first half works
    int ablate = ablate_arg;
    if (ablate_arg & 64) ablate = (blockIdx.x < (gridDim.x + 1) / 2) ? (5 | 8) : (2 | 8);
    if ((ablate_arg & 128) && blockIdx.x >= (gridDim.x + 1) / 2) return;
    // `ablate` (env LDX_ABLATE, tuning only; 0 in production): 1 = no epilogue arithmetic, 2 = one chunk
    // instead of all (no counting), 4 = no stores.  Results are wrong by design when it is non-zero.
    extern __shared__ uint4 lds[];
    unsigned char *bexp = reinterpret_cast<unsigned char *>(lds);   // [2][128][144]
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63u;
    const uint32_t l32 = lane & 31u;
    const uint32_t half = lane >> 5;
    const uint32_t sh = half * 16u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // 64-row units: tile t owns groups g64 in [2t, 2T); unit v <-> small units [8v, 8v+8)
    const uint64_t G64 = (uint64_t)n_slabs * 2u;
    auto base64 = [&](uint64_t t) { return t * G64 - t * (t - 1u); };
    const uint64_t v_begin = u_begin / 8u, v_end = (u_end + 7u) / 8u;   // units that intersect the range
    const uint64_t total = v_end - v_begin;
    const uint64_t b0 = v_begin + total * blockIdx.x / gridDim.x;
    const uint64_t b1 = v_begin + total * (blockIdx.x + 1) / gridDim.x;
    if (b0 >= b1) return;   // block-uniform

    // Stagger: all workgroups do identical work, so the two that share a CU would run their K loops
    // (matrix pipe) and their epilogues (VALU) at the same time and the pipes would take turns.  The
    // second half of the grid (dispatched onto the CUs' second slots) starts half a period late, so one
    // workgroup's epilogue runs beside the other's K loop.  Speed only; any placement is correct.
    const bool late = (ablate & 16) ? (blockIdx.x & 1u) != 0 : ((ablate & 32) ? ((blockIdx.x >> 3) & 1u) != 0
                                                                              : blockIdx.x >= (gridDim.x + 1) / 2);
    if ((ablate & 8) == 0 && late && b1 - b0 >= 8) {
        const uint32_t naps = nchunks / 4 + 8;   // ~ (nchunks * 1024 + 32k) / 2 cycles in naps of 64 * 32
        for (uint32_t k = 0; k < naps; ++k) __builtin_amdgcn_s_sleep(32);
    }

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
    while (v < b1) {   // block-uniform: every wave reaches every barrier
        const uint64_t tb = base64(t), te = base64(t + 1u);
        const uint64_t seg_end = b1 < te ? b1 : te;
        // the j-tile's bits for this thread's expansion share: row tid/2, 8 bytes (tid%2) of each chunk
#if LDX_MFMA_WAVES == 8
        // 512 threads: row tid/4, one 32-bit word (tid%4) of each chunk
        const uint32_t *bsrc = reinterpret_cast<const uint32_t *>(alt + (size_t)t * nchunks * kSlab) + tid;
        constexpr uint32_t kBStride = kSlab * 4u;   // words per chunk
        const uint32_t b_off = (tid >> 2) * kBRow + (tid & 3u) * 32u;
#else
        const uint2 *bsrc = reinterpret_cast<const uint2 *>(alt + (size_t)t * nchunks * kSlab) + tid;
        constexpr uint32_t kBStride = kSlab * 2u;   // uint2 per chunk
        const uint32_t b_off = (tid >> 1) * kBRow + (tid & 1u) * 64u;
#endif

        for (uint64_t pass = v; pass < seg_end; pass += kMfmaWaves) {   // block-uniform
            const uint64_t vv = pass + wave;
            const bool active = vv < seg_end;
            const uint32_t g64 = (uint32_t)((active ? vv : pass) - tb) + 2u * t;
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

            // ---- K loop, software-pipelined at K-step (32 haplotypes) granularity -------------------------
            // During the 8 MFMAs of step s the wave (i) has the B fragments of step s+1 in flight from LDS,
            // (ii) expands the A fragments of step s+1 and (iii) a quarter of its share of the NEXT chunk's
            // B image.  One workgroup barrier per chunk, placed before the last K-step of the chunk: by then
            // every wave has written its share of chunk c+1 (steps 0..1) and issued its last read of chunk c
            // (the prefetch of step 3, done in step 2), so after it the fragments of (c+1, step 0) can be
            // prefetched and the buffer of chunk c may be overwritten by chunk c+2.  sched_barrier(0) between
            // steps keeps hipcc from hoisting a whole chunk's expansions ahead of the first MFMA.
            auto read_bf = [&](v4i (&bf)[4], const unsigned char *buf, int w) {
#pragma unroll
                for (int tt = 0; tt < 4; ++tt)
#ifdef LDX_AB_NOBREAD
                    bf[tt] = v4i{(int)(uintptr_t)buf + w + tt, 1, 1, 1};
#else
                    bf[tt] = *reinterpret_cast<const v4i *>(buf + (32u * tt + l32) * kBRow + w * 32u + half * 16u);
#endif
            };
            auto mma8 = [&](const v4i (&af)[2], const v4i (&bf)[4]) {
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int tt = 0; tt < 4; ++tt)
                        acc[m][tt] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[m], bf[tt], acc[m][tt], 0, 0, 0);
            };
            auto interleave = [&]() {   // 8 x {1 MFMA, 5 VALU}: the VALU work of a step hides behind its MFMAs
                // the next step's four B-fragment reads go FIRST: a whole step (256 cycles) of cover for the LDS
                // latency; left to itself hipcc sinks them to the end of the step and the next step's first
                // MFMAs wait on lgkmcnt
                __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);
                }
            };
            auto lds_barrier = [&]() {   // LDS-only barrier: no vmcnt(0), the global prefetches stay in flight
                __builtin_amdgcn_sched_barrier(0);
#ifndef LDX_AB_NOBARRIER
                __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0)
                __builtin_amdgcn_s_barrier();
#endif
                __builtin_amdgcn_sched_barrier(0);
            };

            __syncthreads();   // the previous pass has finished reading both buffers
            {   // chunk 0 -> buffer 0
                v4i *d0 = reinterpret_cast<v4i *>(bexp + b_off);
#if LDX_MFMA_WAVES == 8
                const uint32_t w0 = bsrc[0];
                d0[0] = expand16(w0);
                d0[1] = expand16(w0 >> 16);
#else
                const uint2 w0 = bsrc[0];
                d0[0] = expand16(w0.x);
                d0[1] = expand16(w0.x >> 16);
                d0[2] = expand16(w0.y);
                d0[3] = expand16(w0.y >> 16);
#endif
            }
            // Global prefetch ring, 3 chunks deep, statically indexed (the chunk loop is unrolled by 3 so no
            // register is ever MOVED: a move of a register with a load in flight is a wait).  During chunk c
            // the loads of chunk c+2 are issued; the A words are first touched at the end of chunk c+1 (the j
            // bits, re-read by every pass of the tile and therefore cache-hot, at its start).  With one chunk of
            // cover the K loop ran at the latency of those loads (~1900 cycles per chunk, matrix pipe 47 % busy
            // for a lone wave) instead of at the 1024 cycles of its 32 MFMAs.
            auto clampc = [&](uint32_t c) { return c < nchunks ? c : nchunks - 1u; };   // surplus loads are discarded
            uint4 ar[3][2];
            decltype(bsrc[0] + 0) br[3];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                br[k] = bsrc[(size_t)clampc(k) * kBStride];
#pragma unroll
                for (int m = 0; m < 2; ++m) ar[k][m] = ai[(size_t)clampc(k) * kSlab + 32 * m];
            }
            __syncthreads();
            v4i af0[2], bf0[4], af1[2], bf1[4];
            read_bf(bf0, bexp, 0);
#pragma unroll
            for (int m = 0; m < 2; ++m) af0[m] = EXPAND_A(ar[0][m].x >> sh);

            const uint32_t nch_run = (ablate & 2) ? 1u : nchunks;
            // one chunk: ring slot CUR holds its A words, slot NXT the next chunk's (A words and B bits),
            // slot FAR receives chunk c+2
#define LDX_CHUNK(CUR, NXT, FAR, cc)                                                                               \
            {                                                                                                      \
                const uint32_t c_ = (cc);                                                                          \
                const unsigned char *rd = bexp + (c_ & 1u) * kBBuf;                                                \
                unsigned char *wr = bexp + ((c_ + 1u) & 1u) * kBBuf;                                               \
                v4i *bdst = reinterpret_cast<v4i *>(wr + b_off);                                                   \
                const uint32_t c3 = clampc(c_ + 2u);                                                               \
                /* B bits FIRST: vmcnt counts in order, so waiting for them (next chunk, step 0) must not */      \
                /* also wait for this batch's A words (needed only at the end of the next chunk) */               \
                br[FAR] = bsrc[(size_t)c3 * kBStride];                                                             \
                __builtin_amdgcn_sched_barrier(0);                                                                 \
                _Pragma("unroll") for (int m = 0; m < 2; ++m) ar[FAR][m] = ai[(size_t)c3 * kSlab + 32 * m];        \
                /* step 0: MFMAs of (c,0); prepare (c,1); first half of this thread's share of B chunk c+1 */     \
                read_bf(bf1, rd, 1);                                                                               \
                _Pragma("unroll") for (int m = 0; m < 2; ++m) af1[m] = EXPAND_A(ar[CUR][m].y >> sh);               \
                LDX_BSHARE_0(br[NXT]);                                                                             \
                mma8(af0, bf0);                                                                                    \
                interleave();                                                                                      \
                __builtin_amdgcn_sched_barrier(0);                                                                 \
                /* step 1: MFMAs of (c,1); prepare (c,2); second half of the B share */                           \
                read_bf(bf0, rd, 2);                                                                               \
                _Pragma("unroll") for (int m = 0; m < 2; ++m) af0[m] = EXPAND_A(ar[CUR][m].z >> sh);               \
                LDX_BSHARE_1(br[NXT]);                                                                             \
                mma8(af1, bf1);                                                                                    \
                interleave();                                                                                      \
                __builtin_amdgcn_sched_barrier(0);                                                                 \
                /* step 2: MFMAs of (c,2); prepare (c,3) */                                                       \
                read_bf(bf1, rd, 3);                                                                               \
                _Pragma("unroll") for (int m = 0; m < 2; ++m) af1[m] = EXPAND_A(ar[CUR][m].w >> sh);               \
                mma8(af0, bf0);                                                                                    \
                interleave();                                                                                      \
                lds_barrier(); /* chunk c+1 complete in `wr`; nobody reads `rd` any more */                       \
                /* step 3: MFMAs of (c,3); prepare (c+1,0) from the other buffer and the next A chunk */          \
                read_bf(bf0, wr, 0);                                                                               \
                _Pragma("unroll") for (int m = 0; m < 2; ++m) af0[m] = EXPAND_A(ar[NXT][m].x >> sh);               \
                mma8(af1, bf1);                                                                                    \
                interleave();                                                                                      \
                __builtin_amdgcn_sched_barrier(0);                                                                 \
            }
#if LDX_MFMA_WAVES == 8
#define LDX_BSHARE_0(bits) bdst[0] = EXPAND_B(bits)
#define LDX_BSHARE_1(bits) bdst[1] = EXPAND_B((bits) >> 16)
#else
#define LDX_BSHARE_0(bits) bdst[0] = EXPAND_B((bits).x); bdst[1] = EXPAND_B((bits).x >> 16)
#define LDX_BSHARE_1(bits) bdst[2] = EXPAND_B((bits).y); bdst[3] = EXPAND_B((bits).y >> 16)
#endif
            for (uint32_t c = 0; c < nch_run; c += 3) {   // block-uniform guards: every wave reaches every barrier
                LDX_CHUNK(0, 1, 2, c)
                if (c + 1 < nch_run) LDX_CHUNK(1, 2, 0, c + 1)
                if (c + 2 < nch_run) LDX_CHUNK(2, 0, 1, c + 2)
            }
#undef LDX_CHUNK
#undef LDX_BSHARE_0
#undef LDX_BSHARE_1

            if (!active) continue;   // wave-uniform; inactive waves only helped with B and the barriers
            // epilogue: acc[m][tt][e] is pair (i, j) with
            //   i = row0 + 32*m + (e & 3) + 8*(e >> 2) + 4*half,  j = 128*t + 32*tt + l32
            double fa2[4], fr2[4];
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) {
                fa2[tt] = fa[t * kSlab + 32u * tt + l32];
                fr2[tt] = fr[t * kSlab + 32u * tt + l32];
            }
            // one coalesced load per statistic for the unit's 64 rows (instead of a dependent global load per
            // row inside the loop: 32 exposed latencies per unit)
            const double sfa = fa[row0 + lane], sfr = fr[row0 + lane], sq = q[row0 + lane];
            // The e-loop is NOT unrolled: 128 pairs x ~90 instructions would be ~90 KB of straight-line code
            // per wave, more than the instruction cache two CUs share.  acc[..][..][e] with a wave-uniform e
            // is a register-indirect move (s_set_gpr_idx_on), not scratch.
#pragma unroll 1
            for (int e = 0; e < 16; ++e) {
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    const uint32_t ri = 32u * m + (e & 3) + 8u * (e >> 2) + 4u * half;   // row inside the unit
                    const uint32_t i = row0 + ri;
                    // lane L preloaded the statistics of row row0 + L: fetch row ri's through the LDS crossbar
                    const double fa1 = __shfl(sfa, (int)ri), fr1 = __shfl(sfr, (int)ri), q1 = __shfl(sq, (int)ri);
                    const uint64_t us = vv * 8u + ri / kGroup;   // the small unit this row belongs to
                    const bool in_range = us >= u_begin && us < u_end;
                    // The four pairs of this row (one per column tile) go through the fast epilogue WITHOUT
                    // branches -- invalid cells (row <= col, pad rows) are computed on whatever the registers
                    // hold and zeroed by a select -- so their four dependent fp64 chains interleave; with a
                    // branch per pair a lone wave spent 545 cycles per pair on a 50-instruction epilogue.
                    uint32_t cnt[4];
                    double f11[4];
                    ldx_ld32 res[4];
                    ldx_ld64 rw[4];
                    bool valid[4], slow[4];
                    bool any_slow = false;
#pragma unroll
                    for (int tt = 0; tt < 4; ++tt) {
                        const uint32_t j = t * kSlab + 32u * tt + l32;
                        valid[tt] = (i > j) && (i < n_snps);
                        cnt[tt] = (uint32_t)acc[m][tt][e];
                        f11[tt] = div_by_n((double)cnt[tt], n, rn);   // calc_ld.py:33
                        rw[tt] = ldx_ld64{0.0, 0.0};
                        if (ablate & 1) {   // tuning: no epilogue arithmetic
                            res[tt] = ldx_ld32{(float)cnt[tt], 0.0f};
                            slow[tt] = false;
                        } else if (kRaw) {   // parity / debugging output: the op-for-op mirror, unrounded values kept
                            const LdRaw lr = ld_epilogue(f11[tt], fa1, fr1, q1, fa2[tt], fr2[tt]);
                            res[tt] = round_pair(lr);
                            if (valid[tt]) rw[tt] = ldx_ld64{lr.rsq, lr.dprime};
                            slow[tt] = false;
                        } else {
                            res[tt] = ld_pair_fast(f11[tt], fa1, fr1, q1, fa2[tt], fr2[tt], slow[tt]);
                            slow[tt] = slow[tt] && valid[tt];
                            any_slow = any_slow || slow[tt];
                        }
                    }
                    if (!kRaw && __builtin_expect(__any(any_slow), 0)) {   // near a rounding tie: the exact mirror
#pragma unroll
                        for (int tt = 0; tt < 4; ++tt)
                            if (slow[tt]) res[tt] = ld_pair_mirror(f11[tt], fa1, fr1, q1, fa2[tt], fr2[tt]);
                    }
                    if (in_range && !(ablate & 4)) {
#pragma unroll
                        for (int tt = 0; tt < 4; ++tt) {
                            const uint32_t jl = 32u * tt + l32;
                            const size_t o = (size_t)(us - u_begin) * LDX_UNIT_PAIRS + (size_t)(ri % kGroup) * kSlab + jl;
                            ldx_ld32 w = res[tt];
                            if (!valid[tt]) w = ldx_ld32{0.0f, 0.0f};
                            if (ablate & 1) w.r_square = (float)cnt[tt];
                            out[o] = w;
                            if (kRaw) raw[o] = rw[tt];
                            if (kN11) n11[o] = valid[tt] ? cnt[tt] : 0u;
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
    const size_t lds = 2u * kBBuf;
    int dev = 0, cus = 256;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess &&
        prop.multiProcessorCount > 0)
        cus = prop.multiProcessorCount;
    const uint64_t total = (unit_end + 7u) / 8u - unit_begin / 8u;
    uint64_t grid = (uint64_t)cus * (8u / kMfmaWaves);   // persistent: 8 waves per CU (two 4-wave workgroups or one of 8)
    const uint64_t max_grid = (total + kMfmaWaves - 1) / kMfmaWaves;
    if (grid > max_grid) grid = max_grid;
    if (grid < 1) grid = 1;
    triangle_mfma_kernel<kRaw, kN11><<<(uint32_t)grid, kMfmaThreads, lds, s>>>(
        (const uint4 *)alt, fa, fr, q, n_snps, n_slabs(n_snps), nch, (double)n_hap, 1.0 / (double)n_hap, unit_begin,
        unit_end, out, out_raw,
        out_n11, getenv("LDX_ABLATE") ? atoi(getenv("LDX_ABLATE")) : 0);
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

// ---- peak-rate probe for the matrix pipe: back-to-back int8 MFMAs on 8 independent accumulators,
// operands in registers, no memory traffic.  variant 0: 32x32x32 (32 K MACs), 1: 16x16x64 (16 K MACs).
namespace ldx {
typedef int v4i_p __attribute__((ext_vector_type(4)));
typedef int v16i_p __attribute__((ext_vector_type(16)));
template <int kVariant>
__global__ void __launch_bounds__(256) probe_mfma_kernel(uint32_t *__restrict__ sink, uint32_t iters)
{
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    v4i_p a = {(int)(tid & 0x01010101u), 0x01000100, 0x00010001, 0x01010000};
    v4i_p b = {0x01010101, (int)((tid >> 3) & 0x01010101u), 0x00000101, 0x01000001};
    uint32_t s = 0;
    if (kVariant == 0) {
        v16i_p acc[8];
#pragma unroll
        for (int k = 0; k < 8; ++k)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[k][e] = 0;
        for (uint32_t it = 0; it < iters; ++it) {
#pragma unroll
            for (int k = 0; k < 8; ++k) acc[k] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc[k], 0, 0, 0);
        }
#pragma unroll
        for (int k = 0; k < 8; ++k)
#pragma unroll
            for (int e = 0; e < 16; ++e) s += (uint32_t)acc[k][e];
    } else {
        v4i_p acc[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] = v4i_p{0, 0, 0, 0};
        for (uint32_t it = 0; it < iters; ++it) {
#pragma unroll
            for (int k = 0; k < 8; ++k) acc[k] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, acc[k], 0, 0, 0);
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) s += (uint32_t)(acc[k].x + acc[k].y + acc[k].z + acc[k].w);
    }
    sink[tid] = s;
}
}  // namespace ldx

extern "C" int ldx_probe_mfma_dev(uint32_t *sink, uint32_t blocks, uint32_t threads, uint32_t iters, int variant,
                                  void *stream)
{
    LDX_REQUIRE(sink && blocks >= 1 && threads >= 64 && threads <= 256 && threads % 64 == 0, "bad argument");
    if (variant == 0)
        ldx::probe_mfma_kernel<0><<<blocks, threads, 0, (hipStream_t)stream>>>(sink, iters);
    else
        ldx::probe_mfma_kernel<1><<<blocks, threads, 0, (hipStream_t)stream>>>(sink, iters);
    LDX_HIP(hipGetLastError());
    return LDX_OK;
}
