// All-pairs alt/alt haplotype counts (calc_ld.py:32) fused with the D / D' / r^2 epilogue
// (calc_ld.py:33-97), for the ld_triangle pair loop (ld_triangle.py:133-230).
//
// Execution model (gfx950, wave64)
//   * A workgroup keeps ONE j-tile -- 128 SNP rows x all haplotype chunks, <= 160 KiB -- resident in
//     LDS, copied linearly from the tiled plane (the HBM image is already the LDS image).
//   * A wavefront processes "units": 8 consecutive i-rows against the 128 j-rows of the tile.  The
//     i-rows are wave-uniform, so their words are fetched with scalar loads and sit in SGPRs; each
//     lane owns j-rows `lane` and `lane + 64` and reads them with conflict-free ds_read_b128.  The
//     inner loop is nothing but v_and_b32 (SGPR x VGPR) + v_bcnt_u32_b32 (accumulating).
//   * 16 pair counts per lane stay in registers and go straight through the fp64 epilogue; one
//     8-byte store per pair, 512 contiguous bytes per wave store.
//   * Workgroups (one per CU, 16 waves) are persistent over a contiguous, equal share of the unit list
//     (t-major), so a tile is staged once per workgroup per tile; units are dealt round-robin to the
//     waves (all units cost the same, so a static deal is as good as a queue and needs no LDS word:
//     at 5008 haplotypes the tile alone is 80 KiB).
#include <atomic>

#include "ldx_common.h"
#include "ldx_tile.h"

namespace ldx {

__device__ inline uint32_t find_tile(uint64_t u, uint64_t G, uint32_t T)
{
    uint32_t lo = 0, hi = T;   // largest t with tile_base(t) <= u
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) / 2;
        if (tile_base(mid, G) <= u) lo = mid; else hi = mid;
    }
    return lo;
}

template <bool kRaw, bool kN11, typename Cell>
__global__ void __launch_bounds__(kThreads)
triangle_kernel(const uint4 *__restrict__ alt, const double *__restrict__ fa, const double *__restrict__ fr,
                const double *__restrict__ q, uint32_t n_snps, uint32_t n_slabs, uint32_t nchunks, double n,
                double rn, uint64_t u_begin, uint64_t u_end, Cell *__restrict__ out, ldx_ld64 *__restrict__ raw,
                uint32_t *__restrict__ n11)
{
    extern __shared__ uint4 lds[];
    uint4 *jt = lds;

    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint64_t G = (uint64_t)n_slabs * kGroupsPerSlab;
    const uint64_t total = u_end - u_begin;
    const uint64_t b0 = u_begin + total * blockIdx.x / gridDim.x;
    const uint64_t b1 = u_begin + total * (blockIdx.x + 1) / gridDim.x;
    if (b0 >= b1) return;   // block-uniform

    uint32_t t = find_tile(b0, G, n_slabs);
    uint64_t u = b0;
    while (u < b1) {   // block-uniform trip count: every wave reaches every barrier
        const uint64_t tb = tile_base(t, G);
        const uint64_t te = tile_base(t + 1u, G);
        const uint64_t seg_end = b1 < te ? b1 : te;
        const uint32_t seg_len = (uint32_t)(seg_end - u);
        block_sync();   // everyone is done with the previous tile
        stage_tile(jt, alt + (size_t)t * nchunks * kSlab, nchunks * kSlab);
        const uint32_t j0 = t * kSlab + lane, j1 = j0 + 64u;
        const double fa2[2] = {fa[j0], fa[j1]};
        const double fr2[2] = {fr[j0], fr[j1]};
        block_sync();

        for (uint32_t k = wave; k < seg_len; k += kWaves) {
            const uint64_t uu = u + k;
            const uint32_t g = (uint32_t)(uu - tb) + t * kGroupsPerSlab;   // i-group
            const uint32_t row0 = g * kGroup;
            const uint4 *ai = alt + ((size_t)(row0 / kSlab) * nchunks) * kSlab + (row0 % kSlab);
            Acc acc;
            count_unit(ai, jt, nchunks, lane, acc);

            const size_t obase = (size_t)(uu - u_begin) * LDX_UNIT_PAIRS;
#pragma unroll
            for (int r = 0; r < (int)kGroup; ++r) {
                const uint32_t i = row0 + r;
                const double fa1 = fa[i], fr1 = fr[i], q1 = q[i];   // wave-uniform -> scalar loads
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    const uint32_t j = jj ? j1 : j0;
                    const bool valid = (i > j) && (i < n_snps);   // ld_triangle.py:149-150
                    const size_t o = obase + cell_offset<Cell>((uint32_t)r, jj * 64u + lane);
                    Cell res = zero_cell<Cell>();
                    ldx_ld64 rw = {0.0, 0.0};
                    if (valid) {
                        const double f11 = div_by_n((double)acc.v[r][jj], n, rn);   // calc_ld.py:33
                        LdK lk;
                        if (kRaw) {   // parity / debugging output: the op-for-op mirror, unrounded values kept
                            const LdRaw lr = ld_epilogue(f11, fa1, fr1, q1, fa2[jj], fr2[jj]);
                            lk = round_pair(lr);
                            rw.r_square = lr.rsq;
                            rw.d_prime = lr.dprime;
                        } else {
                            bool slow;
                            lk = ld_pair_fast(f11, fa1, fr1, q1, fa2[jj], fr2[jj], slow);
                            if (__builtin_expect(__any(slow), 0))
                                if (slow) lk = ld_pair_mirror(f11, fa1, fr1, q1, fa2[jj], fr2[jj]);
                        }
                        res = encode_cell<Cell>(lk);
                    }
                    out[o] = res;
                    if (kRaw) raw[o] = rw;
                    if (kN11) n11[o] = valid ? acc.v[r][jj] : 0u;
                }
            }
        }
        u = seg_end;
        ++t;
    }
}

// ---- rectangular n11 block: panel I rows x panel J rows, dense output (the bit-exact contract) ----
__global__ void __launch_bounds__(kThreads)
pair_counts_kernel(const uint4 *__restrict__ alt_i, const uint4 *__restrict__ alt_j, uint32_t n_i, uint32_t n_j,
                   uint32_t nchunks, uint32_t *__restrict__ n11, size_t ld)
{
    extern __shared__ uint4 lds[];
    uint4 *jt = lds;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t t = blockIdx.x;                          // j-tile
    const uint32_t groups_i = (n_i + kGroup - 1) / kGroup;  // the I plane is padded to whole slabs
    stage_tile(jt, alt_j + (size_t)t * nchunks * kSlab, nchunks * kSlab);
    block_sync();
    for (uint32_t g = blockIdx.y * kWaves + wave; g < groups_i; g += gridDim.y * kWaves) {
        const uint32_t row0 = g * kGroup;
        const uint4 *ai = alt_i + ((size_t)(row0 / kSlab) * nchunks) * kSlab + (row0 % kSlab);
        Acc acc;
        count_unit(ai, jt, nchunks, lane, acc);
#pragma unroll
        for (int r = 0; r < (int)kGroup; ++r) {
            const uint32_t i = row0 + r;
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                const uint32_t j = t * kSlab + jj * 64u + lane;
                if (i < n_i && j < n_j) n11[(size_t)i * ld + j] = acc.v[r][jj];
            }
        }
    }
}

// ---- the epilogue alone, one element per thread ----
template <typename Cell>
__device__ inline bool same_cell(const Cell &a, const Cell &b);
template <>
__device__ inline bool same_cell<ldx_ld32>(const ldx_ld32 &a, const ldx_ld32 &b)
{
    return __float_as_uint(a.r_square) == __float_as_uint(b.r_square) && __float_as_uint(a.d_prime) == __float_as_uint(b.d_prime);
}
template <>
__device__ inline bool same_cell<ldx_k16>(const ldx_k16 &a, const ldx_k16 &b)
{
    return a.r_square == b.r_square && a.d_prime == b.d_prime;
}

// Every production epilogue on the same tuple, for one cell format: the reciprocal-based one of the popcount kernels,
// the count-domain fp64 one of the matrix kernels (general variant, and the "clean" variant where it applies; int and
// float count operands), each with its mirror fallback.  They must agree bit for bit; otherwise the result is poisoned.
template <typename Cell>
__device__ inline Cell all_tiers(double n, double rn, uint32_t c11, double fa1, double fr1, double fa2, double fr2, bool &same,
                                 bool &f32_sure)
{
    f32_sure = false;
    const double f11 = div_by_n((double)c11, n, rn);
    const double q1 = fa1 * fr1;
    bool slow;
    LdK lk = ld_pair_fast(f11, fa1, fr1, q1, fa2, fr2, slow);
    if (slow) lk = ld_pair_mirror(f11, fa1, fr1, q1, fa2, fr2);
    const Cell res = encode_cell<Cell>(lk);
    const Cell mir = encode_cell<Cell>(ld_pair_mirror(f11, fa1, fr1, q1, fa2, fr2));
    same = same_cell(res, mir);
    const FastRow fr_[1] = {fast_row(fa1, fr1, n)};
    const FastCol fc_[1] = {fast_col(fa2, fr2, n)};
    const bool ordinary = fast_ordinary(fa1, fr1, n) && fast_ordinary(fa2, fr2, n);
    {
        const FastConst fk = fast_const(n, 8.0);
        const int cnt_[1] = {(int)(c11 * 8u)};
        Cell g_[1];
        bool sg_[1];
        ld_multi_fast2<1, false, Cell>(cnt_, fk, fr_, fc_, g_, sg_);
        if (sg_[0]) g_[0] = mir;
        same = same && same_cell(g_[0], res);
        if (ordinary) {
            ld_multi_fast2<1, true, Cell>(cnt_, fk, fr_, fc_, g_, sg_);
            if (sg_[0]) g_[0] = mir;
            same = same && same_cell(g_[0], res);
        }
    }
    {
        const FastConst fk = fast_const(n, 1.0);
        const float cnt_[1] = {(float)c11};
        Cell g_[1];
        bool sg_[1];
        ld_multi_fast2<1, false, Cell>(cnt_, fk, fr_, fc_, g_, sg_);
        if (sg_[0]) g_[0] = mir;
        same = same && same_cell(g_[0], res);
        if (ordinary) {
            ld_multi_fast2<1, true, Cell>(cnt_, fk, fr_, fc_, g_, sg_);
            if (sg_[0]) g_[0] = mir;
            same = same && same_cell(g_[0], res);
            // the fp32 first tier of the FP4 kernel: whenever it calls a pair sure, its cell must be the mirror's
            const F32Const f32k = f32_const(n);
            const F32Row r32[1] = {f32_row(fr_[0].a_s * 1e-4, fr_[0].ra, fr_[0].rr, kSnpOrdinary, n)};
            const F32Col c32[1] = {f32_col(fc_[0].a, fc_[0].ra, fc_[0].rr, kSnpOrdinary, n)};
            Cell h_[1];
            float wmax = 0.0f, ymin = 1.0f;
            if (f32_small_n(n)) ld_multi_f32<1, Cell, true>(cnt_, f32k, r32, c32, h_, wmax, ymin);   // the variant the kernel picks for this n
            else {   // the split form the kernel uses for n > 4096
                F32Col cs[1] = {c32[0]};
                float al_[1];
                f32_split_a(cs[0].a, cs[0].a, al_[0]);
                ld_multi_f32<1, Cell, false, true>(cnt_, f32k, r32, cs, h_, wmax, ymin, al_);
            }
            f32_sure = (wmax < f32k.tol) & (ymin > 0.0f);
            if (f32_sure) same = same && same_cell(h_[0], res);
        }
    }
    return res;
}

__global__ void ld_from_counts_kernel(double n, double rn, size_t m, const uint32_t *__restrict__ n11,
                                      const uint32_t *__restrict__ a1, const uint32_t *__restrict__ r1,
                                      const uint32_t *__restrict__ a2, const uint32_t *__restrict__ r2,
                                      ldx_ld64 *raw, double *kout, ldx_ld32 *cells32, ldx_k16 *cells16, uint8_t *flags)
{
    const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= m) return;
    const double fa1 = (double)a1[k] / n, fr1 = (double)r1[k] / n;
    const double fa2 = (double)a2[k] / n, fr2 = (double)r2[k] / n;
    // `raw`, `k` and `flags` come from the op-for-op mirror with a real division, the cells from the production
    // paths of the pair kernels -- so the exhaustive small-n test pins exactly the code that ld_triangle / ld_area run.
    const LdRaw lr = ld_epilogue((double)n11[k] / n, fa1, fr1, fa1 * fr1, fa2, fr2);
    if (raw) raw[k] = ldx_ld64{lr.rsq, lr.dprime};
    if (kout) {
        kout[2 * k] = round4_k(lr.rsq);
        kout[2 * k + 1] = round4_k(lr.dprime);
    }
    bool tier = false;
    if (cells32) {
        bool same, sure32;
        ldx_ld32 res = all_tiers<ldx_ld32>(n, rn, n11[k], fa1, fr1, fa2, fr2, same, sure32);
        if (!same) res = ldx_ld32{__uint_as_float(0x7FC00000u), __uint_as_float(0x7FC00000u)};
        cells32[k] = res;
        tier = sure32;
    }
    if (cells16) {
        bool same, sure32;
        ldx_k16 res = all_tiers<ldx_k16>(n, rn, n11[k], fa1, fr1, fa2, fr2, same, sure32);
        if (!same) res = ldx_k16{0xFFFFu, 0xFFFFu};
        cells16[k] = res;
        tier = sure32;
    }
    // bit 7 (LDX_FLAG_F32_SURE, reported by this entry point only): the fp32 tier would have kept the pair
    if (flags) flags[k] = (uint8_t)(lr.flags | (tier ? LDX_FLAG_F32_SURE : 0u));
}

// ---- LD of an explicit list of pairs: one wavefront per pair, AND + popcount over the chunks of the two rows, the
// op-for-op mirror epilogue with real divisions, k = round4 * 10^4 as doubles (exact for any magnitude) ----
__global__ void __launch_bounds__(256) ld_pairs_kernel(const uint4 *__restrict__ alt, const uint32_t *__restrict__ acnt,
                                                       const uint32_t *__restrict__ rcnt, uint32_t n_snps,
                                                       uint32_t nchunks, double n, const uint32_t *__restrict__ rows,
                                                       const uint32_t *__restrict__ cols, size_t m, double *kout,
                                                       ldx_ld64 *raw, uint8_t *flags, uint32_t *n11)
{
    const size_t p = (size_t)blockIdx.x * 4u + (threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63u;
    if (p >= m) return;
    const uint32_t i = rows[p], j = cols[p];
    if (i >= n_snps || j >= n_snps) {   // out of range: poison, never read outside the plane
        if (lane == 0) {
            if (kout) kout[2 * p] = kout[2 * p + 1] = __builtin_nan("");
            if (raw) raw[p] = ldx_ld64{__builtin_nan(""), __builtin_nan("")};
            if (flags) flags[p] = 0xFFu;
            if (n11) n11[p] = 0xFFFFFFFFu;
        }
        return;
    }
    const uint4 *ri = alt + ((size_t)(i / kSlab) * nchunks) * kSlab + (i % kSlab);
    const uint4 *rj = alt + ((size_t)(j / kSlab) * nchunks) * kSlab + (j % kSlab);
    uint32_t c = 0;
    for (uint32_t ch = lane; ch < nchunks; ch += 64u) {
        const uint4 a = ri[(size_t)ch * kSlab], b = rj[(size_t)ch * kSlab];
        c += __builtin_popcount(a.x & b.x) + __builtin_popcount(a.y & b.y) + __builtin_popcount(a.z & b.z) +
             __builtin_popcount(a.w & b.w);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off);
    if (lane == 0) {
        const double fa1 = (double)acnt[i] / n, fr1 = (double)rcnt[i] / n;   // calc_ld.py:41-44
        const double fa2 = (double)acnt[j] / n, fr2 = (double)rcnt[j] / n;
        const LdRaw lr = ld_epilogue((double)c / n, fa1, fr1, fa1 * fr1, fa2, fr2);
        if (kout) {
            kout[2 * p] = round4_k(lr.rsq);
            kout[2 * p + 1] = round4_k(lr.dprime);
        }
        if (raw) raw[p] = ldx_ld64{lr.rsq, lr.dprime};
        if (flags) flags[p] = (uint8_t)lr.flags;
        if (n11) n11[p] = c;
    }
}

// ---- strips -> dense ld_two_dim (ld_triangle.py:114,223-230) ----
// one value of a cell as the dense float: the float32 nearest to k / 10^4, -0.0f for the int 0, the NaN LDX_LD32_BIG_BITS
// for an escape; *k receives k (+inf for an escape)
__device__ inline float dense_value(const ldx_ld32 &c, int measure, double *k)
{
    const float v = measure == LDX_MEASURE_RSQ ? c.r_square : c.d_prime;
    *k = v != v ? __builtin_inf() : __builtin_rint((double)v * 1e4);
    return v;
}
__device__ inline float dense_value_u16(uint32_t u, double *k)
{
    if (u & LDX_K16_INT0) { *k = 0.0; return -0.0f; }
    if (u == LDX_K16_BIG) { *k = __builtin_inf(); return __uint_as_float(LDX_LD32_BIG_BITS); }
    *k = (double)u;
    return (float)((double)u * 1e-4);
}
__device__ inline float dense_value(const ldx_k16 &c, int measure, double *k)
{
    return dense_value_u16(measure == LDX_MEASURE_RSQ ? c.r_square : c.d_prime, k);
}
__device__ inline float dense_value(const ldx_k16one &c, int, double *k) { return dense_value_u16(c.value, k); }   // (the host checked the measure)

template <typename Cell>
__global__ void triangle_dense_kernel(const Cell *__restrict__ strips, uint32_t n_snps, uint32_t n_slabs,
                                      int measure, int has_thres, double k_thres, uint32_t row_begin,
                                      uint32_t row_end, float *__restrict__ dense, size_t ld)
{
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t i = row_begin + blockIdx.y;
    if (j >= n_snps || i >= row_end) return;
    float v = -0.0f;   // the template's int 0 (ld_triangle.py:114)
    if (i > j) {
        const uint64_t G = (uint64_t)n_slabs * kGroupsPerSlab;
        const uint32_t t = j / kSlab, g = i / kGroup;
        const uint64_t u = tile_base(t, G) + (g - t * kGroupsPerSlab);
        const Cell c = strips[u * LDX_UNIT_PAIRS + cell_offset<Cell>(i % kGroup, j % kSlab)];
        double k;
        v = dense_value(c, measure, &k);
        // ld_triangle.py:223-225 compares the rounded value k/10^4 with the threshold; k_thres is the
        // smallest k whose k/10^4 is not below it.  Sub-threshold cells keep the template's int 0; an escape cell
        // stays an escape (its exact value, fetched by the caller, decides).
        if (has_thres && k < k_thres) v = -0.0f;
    }
    dense[(size_t)(i - row_begin) * ld + j] = v;
}

// ---- peak-rate probe for the inner loop's instruction pair ----
// 16 accumulators x 4 words per round = 64 v_and_b32 (SGPR x VGPR) + 64 accumulating v_bcnt_u32_b32 per
// lane and round, no memory traffic: what the VALU sustains for exactly the inner loop's two opcodes.
__global__ void __launch_bounds__(1024) probe_andpop_kernel(uint32_t *__restrict__ sink, uint32_t iters)
{
    uint32_t acc[16];
    uint32_t b[4];
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
#pragma unroll
    for (int k = 0; k < 16; ++k) acc[k] = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) b[k] = tid * 2654435761u + k * 40503u;
    uint32_t a = __builtin_amdgcn_readfirstlane(blockIdx.x * 747796405u + 2891336453u);
    for (uint32_t it = 0; it < iters; ++it) {
        uint32_t s[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) s[k] = a ^ (0x9E3779B9u * (k + 1));   // SALU
#pragma unroll
        for (int k = 0; k < 16; ++k) {
#pragma unroll
            for (int w = 0; w < 4; ++w) popacc(acc[k], s[k] & b[w]);
        }
        a = a * 1664525u + 1013904223u;
    }
    uint32_t s = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) s += acc[k];
    sink[tid] = s;
}

}  // namespace ldx

using namespace ldx;

// process-wide default kernel of ldx_triangle_dev (read atomically at every call; ldx_triangle_path_dev takes the
// path as an argument instead)
static std::atomic<int> g_triangle_path{LDX_PATH_AUTO};

static bool known_path(int path)
{
    return path == LDX_PATH_AUTO || path == LDX_PATH_POPCOUNT || path == LDX_PATH_MFMA || path == LDX_PATH_FP4;
}

extern "C" int ldx_set_triangle_path(int path)
{
    LDX_REQUIRE(known_path(path), "unknown path");
    g_triangle_path.store(path, std::memory_order_relaxed);
    return LDX_OK;
}

extern "C" int ldx_get_triangle_path(void) { return g_triangle_path.load(std::memory_order_relaxed); }

static size_t tile_lds_bytes(uint32_t nchunks) { return (size_t)nchunks * kSlab * 16u; }

static int ensure_lds(const void *kernel, size_t bytes)
{
    // > 64 KiB of dynamic LDS needs the opt-in attribute
    LDX_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return LDX_OK;
}

static int num_cus() { return device_cus(); }

template <bool kRaw, bool kN11, typename Cell>
static int launch_triangle(const void *alt, const double *fa, const double *fr, const double *q, uint32_t n_snps,
                           uint32_t n_hap, uint64_t unit_begin, uint64_t unit_end, Cell *out, ldx_ld64 *out_raw,
                           uint32_t *out_n11, hipStream_t s)
{
    const uint32_t nch = ldx::n_chunks(n_hap);
    const size_t lds = tile_lds_bytes(nch);
    int rc = ensure_lds((const void *)triangle_kernel<kRaw, kN11, Cell>, lds);
    if (rc) return rc;
    const uint64_t total = unit_end - unit_begin;
    uint64_t grid = (uint64_t)num_cus();   // persistent: one 16-wave workgroup per CU
    const uint64_t max_grid = (total + kWaves - 1) / kWaves;   // at least one unit per wave
    if (grid > max_grid) grid = max_grid;
    triangle_kernel<kRaw, kN11, Cell><<<(uint32_t)grid, kThreads, lds, s>>>(
        (const uint4 *)alt, fa, fr, q, n_snps, ldx::n_slabs(n_snps), nch, (double)n_hap, 1.0 / (double)n_hap, unit_begin,
        unit_end, out, out_raw, out_n11);
    LDX_HIP(hipGetLastError());
    return LDX_OK;
}

extern "C" size_t ldx_triangle_workspace_bytes(void) { return ldx::triangle_mfma_workspace_bytes(); }

extern "C" int ldx_triangle_workspace_init_dev(void *workspace, size_t workspace_bytes, void *stream)
{
    LDX_REQUIRE(workspace, "null pointer");
    LDX_REQUIRE(workspace_bytes >= ldx::triangle_mfma_workspace_bytes(), "workspace too small (see ldx_triangle_workspace_bytes)");
    LDX_HIP(hipMemsetAsync(workspace, 0, ldx::triangle_mfma_workspace_bytes(), (hipStream_t)stream));
    return LDX_OK;
}

extern "C" int ldx_triangle_dev(const void *alt, const double *fa, const double *fr, const double *q,
                                uint32_t n_snps, uint32_t n_hap, uint64_t unit_begin, uint64_t unit_end,
                                ldx_ld32 *out, ldx_ld64 *out_raw, uint32_t *out_n11, void *stream)
{
    return ldx_triangle_ex_dev(alt, fa, fr, q, n_snps, n_hap, unit_begin, unit_end,
                               g_triangle_path.load(std::memory_order_relaxed), LDX_OUT_LD32, out, out_raw, out_n11, nullptr, 0,
                               stream);
}

extern "C" int ldx_triangle_ex_dev(const void *alt, const double *fa, const double *fr, const double *q,
                                   uint32_t n_snps, uint32_t n_hap, uint64_t unit_begin, uint64_t unit_end, int path,
                                   int out_format, void *out, ldx_ld64 *out_raw, uint32_t *out_n11, void *workspace,
                                   size_t workspace_bytes, void *stream)
{
    LDX_REQUIRE(!workspace || workspace_bytes >= ldx::triangle_mfma_workspace_bytes(),
                "workspace too small (see ldx_triangle_workspace_bytes)");
    LDX_REQUIRE(!workspace || ((uintptr_t)workspace & 255u) == 0, "workspace must be 256-byte aligned");
    LDX_REQUIRE(alt && fa && fr && q && out, "null pointer");
    LDX_REQUIRE(known_path(path), "unknown path");
    const bool one_measure = out_format == LDX_OUT_K16_RSQ || out_format == LDX_OUT_K16_DPRIME;
    LDX_REQUIRE(out_format == LDX_OUT_LD32 || out_format == LDX_OUT_K16 || one_measure, "unknown output format");
    LDX_REQUIRE(out_format == LDX_OUT_LD32 || !out_raw, "out_raw needs LDX_OUT_LD32");
    LDX_REQUIRE(!one_measure || !out_n11, "the one-measure formats take no side output");
    LDX_REQUIRE(n_snps >= 1 && n_hap >= 1, "bad shape");
    if (n_hap > LDX_MAX_HAPS) {
        set_error("ldx_triangle_dev: n_hap %u > LDX_MAX_HAPS %u", n_hap, LDX_MAX_HAPS);
        return LDX_E_UNSUPPORTED;
    }
    if (!check_recip(n_hap)) {
        set_error("ldx_triangle_dev: reciprocal division check failed for n = %u", n_hap);
        return LDX_E_UNSUPPORTED;
    }
    const uint64_t U = ldx_triangle_units(n_snps);
    if (unit_end > U) unit_end = U;
    if (unit_begin >= unit_end) return LDX_OK;
    hipStream_t s = (hipStream_t)stream;
    if (path != LDX_PATH_POPCOUNT) {   // AUTO = the FP4 matrix kernel (twice the int8 kernel's counting rate)
        const int rc = triangle_mfma(alt, fa, fr, q, n_snps, n_hap, unit_begin, unit_end, out_format, out, out_raw, out_n11,
                                     path != LDX_PATH_MFMA, workspace, s);
        if (rc != ldx::kNoMatrixPath) return rc;
        if (path != LDX_PATH_AUTO) return LDX_E_UNSUPPORTED;   // an explicit matrix-pipe path: say so (message set)
        // AUTO and a bit plane of 4 GiB or more: the popcount kernel gives the very same cells
    }
    if (out_format == LDX_OUT_K16_RSQ)
        return launch_triangle<false, false>(alt, fa, fr, q, n_snps, n_hap, unit_begin, unit_end, (ldx_k16r *)out, out_raw, out_n11, s);
    if (out_format == LDX_OUT_K16_DPRIME)
        return launch_triangle<false, false>(alt, fa, fr, q, n_snps, n_hap, unit_begin, unit_end, (ldx_k16d *)out, out_raw, out_n11, s);
    if (out_format == LDX_OUT_K16) {
        ldx_k16 *o = (ldx_k16 *)out;
        if (out_n11)
            return launch_triangle<false, true>(alt, fa, fr, q, n_snps, n_hap, unit_begin, unit_end, o, out_raw, out_n11, s);
        return launch_triangle<false, false>(alt, fa, fr, q, n_snps, n_hap, unit_begin, unit_end, o, out_raw, out_n11, s);
    }
    ldx_ld32 *o = (ldx_ld32 *)out;
    if (out_raw && out_n11)
        return launch_triangle<true, true>(alt, fa, fr, q, n_snps, n_hap, unit_begin, unit_end, o, out_raw, out_n11, s);
    if (out_raw)
        return launch_triangle<true, false>(alt, fa, fr, q, n_snps, n_hap, unit_begin, unit_end, o, out_raw, out_n11, s);
    if (out_n11)
        return launch_triangle<false, true>(alt, fa, fr, q, n_snps, n_hap, unit_begin, unit_end, o, out_raw, out_n11, s);
    return launch_triangle<false, false>(alt, fa, fr, q, n_snps, n_hap, unit_begin, unit_end, o, out_raw, out_n11, s);
}

extern "C" int ldx_pair_counts_dev(const void *alt_i, uint32_t n_i, const void *alt_j, uint32_t n_j,
                                   uint32_t n_hap, uint32_t *n11, size_t ld, void *stream)
{
    LDX_REQUIRE(alt_i && alt_j && n11, "null pointer");
    LDX_REQUIRE(n_i >= 1 && n_j >= 1 && n_hap >= 1 && ld >= n_j, "bad shape");
    if (n_hap > LDX_MAX_HAPS) {
        set_error("ldx_pair_counts_dev: n_hap %u > LDX_MAX_HAPS %u", n_hap, LDX_MAX_HAPS);
        return LDX_E_UNSUPPORTED;
    }
    const uint32_t nch = ldx::n_chunks(n_hap);
    const size_t lds = tile_lds_bytes(nch);
    int rc = ensure_lds((const void *)pair_counts_kernel, lds);
    if (rc) return rc;
    const uint32_t tiles_j = ldx::n_slabs(n_j);
    const uint32_t groups_i = (n_i + kGroup - 1) / kGroup;
    uint32_t gy = (groups_i + kWaves - 1) / kWaves;
    const uint32_t want = (uint32_t)(num_cus() + tiles_j - 1) / tiles_j;   // ~1 workgroup per CU in total
    if (gy > want) gy = want;
    if (gy < 1) gy = 1;
    pair_counts_kernel<<<dim3(tiles_j, gy), kThreads, lds, (hipStream_t)stream>>>(
        (const uint4 *)alt_i, (const uint4 *)alt_j, n_i, n_j, nch, n11, ld);
    LDX_HIP(hipGetLastError());
    return LDX_OK;
}

extern "C" int ldx_ld_from_counts_ex_dev(uint32_t n, size_t m, const uint32_t *n11, const uint32_t *a1,
                                         const uint32_t *r1, const uint32_t *a2, const uint32_t *r2, ldx_ld64 *raw,
                                         double *k, ldx_ld32 *cells32, ldx_k16 *cells16, uint8_t *flags, void *stream)
{
    LDX_REQUIRE(n11 && a1 && r1 && a2 && r2, "null pointer");
    LDX_REQUIRE(n >= 1, "n must be positive (the reference raises ZeroDivisionError, calc_ld.py:33)");
    if (m == 0) return LDX_OK;
    if (!check_recip(n)) {
        set_error("ldx_ld_from_counts_dev: reciprocal division check failed for n = %u", n);
        return LDX_E_UNSUPPORTED;
    }
    ld_from_counts_kernel<<<(uint32_t)((m + 255) / 256), 256, 0, (hipStream_t)stream>>>(
        (double)n, 1.0 / (double)n, m, n11, a1, r1, a2, r2, raw, k, cells32, cells16, flags);
    LDX_HIP(hipGetLastError());
    return LDX_OK;
}

extern "C" int ldx_ld_from_counts_dev(uint32_t n, size_t m, const uint32_t *n11, const uint32_t *a1,
                                      const uint32_t *r1, const uint32_t *a2, const uint32_t *r2,
                                      ldx_ld64 *raw, ldx_ld32 *rounded, uint8_t *flags, void *stream)
{
    return ldx_ld_from_counts_ex_dev(n, m, n11, a1, r1, a2, r2, raw, nullptr, rounded, nullptr, flags, stream);
}

extern "C" int ldx_ld_pairs_dev(const void *alt, const uint32_t *acnt, const uint32_t *rcnt, uint32_t n_snps,
                                uint32_t n_hap, const uint32_t *rows, const uint32_t *cols, size_t m, double *k,
                                ldx_ld64 *raw, uint8_t *flags, uint32_t *n11, void *stream)
{
    LDX_REQUIRE(alt && acnt && rcnt && rows && cols, "null pointer");
    LDX_REQUIRE(n_snps >= 1 && n_hap >= 1, "bad shape");
    if (m == 0) return LDX_OK;
    ld_pairs_kernel<<<(uint32_t)((m + 3) / 4), 256, 0, (hipStream_t)stream>>>(
        (const uint4 *)alt, acnt, rcnt, n_snps, ldx::n_chunks(n_hap), (double)n_hap, rows, cols, m, k, raw, flags, n11);
    LDX_HIP(hipGetLastError());
    return LDX_OK;
}

extern "C" int ldx_triangle_dense_ex_dev(const void *strips, int strips_format, uint32_t n_snps, int measure,
                                         int has_thres, double thres, uint32_t row_begin, uint32_t row_end, float *dense,
                                         size_t ld, void *stream)
{
    LDX_REQUIRE(strips && dense, "null pointer");
    LDX_REQUIRE(strips_format == LDX_OUT_LD32 || strips_format == LDX_OUT_K16 || strips_format == LDX_OUT_K16_RSQ ||
                    strips_format == LDX_OUT_K16_DPRIME, "unknown cell format");
    LDX_REQUIRE(strips_format != LDX_OUT_K16_RSQ || measure == LDX_MEASURE_RSQ, "the strips hold r_square only");
    LDX_REQUIRE(strips_format != LDX_OUT_K16_DPRIME || measure == LDX_MEASURE_DPRIME, "the strips hold d_prime only");
    LDX_REQUIRE(row_begin <= row_end && row_end <= n_snps && ld >= n_snps, "bad shape");
    LDX_REQUIRE(measure == LDX_MEASURE_RSQ || measure == LDX_MEASURE_DPRIME, "bad measure");
    if (row_begin == row_end) return LDX_OK;
    const dim3 grid((n_snps + 255u) / 256u, row_end - row_begin);
    const double kt = has_thres ? thres_to_k(thres) : 0.0;
    if (strips_format == LDX_OUT_K16_RSQ || strips_format == LDX_OUT_K16_DPRIME)
        triangle_dense_kernel<<<grid, 256, 0, (hipStream_t)stream>>>((const ldx_k16one *)strips, n_snps, ldx::n_slabs(n_snps),
                                                                    measure, has_thres, kt, row_begin, row_end, dense, ld);
    else if (strips_format == LDX_OUT_K16)
        triangle_dense_kernel<<<grid, 256, 0, (hipStream_t)stream>>>((const ldx_k16 *)strips, n_snps, ldx::n_slabs(n_snps),
                                                                    measure, has_thres, kt, row_begin, row_end, dense, ld);
    else
        triangle_dense_kernel<<<grid, 256, 0, (hipStream_t)stream>>>((const ldx_ld32 *)strips, n_snps, ldx::n_slabs(n_snps),
                                                                    measure, has_thres, kt, row_begin, row_end, dense, ld);
    LDX_HIP(hipGetLastError());
    return LDX_OK;
}

extern "C" int ldx_triangle_dense_dev(const ldx_ld32 *strips, uint32_t n_snps, int measure, int has_thres,
                                      double thres, uint32_t row_begin, uint32_t row_end, float *dense, size_t ld,
                                      void *stream)
{
    return ldx_triangle_dense_ex_dev(strips, LDX_OUT_LD32, n_snps, measure, has_thres, thres, row_begin, row_end, dense,
                                     ld, stream);
}

extern "C" int ldx_probe_andpop_dev(uint32_t *sink, uint32_t blocks, uint32_t threads, uint32_t iters, void *stream)
{
    LDX_REQUIRE(sink && blocks >= 1 && threads >= 64 && threads <= 1024 && threads % 64 == 0, "bad argument");
    probe_andpop_kernel<<<blocks, threads, 0, (hipStream_t)stream>>>(sink, iters);
    LDX_HIP(hipGetLastError());
    return LDX_OK;
}
