// Windowed scan around query SNPs: the loop of ld_area.py:152-276 on the packed panel.
//
// Three launches on the caller's stream, all inside ldx_area_dev:
//   1. area_gather: copy the query rows into a compact tiled "query panel" (so the scalar side of the
//      pair kernel walks consecutive rows), with their positions and frequency vectors.
//   2. area_plan:   per j-tile of the panel, the range of query groups whose window can reach the tile
//      (queries ascend by position, so it is one binary search each way), then a prefix sum -> unit list.
//   3. area_scan:   persistent workgroups over equal shares of the unit list; per unit 8 queries x 128
//      opposing rows through the same LDS-tile / scalar-row inner loop as ld_triangle, the fp64 epilogue
//      with var_1 = query, var_2 = opposing (ld_area.py:242-243), the window / self filter
//      (ld_area.py:174-177,222), the threshold on the ROUNDED measure (ld_area.py:248) and hit append.
// Hits are appended through per-wave slot batches (one global atomic per 256 slots); unused slots are
// marked invalid (query == UINT32_MAX).
#include <atomic>

#include "ldx_common.h"
#include "ldx_tile.h"

namespace ldx {

constexpr uint32_t kBatch = 256;   // hit slots a wave reserves per atomic
constexpr uint32_t kInvalid = 0xFFFFFFFFu;

struct AreaWs {          // carved out of the caller's workspace
    uint4 *qalt;         // tiled plane of the gathered query rows
    int64_t *qpos;       // [q_pad]  position of query k (INT64_MAX beyond n_query)
    uint32_t *qrow;      // [q_pad]  panel row of query k
    double *qfa, *qfr, *qq;
    uint32_t *g_begin;   // [T] first query group of tile t
    uint64_t *unit_base; // [T + 1] prefix sum of group counts
};

static size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

static size_t carve(AreaWs &w, void *base, uint32_t n_snps, uint32_t n_hap, uint32_t n_query)
{
    const uint32_t qpad = ldx_padded_snps(n_query), T = ldx::n_slabs(n_snps);
    size_t off = 0;
    char *b = (char *)base;
    auto take = [&](size_t bytes) { void *p = b ? b + off : nullptr; off = align_up(off + bytes, 256); return p; };
    w.qalt = (uint4 *)take(ldx_plane_bytes(n_query, n_hap));
    w.qpos = (int64_t *)take(qpad * sizeof(int64_t));
    w.qrow = (uint32_t *)take(qpad * sizeof(uint32_t));
    w.qfa = (double *)take(qpad * sizeof(double));
    w.qfr = (double *)take(qpad * sizeof(double));
    w.qq = (double *)take(qpad * sizeof(double));
    w.g_begin = (uint32_t *)take(T * sizeof(uint32_t));
    w.unit_base = (uint64_t *)take((T + 1) * sizeof(uint64_t));
    return off;
}

// one wavefront per query row: copy its chunks from the panel plane into the query plane
__global__ void __launch_bounds__(256) area_gather_kernel(const uint4 *__restrict__ alt, const double *__restrict__ fa,
                                                          const double *__restrict__ fr, const double *__restrict__ q,
                                                          const int64_t *__restrict__ pos,
                                                          const uint32_t *__restrict__ queries, uint32_t n_query,
                                                          uint32_t q_pad, uint32_t nchunks, AreaWs w)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t k = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (k >= q_pad) return;
    const bool real = k < n_query;
    const uint32_t row = real ? queries[k] : 0u;
    const uint4 zero = {0u, 0u, 0u, 0u};
    const uint4 *src = alt + ((size_t)(row / kSlab) * nchunks) * kSlab + (row % kSlab);
    uint4 *dst = w.qalt + ((size_t)(k / kSlab) * nchunks) * kSlab + (k % kSlab);
    for (uint32_t c = lane; c < nchunks; c += 64u) dst[(size_t)c * kSlab] = real ? src[(size_t)c * kSlab] : zero;
    if (lane == 0) {
        w.qpos[k] = real ? pos[row] : INT64_MAX;
        w.qrow[k] = real ? row : kInvalid;
        w.qfa[k] = real ? fa[row] : 0.0;
        w.qfr[k] = real ? fr[row] : 0.0;
        w.qq[k] = real ? q[row] : 0.0;
    }
}

// single workgroup: per tile the query-group range, then an exclusive scan (T <= a few thousand)
__global__ void __launch_bounds__(1024) area_plan_kernel(const int64_t *__restrict__ pos, uint32_t n_snps, uint32_t T,
                                                         uint32_t n_query, int64_t flank, AreaWs w)
{
    __shared__ uint64_t carry;
    __shared__ uint32_t wsum[16];
    if (threadIdx.x == 0) { carry = 0; w.unit_base[0] = 0; }
    block_sync();
    for (uint32_t t0 = 0; t0 < T; t0 += 1024u) {
        const uint32_t t = t0 + threadIdx.x;
        uint32_t cnt = 0;
        if (t < T) {
            const uint32_t last = (t + 1u) * kSlab < n_snps ? (t + 1u) * kSlab - 1u : n_snps - 1u;
            const int64_t pmin = pos[t * kSlab], pmax = pos[last];
            // qa = first query with pos_q + flank >= pmin ; qb = first query with max(0, pos_q - flank) >= pmax
            uint32_t lo = 0, hi = n_query;
            while (lo < hi) { const uint32_t m = (lo + hi) / 2; if (w.qpos[m] + flank >= pmin) hi = m; else lo = m + 1; }
            const uint32_t qa = lo;
            lo = 0; hi = n_query;
            while (lo < hi) {
                const uint32_t m = (lo + hi) / 2;
                int64_t low = w.qpos[m] - flank;
                if (low < 0) low = 0;
                if (low >= pmax) hi = m; else lo = m + 1;
            }
            const uint32_t qb = lo;
            if (qb > qa) {
                const uint32_t ga = qa / kGroup, gb = (qb + kGroup - 1) / kGroup;
                w.g_begin[t] = ga;
                cnt = gb - ga;
            } else {
                w.g_begin[t] = 0;
            }
        }
        // workgroup inclusive scan of cnt
        uint32_t x = cnt;
        const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const uint32_t y = __shfl_up(x, off); if (lane >= (uint32_t)off) x += y; }
        if (lane == 63) wsum[wv] = x;
        block_sync();
        uint32_t pre = 0;
        for (uint32_t k = 0; k < wv; ++k) pre += wsum[k];
        const uint64_t incl = carry + pre + x;
        if (t < T) w.unit_base[t + 1] = incl;
        block_sync();
        if (threadIdx.x == 1023) carry = incl;
        block_sync();
    }
}

__global__ void __launch_bounds__(kThreads)
area_scan_kernel(const uint4 *__restrict__ alt, const uint4 *__restrict__ qalt, const double *__restrict__ fa,
                 const double *__restrict__ fr, const int64_t *__restrict__ pos, const int64_t *__restrict__ qpos,
                 const uint32_t *__restrict__ qrows, const double *__restrict__ qfa, const double *__restrict__ qfr,
                 const double *__restrict__ qq, const uint32_t *__restrict__ g_begin,
                 const uint64_t *__restrict__ unit_base, uint32_t n_snps, uint32_t T, uint32_t nchunks, double n,
                 int64_t flank, int measure, double k_thres, ldx_hit *__restrict__ hits, uint64_t hit_cap,
                 unsigned long long *__restrict__ n_hits, uint32_t *__restrict__ query_counts)
{
    extern __shared__ uint4 lds[];
    uint4 *jt = lds;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint64_t total = unit_base[T];
    const uint64_t b0 = total * blockIdx.x / gridDim.x;
    const uint64_t b1 = total * (blockIdx.x + 1) / gridDim.x;

    // per-wave hit slots: [slot, slot_end) is the unfilled part of the current batch
    uint64_t slot = 0, slot_end = 0;

    if (b0 < b1) {
        uint32_t lo = 0, hi = T;   // largest t with unit_base[t] <= b0
        while (hi - lo > 1) { const uint32_t m = (lo + hi) / 2; if (unit_base[m] <= b0) lo = m; else hi = m; }
        uint32_t t = lo;
        uint64_t u = b0;
        while (u < b1) {   // block-uniform
            const uint64_t tb = unit_base[t], te = unit_base[t + 1u];
            if (te <= u) { ++t; continue; }   // empty tile
            const uint64_t seg_end = b1 < te ? b1 : te;
            const uint32_t seg_len = (uint32_t)(seg_end - u);
            block_sync();
            stage_tile(jt, alt + (size_t)t * nchunks * kSlab, nchunks * kSlab);
            const uint32_t o0 = t * kSlab + lane, o1 = o0 + 64u;
            const bool ov[2] = {o0 < n_snps, o1 < n_snps};
            const double fa2[2] = {fa[o0], fa[o1]};
            const double fr2[2] = {fr[o0], fr[o1]};
            const int64_t po[2] = {ov[0] ? pos[o0] : 0, ov[1] ? pos[o1] : 0};
            block_sync();
            const uint32_t gfirst = g_begin[t];

            for (uint32_t k = wave; k < seg_len; k += kWaves) {
                const uint32_t g = gfirst + (uint32_t)(u + k - tb);   // query group
                const uint32_t q0 = g * kGroup;
                const uint4 *ai = qalt + ((size_t)(q0 / kSlab) * nchunks) * kSlab + (q0 % kSlab);
                Acc acc;
                count_unit(ai, jt, nchunks, lane, acc);
#pragma unroll
                for (int r = 0; r < (int)kGroup; ++r) {
                    const uint32_t qi = q0 + r;                       // wave-uniform
                    const uint32_t qrow = qrows[qi];
                    const int64_t pq = qpos[qi];
                    int64_t low = pq - flank;
                    if (low < 0) low = 0;                             // ld_area.py:174-176
                    const int64_t high = pq + flank;                // ld_area.py:177
                    const double fa1 = qfa[qi], fr1 = qfr[qi], q1 = qq[qi];
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj) {
                        const uint32_t o = jj ? o1 : o0;
                        bool keep = qrow != kInvalid && ov[jj] && o != qrow && low < po[jj] && po[jj] <= high;
                        ldx_ld32 res = {0.0f, 0.0f};
                        if (keep) {
                            const LdRaw lr = ld_epilogue((double)acc.v[r][jj] / n, fa1, fr1, q1, fa2[jj], fr2[jj]);
                            const LdK lk = round_pair(lr);
                            keep = (measure == LDX_MEASURE_RSQ ? lk.kr : lk.kd) >= k_thres;   // ld_area.py:248
                            res = encode_cell<ldx_ld32>(lk);
                        }
                        const unsigned long long m = __ballot(keep);
                        if (m) {   // wave-uniform
                            const uint32_t cnt = __builtin_popcountll(m);
                            if (slot + cnt > slot_end) {
                                // close the old batch (mark what is left invalid), open a new one
                                for (uint64_t s = slot + lane; s < slot_end; s += 64u)
                                    if (s < hit_cap) hits[s].query = kInvalid;
                                unsigned long long base = 0;
                                if (lane == 0) base = atomicAdd(n_hits, (unsigned long long)kBatch);
                                base = ((unsigned long long)__builtin_amdgcn_readfirstlane((uint32_t)(base >> 32)) << 32) |
                                       __builtin_amdgcn_readfirstlane((uint32_t)base);
                                slot = base;
                                slot_end = base + kBatch;
                            }
                            if (keep) {
                                const uint64_t s = slot + __builtin_popcountll(m & ((1ull << lane) - 1ull));
                                if (s < hit_cap) {
                                    hits[s] = ldx_hit{qrow, o, res.r_square, res.d_prime};
                                    if (query_counts) atomicAdd(&query_counts[qrow], 1u);   // only stored hits
                                }
                            }
                            slot += cnt;
                        }
                    }
                }
            }
            u = seg_end;
            ++t;
        }
    }
    for (uint64_t s = slot + lane; s < slot_end; s += 64u)
        if (s < hit_cap) hits[s].query = kInvalid;
}

// ---- finishing on the device: raw slots (arbitrary order, with unused slots) -> hits in (query, opposing) order ----
// count per query -> exclusive scan (the CSR row index of the result) -> scatter -> per-query ordering by opposing row.
// No host round trip: the launches cover the whole slot buffer and read the reserved count from device memory.
__global__ void area_count_kernel(const ldx_hit *__restrict__ raw, const unsigned long long *__restrict__ n_reserved,
                                  uint64_t cap, uint32_t n_snps, uint32_t *__restrict__ counts)
{
    const uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t lim = *n_reserved < cap ? *n_reserved : cap;
    if (s >= lim) return;
    const uint32_t qv = raw[s].query;
    if (qv < n_snps) atomicAdd(&counts[qv], 1u);   // kInvalid marks an unused slot
}

// single workgroup: offsets[k] = sum of counts[0..k), offsets[n_snps] = all hits; cursor = a copy for the scatter;
// summary = {hits, slots reserved} (slots reserved > capacity: the caller retries with a larger buffer)
__global__ void __launch_bounds__(1024) area_offsets_kernel(const uint32_t *__restrict__ counts, uint32_t n_snps,
                                                            uint32_t *__restrict__ offsets, uint32_t *__restrict__ cursor,
                                                            const unsigned long long *__restrict__ n_reserved,
                                                            unsigned long long *__restrict__ summary,
                                                            uint32_t *__restrict__ n_long)
{
    if (threadIdx.x == 0) store_agent(n_long, 0u);   // the ordering kernels' list of long queries starts empty (no memset node)
    // Exclusive scan of the per-SNP hit counts by ONE workgroup in three barrier-separated phases per chunk of
    // 8192 x 1024 counts: (A) a wave at a time sums 1024-count tiles (every lane its own 64-byte line, the lane
    // totals reduced by shuffles), (B) the tile totals -- a table in LDS -- are scanned by the block, (C) the waves
    // go over their tiles again and write the offsets.  Sixteen waves work on independent tiles between the barriers;
    // the first version scanned 1024 counts per step with one exposed load latency and three barriers per step
    // (110 us at 100 000 SNPs, a third of the band kernel's time).
    constexpr uint32_t kTile = 1024u, kTiles = 8192u, kBatch = 4u;
    __shared__ uint32_t tile_tot[kTiles];
    __shared__ uint32_t wsum[16];
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    auto wave_scan = [&](uint32_t x) {   // inclusive
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const uint32_t y = __shfl_up(x, off); if (lane >= (uint32_t)off) x += y; }
        return x;
    };
    auto load16 = [&](uint32_t k0, uint32_t (&c)[16]) {   // counts[k0 .. k0 + 16), zero past the end
        if (k0 < n_snps && n_snps - k0 >= 16u) {
            const uint4 *src = reinterpret_cast<const uint4 *>(counts + k0);   // k0 is a multiple of 16: 64-byte aligned
#pragma unroll
            for (int q = 0; q < 4; ++q) { const uint4 v = src[q]; c[4 * q] = v.x; c[4 * q + 1] = v.y; c[4 * q + 2] = v.z; c[4 * q + 3] = v.w; }
        } else {
#pragma unroll
            for (int j = 0; j < 16; ++j) c[j] = (k0 < n_snps && (uint32_t)j < n_snps - k0) ? counts[k0 + j] : 0u;
        }
    };
    uint32_t carry = 0;   // counts before this chunk (every thread keeps it)
    for (uint32_t base = 0; base < n_snps; base += kTiles * kTile) {
        const uint32_t left = n_snps - base;
        const uint32_t n_tiles = left >= kTiles * kTile ? kTiles : (left + kTile - 1u) / kTile;
        for (uint32_t t0 = wv; t0 < n_tiles; t0 += 16u * kBatch) {   // (A): the loads of kBatch tiles in flight together
            uint32_t c[kBatch][16];
#pragma unroll
            for (uint32_t b = 0; b < kBatch; ++b) load16(base + (t0 + 16u * b) * kTile + lane * 16u, c[b]);   // zeros past the end
#pragma unroll
            for (uint32_t b = 0; b < kBatch; ++b) {
                uint32_t sum = 0;
#pragma unroll
                for (int j = 0; j < 16; ++j) sum += c[b][j];
                const uint32_t incl = wave_scan(sum);
                if (lane == 63 && t0 + 16u * b < n_tiles) tile_tot[t0 + 16u * b] = incl;
            }
        }
        block_sync();
        {   // (B) tile_tot -> exclusive prefix, in place: eight consecutive entries per thread
            uint32_t v[8], sum = 0;
#pragma unroll
            for (int j = 0; j < 8; ++j) { const uint32_t t = threadIdx.x * 8u + j; v[j] = t < n_tiles ? tile_tot[t] : 0u; sum += v[j]; }
            const uint32_t incl = wave_scan(sum);
            if (lane == 63) wsum[wv] = incl;
            block_sync();
            uint32_t pre = 0, total = 0;
#pragma unroll
            for (uint32_t w = 0; w < 16; ++w) { const uint32_t x = wsum[w]; pre += w < wv ? x : 0u; total += x; }
            uint32_t at = carry + pre + incl - sum;
#pragma unroll
            for (int j = 0; j < 8; ++j) { const uint32_t t = threadIdx.x * 8u + j; if (t < n_tiles) tile_tot[t] = at; at += v[j]; }
            carry += total;
        }
        block_sync();
        for (uint32_t t0 = wv; t0 < n_tiles; t0 += 16u * kBatch) {   // (C)
            uint32_t c[kBatch][16];
#pragma unroll
            for (uint32_t b = 0; b < kBatch; ++b) load16(base + (t0 + 16u * b) * kTile + lane * 16u, c[b]);
#pragma unroll
            for (uint32_t b = 0; b < kBatch; ++b) {
                const uint32_t t = t0 + 16u * b;
                if (t >= n_tiles) break;   // wave-uniform
                const uint32_t k0 = base + t * kTile + lane * 16u;
                uint32_t sum = 0;
#pragma unroll
                for (int j = 0; j < 16; ++j) sum += c[b][j];
                uint32_t at = tile_tot[t] + wave_scan(sum) - sum;
                if (k0 < n_snps && n_snps - k0 >= 16u) {
                    uint4 *d0 = reinterpret_cast<uint4 *>(offsets + k0), *d1 = reinterpret_cast<uint4 *>(cursor + k0);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        uint4 o;
                        o.x = at; at += c[b][4 * q];
                        o.y = at; at += c[b][4 * q + 1];
                        o.z = at; at += c[b][4 * q + 2];
                        o.w = at; at += c[b][4 * q + 3];
                        d0[q] = o;
                        d1[q] = o;
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < 16; ++j)
                        if (k0 < n_snps && (uint32_t)j < n_snps - k0) { offsets[k0 + j] = at; cursor[k0 + j] = at; at += c[b][j]; }
                }
            }
        }
        block_sync();   // tile_tot is rewritten by the next chunk
    }
    if (threadIdx.x == 0) {
        offsets[n_snps] = carry;
        summary[0] = carry;
        summary[1] = *n_reserved;
    }
}

// The same exclusive scan for large n in two short multi-block kernels (the single-workgroup kernel above takes 38 us at
// 100 000 SNPs: one CU's worth of bandwidth): (1) every block of 1024 counts writes its total; (2) every block adds up the
// totals of the blocks before it (at most a few thousand words, L2-resident), scans its own 1024 counts and writes
// offsets and cursor; the last block writes offsets[n_snps] and the summary.
constexpr uint32_t kScanBlock = 1024u;   // counts per block: 256 threads x 4

__global__ void __launch_bounds__(256) area_block_sums_kernel(const uint32_t *__restrict__ counts, uint32_t n_snps,
                                                              uint32_t *__restrict__ block_tot, uint32_t *__restrict__ n_long)
{
    __shared__ uint32_t wsum[4];
    if (blockIdx.x == 0 && threadIdx.x == 0) store_agent(n_long, 0u);   // the ordering kernels' list of long queries starts empty
    const uint32_t k0 = blockIdx.x * kScanBlock + threadIdx.x * 4u;
    uint32_t sum = 0;
#pragma unroll
    for (uint32_t j = 0; j < 4u; ++j) sum += k0 + j < n_snps ? counts[k0 + j] : 0u;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_down(sum, off);
    if ((threadIdx.x & 63u) == 0) wsum[threadIdx.x >> 6] = sum;
    block_sync();
    if (threadIdx.x == 0) block_tot[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

__global__ void __launch_bounds__(256) area_block_scan_kernel(const uint32_t *__restrict__ counts, uint32_t n_snps,
                                                              const uint32_t *__restrict__ block_tot,
                                                              uint32_t *__restrict__ offsets, uint32_t *__restrict__ cursor,
                                                              const unsigned long long *__restrict__ n_reserved,
                                                              unsigned long long *__restrict__ summary)
{
    __shared__ uint32_t wsum[4], wpre[4];
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    uint32_t before = 0;                                   // hits of the blocks before this one
    for (uint32_t b = threadIdx.x; b < blockIdx.x; b += 256u) before += block_tot[b];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) before += __shfl_down(before, off);
    if (lane == 0) wpre[wv] = before;
    const uint32_t k0 = blockIdx.x * kScanBlock + threadIdx.x * 4u;
    uint32_t c[4], sum = 0;
#pragma unroll
    for (uint32_t j = 0; j < 4u; ++j) { c[j] = k0 + j < n_snps ? counts[k0 + j] : 0u; sum += c[j]; }
    uint32_t incl = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const uint32_t y = __shfl_up(incl, off); if (lane >= (uint32_t)off) incl += y; }
    if (lane == 63) wsum[wv] = incl;
    block_sync();
    uint32_t at = wpre[0] + wpre[1] + wpre[2] + wpre[3] + incl - sum;
    for (uint32_t w = 0; w < wv; ++w) at += wsum[w];
#pragma unroll
    for (uint32_t j = 0; j < 4u; ++j) {
        if (k0 + j < n_snps) { offsets[k0 + j] = at; cursor[k0 + j] = at; }
        at += c[j];
    }
    if (blockIdx.x == gridDim.x - 1u && threadIdx.x == 255u) {   // the last thread of the last block holds the grand total
        offsets[n_snps] = at;
        summary[0] = at;
        summary[1] = *n_reserved;
    }
}

__global__ void area_scatter_kernel(const ldx_hit *__restrict__ raw, const unsigned long long *__restrict__ n_reserved,
                                    uint64_t cap, uint32_t n_snps, uint32_t *__restrict__ cursor, ldx_hit *__restrict__ sorted)
{
    const uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t lim = *n_reserved < cap ? *n_reserved : cap;
    if (s >= lim) return;
    const ldx_hit h = raw[s];
    if (h.query < n_snps) sorted[atomicAdd(&cursor[h.query], 1u)] = h;
}

// A query's hits into ascending opposing row = VCF order (ld_area.py:215-217).  The scatter kernel places them in the
// order its atomics happened to run, i.e. in no order.  One thread per query sorts the usual handful in place
// (insertion sort, <= kOrderShort hits); a query with more (low thresholds: up to a whole window, thousands) goes on a
// list and is ordered by a whole workgroup (area_order_long_kernel) -- a single thread would need ~n^2 / 2 dependent
// global-memory moves there, milliseconds to seconds for one slow lane.
constexpr uint32_t kOrderShort = 32u, kOrderTile = 2048u;

__global__ void area_order_kernel(const uint32_t *__restrict__ offsets, uint32_t n_snps, ldx_hit *__restrict__ sorted,
                                  uint32_t *__restrict__ long_list, uint32_t *__restrict__ n_long)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_snps) return;
    const uint32_t b = offsets[k], e = offsets[k + 1u];
    if (e - b > kOrderShort) {
        long_list[atomicAdd(n_long, 1u)] = k;
        return;
    }
    for (uint32_t i = b + 1u; i < e; ++i) {
        const ldx_hit h = sorted[i];
        uint32_t j = i;
        while (j > b && sorted[j - 1u].oppos > h.oppos) {
            sorted[j] = sorted[j - 1u];
            --j;
        }
        sorted[j] = h;
    }
}

// Long segments, one workgroup at a time per listed query: rank sort.  A hit's place is the number of hits of its query
// with a smaller opposing row (the rows of one query are distinct; ties, which cannot happen, would be broken by
// position); every thread ranks one hit per round against all keys, staged through LDS in tiles, and writes it to that
// place in `scratch` (the raw slot buffer, consumed by the scatter kernel before); the segment is then copied back.
// n^2 / 256 key compares per thread instead of n^2 / 2 dependent moves in one.
__global__ void __launch_bounds__(256) area_order_long_kernel(const uint32_t *__restrict__ offsets,
                                                              const uint32_t *__restrict__ long_list,
                                                              const uint32_t *__restrict__ n_long, ldx_hit *__restrict__ sorted,
                                                              ldx_hit *__restrict__ scratch)
{
    __shared__ uint32_t keys[kOrderTile];
    const uint32_t count = *n_long;
    for (uint32_t w = blockIdx.x; w < count; w += gridDim.x) {   // block-uniform
        const uint32_t k = long_list[w], b = offsets[k], n = offsets[k + 1u] - b;
        for (uint32_t i0 = 0; i0 < n; i0 += blockDim.x) {        // block-uniform: every thread reaches every barrier
            const uint32_t i = i0 + threadIdx.x;
            ldx_hit h{};
            if (i < n) h = sorted[b + i];
            uint32_t rank = 0;
            for (uint32_t t0 = 0; t0 < n; t0 += kOrderTile) {
                const uint32_t tn = n - t0 < kOrderTile ? n - t0 : kOrderTile;
                block_sync();
                for (uint32_t j = threadIdx.x; j < tn; j += blockDim.x) keys[j] = sorted[b + t0 + j].oppos;
                block_sync();
                if (i < n)
                    for (uint32_t j = 0; j < tn; ++j) {
                        const uint32_t kj = keys[j];
                        rank += (kj < h.oppos) || (kj == h.oppos && t0 + j < i);
                    }
            }
            if (i < n) scratch[b + rank] = h;
        }
        __threadfence_block();
        block_sync();                                            // the ordered segment is complete in `scratch`
        for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) sorted[b + i] = scratch[b + i];
        block_sync();
    }
}

}  // namespace ldx

using namespace ldx;

extern "C" size_t ldx_area_finish_workspace_bytes(uint32_t n_snps)
{
    return 3u * ((((size_t)n_snps + 1u) * 4u + 255u) / 256u * 256u);   // counts, cursor, {number of long queries, their rows}
}

extern "C" int ldx_area_finish_dev(ldx_hit *raw, const uint64_t *n_reserved, uint64_t hit_cap, uint32_t n_snps,
                                   ldx_hit *sorted, uint32_t *offsets, uint64_t *summary, void *workspace,
                                   size_t workspace_bytes, void *stream)
{
    return ldx_area_finish_ex_dev(raw, n_reserved, hit_cap, n_snps, sorted, offsets, summary, workspace, workspace_bytes, 0,
                                  stream);
}

extern "C" uint32_t *ldx_area_finish_counts(void *finish_workspace) { return (uint32_t *)finish_workspace; }

namespace ldx {
// the caller's copy of a finished scan (ldx_area_results_dev): one grid-stride pass over max(hits, offsets) elements
__global__ void area_results_kernel(const ldx_hit *__restrict__ sorted, uint64_t n_hits, int64_t *__restrict__ query,
                                    int64_t *__restrict__ oppos, float2 *__restrict__ values,
                                    const uint32_t *__restrict__ offsets_src, uint32_t *__restrict__ offsets_dst,
                                    uint32_t n_offsets, const uint32_t *__restrict__ word_src, uint32_t *__restrict__ word_dst)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x, first = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (uint64_t k = first; k < n_hits; k += stride) {
        const ldx_hit h = sorted[k];
        query[k] = (int64_t)h.query;
        oppos[k] = (int64_t)h.oppos;
        values[k] = float2{h.r_square, h.d_prime};
    }
    if (offsets_dst)
        for (uint64_t k = first; k < n_offsets; k += stride) offsets_dst[k] = offsets_src[k];
    if (word_dst && first == 0) *word_dst = *word_src;
}
}  // namespace ldx

extern "C" int ldx_area_results_dev(const ldx_hit *sorted, uint64_t n_hits, int64_t *query, int64_t *oppos, float *values,
                                    const uint32_t *offsets_src, uint32_t *offsets_dst, uint32_t n_offsets,
                                    const uint32_t *word_src, uint32_t *word_dst, void *stream)
{
    LDX_REQUIRE(n_hits == 0 || (sorted && query && oppos && values), "null pointer");
    LDX_REQUIRE(!offsets_dst || offsets_src, "offsets_dst without offsets_src");
    LDX_REQUIRE(!word_dst || word_src, "word_dst without word_src");
    uint64_t work = n_hits;
    if (offsets_dst && n_offsets > work) work = n_offsets;
    if (word_dst && !work) work = 1u;
    if (!work) return LDX_OK;
    uint64_t blocks = (work + 255u) / 256u;
    if (blocks > 4096u) blocks = 4096u;
    ldx::area_results_kernel<<<(uint32_t)blocks, 256, 0, (hipStream_t)stream>>>(sorted, n_hits, query, oppos,
                                                                               reinterpret_cast<float2 *>(values), offsets_src,
                                                                               offsets_dst, n_offsets, word_src, word_dst);
    LDX_HIP(hipGetLastError());
    return LDX_OK;
}

extern "C" int ldx_area_finish_ex_dev(ldx_hit *raw, const uint64_t *n_reserved, uint64_t hit_cap, uint32_t n_snps,
                                      ldx_hit *sorted, uint32_t *offsets, uint64_t *summary, void *workspace,
                                      size_t workspace_bytes, int counts_ready, void *stream)
{
    LDX_REQUIRE(n_reserved && offsets && summary && workspace, "null pointer");
    LDX_REQUIRE((raw && sorted) || hit_cap == 0, "hit buffers are null but hit_cap > 0");
    LDX_REQUIRE(n_snps >= 1 && hit_cap < (1ull << 32), "bad shape");
    LDX_REQUIRE(workspace_bytes >= ldx_area_finish_workspace_bytes(n_snps), "workspace too small");
    LDX_REQUIRE(((uintptr_t)workspace & 255u) == 0, "workspace must be 256-byte aligned");
    LDX_REQUIRE(((uintptr_t)offsets & 15u) == 0, "offsets must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    const size_t vec = (((size_t)n_snps + 1u) * 4u + 255u) / 256u * 256u;
    uint32_t *counts = (uint32_t *)workspace, *cursor = (uint32_t *)((char *)workspace + vec);
    uint32_t *n_long = (uint32_t *)((char *)workspace + 2u * vec), *long_list = n_long + 1;   // [1 + n_snps]
    // counts_ready: the scan counted the hits per query row as it stored them (ldx_area_scan_dev with
    // query_counts = ldx_area_finish_counts(workspace)): no memset, no pass over the slot buffer
    if (!counts_ready) LDX_HIP(hipMemsetAsync(counts, 0, ((size_t)n_snps + 1u) * 4u, s));
    const uint32_t slot_blocks = (uint32_t)((hit_cap + 255u) / 256u);
    if (slot_blocks && !counts_ready) {
        area_count_kernel<<<slot_blocks, 256, 0, s>>>(raw, (const unsigned long long *)n_reserved, hit_cap, n_snps, counts);
        LDX_HIP(hipGetLastError());
    }
    const uint32_t scan_blocks = (n_snps + kScanBlock - 1u) / kScanBlock;
    if (scan_blocks >= 8u && scan_blocks <= n_snps) {   // block totals live behind the list of long queries (n_snps + 1 words: never full)
        uint32_t *block_tot = long_list + (n_snps - scan_blocks);
        area_block_sums_kernel<<<scan_blocks, 256, 0, s>>>(counts, n_snps, block_tot, n_long);
        LDX_HIP(hipGetLastError());
        area_block_scan_kernel<<<scan_blocks, 256, 0, s>>>(counts, n_snps, block_tot, offsets, cursor,
                                                          (const unsigned long long *)n_reserved, (unsigned long long *)summary);
    } else {
        area_offsets_kernel<<<1, 1024, 0, s>>>(counts, n_snps, offsets, cursor, (const unsigned long long *)n_reserved,
                                               (unsigned long long *)summary, n_long);
    }
    LDX_HIP(hipGetLastError());
    if (slot_blocks) {
        area_scatter_kernel<<<slot_blocks, 256, 0, s>>>(raw, (const unsigned long long *)n_reserved, hit_cap, n_snps, cursor,
                                                        sorted);
        LDX_HIP(hipGetLastError());
        area_order_kernel<<<(n_snps + 255u) / 256u, 256, 0, s>>>(offsets, n_snps, sorted, long_list, n_long);
        LDX_HIP(hipGetLastError());
        // queries with more than kOrderShort hits (none at the usual thresholds: the kernel then returns at once); the raw
        // slot buffer has been consumed by the scatter kernel and serves as the out-of-place target
        const uint32_t long_blocks = n_snps < 1024u ? n_snps : 1024u;
        area_order_long_kernel<<<long_blocks, 256, 0, s>>>(offsets, long_list, n_long, sorted, raw);
        LDX_HIP(hipGetLastError());
    }
    return LDX_OK;
}

extern "C" size_t ldx_area_band_passes_offset(uint32_t n_snps)
{
    // byte offset, inside the workspace of ldx_area_dev, of the uint32 that holds the number of passes the matrix-pipe
    // band evaluated (4 units of 64 rows x 128 columns each): instrumentation for tests and sharding studies
    return ((size_t)n_snps + 255u) / 256u * 256u + (size_t)ldx::n_slabs(n_snps) * 4u;
}

// which kernel runs ld_area: LDX_PATH_AUTO = the matrix-pipe band when at least 1/16 of the SNPs are queries (it
// evaluates every pair of the band once for both orders, whatever the query list: 0.52 ms at 100k SNPs, +-1000
// neighbours, against 8.2 ms x (queries / SNPs) for the popcount scan), the scan of query rows otherwise
static std::atomic<int> g_area_path{LDX_PATH_AUTO};   // process-wide default, read atomically at every call

extern "C" int ldx_set_area_path(int path)
{
    LDX_REQUIRE(path == LDX_PATH_AUTO || path == LDX_PATH_POPCOUNT || path == LDX_PATH_MFMA || path == LDX_PATH_FP4,
                "unknown path");
    g_area_path.store(path, std::memory_order_relaxed);
    return LDX_OK;
}

extern "C" int ldx_get_area_path(void) { return g_area_path.load(std::memory_order_relaxed); }

extern "C" size_t ldx_area_workspace_bytes(uint32_t n_snps, uint32_t n_hap, uint32_t n_query)
{
    AreaWs w;
    const size_t a = carve(w, nullptr, n_snps, n_hap, n_query ? n_query : 1), b = area_mfma_workspace_bytes(n_snps);
    return a > b ? a : b;
}

extern "C" int ldx_area_dev(const void *alt, const double *fa, const double *fr, const double *q, uint32_t n_snps,
                            uint32_t n_hap, const int64_t *positions, const uint32_t *queries, uint32_t n_query,
                            int64_t flank, int measure, double thres, ldx_hit *hits, uint64_t hit_cap,
                            uint64_t *n_hits, void *workspace, size_t workspace_bytes, void *stream)
{
    return ldx_area_scan_dev(alt, fa, fr, q, n_snps, n_hap, positions, queries, n_query, flank, measure, thres, hits, hit_cap,
                             n_hits, nullptr, workspace, workspace_bytes, stream);
}

extern "C" int ldx_area_scan_dev(const void *alt, const double *fa, const double *fr, const double *q, uint32_t n_snps,
                            uint32_t n_hap, const int64_t *positions, const uint32_t *queries, uint32_t n_query,
                            int64_t flank, int measure, double thres, ldx_hit *hits, uint64_t hit_cap,
                            uint64_t *n_hits, uint32_t *query_counts, void *workspace, size_t workspace_bytes, void *stream)
{
    LDX_REQUIRE(alt && fa && fr && q && positions && queries && n_hits && workspace, "null pointer");
    LDX_REQUIRE(hits || hit_cap == 0, "hits is null but hit_cap > 0");
    LDX_REQUIRE(n_snps >= 1 && n_hap >= 1 && n_query >= 1 && flank >= 0, "bad shape");
    LDX_REQUIRE(measure == LDX_MEASURE_RSQ || measure == LDX_MEASURE_DPRIME, "bad measure");
    LDX_REQUIRE(((uintptr_t)workspace & 255u) == 0, "workspace must be 256-byte aligned");
    if (n_hap > LDX_MAX_HAPS) {
        set_error("ldx_area_dev: n_hap %u > LDX_MAX_HAPS %u", n_hap, LDX_MAX_HAPS);
        return LDX_E_UNSUPPORTED;
    }
    AreaWs w;
    const size_t need = carve(w, workspace, n_snps, n_hap, n_query);
    LDX_REQUIRE(workspace_bytes >= need && workspace_bytes >= area_mfma_workspace_bytes(n_snps),
                "workspace too small (see ldx_area_workspace_bytes)");
    hipStream_t s = (hipStream_t)stream;
    if (query_counts) LDX_HIP(hipMemsetAsync(query_counts, 0, ((size_t)n_snps + 1u) * 4u, s));
    const int path = g_area_path.load(std::memory_order_relaxed);
    if (path == LDX_PATH_MFMA || path == LDX_PATH_FP4 ||
        (path == LDX_PATH_AUTO && (uint64_t)n_query * 16u >= n_snps && n_snps >= 2)) {
        const int rc = area_mfma(alt, fa, fr, q, n_snps, n_hap, positions, queries, n_query, flank, measure, thres, hits, hit_cap,
                                 n_hits, query_counts, workspace, path != LDX_PATH_MFMA, s);
        if (rc != ldx::kNoMatrixPath) return rc;
        if (path != LDX_PATH_AUTO) return LDX_E_UNSUPPORTED;
        // AUTO and a bit plane of 4 GiB or more: the popcount scan below finds the very same hits
    }
    const uint32_t qpad = ldx_padded_snps(n_query), T = ldx::n_slabs(n_snps), nch = ldx::n_chunks(n_hap);
    LDX_HIP(hipMemsetAsync(n_hits, 0, sizeof(uint64_t), s));
    area_gather_kernel<<<(qpad + 3u) / 4u, 256, 0, s>>>((const uint4 *)alt, fa, fr, q, positions, queries, n_query, qpad,
                                                       nch, w);
    LDX_HIP(hipGetLastError());
    area_plan_kernel<<<1, 1024, 0, s>>>(positions, n_snps, T, n_query, flank, w);
    LDX_HIP(hipGetLastError());
    const size_t lds = (size_t)nch * kSlab * 16u;
    LDX_HIP(hipFuncSetAttribute((const void *)area_scan_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int cus = device_cus();
    area_scan_kernel<<<cus, kThreads, lds, s>>>((const uint4 *)alt, w.qalt, fa, fr, positions, w.qpos, w.qrow, w.qfa,
                                                w.qfr, w.qq, w.g_begin, w.unit_base, n_snps, T, nch, (double)n_hap,
                                                flank, measure, thres_to_k(thres), hits, hit_cap,
                                                (unsigned long long *)n_hits, query_counts);
    LDX_HIP(hipGetLastError());
    return LDX_OK;
}
