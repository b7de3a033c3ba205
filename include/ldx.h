/*
 * ldx.h -- C ABI of the MI355X (gfx950) pairwise-LD engine.
 *
 * Drop-in boundary for the hot path of PlatonB/ld-tools: everything the reference does in
 * backend/calc_ld.py:3-99, batched over the pair loops that drive it
 * (ld_triangle.py:133-230, ld_area.py:152-276).  The reference is pure Python and has no FFI
 * of its own; the binding a maintainer adds is the ctypes stub shown in INTEGRATION.md.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes; no C++ types, no exceptions cross the boundary.
 *   - every function returns 0 on success or a negative LDX_E_* code; ldx_last_error() gives
 *     the message of the last failure on the calling thread.
 *   - `_dev` entry points take DEVICE pointers (e.g. torch tensors' data_ptr()) and a
 *     hipStream_t passed as void* (NULL = the null stream).  They enqueue work and return;
 *     they never allocate, free or synchronise.  The caller owns every buffer.
 *   - `_host` entry points take HOST pointers, run the same kernels on the current device and
 *     synchronise before returning.  There is no CPU fallback anywhere in this library.
 *
 * Packed panel layout in HBM ("tiled plane")
 *   A plane holds one bit per (SNP, haplotype).  Rows are grouped into slabs of LDX_SLAB_ROWS
 *   SNPs; the haplotype axis is cut into chunks of 128 haplotypes (16 bytes), allocated in pairs
 *   (n_chunks = 2 * ceil(n_hap / 256): the matrix kernel consumes two chunks per K-block).  Element
 *   (slab s, chunk c, row r) is the 16-byte group at byte offset
 *       ((s * n_chunks + c) * LDX_SLAB_ROWS + r) * 16,
 *   bit (h % 128) of it (little-endian, 32-bit words) being haplotype h = 128*c + (h % 128) of
 *   SNP 128*s + r.  One slab is therefore a contiguous n_chunks*2 KiB image that a workgroup
 *   copies linearly into LDS, and 8 consecutive SNPs of one chunk are 128 contiguous bytes that a
 *   wavefront fetches with scalar loads.  Pad bits and pad rows are zero.
 *   The `alt` plane has bit = 1 where the allele code is 1, the `ref` plane where it is 0
 *   (calc_ld.py:37-40 counts them separately; any other code is in neither plane).
 */
#ifndef LDX_H
#define LDX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LDX_VERSION 102            /* 0.1.2: ldx_triangle_ex_dev takes the pass scheduler's workspace (the library keeps no per-stream
                                      state); one-measure cell formats LDX_OUT_K16_RSQ / LDX_OUT_K16_DPRIME */
#define LDX_SLAB_ROWS 128u         /* SNP rows per slab == SNP columns per j-tile */
#define LDX_GROUP_ROWS 8u          /* SNP rows a wavefront pairs against one j-tile per unit */
#define LDX_CHUNK_HAPS 128u        /* haplotypes per 16-byte chunk */
#define LDX_UNIT_PAIRS (LDX_SLAB_ROWS * LDX_GROUP_ROWS)   /* 1024 result cells per unit */
/* Cell order INSIDE a unit (8 rows x 128 columns), r8 = row % 8, c = column % 128: rows one after the other (128 cells
 * each).  Inside a row the order follows what ONE lane of the matrix kernel holds -- a 32 x 32 MFMA tile puts column l
 * of each of the four column tiles into lane l, i.e. the four columns {c0, c0 + 32, c0 + 64, c0 + 96} -- so that the lane
 * writes them with 16-byte stores and a wave's store instruction covers whole contiguous runs:
 *   4-byte cells (ldx_k16):  the lane's four cells are adjacent: element 4 * (c % 32) + c / 32  (one store per row);
 *   8-byte cells (ldx_ld32): the lane's cells of column tiles (0, 1) and (2, 3) are adjacent pairs:
 *                            element 64 * (c / 64) + 2 * (c % 32) + (c / 32) % 2              (two stores per row).
 * Why: the kernel's result stream was bound by the number of store INSTRUCTIONS a CU can issue, not by bytes (round 4:
 * no stores at all -19 % kernel time at 50 000 x 1008, eight 4-byte stores -> two 16-byte stores per step -12 %;
 * tools/probes/wrbw.hip, profiles/r04/store_issue_*.log).  Side outputs of a launch (n11, unrounded values) use the order
 * of that launch's cell format.  Every producer and consumer of strip output goes through these macros
 * (ld_tools_amd/_lib.py: cell_offset).  Round 3's column-quad-major attempt (quads of ADJACENT columns, which needed the
 * MFMA operand roles swapped) was slower; this order needs no change to the arithmetic. */
#define LDX_CELL_OFFSET4(r8, c) ((r8) * LDX_SLAB_ROWS + (((c) & 31u) << 2) + ((c) >> 5))
#define LDX_CELL_OFFSET8(r8, c) ((r8) * LDX_SLAB_ROWS + (((c) >> 6) << 6) + (((c) & 31u) << 1) + (((c) >> 5) & 1u))
#define LDX_CELL_OFFSET(out_format, r8, c) ((out_format) == LDX_OUT_LD32 ? LDX_CELL_OFFSET8(r8, c) : LDX_CELL_OFFSET4(r8, c))
#define LDX_MAX_HAPS 10240u        /* one j-tile (128 rows, all chunks) must fit 160 KiB of LDS */

/* error codes */
#define LDX_OK 0
#define LDX_E_ARG (-1)             /* bad argument (null pointer, size out of range, ...) */
#define LDX_E_HIP (-2)             /* a HIP runtime call failed; see ldx_last_error() */
#define LDX_E_UNSUPPORTED (-3)     /* n_hap > LDX_MAX_HAPS, wrong device architecture, ... */
#define LDX_E_OVERFLOW (-4)        /* hit buffer too small (ldx_area_*): count is still returned */

/* per-pair flag bits: which results are the reference's *int* 0 rather than a float */
#define LDX_FLAG_DPRIME_INT0 1u    /* calc_ld.py:68-69,75-76 (ZeroDivisionError branch) */
#define LDX_FLAG_RSQ_INT0 2u       /* calc_ld.py:89-90 (unrounded d_prime == 0) */
#define LDX_FLAG_F32_SURE 0x80u     /* ldx_ld_from_counts_ex_dev only: the fp32 epilogue tier would keep this pair */

/* measures (ld_triangle -l / ld_area -l: ld_triangle_cli_en.py:52, ld_area_cli_en.py:50) */
#define LDX_MEASURE_RSQ 0
#define LDX_MEASURE_DPRIME 1

/* Result cells.  The reference returns round(r_square, 4) and round(d_prime, 4) (calc_ld.py:94-95), i.e. k / 10^4
 * for an integer k, or the *int* 0 in the degenerate branches.  Two cell formats carry (k, int-0 mark) per value:
 *   ldx_ld32 (8 bytes/pair): the float32 nearest to k / 10^4; int 0 is -0.0f (sign bit set), a float 0.0 is +0.0f.
 *            k = rint(value * 10^4) is exact while value < 1024 (float32 ulp below 10^-4).  Larger values -- they
 *            arise only with missing codes, where a + r < n lets the bound of D' vanish -- are stored as the quiet
 *            NaN LDX_LD32_BIG_BITS: fetch the exact value with ldx_ld_pairs_dev.
 *   ldx_k16  (4 bytes/pair): k itself in bits 0..14 of a uint16 (host: k / 10^4 in double IS Python's round(x, 4)),
 *            bit 15 set (LDX_K16_INT0) for the int 0; k >= 32767 (value >= 3.2767) is stored as LDX_K16_BIG: fetch
 *            the exact value with ldx_ld_pairs_dev.
 * str() of every value can be reproduced from either. */
typedef struct { float r_square; float d_prime; } ldx_ld32;
typedef struct { uint16_t r_square; uint16_t d_prime; } ldx_k16;
typedef struct { double r_square; double d_prime; } ldx_ld64;   /* unrounded, for parity checks */
#define LDX_LD32_BIG_BITS 0x7FC00B16u   /* ldx_ld32 value >= 1024: a quiet NaN with this payload */
#define LDX_K16_INT0 0x8000u
#define LDX_K16_BIG 0x7FFFu
#define LDX_OUT_LD32 0
#define LDX_OUT_K16 1
/* ONE measure per pair, 2 bytes (round 6): the r_square half or the d_prime half of ldx_k16 alone -- what a caller that writes
 * one measure needs (ld_triangle.py:223-230,344-360 fill and print ld_two_dim for the ONE measure -l names).  Same element
 * order as ldx_k16 (LDX_CELL_OFFSET4), same k / int-0 bit / escape.  The kernel skips the other value's arithmetic, margin
 * and bytes.  No side outputs (out_raw, out_n11) with these formats. */
typedef struct { uint16_t value; } ldx_k16one;
#define LDX_OUT_K16_RSQ 2
#define LDX_OUT_K16_DPRIME 3

/* one ld_area hit (ld_area.py:261-271): query/opposing SNP row indices and rounded values */
typedef struct {
    uint32_t query;      /* row index of var_1 (the query) in the panel */
    uint32_t oppos;      /* row index of var_2 (the opposing variant) */
    float r_square;      /* as ldx_ld32 */
    float d_prime;
} ldx_hit;

/* ---- library / device ---------------------------------------------------------------- */
int ldx_version(void);
const char *ldx_last_error(void);
int ldx_device_count(void);                 /* number of visible HIP devices, <0 on error */
int ldx_device_arch(int device, char *buf, size_t buflen);   /* e.g. "gfx950" */

/* ---- geometry helpers (pure arithmetic, usable without a GPU) ------------------------- */
uint32_t ldx_n_slabs(uint32_t n_snps);                     /* ceil(n_snps / 128) */
uint32_t ldx_n_chunks(uint32_t n_hap);                     /* 2 * ceil(n_hap / 256): chunks come in pairs */
size_t ldx_plane_bytes(uint32_t n_snps, uint32_t n_hap);   /* bytes of one tiled plane */
uint32_t ldx_padded_snps(uint32_t n_snps);                 /* n_slabs * 128 */
/* Triangle work units.  Unit u of the strict lower triangle pairs the 8 rows of group g with the
 * 128 columns of j-tile t, u = t*G - 8*t*(t-1) + (g - 16*t), G = padded_snps/8, g >= 16*t.  Result
 * cell (row i, column j), i > j, lives at element u*1024 + LDX_CELL_OFFSET(format, i % 8, j % 128) of the strip
 * output, t = j/128, g = i/8 (ldx_triangle_cell_index does the arithmetic). */
uint64_t ldx_triangle_units(uint32_t n_snps);
uint64_t ldx_triangle_unit_of(uint32_t n_snps, uint32_t row, uint32_t col);  /* requires row > col */
uint64_t ldx_triangle_tile_base(uint32_t n_snps, uint32_t tile);             /* first unit of j-tile */
uint64_t ldx_triangle_cell_index(uint32_t n_snps, uint32_t row, uint32_t col, int out_format); /* element of cell (row > col) in the full strip output of that cell format (LDX_OUT_*) */

/* ---- packing: the genotype lists of ld_triangle.py:160-186 / ld_area.py:182-187,230-235 ---- */
/* codes: int8 [n_snps][ld_codes], 1 = ALT, 0 = REF, anything else = neither (None, 2nd ALT...).
 * alt/ref: tiled planes of ldx_plane_bytes(); ref may be NULL.  acnt/rcnt: uint32 [padded_snps]
 * = per-SNP counts of code 1 / code 0 (calc_ld.py:37-40); rcnt may be NULL iff ref is. */
int ldx_pack_codes_dev(const int8_t *codes, uint32_t n_snps, uint32_t n_hap, size_t ld_codes,
                       void *alt, void *ref, uint32_t *acnt, uint32_t *rcnt, void *stream);
/* row-major bit planes (uint32 words, W32 = ld_words per row, little-endian bit order) -> tiled */
int ldx_tile_plane_dev(const uint32_t *rowmajor, uint32_t n_snps, uint32_t n_hap, size_t ld_words,
                       void *tiled, uint32_t *cnt, void *stream);
/* per-SNP frequency vectors used by the epilogue: fa = a/n, fr = r/n, q = fa*fr (calc_ld.py:41-44,
 * and the first product of :87-88).  double [padded_snps] each. */
int ldx_snp_stats_dev(const uint32_t *acnt, const uint32_t *rcnt, uint32_t n_snps, uint32_t n_hap,
                      double *fa, double *fr, double *q, void *stream);

/* round(a/n, 4) per SNP as a double: var_1/var_2_alt_freq of calc_ld.py:96-97 and the query's alt_freq of
 * ld_area.py:188-189.  freq4: double [n_snps]. */
int ldx_alt_freq4_dev(const uint32_t *acnt, uint32_t n_snps, uint32_t n_hap, double *freq4, void *stream);

/* ---- bit-exact contract: the alt/alt haplotype count of calc_ld.py:32 ------------------ */
/* n11[i][j] = popcount(alt_i[row i] & alt_j[row j]) for all rows of panel I against all rows of
 * panel J (may be the same plane).  n11 is dense row-major uint32 [n_i][ld]. */
int ldx_pair_counts_dev(const void *alt_i, uint32_t n_i, const void *alt_j, uint32_t n_j,
                        uint32_t n_hap, uint32_t *n11, size_t ld, void *stream);

/* ---- the epilogue alone: calc_ld.py:33-97 from the six integers ------------------------ */
/* Element k uses (n, n11[k], a1[k], r1[k], a2[k], r2[k]).  Any output may be NULL. */
int ldx_ld_from_counts_dev(uint32_t n, size_t m, const uint32_t *n11, const uint32_t *a1,
                           const uint32_t *r1, const uint32_t *a2, const uint32_t *r2,
                           ldx_ld64 *raw, ldx_ld32 *rounded, uint8_t *flags, void *stream);
/* The same with every output form: k = round(x, 4) * 10^4 as doubles [m][2] (r_square, d_prime; exact for any
 * magnitude), and the two cell formats.  The cells come from the production epilogues of the pair kernels (every tier:
 * fp32 / count-domain fp64 / op-for-op mirror), which must agree with each other bit for bit -- a disagreement poisons
 * the cell (NaN / 0xFFFF) so that the exhaustive tests fail loudly.  Any output may be NULL. */
int ldx_ld_from_counts_ex_dev(uint32_t n, size_t m, const uint32_t *n11, const uint32_t *a1,
                              const uint32_t *r1, const uint32_t *a2, const uint32_t *r2,
                              ldx_ld64 *raw, double *k, ldx_ld32 *cells32, ldx_k16 *cells16, uint8_t *flags,
                              void *stream);

/* ---- LD of an explicit list of pairs of one panel: calc_ld.py:30-97 per pair, exact for any magnitude ---- */
/* Pair p = (var_1 = rows[p], var_2 = cols[p]).  k: double [m][2] = round(x, 4) * 10^4 for (r_square, d_prime);
 * raw: unrounded; flags: LDX_FLAG_*; n11: the alt/alt haplotype counts.  Any output may be NULL.  This is how the
 * escape cells of the two formats are resolved, and what ld_lite-style single lookups use. */
int ldx_ld_pairs_dev(const void *alt, const uint32_t *acnt, const uint32_t *rcnt, uint32_t n_snps, uint32_t n_hap,
                     const uint32_t *rows, const uint32_t *cols, size_t m, double *k, ldx_ld64 *raw,
                     uint8_t *flags, uint32_t *n11, void *stream);

/* ---- ld_triangle: all row > col pairs (ld_triangle.py:133-230) ------------------------- */
/* Computes work units [unit_begin, unit_end) (clamped to ldx_triangle_units()).  var_1 = row,
 * var_2 = col as at ld_triangle.py:193-194.  Outputs are indexed from unit_begin:
 * out[(u - unit_begin)*1024 + ...].  Cells with row <= col or row >= n_snps are written as zero.
 * out_raw / out_n11 may be NULL.  fa/fr/q from ldx_snp_stats_dev. */
int ldx_triangle_dev(const void *alt, const double *fa, const double *fr, const double *q,
                     uint32_t n_snps, uint32_t n_hap, uint64_t unit_begin, uint64_t unit_end,
                     ldx_ld32 *out, ldx_ld64 *out_raw, uint32_t *out_n11, void *stream);
/* Which kernel ldx_triangle_dev launches.  All produce identical results (tests compare them cell for
 * cell): POPCOUNT = v_and_b32 + v_bcnt_u32_b32 on the bit-packed rows; MFMA = int8 G.G^T on the matrix
 * cores (v_mfma_i32_32x32x32_i8) with the bits expanded to bytes in registers; FP4 = the same contraction on
 * v_mfma_f32_32x32x64_f8f6f4 with the bits expanded to FP4 nibbles (twice the int8 rate; fp32 accumulation of
 * 0/1 products is exact below 2^24).  AUTO picks the one that measures fastest (FP4).
 * ldx_set_triangle_path sets the process-wide default that ldx_triangle_dev reads (atomically) at every call;
 * ldx_triangle_path_dev takes the path per call. */
#define LDX_PATH_AUTO 0
#define LDX_PATH_POPCOUNT 1
#define LDX_PATH_MFMA 2
#define LDX_PATH_FP4 3
int ldx_set_triangle_path(int path);
int ldx_get_triangle_path(void);
/* ldx_triangle_dev with the kernel path and the cell format per call.  out: ldx_ld32, ldx_k16 or ldx_k16one cells
 * (out_format = LDX_OUT_LD32 / LDX_OUT_K16 / LDX_OUT_K16_RSQ / LDX_OUT_K16_DPRIME), indexed as in ldx_triangle_dev.  out_raw
 * needs LDX_OUT_LD32; the one-measure formats take neither side output and run on the FP4 or the popcount kernel.
 * workspace (ABI 102): ldx_triangle_workspace_bytes() bytes of device memory, 256-byte aligned, that hold the matrix
 * kernel's pass scheduler (two ticket counters).  Contract:
 *   - ZERO it once before the first launch that uses it (hipMemsetAsync, torch.zeros, or ldx_triangle_workspace_init_dev);
 *     every launch leaves it zeroed again -- re-armed by its last workgroup -- so it is never touched by the host afterwards;
 *   - ONE workspace per launch that may be in flight: launches that may overlap (different streams, parallel branches of a
 *     graph, two graphs replayed at once) need different workspaces; consecutive launches of one stream, or consecutive
 *     nodes of one graph, may share one.  ld_tools_amd.ld_triangle keeps one per result buffer (TriangleResult.ws);
 *   - the library keeps NO scheduling state of its own (no per-stream slots, no limit on streams or captured launches, no
 *     allocation, nothing to leak); a launch recorded under stream capture is like any other;
 *   - workspace = NULL is allowed: the passes are then dealt round-robin instead of drawn from the counter (identical
 *     cells; measured 6.5-7.4 % slower on panels of more than one round of passes -- 10 000 x 5008, 40 000 x 5008,
 *     50 000 x 1008 -- and 3 % faster below one round, profiles/r06/round_robin_without_workspace.log) -- what
 *     ldx_triangle_dev does.
 * The popcount path ignores the workspace. */
size_t ldx_triangle_workspace_bytes(void);
int ldx_triangle_workspace_init_dev(void *workspace, size_t workspace_bytes, void *stream);   /* = hipMemsetAsync(workspace, 0, ...) */
int ldx_triangle_ex_dev(const void *alt, const double *fa, const double *fr, const double *q,
                        uint32_t n_snps, uint32_t n_hap, uint64_t unit_begin, uint64_t unit_end,
                        int path, int out_format, void *out, ldx_ld64 *out_raw, uint32_t *out_n11,
                        void *workspace, size_t workspace_bytes, void *stream);

/* Strip output -> dense row-major float32 [n_rows][ld] matrix of one measure with the
 * ld_two_dim semantics of ld_triangle.py:114,223-230: cell = rounded measure, or 0 when
 * row <= col or (has_thres and rounded measure < thres).  Rows [row_begin, row_end). */
int ldx_triangle_dense_dev(const ldx_ld32 *strips, uint32_t n_snps, int measure, int has_thres,
                           double thres, uint32_t row_begin, uint32_t row_end, float *dense,
                           size_t ld, void *stream);
/* The same for any cell format (strips_format = LDX_OUT_*; a one-measure format holds ONE measure: `measure` must be it).
 * Escape cells (values the format cannot hold) come out as the NaN LDX_LD32_BIG_BITS whatever the threshold: resolve them
 * with ldx_ld_pairs_dev. */
int ldx_triangle_dense_ex_dev(const void *strips, int strips_format, uint32_t n_snps, int measure, int has_thres,
                              double thres, uint32_t row_begin, uint32_t row_end, float *dense,
                              size_t ld, void *stream);

/* ---- ld_area: windowed scan around query SNPs (ld_area.py:152-276) --------------------- */
/* positions: int64 [n_snps] ascending 1-based coordinates (VCF order).  queries: uint32 row
 * indices [n_query].  For each query q the opposing set is o != q with
 * max(0, pos_q - flank) < pos_o <= pos_q + flank (pysam fetch semantics, ld_area.py:174-177,
 * 215-217).  var_1 = query, var_2 = opposing (ld_area.py:242-243).  A hit is kept when the
 * ROUNDED measure >= thres (ld_area.py:248).  `queries` must ascend STRICTLY (distinct rows; so their positions ascend):
 * n_query == n_snps therefore means "every SNP is a query", which the matrix-pipe band takes as such.
 * hits: capacity hit_cap, written in arbitrary order (sort by (query, oppos) for VCF order).
 * Wavefronts reserve hit slots in batches of 256: *n_hits (device uint64) receives the number of
 * slots RESERVED, unused slots carry query == UINT32_MAX and must be skipped.  If *n_hits exceeds
 * hit_cap only the first hit_cap slots were stored: retry with a larger buffer.
 * workspace: ldx_area_workspace_bytes() bytes, 256-byte aligned, scratch for the gathered query
 * panel, the unit plan and the band kernel's ticket counters: one workspace per scan in flight (two scans that may run
 * at the same time -- different streams, or two graphs that hold a scan each -- need two). */
int ldx_area_dev(const void *alt, const double *fa, const double *fr, const double *q,
                 uint32_t n_snps, uint32_t n_hap, const int64_t *positions,
                 const uint32_t *queries, uint32_t n_query, int64_t flank, int measure,
                 double thres, ldx_hit *hits, uint64_t hit_cap, uint64_t *n_hits, void *workspace,
                 size_t workspace_bytes, void *stream);
size_t ldx_area_workspace_bytes(uint32_t n_snps, uint32_t n_hap, uint32_t n_query);
/* The same scan, which also counts the stored hits per query row as it appends them: query_counts = device uint32
 * [n_snps + 1] (zeroed here) or NULL.  With query_counts = ldx_area_finish_counts(finish workspace) the finishing step
 * below (ldx_area_finish_ex_dev, counts_ready = 1) needs neither its memset nor its pass over the slot buffer. */
int ldx_area_scan_dev(const void *alt, const double *fa, const double *fr, const double *q,
                      uint32_t n_snps, uint32_t n_hap, const int64_t *positions,
                      const uint32_t *queries, uint32_t n_query, int64_t flank, int measure,
                      double thres, ldx_hit *hits, uint64_t hit_cap, uint64_t *n_hits, uint32_t *query_counts,
                      void *workspace, size_t workspace_bytes, void *stream);
/* Finish a scan on the device, without a host round trip: raw slots (arbitrary order, unused slots marked) -> hits
 * sorted by (query row, opposing row) = the reference's output order (ld_area.py:152,215-217), plus the per-row index
 * offsets[n_snps + 1] (hits of query row q are sorted[offsets[q] .. offsets[q + 1])).  n_reserved: the device counter
 * ldx_area_dev filled; sorted: capacity hit_cap; summary: device uint64 [2] = {number of hits, slots reserved}.
 * If summary[1] > hit_cap the scan overflowed its buffer: run both again with hit_cap >= summary[1].
 * `raw` is CONSUMED: once its slots have been scattered it serves as the scratch the ordering of long hit lists writes to.
 * offsets must be 16-byte aligned, workspace 256-byte aligned. */
int ldx_area_finish_dev(ldx_hit *raw, const uint64_t *n_reserved, uint64_t hit_cap, uint32_t n_snps,
                        ldx_hit *sorted, uint32_t *offsets, uint64_t *summary, void *workspace,
                        size_t workspace_bytes, void *stream);
size_t ldx_area_finish_workspace_bytes(uint32_t n_snps);
int ldx_area_finish_ex_dev(ldx_hit *raw, const uint64_t *n_reserved, uint64_t hit_cap, uint32_t n_snps,
                           ldx_hit *sorted, uint32_t *offsets, uint64_t *summary, void *workspace,
                           size_t workspace_bytes, int counts_ready, void *stream);
uint32_t *ldx_area_finish_counts(void *finish_workspace);   /* where the finishing step keeps its per-query counts */
/* The caller's copy of a finished scan in ONE launch: the first n_hits sorted hits split into query rows, opposing rows
 * (int64 each) and the value pairs (float [n_hits][2]: r_square, d_prime), and -- each optional, NULL to skip -- a copy of
 * the n_offsets words of the offsets index and of one 32-bit instrumentation word.  (ld_area.py:261-276 reads exactly these
 * per query; a driver that keeps the scan's buffers for the next table needs its own copy of the result.) */
int ldx_area_results_dev(const ldx_hit *sorted, uint64_t n_hits, int64_t *query, int64_t *oppos, float *values,
                         const uint32_t *offsets_src, uint32_t *offsets_dst, uint32_t n_offsets,
                         const uint32_t *word_src, uint32_t *word_dst, void *stream);
/* instrumentation: byte offset, inside the workspace of ldx_area_dev, of the uint32 count of passes (4 units of 64 rows
 * x 128 columns) the matrix-pipe band evaluated */
size_t ldx_area_band_passes_offset(uint32_t n_snps);
/* kernel behind ldx_area_dev: LDX_PATH_AUTO (FP4 matrix-pipe band when >= 1/16 of the SNPs are queries, popcount scan
 * otherwise), LDX_PATH_POPCOUNT, LDX_PATH_MFMA (int8 band), LDX_PATH_FP4; the hit sets are identical */
int ldx_set_area_path(int path);
int ldx_get_area_path(void);

/* ---- synthetic panels: deterministic, identical on host and device ------ */
/* codes int8 [n_snps][ld_codes] receive global SNPs [snp_offset, snp_offset + n_snps) (a rank's
 * shard).  thresholds: per-SNP ALT probability * 2^64 (computed on the host, see
 * ld_tools_amd/synth.py), covering whole LD blocks: thresholds[k] belongs to global SNP
 * (snp_offset / block_len) * block_len + k, up to the end of the block holding the last SNP.
 * rho_thr: within-block copy probability * 2^64.  miss_thr: probability * 2^64 of code 2. */
int ldx_synth_codes_dev(int8_t *codes, uint32_t n_snps, uint32_t n_hap, size_t ld_codes,
                        uint64_t seed, const uint64_t *thresholds, uint64_t rho_thr,
                        uint32_t block_len, uint64_t miss_thr, uint32_t snp_offset, void *stream);
/* The same with SNPs that are not "ordinary" (what a sub-panel of the ALL-panel variants holds: ld_area.py:215-225 takes
 * every rs variant of the window, monomorphic in the sub-panel or not): mono_thr = probability * 2^64 that a SNP is
 * monomorphic (every code 0; one in eight of them every code 1), miss_rows_thr = probability * 2^64 that a SNP carries the
 * miss_thr codes at all (2^64 - 1: every SNP, as ldx_synth_codes_dev). */
int ldx_synth_codes_ex_dev(int8_t *codes, uint32_t n_snps, uint32_t n_hap, size_t ld_codes,
                           uint64_t seed, const uint64_t *thresholds, uint64_t rho_thr,
                           uint32_t block_len, uint64_t miss_thr, uint32_t snp_offset,
                           uint64_t mono_thr, uint64_t miss_rows_thr, void *stream);

/* ---- host-pointer conveniences (same kernels; allocate, copy, synchronise) ------------- */
/* calc_ld for ONE pair of code vectors of lengths h1, h2 (zip semantics of calc_ld.py:30-31:
 * n = min(h1, h2) for the haplotype count, allele counts over the full vectors; any lengths >= 1).
 * counts[6] = {n, n11, a1, r1, a2, r2}; raw = unrounded; rounded = round(x, 4) as doubles, exact for any
 * magnitude; freq4[2] = round4(fa1), round4(fa2).  One H2D copy, ONE kernel (counts straight from the codes,
 * epilogue, rounding) and one 80-byte D2H copy per call, through device scratch cached per host thread. */
int ldx_calc_ld_host(const int8_t *g1, uint32_t h1, const int8_t *g2, uint32_t h2,
                     uint32_t counts[6], ldx_ld64 *raw, ldx_ld64 *rounded, double freq4[2],
                     uint8_t *flags);

/* ---- instrumentation ------------------------------------------------------------------- */
/* Peak-rate probe for the v_and_b32 + v_bcnt_u32_b32 pair (the inner loop's two instructions):
 * runs `iters` rounds of 64 AND + 64 BCNT per lane on `blocks` x `threads` threads (threads a multiple
 * of 64, <= 1024), no memory traffic, and writes a checksum to sink[blocks*threads]. */
int ldx_probe_andpop_dev(uint32_t *sink, uint32_t blocks, uint32_t threads, uint32_t iters, void *stream);

/* Peak-rate probe for the matrix pipe: `iters` rounds of 8 back-to-back int8 MFMAs per wave on independent
 * accumulators, operands in registers.  variant 0 = v_mfma_i32_32x32x32_i8 (32768 MACs each),
 * 1 = v_mfma_i32_16x16x64_i8 (16384 MACs each).  threads <= 256. */
int ldx_probe_mfma_dev(uint32_t *sink, uint32_t blocks, uint32_t threads, uint32_t iters, int variant,
                       void *stream);

/* Tests / tuning: force the number of passes a matrix-kernel launch hands out as two half-height tickets
 * (n_short >= 0), or restore the launch heuristic (n_short < 0).  Results do not depend on it. */
int ldx_debug_force_short_passes(int n_short);
/* Tuning builds (-DLDX_TUNING) count events of the fp32 epilogue tier: out[0] = units it handled, [1] = lane-steps parked
 * for the fp64 tier, [2] = units redone because the queue overflowed, [3] = mirror evaluations behind the fp64 tier.
 * Product builds leave the counters at zero.  Synchronises the device. */
int ldx_debug_counters(uint64_t out[8], int reset);

#ifdef __cplusplus
}
#endif
#endif /* LDX_H */
