// ld_triangle / ld_area on the matrix cores: n11 = G . G^T fused with the LD epilogue (calc_ld.py:30-97 for 8192 pairs per
// wave unit).  One kernel template, triangle_mfma_kernel, in two counting forms:
//   * kFp4 = true (the default path): v_mfma_f32_32x32x64_f8f6f4 with FP4 (E2M1) operands.  A haplotype bit becomes the
//     nibble of its own 4-bit group with one AND per eight haplotypes (expand32_a4 / expand32_b4): the A operand keeps the bit
//     at position p of the nibble, the B operand at 2 - p, every co-occurrence multiplies to exactly 1 and the fp32
//     accumulators hold n11 exactly (< 2^24).  64 haplotypes per 32 cycles and SIMD: twice the int8 rate.
//   * kFp4 = false (LDX_PATH_MFMA, the comparison path): v_mfma_i32_32x32x32_i8, bits expanded to bytes; accumulators hold
//     8 * n11.
// Structure (both forms)
//   * 256-thread workgroups (4 waves), two per CU (<= 256 VGPRs each): the partner of a wave on its SIMD belongs to the
//     OTHER workgroup -- one runs its VALU epilogue while the other feeds the matrix pipe.
//   * a wave's unit is 64 i-rows x 128 j-rows: 2 x 4 accumulator tiles of 32x32 (128 VGPRs); a PASS is four consecutive
//     units of one j-tile, one per wave; passes are handed out by a ticket counter (dynamic: workgroups do not run at the
//     same speed) that lives in the CALLER's workspace (ldx_triangle_workspace_bytes; the band: ldx_area_workspace_bytes)
//     -- the library keeps no per-stream or per-process scheduling state (round 6).
//   * B side (the 128 j-rows, shared by the 4 waves): per K-block (256 haplotypes, FP4; 128, int8) the workgroup expands
//     the j-tile's bits ONCE into a double-buffered LDS image; one LDS-only barrier per K-block.  Fragments are plain
//     ds_read_b128.  (Round 6 built and measured two barrier-free K loops for the ld_area band -- a triple-buffered image
//     with progress flags, and waves that expand the j-rows themselves, also as fully independent workers: slower / flat;
//     patches, A/B logs, stamps and counters under profiles/r06/, the account in HISTORY.md.)
//   * A side (a wave's own 64 rows): each lane loads the 16 bytes of "its" row per K-block (row = lane % 32 of each 32-row
//     tile, lane half = chunk of the K-block) three blocks ahead with hand-issued global loads (hand-counted s_waitcnt
//     vmcnt) and expands them in registers.
//   * Row / column placement follows the documented 32x32 C/D map (col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) +
//     4 * (lane >> 5)); tests compare every cell with the popcount kernel and the oracle.
//   * Epilogue tiers (ldx_common.h): fp32 (FP4 triangle, units inside the triangle) -> fp64 count-domain -> op-for-op
//     mirror; the band screens in fp32, prefilters in fp64 and evaluates queued candidates one pair per lane.
// DESIGN.md section 3 describes the kernels as they are; HISTORY.md has the measurements behind each choice.
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <type_traits>

#include "ldx_common.h"
#include "ldx_tile.h"

namespace ldx {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));
typedef unsigned v2u __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

// global loads the compiler does not track (see the K loop): the caller waits with s_waitcnt vmcnt(N).
// The K loop's form: a SCALAR base (the plane + the K-block's offset: SALU) and a 32-bit per-lane byte offset that does not
// change during a pass -- no 64-bit vector address arithmetic per block (four v_lshl_add_u64 / v_lshlrev_b64 of the block's
// ~69 VALU instructions) and one address register instead of three pairs.  Planes are < 4 GiB (launch_mfma checks).
// HAZARD: a VMEM instruction that reads an SGPR which a VALU instruction wrote (v_readlane_b32 reloading a spilled SGPR,
// v_readfirstlane_b32) needs five wait states in between on gfx9; hipcc inserts them for its own instructions but does not
// look inside inline asm.  The first form of these loads took the base straight from the compiler and one instantiation
// (int8 + n11 plane: its bases are reloaded from spill lanes right in front of the loads) read a stale register and
// faulted.  So the base goes through an s_mov_b64 INSIDE the asm: the VMEM instruction then reads SALU-written registers,
// which is interlocked.  tests/test_abi_and_host.py scans the shipped code object for the pattern.
__device__ __forceinline__ void gload16_s(v4u &dst, uint32_t voff, const void *sbase)
{
    const void *t;
    asm volatile("s_mov_b64 %1, %3\n\tglobal_load_dwordx4 %0, %2, %1" : "=v"(dst), "=&s"(t) : "v"(voff), "s"(sbase));
}
__device__ __forceinline__ void gload16x2_s(v4u &dst0, v4u &dst1, uint32_t voff, const void *sbase)   // rows r and r + 32
{
    const void *t;
    asm volatile("s_mov_b64 %2, %4\n\tglobal_load_dwordx4 %0, %3, %2\n\tglobal_load_dwordx4 %1, %3, %2 offset:512"
                 : "=&v"(dst0), "=&v"(dst1), "=&s"(t) : "v"(voff), "s"(sbase));   // early-clobber: the second load still reads voff
}
__device__ __forceinline__ void gload8_s(v2u &dst, uint32_t voff, const void *sbase)
{
    const void *t;
    asm volatile("s_mov_b64 %1, %3\n\tglobal_load_dwordx2 %0, %2, %1" : "=v"(dst), "=&s"(t) : "v"(voff), "s"(sbase));
}

#ifndef LDX_STEP_UNROLL
#define LDX_STEP_UNROLL 1   // fp32 tier: steps per trip of the step loop (epilogue_f32)
#endif
#ifndef LDX_VALU_PER_MFMA
#define LDX_VALU_PER_MFMA 5   // K loop: VALU slots scheduled behind each MFMA (a step has ~28 VALU for its 8 MFMAs; 5 measured better than 3 or 4)
#endif
constexpr int kMfmaWaves = 4;   // waves per workgroup; two workgroups per CU
constexpr int kMfmaThreads = kMfmaWaves * 64;
constexpr uint32_t kRows64 = 64;              // i-rows per wave unit
constexpr uint32_t kBRow = 144;               // bytes per expanded j-row in LDS (128 + 16 pad)
constexpr uint32_t kBBuf = kSlab * kBRow;     // one buffer: 18 KiB (int8 image; the FP4 image [4 steps][2 halves][128 rows][16 B] is 16 KiB)
constexpr uint32_t kStat = 6;                  // doubles per SNP in the LDS operand tables (16-byte aligned rows)
// tuning builds: LDX_STAMP.  -DLDX_STAMPS_ONLY (with -DLDX_TUNING) keeps the stamps but compiles `ablate` to 0 and the event
// counters out, so that the stamped code is the product's; the last 20 words of a wave's record then hold s_memtime at the
// top of each of the sixteen fp32-tier steps of its SECOND pass and at the end of the loop.
[[maybe_unused]] constexpr uint32_t kStampPasses = 40, kStampStride = 6 + 4 * kStampPasses + 20;

// Bits to int8 operands.  The matrix pipe only needs A[k] * B[k] to be the SAME constant for every haplotype k
// that both rows carry, not 1: the A side turns hap bit i of a nibble into the byte 1 << i (a byte replicate
// + one AND per four haplotypes), the B side into 8 >> i (multiply-spread), so every co-occurrence adds 8 and
// the accumulators hold 8 * n11 (< 2^31 for every panel whose counts fit 28 bits).

// B side: 16 haplotype bits (bits 0..15 of `bits`) -> 16 bytes {8,4,2,1}[k % 4] or 0.
// nibble bit i -> bit 7i + 3: byte i, bit 3 - i (multiplier bits at 6i + 3; no two products coincide)
__device__ __forceinline__ v4i expand16(uint32_t bits)
{
    v4i r;
    r.x = (int)(((bits & 0xFu) * 0x00208208u) & 0x01020408u);
    r.y = (int)((((bits >> 4) & 0xFu) * 0x00208208u) & 0x01020408u);
    r.z = (int)((((bits >> 8) & 0xFu) * 0x00208208u) & 0x01020408u);
    r.w = (int)((((bits >> 12) & 0xFu) * 0x00208208u) & 0x01020408u);
    return r;
}

// A side: 16 haplotype bits of `word` (its low or high half, chosen by the byte selectors) -> 16 bytes
// {1,2,4,8}[k % 4] or 0: v_perm_b32 replicates a byte four times, the mask keeps bit i in byte i.
__device__ __forceinline__ v4i expand16_a(uint32_t word, uint32_t sel0, uint32_t sel1)
{
    const uint32_t p0 = __builtin_amdgcn_perm(word, word, sel0), p1 = __builtin_amdgcn_perm(word, word, sel1);
    v4i r;
    r.x = (int)(p0 & 0x08040201u);
    r.y = (int)((p0 >> 4) & 0x08040201u);
    r.z = (int)(p1 & 0x08040201u);
    r.w = (int)((p1 >> 4) & 0x08040201u);
    return r;
}

// ---- FP4 (E2M1) operands for v_mfma_f32_32x32x64_f8f6f4: twice the int8 rate (64 haplotypes per 32 cycles) -----------
// A nibble with ONE bit at position p in {0, 1, 2} reads 0.5 / 1 / 2 (position 3 is the sign: -0), so a haplotype bit
// becomes the nibble of its own 4-bit group with NO data movement -- one AND per eight haplotypes -- as long as the
// other operand carries the bit at position 2 - p: every co-occurrence multiplies to 1 and the fp32 accumulators hold
// n11 exactly (< 2^24).  The K order inside an instruction is free (both operands use the same one): register v of an
// operand holds haplotypes {v, v + 4, ..., v + 28} of the 32-bit word for v < 2, and {2, 6, ...} / {3, 7, ...} of the
// word shifted down by two for v = 2, 3.  5 VALU per 32 haplotypes on the A side, 6 on the B side (the int8 forms
// above: 14 and 24).  tools/probes/fp4rate.hip checks the exactness and the rate on the device.
__device__ __forceinline__ v4i expand32_a4(uint32_t w)
{
    const uint32_t t = w >> 2;
    return v4i{(int)(w & 0x11111111u), (int)(w & 0x22222222u), (int)(t & 0x11111111u), (int)(t & 0x22222222u)};   // 0.5, 1, 0.5, 1
}
__device__ __forceinline__ v4i expand32_b4(uint32_t w)
{
    const uint32_t t = w >> 2;
    return v4i{(int)((w << 2) & 0x44444444u), (int)(w & 0x22222222u), (int)((w) & 0x44444444u), (int)(t & 0x22222222u)};   // 2, 1, 2, 1
}

// tuning-only build variants (python ld_tools_amd/build.py --out libldx_x.so -DLDX_AB_...; results are wrong)
#ifdef LDX_AB_NOAEXP
#define EXPAND_A(x) v4i{(int)(x), 1, 1, 1}
#else
#define EXPAND_A(x) expand16_a(x, sel0, sel1)
#endif
#ifdef LDX_AB_NOBEXP
#define EXPAND_B(x) v4i{(int)(x), 1, 1, 1}
#else
#define EXPAND_B(x) expand16(x)
#endif
#ifdef LDX_AB_NOBWRITE   // no expansion and no LDS write of the j-tile image (what a pre-expanded, DMA-fed image would save)
#define BWRITE(dst, x) asm volatile("" : : "v"(x))
#else
#define BWRITE(dst, x) dst = (kFp4 ? expand32_b4(x) : EXPAND_B(x))
#endif

// In-chunk stamps (build with -DLDX_CHUNK_STAMPS on top of -DLDX_TUNING): s_memtime at six points of ONE chunk
// (index LDX_CHUNK_STAMPS) of a wave's second pass, kept in SGPRs and written after the K loop.  Issued by
// inline asm so that hipcc does not drain lgkmcnt for them; an outstanding s_memtime only makes the compiler's
// own lgkmcnt waits more conservative.
#ifdef LDX_CHUNK_STAMPS
#define LDX_CSTAMP(k)                                                                  \
    if (c_ == (uint32_t)(LDX_CHUNK_STAMPS) && npass == 1) asm volatile("s_memtime %0" : "=s"(cst[k]));
#else
#define LDX_CSTAMP(k)
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

// passes of the full triangle before j-tile t: tile s has 2 * (n_slabs - s) units = ceil((n_slabs - s) / 2) passes
__host__ __device__ inline uint32_t mfma_pass_base(uint32_t t, uint32_t n_slabs)
{
    auto c = [](uint32_t k) { return ((k + 1u) / 2u) * ((k + 2u) / 2u); };   // sum of ceil(m / 2), m = 1..k
    return c(n_slabs) - c(n_slabs - t);
}

// Ticket counters of the dynamic pass scheduler: {next ticket, workgroups finished} at words 0 and 1 of the CALLER's
// workspace (ldx_triangle_workspace_bytes(); zeroed once by the caller, re-armed by the last workgroup out of every launch
// with agent-scope stores), so launches that may overlap simply use different workspaces and the library keeps no
// scheduling state of its own.  Round 6: rounds 1-5 kept a pool of 256 slots per device keyed by
// (device, hipStream_t) plus 65 536 private sets for captured launches, guarded by sequence numbers in pinned host memory,
// hipStreamQuery and a 50-us rule against recycled stream handles -- all gone.  Without a workspace (NULL) the passes are
// dealt round-robin (workgroup b takes tickets b, b + grid, ...): no counter at all, identical cells.
// Tuning builds only: behind the counters one K-loop token per physical CU (indexed by XCC / SE / SH / CU id).  The token
// lets one workgroup per CU into its K loop at a time; measured effect on the wall clock: none (HISTORY.md, round 2).
constexpr uint32_t kCuSlots = 2048;
constexpr uint32_t kSchedWords = 3u + kCuSlots;
constexpr size_t kTriWorkspaceBytes = (kSchedWords * 4u + 255u) / 256u * 256u;
__device__ unsigned long long g_dbg[8];   // tuning builds (-DLDX_TUNING): event counters, see ldx_debug_counters
#if defined(LDX_TUNING) && !defined(LDX_STAMPS_ONLY)
#define LDX_COUNT(slot, v) do { if (lane == 0) atomicAdd(&g_dbg[slot], (unsigned long long)(v)); } while (0)
#else
#define LDX_COUNT(slot, v)
#endif
static std::atomic<int> g_forced_short{-1};   // >= 0: number of halved passes per launch (ldx_debug_force_short_passes)


// Arguments of the banded (ld_area) use of the kernel: the same passes, K loop and operand staging; the pass list is
// cut to the units a window of +-flank can reach, and the epilogue turns every pair into up to two thresholded
// hits, (query = row, opposing = column) and (query = column, opposing = row)  (ld_area.py:152-276).
struct AreaArgs {
    const int64_t *pos;            // [n_snps] ascending 1-based positions
    const uint8_t *is_query;       // [n_snps] 1 = the SNP is a query; null = every SNP is
    const uint32_t *pass_base;     // [T + 1] prefix sum of passes per j-tile (pass_base[T] = all passes)
    const uint32_t *g_begin;       // [T] first 64-row group of tile t that can hold a hit (2t unless the queries end before the tile)
    const uint32_t *g_end;         // [T] one past the last such group
    const uint32_t *order;         // [passes] ticket -> pass (area_band_plan_kernel: per XCD range the tiles' FIRST passes first), or null
    ldx_hit *hits;
    uint32_t *counts;              // [n_snps] or null: hits per query row, counted as they are appended (ldx_area_scan_dev)
    // (kernel arguments live in scalar registers: two more of them once pushed the band kernel's register allocation into
    // spilling INSIDE its K loop, whose hand-counted s_waitcnt vmcnt a scratch access breaks -- tests/test_abi_and_host.py
    // scans for that; keep this struct at its size)
    unsigned long long *n_hits;    // band: the hit-slot counter
    uint64_t hit_cap;              // band
    double flank, k_thres;
    int measure;
    F32Const f32;                  // the fp32 epilogue tier's constants (triangle launches use only this member and the next)
};
// an entry of the band's ticket order: kAreaDecoded | tile << 12 | pass inside the tile (both < 4096 whenever the order exists:
// area_order_entries); plain pass indices (no order: panels beyond ~512 000 SNPs) are < 2^30
constexpr uint32_t kAreaDecoded = 1u << 30;
constexpr uint32_t kHitBatch = 256;   // hit slots a wave reserves per atomic (as in ldx_area.hip)
constexpr uint32_t kAreaQueue = 256;  // band: candidate pairs a wave collects before it evaluates them, one per lane
#ifdef LDX_MM1   // tuning build with three workgroups per CU: 53 KB of LDS each
constexpr uint32_t kQueueCap = 32;
#else
constexpr uint32_t kQueueCap = 128;   // fp32 tier: lane-steps a wave can park for the fp64 tier
#endif
// dynamic LDS of the kernel: the two j-tile image buffers, the fp64 operand tables, tickets, and for the FP4 triangle
// kernel the fp32 tables and the four queues
constexpr size_t mfma_lds_bytes(uint32_t stat_rows, bool f32_tier, bool band_f32 = false)
{
    return 2u * kBBuf + (kSlab + kMfmaWaves * stat_rows) * kStat * sizeof(double) + 32u +
           (f32_tier ? (kSlab + kMfmaWaves * kRows64) * 16u + kMfmaWaves * kQueueCap * 36u + kMfmaWaves * 64u * 4u : 0u) +
           (band_f32 ? (kSlab + kMfmaWaves * kRows64) * 16u + kMfmaWaves * kAreaQueue * 8u : 0u);   // the band's float32 screening
                                                                     // tables (same place as the tier's) and candidate queues
}

// tuning build -DLDX_MM1: every ticket half-height (32-row accumulator tiles only), three workgroups per CU
#ifdef LDX_MM1
constexpr int kWgPerCu = 3;
constexpr uint32_t kStatRows = 32;
#else
constexpr int kWgPerCu = 2;
constexpr uint32_t kStatRows = kRows64;
#endif
// Result cells are written once and never read back by the kernel: non-temporal stores (the `nt` bit) keep them from
// displacing the bit planes in the L2s.  Measured: -3.4 % at 50 000 x 1008 (the write-heavy shape), -0.9 % at 40 000 x 5008.
template <typename Cell>
__device__ __forceinline__ void store_cell(Cell *p, Cell v)
{
    if constexpr (sizeof(Cell) == 2)
        __builtin_nontemporal_store(__builtin_bit_cast(uint16_t, v), reinterpret_cast<uint16_t *>(p));
    else if constexpr (sizeof(Cell) == 4)
        __builtin_nontemporal_store(__builtin_bit_cast(uint32_t, v), reinterpret_cast<uint32_t *>(p));
    else
        __builtin_nontemporal_store(__builtin_bit_cast(unsigned long long, v), reinterpret_cast<unsigned long long *>(p));
}

// A lane's four 2-byte cells of one row (the one-measure formats), already packed two per word: one 8-byte store
template <typename Cell>
__device__ __forceinline__ void store_words2_saddr(Cell *sbase, uint32_t voff_bytes, uint32_t w0, uint32_t w1)
{
    Cell *t;   // (the base through an s_mov_b64 inside the asm: see gload16_s)
    const v2u v = {w0, w1};
    asm volatile("s_mov_b64 %0, %3\n\tglobal_store_dwordx2 %1, %2, %0 nt" : "=&s"(t) : "v"(voff_bytes), "v"(v), "s"(sbase) : "memory");
}

// The same store with a SCALAR base and a 32-bit per-lane byte offset (`global_store_dword voff, vdata, s[base:base+1]`):
// half the address registers the 64-bit-vaddr form reads (hipcc does not select this form for these stores by itself):
// -1 % on every shape.  Inline asm: the compiler's own vmcnt bookkeeping does not see these stores, which only makes its
// later waits stricter than they need be (a store returns nothing to wait for).
template <int kOffset, typename Cell>
__device__ __forceinline__ void store_cell_saddr(Cell *sbase, uint32_t voff_bytes, Cell v)
{
    Cell *t;   // (the base through an s_mov_b64 inside the asm: see gload16_s)
    if constexpr (sizeof(Cell) == 4)
        asm volatile("s_mov_b64 %0, %3\n\tglobal_store_dword %1, %2, %0 offset:%4 nt" : "=&s"(t) : "v"(voff_bytes), "v"(__builtin_bit_cast(uint32_t, v)), "s"(sbase), "n"(kOffset) : "memory");
    else
        asm volatile("s_mov_b64 %0, %3\n\tglobal_store_dwordx2 %1, %2, %0 offset:%4 nt" : "=&s"(t) : "v"(voff_bytes), "v"(__builtin_bit_cast(unsigned long long, v)), "s"(sbase), "n"(kOffset) : "memory");
}

// A lane's four cells of one row -- adjacent in memory (LDX_CELL_OFFSET4 / 8) -- as ONE 16-byte store (two for 8-byte cells),
// scalar row base + per-lane byte offset, non-temporal.  The CU's vector-memory unit takes a wave's store instruction at
// the same cost whatever its width, and eight 4-byte stores per step and wave held the fp32 tier's step at ~1450 cycles
// whatever its arithmetic cost (tools/probes/epi.hip: 990 alone; no stores at all: -19 % kernel time at 50 000 x 1008).
template <typename Cell>
__device__ __forceinline__ void store_cells4_saddr(Cell *sbase, uint32_t voff_bytes, Cell c0, Cell c1, Cell c2, Cell c3)
{
    Cell *t;   // (the base through an s_mov_b64 inside the asm: see gload16_s)
    if constexpr (sizeof(Cell) == 4) {
        const v4u v = {__builtin_bit_cast(uint32_t, c0), __builtin_bit_cast(uint32_t, c1), __builtin_bit_cast(uint32_t, c2),
                       __builtin_bit_cast(uint32_t, c3)};
        asm volatile("s_mov_b64 %0, %3\n\tglobal_store_dwordx4 %1, %2, %0 nt" : "=&s"(t) : "v"(voff_bytes), "v"(v), "s"(sbase) : "memory");
    } else {
        const v2u a = __builtin_bit_cast(v2u, c0), b = __builtin_bit_cast(v2u, c1), c = __builtin_bit_cast(v2u, c2),
                  d = __builtin_bit_cast(v2u, c3);
        const v4u lo = {a.x, a.y, b.x, b.y}, hi = {c.x, c.y, d.x, d.y};
        asm volatile("s_mov_b64 %0, %4\n\tglobal_store_dwordx4 %1, %2, %0 nt\n\tglobal_store_dwordx4 %1, %3, %0 offset:512 nt"   // (64 cells on)
                     : "=&s"(t) : "v"(voff_bytes), "v"(lo), "v"(hi), "s"(sbase) : "memory");
    }
}

// kFp4: the counting runs on v_mfma_f32_32x32x64_f8f6f4 with FP4 operands (expand32_a4 / expand32_b4) instead of
// v_mfma_i32_32x32x32_i8: a K-block is then 256 haplotypes -- two 128-haplotype chunks, one per lane half -- in four
// steps of 64, so the loop below keeps its shape (per step 8 MFMAs, 4 fragment reads, one quarter of the thread's share
// of a later block's j-tile image) with `nblocks` = nchunks / 2 iterations and ~16 instead of ~28 VALU per step.
template <bool kRaw, bool kN11, bool kArea = false, bool kFp4 = false, typename Cell = ldx_ld32>
__global__ void __launch_bounds__(kMfmaThreads, kArea ? 2 : kWgPerCu)
triangle_mfma_kernel(const uint4 *__restrict__ alt, const double *__restrict__ fa, const double *__restrict__ fr,
                     const double *__restrict__ q, uint32_t n_snps, uint32_t n_slabs, uint32_t nchunks, double n,
                     double rn, uint64_t u_begin, uint64_t u_end, Cell *__restrict__ out, ldx_ld64 *__restrict__ raw,
                     uint32_t *__restrict__ n11, uint32_t p_begin, uint32_t p_end_arg, uint32_t n_short, uint32_t *sched,
                     int ablate_arg, unsigned long long *stamps, AreaArgs aa)
{
    const uint32_t p_end = kArea ? aa.pass_base[n_slabs] : p_end_arg;   // area: the plan kernel's total
    // Tickets: first the passes [p_begin, p_end - n_short) whole, then each of the last n_short passes as TWO
    // half-height tickets (rows 0..31 / 32..63 of the pass's four units): finer work items for the end of the
    // launch, where whole passes leave the chip partly idle (see launch_mfma).
    const uint32_t n_norm = p_end - p_begin - n_short, n_tickets = n_norm + 2u * n_short;
    // `ablate` (tuning builds only, -DLDX_TUNING + env LDX_ABLATE; a compile-time 0 in the product, so that
    // none of its tests survives as a branch): 1 = no epilogue arithmetic, 2 = one chunk instead of all (no
    // counting), 4 = no stores, 16 = no epilogue priority, 64 = first half of the grid K loop only / second half
    // epilogue only, 128 = only the first half of the grid works, 512 = no clean-unit epilogue.  Results are wrong
    // by design for bits 1, 2, 4, 64, 128.
#if defined(LDX_TUNING) && !defined(LDX_STAMPS_ONLY)
    int ablate = ablate_arg;
    if (ablate_arg & 64) ablate = (blockIdx.x < (gridDim.x + 1) / 2) ? 5 : 2;
    if ((ablate_arg & 128) && blockIdx.x >= (gridDim.x + 1) / 2) return;
#else
    constexpr int ablate = 0;
    (void)ablate_arg;
    (void)stamps;
#endif
    extern __shared__ uint4 lds[];
    unsigned char *bexp = reinterpret_cast<unsigned char *>(lds);   // [2][128][144]
    // per-SNP operands of the fast epilogue (ldx_common.h, FastCol / FastRow): the j-tile's 128 columns, written
    // once per tile, and this wave's 64 rows, written once per pass
    double *cstat = reinterpret_cast<double *>(bexp + 2u * kBBuf);   // [128][kStat]: FastCol (+ position, is_query for ld_area)
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63u;
    const uint32_t l32 = lane & 31u;
    const uint32_t half = lane >> 5;
    const uint32_t sel0 = half ? 0x02020202u : 0x00000000u;   // v_perm selectors: this lane's two bytes of an A word
    const uint32_t sel1 = half ? 0x03030303u : 0x01010101u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    double *rstat = cstat + kSlab * kStat + wave * ((kArea ? kRows64 : kStatRows) * kStat);   // [64][kStat], private to the wave
    const FastConst fk = fast_const(n, kFp4 ? 1.0 : 8.0);   // the int8 accumulators hold 8 * n11, the FP4 ones n11
    // In-kernel stamps (tuning builds, env LDX_STAMPS=file): per wave {HW_ID | XCC_ID << 32, realtime, passes}
    // and per pass {start, prologue done, K loop done, epilogue done} in shader cycles; written to a buffer nothing else reads.
#ifdef LDX_TUNING
    uint32_t npass = 0;
    unsigned long long *const my_stamps = stamps ? stamps + (size_t)(blockIdx.x * kMfmaWaves + wave) * kStampStride : nullptr;
    if (my_stamps && lane == 0) {
        my_stamps[0] = (unsigned long long)__builtin_amdgcn_s_getreg(63492) |
                       ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32);
        my_stamps[1] = __builtin_amdgcn_s_memrealtime();
        my_stamps[2] = __builtin_amdgcn_s_memtime();
    }
#define LDX_STAMP(slot)                                                                                    \
    do {                                                                                                   \
        if (my_stamps && lane == 0 && npass < kStampPasses) my_stamps[6 + npass * 4 + (slot)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define LDX_STAMP(slot)
#endif

    // 64-row units: tile t owns groups g64 in [2t, 2T); unit v <-> small units [8v, 8v+8)
    const uint64_t G64 = (uint64_t)n_slabs * 2u;
    auto base64 = [&](uint64_t t) { return t * G64 - t * (t - 1u); };
    const uint64_t v_begin = u_begin / 8u, v_end = (u_end + 7u) / 8u;   // units that intersect the range

    // Work items are PASSES: four consecutive units of one j-tile, one per wave.  Pass p of the whole triangle
    // (mfma_pass_base below: tile t's passes start at pass_base(t)) is handed out dynamically -- a ticket
    // counter in global memory, one atomic per workgroup and pass, drawn one pass ahead -- because workgroups
    // do not run at the same speed: two that share a CU and fall into step (K loop beside K loop, epilogue
    // beside epilogue) take ~1.5x as long per pass as two in antiphase, and with an equal static share the
    // slowest pair set the kernel time (max wave lifetime 826k cycles against a median of 533k at 10k SNPs).
    uint32_t *tickets = reinterpret_cast<uint32_t *>(cstat + kSlab * kStat + kMfmaWaves * ((kArea ? kRows64 : kStatRows) * kStat));   // [2]
    uint32_t *cols_odd = tickets + 2;   // [2]: per wave of the column stagers, != 0 if one of its columns is not "ordinary"
    // fp32 tier (ldx_common.h, ld_multi_f32): its per-SNP tables and each wave's queue of lane-steps for the fp64 tier
    constexpr bool kF32Tier = kFp4 && !kRaw && !kN11 && !kArea;
    constexpr bool kBandF32 = kFp4 && kArea;   // the band screens its steps in float32 first (area_epilogue)
    float *ctab32 = reinterpret_cast<float *>(tickets + 8);                 // [128][4]: F32Col
    float *rtab32 = ctab32 + kSlab * 4u + wave * (kRows64 * 4u);            // [64][4]: F32Row, private to the wave
    uint32_t *qid = reinterpret_cast<uint32_t *>(ctab32 + kSlab * 4u + kMfmaWaves * kRows64 * 4u) + wave * kQueueCap;   // [kQueueCap]
    float *qcnt = reinterpret_cast<float *>(reinterpret_cast<uint32_t *>(ctab32 + kSlab * 4u + kMfmaWaves * kRows64 * 4u) +
                                            kMfmaWaves * kQueueCap) + wave * (kQueueCap * 8u);   // [kQueueCap][8] counts
    uint32_t *ulist = reinterpret_cast<uint32_t *>(ctab32 + kSlab * 4u + kMfmaWaves * kRows64 * 4u) + kMfmaWaves * kQueueCap * 9u +
                      wave * 64u;   // [64]: parked pairs deferred to the mirror batch (entry << 3 | pair)
    uint32_t *aq_id = reinterpret_cast<uint32_t *>(ctab32 + kSlab * 4u + kMfmaWaves * kRows64 * 4u) + wave * kAreaQueue;   // band: [kAreaQueue]
    uint32_t *aq_cnt = reinterpret_cast<uint32_t *>(ctab32 + kSlab * 4u + kMfmaWaves * kRows64 * 4u) + kMfmaWaves * kAreaQueue + wave * kAreaQueue;
    const F32Const fc32 = aa.f32;   // computed on the host (f32_const): kernel arguments live in scalar registers
    // The band (ld_area) hands its passes out PER XCD: the pass list -- j-tile-major, i.e. sorted by position -- is cut into
    // eight contiguous ranges, one per XCD, each with its own counter (sched[2 + 32 x]: the K-loop-token words, which the
    // band never uses; a cache line apart); a workgroup draws from the range of the XCD it runs on and, when that is
    // exhausted, from the next ones.  The +-flank windows of consecutive tiles overlap almost entirely, so an XCD that
    // walks a contiguous stretch of tiles finds their rows in its own 4 MiB L2 instead of re-streaming them from the
    // Infinity Cache (round 3: 450 MB of traffic for a 64 MB plane).  The triangle keeps the single counter: every one of its
    // tiles needs all rows below it, which no L2 holds.
    const uint32_t my_xcd = kArea ? (__builtin_amdgcn_s_getreg(63508) & 7u) : 0u;   // XCC_ID
    uint32_t static_next = blockIdx.x;   // (thread 0 only) the last ticket of the round-robin order: launches without a workspace
    auto draw = [&]() -> uint32_t {
        if constexpr (kArea) {
            for (uint32_t k = 0; k < 8u; ++k) {   // the ranges are cut at TILES (eighths of the tile list): the plan's order is per range
                const uint32_t x = (my_xcd + k) & 7u;
                const uint32_t lo = aa.pass_base[(uint32_t)((uint64_t)n_slabs * x / 8u)], hi = aa.pass_base[(uint32_t)((uint64_t)n_slabs * (x + 1u) / 8u)];
                if (lo == hi) continue;
                const uint32_t got = atomicAdd(&sched[2u + 32u * x], 1u);
                if (got < hi - lo) return aa.order ? aa.order[lo + got] : lo + got;   // (a decoded entry, kAreaDecoded, or a plain pass)
            }
            return 0xFFFFFFFFu;   // every range is exhausted
        } else {
            // the first gridDim.x tickets are the workgroups' own indices (no atomic round trip before a workgroup's first
            // pass: ~1.5 us per launch); the counter hands out the rest (the grid never exceeds the tickets: launch_mfma).
            // No workspace (sched == null): round-robin, workgroup b takes tickets b, b + grid, b + 2 grid ...
            if (!sched) return static_next += gridDim.x;
            return gridDim.x + atomicAdd(&sched[0], 1u);
        }
    };
    // this CU's K-loop token: OFF in product builds.  Tuning builds switch it on with LDX_ABLATE bit 4096 to make the
    // stamps readable (K loop alone on the matrix pipe: 29.3k cycles per unit).  It buys nothing on the wall clock: the
    // two waves of a SIMD are bound by their combined instruction issue.
    uint32_t *ktok = nullptr;
    if (!kArea && sched && (ablate_arg & 4096)) {
        const uint32_t hw = __builtin_amdgcn_s_getreg(63492), xcc = __builtin_amdgcn_s_getreg(63508);   // HW_ID, XCC_ID
        ktok = sched + 2u + ((((xcc & 7u) * 8u + ((hw >> 13) & 7u)) * 2u + ((hw >> 12) & 1u)) * 16u + ((hw >> 8) & 15u));
    }
    uint32_t parity = 0;
    if (tid == 0) tickets[0] = kArea ? draw() : blockIdx.x;

    uint64_t hit_slot = 0, hit_slot_end = 0;   // area: this wave's unfilled part of its current batch of hit slots
    uint32_t t_prev = 0xFFFFFFFFu;
    for (;;) {   // block-uniform: every wave reaches every barrier
        block_sync();   // the ticket is in LDS; every wave is past its previous epilogue (both B buffers free)
        const uint32_t ticket = tickets[parity];
        parity ^= 1u;
        if (kArea ? ticket == 0xFFFFFFFFu : ticket >= n_tickets) {   // block-uniform; the last workgroup out re-arms the counters
            if (tid == 0 && sched && atomicAdd(&sched[1], 1u) == gridDim.x - 1u) {
                // (agent-scope atomic stores, never plain ones: store_agent, ldx_common.h.  They go through to the memory side
                // and are complete when acknowledged, so the order between them needs a wait for the acknowledgement, not a
                // fence: __threadfence() -- and a release store -- write back the XCD's whole L2, result cells included, at
                // the end of every launch: +2-3 % at 10 000 SNPs, profiles/r05/kernel_end_ab.log)
                store_agent(&sched[1], 0u);
                if (kArea)
                    for (uint32_t x = 0; x < 8u; ++x) store_agent(&sched[2u + 32u * x], 0u);   // the per-XCD counters (over-drawn at the end)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                store_agent(&sched[0], 0u);   // the workspace is re-armed for the next launch that uses it
            }
            break;
        }
        const bool short_pass = !kArea && ticket >= n_norm;                    // block-uniform
        const uint32_t hsel = short_pass ? (ticket - n_norm) & 1u : 0u;        // which 32-row half
        // The triangle's tickets walk the j-tiles from the LAST to the first (inside a tile forwards): the last tiles are the
        // small ones -- one or two passes each, all on the diagonal, i.e. through the slow general epilogue -- and a launch
        // that ends with them ends with its longest items; reversed, it ends with the uniform passes of tile 0, which the
        // half-height tickets then split (round 4; the band keeps its position order for the L2).
        const uint32_t seq = short_pass ? n_norm + ((ticket - n_norm) >> 1) : ticket;   // position in the ticket order
        uint32_t p = p_begin + seq;
        auto pbase = [&](uint32_t tile) { return kArea ? aa.pass_base[tile] : mfma_pass_base(tile, n_slabs); };
        auto tile_of_pass = [&](uint32_t pp) {
            uint32_t lo = 0, hi = n_slabs;   // largest t with pass_base(t) <= pp
            while (hi - lo > 1) {
                const uint32_t mid = (lo + hi) / 2;
                if (pbase(mid) <= pp) lo = mid; else hi = mid;
            }
            return lo;
        };
        uint32_t t;
        if (kArea && (ticket & kAreaDecoded)) {   // the plan kernel's ticket order carries the tile and the pass inside it: two
            t = (ticket >> 12) & 0xFFFu;           // independent loads instead of a binary search through pass_base (~10
            p = pbase(t) + (ticket & 0xFFFu);      // dependent scalar loads, ~3k cycles at the top of every pass: round 6)
        } else
#ifndef LDX_AB_FORWARD_TILES
        if constexpr (!kArea) {
            const uint32_t mirrored = p_end - 1u - seq;   // the pass at the same distance from the end
            t = tile_of_pass(mirrored);
            const uint32_t lo_t = pbase(t) > p_begin ? pbase(t) : p_begin;
            const uint32_t hi_t = (t + 1u < n_slabs && pbase(t + 1u) < p_end) ? pbase(t + 1u) : p_end;
            p = lo_t + (hi_t - 1u - mirrored);
        } else
#endif
        {
            t = tile_of_pass(p);
        }
        const bool new_tile = t != t_prev;
        t_prev = t;
        const uint64_t tb = base64(t), te = base64(t + 1u);
        // triangle: the units of the tile inside [v_begin, v_end); area: the units [g_begin, g_end) the plan kept for the tile
        const uint64_t area_first = kArea ? tb + (aa.g_begin[t] - 2u * t) : 0u;
        const uint64_t pass = (kArea ? area_first : tb) + (uint64_t)(p - pbase(t)) * kMfmaWaves;   // its first unit
        const uint64_t seg_begin = kArea ? area_first : (v_begin > tb ? v_begin : tb);
        const uint64_t seg_end = kArea ? tb + (aa.g_end[t] > 2u * t ? aa.g_end[t] - 2u * t : 0u) : (v_end < te ? v_end : te);
        // the j-tile's bits for this thread's expansion share.  int8: row tid/2, 8 bytes (tid%2) of each chunk;
        // FP4: row tid%128, the whole 16 bytes of chunk 2b + tid/128 of K-block b (that lane half's chunk)
        const uint32_t b_off = kFp4 ? ((tid >> 7) * kSlab + (tid & 127u)) * 16u : (tid >> 1) * kBRow + (tid & 1u) * 64u;
        const uint32_t nblocks = kFp4 ? nchunks / 2u : nchunks;   // K-blocks per unit

        auto pass_body = [&](auto mm_c) {
            constexpr int MM = decltype(mm_c)::value;   // 32-row accumulator tiles per wave: 2 (a 64-row unit) or 1 (half)
            const uint64_t vv = pass + wave;
            const bool active = vv >= seg_begin && vv < seg_end;
            const uint32_t g64 = (uint32_t)((vv < te ? vv : pass) - tb) + 2u * t;   // a row group inside the panel either way
            const uint32_t roff = MM == 1 ? 32u * hsel : 0u;                        // half-height: rows roff .. roff+31 of the unit
            const uint32_t row0 = g64 * kRows64 + roff;
            // this lane's A rows: row0 + 32*m + l32 (both inside one slab: 64 | 128)
            // FP4: each lane half streams its own chunk of a K-block (half 0: chunk 2b, half 1: chunk 2b + 1)
            const uint4 *ai = alt + ((size_t)(row0 / kSlab) * nchunks) * kSlab + (row0 % kSlab) + l32 + (kFp4 ? half * kSlab : 0u);
            constexpr uint32_t kAStride = kFp4 ? 2u * kSlab : kSlab;   // uint4 per K-block

            typedef std::conditional_t<kFp4, v16f, v16i> acc_t;
            typedef std::conditional_t<kFp4, float, int> accel_t;   // one accumulator element: n11 (FP4) or 8 * n11 (int8)
            auto count_of = [](accel_t x) { if constexpr (kFp4) return (uint32_t)x; else return (uint32_t)x >> 3; };
            acc_t acc[MM][4];
#pragma unroll
            for (int m = 0; m < MM; ++m)
#pragma unroll
                for (int tt = 0; tt < 4; ++tt)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[m][tt][e] = 0;

            // ---- K loop, software-pipelined at K-step (32 haplotypes) granularity -------------------------
            // During the 8 MFMAs of step s the wave (i) has the B fragments of step s+1 in flight from LDS,
            // (ii) expands the A fragments of step s+1 and (iii) a quarter of its share of a later chunk's
            // B image (steps 0-2: quarters 1-3 of chunk c+1; step 3, after the barrier: quarter 0 of chunk c+2 into
            // the buffer chunk c just vacated) -- 28 VALU + 1 ds_write_b128 in EVERY step: a step hides ~28 VALU
            // behind its 8 MFMAs and pays ~7 cycles for each one beyond (tools/probes/steprate.hip).  One workgroup barrier per chunk, placed before the last K-step of the chunk: by then
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
                    bf[tt] = kFp4 ? *reinterpret_cast<const v4i *>(buf + (((uint32_t)w * 2u + half) * kSlab + 32u * tt + l32) * 16u)
                                  : *reinterpret_cast<const v4i *>(buf + (32u * tt + l32) * kBRow + w * 32u + half * 16u);
#endif
            };
            auto mma8 = [&](const v4i (&af)[MM], const v4i (&bf)[4]) {
#pragma unroll
                for (int m = 0; m < MM; ++m)
#pragma unroll
                    for (int tt = 0; tt < 4; ++tt)
#ifdef LDX_AB_NOMFMA
                        asm volatile("" : "+v"(acc[m][tt]) : "v"(af[m]), "v"(bf[tt]));   // operands stay live, no work
#else
                    {
                        if constexpr (kFp4) {
                            const v8i a8v = {af[m].x, af[m].y, af[m].z, af[m].w, 0, 0, 0, 0};
                            const v8i b8v = {bf[tt].x, bf[tt].y, bf[tt].z, bf[tt].w, 0, 0, 0, 0};
                            // cbsz = blgp = 4: FP4 operands (4 registers each); scale operands 0 = the unscaled instruction
                            acc[m][tt] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8v, b8v, acc[m][tt], 4, 4, 0, 0, 0, 0);
                        } else {
                            acc[m][tt] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[m], bf[tt], acc[m][tt], 0, 0, 0);
                        }
                    }
#endif
            };
            auto interleave = [&]() {   // 8 x {1 MFMA, up to 5 VALU}: the VALU work of a step hides behind its MFMAs
                // the next step's four B-fragment reads go FIRST: a whole step (256 cycles) of cover for the LDS
                // latency; left to itself hipcc sinks them to the end of the step and the next step's first
                // MFMAs wait on lgkmcnt
                __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
                for (int k = 0; k < 4 * MM; ++k) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, LDX_VALU_PER_MFMA, 0);
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

            LDX_STAMP(0);
            // Global loads of the K loop are issued by hand (global_load in inline asm) and waited for by hand
            // (s_waitcnt vmcnt(N) with N = the loads allowed to stay in flight): hipcc's own counter tracking gives
            // up at the loop's control-flow joins and emitted vmcnt(0) right after issuing a batch, i.e. one full
            // memory latency per chunk (K loop 1960 cycles per chunk instead of ~1150).  Rules that keep this
            // sound: every asm load is followed, before the first C++ use of its register, by a wait that covers
            // it and by ring_touch(), which ties the register to that point of the program order; the ring is
            // statically indexed (chunk loop unrolled by 3), so no register with a load in flight is ever copied.
            v4u ar[3][MM];   // A words: ring slot k holds the chunk c with c % 3 == k
            auto touch_ring = [&](int k) {   // ties the slot's registers to this point of the program order
                if constexpr (MM == 2) asm volatile("" : "+v"(ar[k][0]), "+v"(ar[k][1]));
                else asm volatile("" : "+v"(ar[k][0]));
            };
            typedef std::conditional_t<kFp4, v4u, v2u> bring_t;
            bring_t br[3];   // this thread's bits of the j-tile's K-block (B expansion share): 8 bytes (int8) / 16 bytes (FP4)
            auto clampc = [&](uint32_t c) { return c < nblocks ? c : nblocks - 1u; };   // surplus loads are discarded
            // per-lane byte offsets from the plane (constant during the pass) + the K-block's scalar offset
            constexpr uint32_t kBlockBytes = kAStride * 16u;   // one K-block of a slab: 4096 bytes (FP4: two chunks), 2048 (int8)
            const uint32_t b_voff = kFp4 ? (uint32_t)((((size_t)t * nchunks + (tid >> 7)) * kSlab + (tid & 127u)) * 16u)
                                         : (uint32_t)(((size_t)t * nchunks * kSlab) * 16u + tid * 8u);
            const uint32_t a_voff = (uint32_t)((size_t)(ai - alt) * 16u);
            auto block_base = [&](uint32_t blk) { return reinterpret_cast<const unsigned char *>(alt) + (size_t)blk * kBlockBytes; };
            auto load_b = [&](bring_t &dst, uint32_t blk) {
                if constexpr (kFp4) gload16_s(dst, b_voff, block_base(blk));
                else gload8_s(dst, b_voff, block_base(blk));
            };
            auto load_a = [&](v4u (&dst)[MM], uint32_t blk) {
                if constexpr (MM == 2) gload16x2_s(dst[0], dst[1], a_voff, block_base(blk));
                else gload16_s(dst[0], a_voff, block_base(blk));
            };
            // quarter q (K step q) of this thread's share of a K-block, expanded into the image at `buf`
            auto bquarter = [&](unsigned char *buf, const bring_t &bits, int q) {
                if constexpr (kFp4) {
                    BWRITE(*reinterpret_cast<v4i *>(buf + b_off + (uint32_t)q * (2u * kSlab * 16u)), bits[q]);
                } else {
                    const uint32_t w = q < 2 ? bits.x : bits.y;
                    BWRITE(*reinterpret_cast<v4i *>(buf + b_off + (uint32_t)q * 16u), (q & 1) ? w >> 16 : w);
                }
            };
            auto expand_a = [&](uint32_t x) {   // one K step of this lane's A row: its 16 (int8) / 32 (FP4) haplotype bits of word x
                if constexpr (kFp4) return expand32_a4(x);
                else return EXPAND_A(x);
            };
            {
                bring_t w0;
                load_b(w0, 0u);
                load_b(br[1], clampc(1));
#pragma unroll
                for (int k = 0; k < 2; ++k) load_a(ar[k], clampc(k));
                asm volatile("s_waitcnt vmcnt(0)");
                asm volatile("" : "+v"(w0), "+v"(br[1]));
                touch_ring(0);
                touch_ring(1);
#pragma unroll
                for (int q = 0; q < 4; ++q) bquarter(bexp, w0, q);   // K-block 0 -> buffer 0
                // quarter 0 of K-block 1 -> buffer 1 (in the loop it is written during step 3 of the block before)
                bquarter(bexp + kBBuf, br[1], 0);
            }
            bool rows_ordinary = false;
            if (!kRaw) {   // epilogue operands -> LDS (every wave is past its previous epilogue: barrier above)
                if (new_tile && tid < kSlab) {
                    const uint32_t j = t * kSlab + tid;
                    const FastCol c = fast_col(fa[j], fr[j], n);
                    const int ccls = snp_class(fa[j], fr[j], n);
                    const bool odd = ccls != kSnpOrdinary;   // (degenerate SNPs included: the fp64 tiers then run their general variant)
                    if (lane == 0) cols_odd[wave] = 0u;
                    if (__any(odd) && lane == 0) cols_odd[wave] = 1u;
                    typedef double d2 __attribute__((ext_vector_type(2)));
                    d2 *dst = reinterpret_cast<d2 *>(cstat + tid * kStat);
                    dst[0] = d2{c.a, c.ra};
                    dst[1] = d2{c.rr, c.rq};
                    if (kArea) dst[2] = j < n_snps ? d2{(double)aa.pos[j], (double)(aa.is_query ? aa.is_query[j] : (uint8_t)1)} : d2{0.0, 0.0};
                    if constexpr (kF32Tier) {
                        const F32Col c32 = f32_col(c.a, c.ra, c.rr, ccls, n);
                        *reinterpret_cast<v4f *>(ctab32 + tid * 4u) = v4f{c32.a, c32.ra, c32.rr, c32.s};
                    }
                    if constexpr (kBandF32) {   // the band's float32 screen (area_epilogue): a = ah + al, and the column's share of
                        // the threshold, sqrt(k) / s2 with s2 = 10 / sqrt(a r), a hair low; +inf for a count of 0 (r^2 is the int 0
                        // there: never a hit)
                        float ah, al;
                        f32_split_a((float)c.a, ah, al);
                        const double ar = 1.0 / c.rq;   // a r (0 for a count of 0)
                        const double kc = aa.k_thres - 2.5;
                        const double cthr = ar > 0.0 ? 0.1 * __builtin_sqrt((kc > 0.0 ? kc : 0.0) * ar) * (1.0 - 0x1p-20) : __builtin_inf();
                        *reinterpret_cast<v4f *>(ctab32 + tid * 4u) = v4f{ah, al, (float)cthr, 0.0f};
                    }
                }
                const uint32_t i = row0 + (MM == 1 ? l32 : lane);   // a half-height unit has 32 rows: stay inside the padded vectors
                const FastRow r = fast_row(fa[i], fr[i], n);
                const int rcls = snp_class(fa[i], fr[i], n);
                rows_ordinary = __all(rcls == kSnpOrdinary);
                typedef double d2 __attribute__((ext_vector_type(2)));
                d2 *dst = reinterpret_cast<d2 *>(rstat + (lane < (kArea ? kRows64 : kStatRows) ? lane : 0u) * kStat);
                if (lane < (kArea ? kRows64 : kStatRows)) {
                    dst[0] = d2{r.a_s, r.ra};
                    dst[1] = d2{r.rr, r.rq_s};
                }
                if constexpr (kF32Tier) {   // (a half-height unit's lanes 32-63 repeat rows 0-31 into slots nobody reads)
                    const F32Row r32 = f32_row(r.a_s * 1e-4, r.ra, r.rr, rcls, n);   // 1e4 a / 1e4: exact (a < 2^32)
                    *reinterpret_cast<v4f *>(rtab32 + lane * 4u) = v4f{r32.a, r32.ra_s, r32.rr_s, r32.s};
                }
                if constexpr (kBandF32) {   // {a, s1 = 10 / sqrt(a r)}; 0 for a count of 0 (such a row is never a candidate)
                    const double s1 = 10.0 * __builtin_sqrt(r.ra * r.rr);
                    *reinterpret_cast<v4f *>(rtab32 + lane * 4u) = v4f{(float)(r.a_s * 1e-4), s1 < __builtin_inf() ? (float)s1 : 0.0f, 0.0f, 0.0f};
                }
                if (kArea) dst[2] = i < n_snps ? d2{(double)aa.pos[i], (double)(aa.is_query ? aa.is_query[i] : (uint8_t)1)} : d2{0.0, 0.0};
            }
            if (ktok && tid == 0)   // one workgroup per CU in its K loop at a time (see g_sched)
                while (atomicCAS(ktok, 0u, 1u) != 0u) __builtin_amdgcn_s_sleep(4);
            block_sync();
            LDX_STAMP(1);
            if (ablate & 32) __builtin_amdgcn_s_setprio(2);   // tuning: the K-loop wave outranks the epilogue wave instead
#if defined(LDX_AB_BANDPRIO)   // tuning: in the band (both waves of a SIMD in their K loops most of the time) one workgroup of the CU outranks the other
            if (kArea) { if (LDX_AB_BANDPRIO == 1 ? (blockIdx.x >= gridDim.x / 2u) : (blockIdx.x & 1u)) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0); }
#endif
            v4i af0[MM], bf0[4], af1[MM], bf1[4];
            read_bf(bf0, bexp, 0);
#pragma unroll
            for (int m = 0; m < MM; ++m) af0[m] = expand_a(ar[0][m].x);

            const uint32_t nch_run = (ablate & 2) ? 1u : nblocks;
#ifdef LDX_CHUNK_STAMPS
            unsigned long long cst[6] = {0, 0, 0, 0, 0, 0};
#endif
            // one chunk: ring slot CUR holds its A words, slot NXT the next chunk's (A words and B bits),
            // slot FAR receives chunk c+2
#define LDX_CHUNK(CUR, NXT, FAR, cc)                                                                               \
            {                                                                                                      \
                const uint32_t c_ = (cc);                                                                          \
                const unsigned char *rd = bexp + (c_ & 1u) * kBBuf;                                                \
                unsigned char *wr = bexp + ((c_ + 1u) & 1u) * kBBuf;                                               \
                const uint32_t c3 = clampc(c_ + 2u);                                                               \
                LDX_CSTAMP(0)                                                                                      \
                /* loads of K-block c+2, B bits FIRST (vmcnt counts in issue order).  In flight now, oldest first: */ \
                /* block c+1's {B, A, A} and this batch's {B, A, A}; step 0 needs the former B: 5 may stay */       \
                load_b(br[FAR], c3);                                                                               \
                load_a(ar[FAR], c3);                                                                               \
                asm volatile("s_waitcnt vmcnt(%0)" : : "n"(MM == 2 ? 5 : 3));   /* all but the 1 + MM newest pairs */ \
                asm volatile("" : "+v"(br[NXT]));                                                                  \
                /* step 0: MFMAs of (c,0); prepare (c,1); quarter 1 of this thread's share of B block c+1 */      \
                read_bf(bf1, rd, 1);                                                                               \
                _Pragma("unroll") for (int m = 0; m < MM; ++m) af1[m] = expand_a(ar[CUR][m].y);               \
                bquarter(wr, br[NXT], 1);                                                                          \
                mma8(af0, bf0);                                                                                    \
                interleave();                                                                                      \
                __builtin_amdgcn_sched_barrier(0);                                                                 \
                LDX_CSTAMP(1)                                                                                      \
                /* step 1: MFMAs of (c,1); prepare (c,2); quarter 2 of the B share */                             \
                read_bf(bf0, rd, 2);                                                                               \
                _Pragma("unroll") for (int m = 0; m < MM; ++m) af0[m] = expand_a(ar[CUR][m].z);               \
                bquarter(wr, br[NXT], 2);                                                                          \
                mma8(af1, bf1);                                                                                    \
                interleave();                                                                                      \
                __builtin_amdgcn_sched_barrier(0);                                                                 \
                LDX_CSTAMP(2)                                                                                      \
                /* step 2: MFMAs of (c,2); prepare (c,3); quarter 3 of the B share */                             \
                read_bf(bf1, rd, 3);                                                                               \
                _Pragma("unroll") for (int m = 0; m < MM; ++m) af1[m] = expand_a(ar[CUR][m].w);               \
                bquarter(wr, br[NXT], 3);                                                                          \
                mma8(af0, bf0);                                                                                    \
                interleave();                                                                                      \
                __builtin_amdgcn_sched_barrier(0);                                                                 \
                LDX_CSTAMP(3)                                                                                      \
                lds_barrier(); /* block c+1 complete in `wr`; nobody reads `rd` any more */                       \
                LDX_CSTAMP(4)                                                                                      \
                /* step 3: MFMAs of (c,3); prepare (c+1,0) from the other buffer and the next A block, whose */   \
                /* words must have landed; quarter 0 of the B share of block c+2 goes into the buffer of block c, */ \
                /* which nobody reads any more.  Only this block's two A loads may still be in flight. */           \
                asm volatile("s_waitcnt vmcnt(%0)" : : "n"(MM));                                                   \
                touch_ring(NXT);                                                                                   \
                asm volatile("" : "+v"(br[FAR]));                                                                  \
                read_bf(bf0, wr, 0);                                                                               \
                _Pragma("unroll") for (int m = 0; m < MM; ++m) af0[m] = expand_a(ar[NXT][m].x);               \
                bquarter(bexp + (c_ & 1u) * kBBuf, br[FAR], 0);                                                    \
                mma8(af1, bf1);                                                                                    \
                interleave();                                                                                      \
                __builtin_amdgcn_sched_barrier(0);                                                                 \
                LDX_CSTAMP(5)                                                                                      \
            }
            // A wave whose unit lies outside the tile's segment (the last pass of a tile: 2 of 4 units every other tile of the
            // triangle, ~1.5 of 20 in the band's five passes per tile at +-1000 rows) has nothing to count: it keeps up its
            // share of the j-tile image -- the same loads and the same image writes on the same side of the per-block
            // barrier -- and issues no fragment reads and no MFMAs, which would only compete with the CU's other workgroup.
#define LDX_CHUNK_IDLE(NXT, FAR, cc)                                                                               \
            {                                                                                                      \
                const uint32_t c_ = (cc);                                                                          \
                unsigned char *wr = bexp + ((c_ + 1u) & 1u) * kBBuf;                                               \
                load_b(br[FAR], clampc(c_ + 2u));                                                                  \
                asm volatile("s_waitcnt vmcnt(1)");   /* in flight: B(c+1), B(c+2) */                              \
                asm volatile("" : "+v"(br[NXT]));                                                                  \
                bquarter(wr, br[NXT], 1);                                                                          \
                bquarter(wr, br[NXT], 2);                                                                          \
                bquarter(wr, br[NXT], 3);                                                                          \
                lds_barrier();                                                                                     \
                asm volatile("s_waitcnt vmcnt(0)");                                                                \
                asm volatile("" : "+v"(br[FAR]));                                                                  \
                bquarter(bexp + (c_ & 1u) * kBBuf, br[FAR], 0);                                                    \
            }
            if (__builtin_expect(!active, 0)) {   // wave-uniform
                for (uint32_t c = 0; c < nch_run; c += 3) {
                    LDX_CHUNK_IDLE(1, 2, c)
                    if (c + 1 < nch_run) LDX_CHUNK_IDLE(2, 0, c + 1)
                    if (c + 2 < nch_run) LDX_CHUNK_IDLE(0, 1, c + 2)
                }
            } else {
            // Round 5 tried PEELING the first K-block out of this loop, so that hipcc folds the accumulators' zero
            // initialisation into the C operand of the unit's first MFMAs (128 v_mov_b32 per unit and wave less: one
            // lane-instruction per pair).  It does fold them -- and the kernel got SLOWER where a launch is a few rounds of
            // passes: +3.2 % at 8 000 SNPs, +2.7 % at 10 000, nothing at 40 000 or at 50 000 x 1008
            // (profiles/r05/peeled_first_k_block_ab.log: same box, interleaved, three rounds; -DLDX_AB_PEEL builds it).
#ifdef LDX_AB_PEEL
            if constexpr (!kArea) {
                LDX_CHUNK(0, 1, 2, 0u)
                for (uint32_t c = 1; c < nch_run; c += 3) {
                    LDX_CHUNK(1, 2, 0, c)
                    if (c + 1 < nch_run) LDX_CHUNK(2, 0, 1, c + 1)
                    if (c + 2 < nch_run) LDX_CHUNK(0, 1, 2, c + 2)
                }
            } else
#endif
            {
                for (uint32_t c = 0; c < nch_run; c += 3) {   // block-uniform guards: every wave reaches every barrier
                    LDX_CHUNK(0, 1, 2, c)
                    if (c + 1 < nch_run) LDX_CHUNK(1, 2, 0, c + 1)
                    if (c + 2 < nch_run) LDX_CHUNK(2, 0, 1, c + 2)
                }
            }
            }
#undef LDX_CHUNK
#undef LDX_CHUNK_IDLE
            // drain the surplus loads of the last two chunks: their ring registers are about to be reused
            asm volatile("s_waitcnt vmcnt(0)");
            touch_ring(0);
            touch_ring(1);
            touch_ring(2);
            asm volatile("" : "+v"(br[0]), "+v"(br[1]), "+v"(br[2]));

#ifdef LDX_CHUNK_STAMPS
            asm volatile("s_waitcnt lgkmcnt(0)");
            if (my_stamps && lane == 0 && npass == 1)
                for (int k = 0; k < 6; ++k) my_stamps[6 + 4 * (kStampPasses - 2) + k] = cst[k];   // slots of passes 38, 39
#endif
            LDX_STAMP(2);
            // the next pass's ticket: drawn here (after the K loop's hand-counted loads), stored to LDS after the
            // epilogue, so the atomic's latency hides behind it
            uint32_t next_ticket = 0;
            if (tid == 0) {
                if (ktok) atomicExch(ktok, 0u);   // the other workgroup of the CU may start its K loop
                next_ticket = draw();
            }
            // The epilogue wave outranks the SIMD's other wave (in its K loop, matrix-pipe-bound with issue slots
            // to spare) in instruction arbitration: +2 % at 40k SNPs.
            if (ablate & 32) __builtin_amdgcn_s_setprio(0);
            else if (!(ablate & 16)) __builtin_amdgcn_s_setprio(3);
            if (!active) {   // wave-uniform; inactive waves only helped with B and the barriers
                if (tid == 0) tickets[parity] = next_ticket;
                return;
            }
            // epilogue: acc[m][tt][e] is 8 * n11 of pair (i, j) with
            //   i = row0 + 32*m + (e & 3) + 8*(e >> 2) + 4*half,  j = 128*t + 32*tt + l32
            typedef double d2 __attribute__((ext_vector_type(2)));
            double fa2[4], fr2[4];     // kRaw: the mirror's column operands
            double sfa = 0.0, sfr = 0.0, sq = 0.0;
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) {
                if (kRaw) {
                    fa2[tt] = fa[t * kSlab + 32u * tt + l32];
                    fr2[tt] = fr[t * kSlab + 32u * tt + l32];
                }
            }
            if (kRaw) {   // one coalesced load per statistic for the unit's 64 rows, handed out by shuffles below
                const uint32_t il = row0 + (MM == 1 ? l32 : lane);
                sfa = fa[il];
                sfr = fr[il];
                sq = q[il];
            }
            // Two variants of the loop, chosen per wave and pass: the general one, and a "clean" one for units that
            // lie wholly below the diagonal, inside the panel and inside [u_begin, u_end), whose 64 rows and 128
            // columns are all ordinary SNPs (polymorphic, no missing codes): no validity tests and selects, no
            // degenerate handling, no magnitude guard -- ~10 of 43 VALU instructions per pair less.
            auto epilogue = [&](auto clean_c) {
            // The e-loop is NOT unrolled: 128 pairs x ~50 instructions would be ~50 KB of straight-line code
            // per wave, most of the instruction cache two CUs share.  acc[..][..][e] with a wave-uniform e
            // is a register-indirect move (s_set_gpr_idx_on), not scratch.
#pragma unroll 1
            for (int e = 0; e < 16; ++e) {
                constexpr bool kClean = decltype(clean_c)::value;
                // The pairs of this step (MM rows x four column tiles) go through the epilogue WITHOUT branches --
                // invalid cells (row <= col, pad rows) are computed on whatever the registers hold and zeroed by a
                // select -- two at a time with their dependent fp64 chains interleaved (ld_multi_fast2): beside
                // another wave's K loop an epilogue gets few issue slots and must not also wait on its own latencies.
                uint32_t ri[MM], cnt[MM][4];
                uint64_t us[MM];
                bool in_range[MM], valid[MM][4], slow[MM][4];
                Cell res[MM][4];
                ldx_ld64 rw[MM][4];
                bool any_slow = false;
#pragma unroll
                for (int m = 0; m < MM; ++m) {
                    ri[m] = 32u * m + (e & 3) + 8u * (e >> 2) + 4u * half;   // row inside this wave's tile(s)
                    us[m] = vv * 8u + (ri[m] + roff) / kGroup;               // the small unit this row belongs to
                    in_range[m] = kClean || (us[m] >= u_begin && us[m] < u_end);
                }
                if (kRaw) {   // parity / debugging output: the op-for-op mirror, unrounded values kept
#pragma unroll
                    for (int m = 0; m < MM; ++m) {
                        const uint32_t i = row0 + ri[m];
                        const double fa1 = __shfl(sfa, (int)ri[m]), fr1 = __shfl(sfr, (int)ri[m]), q1 = __shfl(sq, (int)ri[m]);
#pragma unroll
                        for (int tt = 0; tt < 4; ++tt) {
                            const uint32_t j = t * kSlab + 32u * tt + l32;
                            valid[m][tt] = (i > j) && (i < n_snps);
                            cnt[m][tt] = count_of(acc[m][tt][e]);
                            const LdRaw lr = ld_epilogue((double)cnt[m][tt] / n, fa1, fr1, q1, fa2[tt], fr2[tt]);   // calc_ld.py:33
                            res[m][tt] = encode_cell<Cell>(round_pair(lr));
                            rw[m][tt] = valid[m][tt] ? ldx_ld64{lr.rsq, lr.dprime} : ldx_ld64{0.0, 0.0};
                        }
                    }
                } else {
                    // Two chains at a time, staged: with 128 live accumulators there is room for the operands and
                    // temporaries of two interleaved pairs, not four (hipcc then spills accumulator tiles).  A 64-row
                    // unit pairs its two rows on one column (MM == 2), a half-height one two columns on its row;
                    // operands come from LDS every step, the next pair's columns one pair ahead.
                    FastRow fr2x[MM];
#pragma unroll
                    for (int m = 0; m < MM; ++m) {
                        const d2 *rs = reinterpret_cast<const d2 *>(rstat + ri[m] * kStat);   // two addresses per wave: broadcast
                        const d2 r01 = rs[0], r23 = rs[1];
                        fr2x[m] = FastRow{r01.x, r01.y, r23.x, r23.y};
                    }
                    constexpr int kPairs = MM == 2 ? 4 : 2;
                    auto chain_m = [](int, int k) { return MM == 2 ? k : 0; };
                    auto chain_tt = [](int pp, int k) { return MM == 2 ? pp : 2 * pp + k; };
                    auto load_col = [&](int tt) {
                        const d2 *cs = reinterpret_cast<const d2 *>(cstat + (32u * tt + l32) * kStat);
                        const d2 c01 = cs[0], c23 = cs[1];
                        return FastCol{c01.x, c01.y, c23.x, c23.y};
                    };
                    FastCol nxt[2];
                    nxt[0] = load_col(chain_tt(0, 0));
                    nxt[1] = MM == 2 ? nxt[0] : load_col(chain_tt(0, 1));
#pragma unroll
                    for (int pp = 0; pp < kPairs; ++pp) {
                        const FastCol fcx[2] = {nxt[0], nxt[1]};
                        if (pp + 1 < kPairs) {   // the next pair's columns: their LDS latency hides behind this pair's arithmetic
                            nxt[0] = load_col(chain_tt(pp + 1, 0));
                            nxt[1] = MM == 2 ? nxt[0] : load_col(chain_tt(pp + 1, 1));
                        }
                        FastRow frk[2];
                        accel_t a8[2];
                        Cell r2[2];
                        bool s2[2];
#pragma unroll
                        for (int k = 0; k < 2; ++k) {
                            const int m = chain_m(pp, k), tt = chain_tt(pp, k);
                            frk[k] = fr2x[m];
                            const uint32_t i = row0 + ri[m], j = t * kSlab + 32u * tt + l32;
                            valid[m][tt] = kClean || ((i > j) && (i < n_snps));
                            a8[k] = acc[m][tt][e];
                        }
                        if (ablate & 1) {   // tuning: no epilogue arithmetic
#pragma unroll
                            for (int k = 0; k < 2; ++k) {
                                res[chain_m(pp, k)][chain_tt(pp, k)] = encode_cell<Cell>((double)a8[k], 0.0, false, false);
                                slow[chain_m(pp, k)][chain_tt(pp, k)] = false;
                            }
                        } else {
                            ld_multi_fast2<2, kClean, Cell>(a8, fk, frk, fcx, r2, s2);
#pragma unroll
                            for (int k = 0; k < 2; ++k) {
                                const int m = chain_m(pp, k), tt = chain_tt(pp, k);
                                res[m][tt] = r2[k];
                                slow[m][tt] = s2[k] && valid[m][tt];
                                any_slow = any_slow || slow[m][tt];
                            }
                        }
                    }
                    if (__builtin_expect(__any(any_slow), 0)) {   // near a rounding tie, Dn == 0, ...: the exact mirror
#pragma unroll
                        for (int m = 0; m < MM; ++m)
#pragma unroll
                            for (int tt = 0; tt < 4; ++tt)
                                if (slow[m][tt]) {
                                    const uint32_t i = row0 + ri[m], j = t * kSlab + 32u * tt + l32;
                                    res[m][tt] = encode_cell<Cell>(
                                        ld_pair_mirror((double)count_of(acc[m][tt][e]) / n, fa[i], fr[i], q[i], fa[j], fr[j]));
                                }
                    }
                }
#pragma unroll
                for (int m = 0; m < MM; ++m)
                    if (in_range[m] && !(ablate & 4)) {
#pragma unroll
                        for (int tt = 0; tt < 4; ++tt) {
                            const uint32_t jl = 32u * tt + l32;
                            const size_t o = (size_t)(us[m] - u_begin) * LDX_UNIT_PAIRS + cell_offset<Cell>((ri[m] + roff) % kGroup, jl);
                            Cell w = res[m][tt];
                            if (!kClean && !valid[m][tt]) w = zero_cell<Cell>();
                            store_cell(out + o, w);
                            if (kRaw) raw[o] = rw[m][tt];
                            if (kN11) n11[o] = valid[m][tt] ? count_of(acc[m][tt][e]) : 0u;
                        }
                    }
            }
            };
            // ---- fp32 first tier (FP4 triangle kernel, units of ordinary SNPs wholly inside the triangle and the range) ----
            // Per step each lane runs its 8 pairs through ld_multi_f32 and tests ONCE whether all of them are provably
            // rounded like the reference; if so it stores its 8 cells, if not it parks the step (id + 8 counts) in the
            // wave's LDS queue.  The queue is drained AFTER the sixteen steps -- the accumulators are dead by then, so the
            // fp64 tier (ld_multi_fast2, and the op-for-op mirror behind it) has the registers it wants -- eight entries
            // at a time, one parked PAIR per lane: the rare path runs at full lane occupancy.  A unit that parks more steps
            // than the queue holds is redone as a whole by the fp64 epilogue (returns false).
            auto epilogue_f32 = [&](bool all_ordinary) -> bool {
              if constexpr (kF32Tier) {
                if (ablate & 1) return true;   // tuning: no epilogue at all
                // this unit's cells: a wave-uniform base (scalar registers) + a per-lane constant + a per-step scalar offset
                const uint64_t ub = vv * 8u - u_begin;
                Cell *const ubase = out + (((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)(ub >> 32)) << 32) |
                                           __builtin_amdgcn_readfirstlane((uint32_t)ub)) * LDX_UNIT_PAIRS;
                // the lane's half / column, recomputed HERE from an opaque copy of the lane id: values derived from it at the
                // top of the kernel would be hoisted out of the pass loop, spilled around the K loop (256 registers) and
                // reloaded from scratch inside the step loop -- behind a vmcnt(0) that also waits for the step's stores
                uint32_t ln = lane;
                asm volatile("" : "+v"(ln));
                const uint32_t l32e = ln & 31u, halfe = ln >> 5;
                // rows e and e + 4 of a group of 8 belong to the two lane halves; a lane's four columns (l32 + 32 tt) are adjacent
                // cells of the row (include/ldx.h: 4 l32 .. 4 l32 + 3 for 4-byte cells; the pairs 2 l32, 2 l32 + 1 and 64 + 2 l32,
                // 65 + 2 l32 for 8-byte cells): one 16-byte store per row, or two
                const uint32_t lane_off = halfe * 4u * kSlab + (sizeof(Cell) <= 4 ? 4u : 2u) * l32e;
                const uint32_t lane_off_b = lane_off * (uint32_t)sizeof(Cell);
                const float *const rt = rtab32 + halfe * 16u, *const ct = ctab32 + l32e * 4u;
                const uint32_t grp0 = roff / kGroup;   // first 8-row group of this wave's rows inside the unit (scalar)
                uint32_t qn = 0;   // parked steps (wave-uniform)
                F32Col cols[4];    // this lane's four columns: held for the sixteen steps (16 registers; no spills at 256)
#pragma unroll
                for (int tt = 0; tt < 4; ++tt) {
                    const v4f v = *reinterpret_cast<const v4f *>(ct + 32u * tt * 4u);
                    cols[tt] = F32Col{v.x, v.y, v.z, v.w};
                }
                // Degenerate SNPs of this unit (ldx_common.h, snp_class / f32_row): lane masks, all scalar.  colmask[tt] = lanes
                // whose column of tile tt is degenerate (from the table entries this lane just read); rdeg = bit per row of the
                // wave's 64 (lane l reads row l's entry once).  A unit without any runs the loop it always ran.
                uint64_t colmask[4], rdeg;
#pragma unroll
                for (int tt = 0; tt < 4; ++tt) colmask[tt] = __builtin_amdgcn_ballot_w64(f32_entry_degenerate(cols[tt].ra, cols[tt].s));
                {
                    const v4f rv = *reinterpret_cast<const v4f *>(rtab32 + (MM == 1 ? l32e : ln) * 4u);
                    rdeg = __builtin_amdgcn_ballot_w64(f32_entry_degenerate(rv.y, rv.w));
                }
#ifdef LDX_AB_NODEGSEL   // tuning: never force (the cells of degenerate SNPs come out as 0 / 0: wrong by design)
                const bool any_deg = false;
#else
                const bool any_deg = (rdeg | colmask[0] | colmask[1] | colmask[2] | colmask[3]) != 0ull;   // wave-uniform
#endif
                float cal[4] = {0.0f, 0.0f, 0.0f, 0.0f};   // n > 4096: the columns' counts split (ldx_common.h, f32_split_a): a = ah + al
                if (!f32_small_n((double)fc32.n)) {
#pragma unroll
                    for (int tt = 0; tt < 4; ++tt) f32_split_a(cols[tt].a, cols[tt].a, cal[tt]);
                }
                // the sixteen steps, in four instantiations: n <= 4096 needs no error term for the product a1 a2 (ldx_common.h);
                // kDeg forces the cells of degenerate rows / columns to the int-0 code (one v_cndmask per cell)
                auto steps = [&](auto small_c, auto deg_c) -> bool {
                constexpr bool kDeg = decltype(deg_c)::value;
                [[maybe_unused]] const uint32_t forced = sizeof(Cell) == 4 ? ((uint32_t)LDX_K16_INT0 << 16 | LDX_K16_INT0) : 0x80000000u;   // int 0, int 0
                // LDX_STEP_UNROLL steps per trip of the loop (1, 2, 4, 8 or 16; the rows of a step are (e & 3) + 8 (e >> 2) + 32 m:
                // unrolled by 4 the row inside its group of eight is static -- LDS and store offsets become immediates, the
                // scalar address arithmetic happens once per four steps --, unrolled by 16 the accumulator index is static too)
                auto load_rows = [&](int e, F32Row (&dst)[MM]) {
#pragma unroll
                    for (int m = 0; m < MM; ++m) {   // two addresses per wave: broadcast
                        const v4f v = *reinterpret_cast<const v4f *>(rt + (32u * m + (e & 3) + 8u * (e >> 2)) * 4u);
                        dst[m] = F32Row{v.x, v.y, v.z, v.w};
                    }
                };
#ifdef LDX_ROW_PREFETCH   // the rows of step e + 1 are read at the top of step e (two buffers; needs an even LDX_STEP_UNROLL)
                static_assert(LDX_STEP_UNROLL % 2 == 0, "LDX_ROW_PREFETCH alternates two row buffers: unroll by an even number");
                F32Row rowbuf[2][MM];
                load_rows(0, rowbuf[0]);
#endif
#pragma unroll 1
                for (int e_ = 0; e_ < 16; e_ += LDX_STEP_UNROLL) {
#pragma unroll
                for (int r_ = 0; r_ < LDX_STEP_UNROLL; ++r_) {
                    const int e = e_ + r_;
#if defined(LDX_TUNING) && defined(LDX_STAMPS_ONLY)
                    if (my_stamps && lane == 0 && npass == 1) my_stamps[6 + 4 * kStampPasses + e] = __builtin_amdgcn_s_memtime();
#endif
#ifdef LDX_ROW_PREFETCH
                    F32Row (&rows)[MM] = rowbuf[r_ & 1];
                    load_rows((e + 1) & 15, rowbuf[(r_ + 1) & 1]);   // (after the last step: step 0's rows again, unused)
#else
                    F32Row rows[MM];
                    load_rows(e, rows);
#endif
                    constexpr int kOne = cell_measure<Cell>::value;   // >= 0: a one-measure format (2-byte cells)
                    Cell cell[4 * MM];
                    uint32_t cellw[2 * MM];   // one-measure formats: a row's four cells as two words
                    float wmax = 0.0f, ymin = 1.0f;
#pragma unroll
                    for (int g = 0; g < MM; ++g) {   // four interleaved chains at a time: (m, tt) = (g, 0..3)
                        float c4[4];
                        F32Row r4[4];
#pragma unroll
                        for (int tt = 0; tt < 4; ++tt) {
                            c4[tt] = acc[g][tt][e];
                            r4[tt] = rows[g];
                        }
                        uint64_t rowmask = 0;
                        if constexpr (kDeg) {   // this step's two rows of tile g: lanes 0-31 hold row ri, lanes 32-63 row ri + 4
                            const uint32_t ri0 = 32u * g + (e & 3) + 8u * (e >> 2);
                            rowmask = (((rdeg >> ri0) & 1ull) ? 0x00000000FFFFFFFFull : 0ull) |
                                      (((rdeg >> (ri0 + 4u)) & 1ull) ? 0xFFFFFFFF00000000ull : 0ull);
                        }
                        if constexpr (kOne >= 0) {
                            uint32_t b4[4];
                            ld_multi_f32_one<4, kOne, decltype(small_c)::value, !decltype(small_c)::value>(c4, fc32, r4, cols, b4, wmax, ymin, cal);
                            if constexpr (kDeg) {
#pragma unroll
                                for (int tt = 0; tt < 4; ++tt) b4[tt] = select_lanes(b4[tt], (uint32_t)LDX_K16_INT0, rowmask | colmask[tt]);
                            }
                            // the cells are the low halves of 2^23 + k: two per byte permute
                            cellw[2 * g] = __builtin_amdgcn_perm(b4[1], b4[0], 0x05040100u);
                            cellw[2 * g + 1] = __builtin_amdgcn_perm(b4[3], b4[2], 0x05040100u);
                        } else {
                            Cell o4[4];
                            ld_multi_f32<4, Cell, decltype(small_c)::value, !decltype(small_c)::value>(c4, fc32, r4, cols, o4, wmax, ymin, cal);
                            if constexpr (kDeg) {
#pragma unroll
                                for (int tt = 0; tt < 4; ++tt) {
                                    const uint64_t m = rowmask | colmask[tt];
                                    if constexpr (sizeof(Cell) == 4) {
                                        o4[tt] = __builtin_bit_cast(Cell, select_lanes(__builtin_bit_cast(uint32_t, o4[tt]), forced, m));
                                    } else {
                                        const v2u w = __builtin_bit_cast(v2u, o4[tt]);
                                        o4[tt] = __builtin_bit_cast(Cell, v2u{select_lanes(w.x, forced, m), select_lanes(w.y, forced, m)});
                                    }
                                }
                            }
#pragma unroll
                            for (int tt = 0; tt < 4; ++tt) cell[g * 4 + tt] = o4[tt];
                        }
                    }
                    const bool sure = ((wmax < fc32.tol) & (ymin > 0.0f)) | ((ablate & 1024) != 0);   // tuning: 1024 = never park
#ifdef LDX_AB_NOSTORE   // tuning: the cells are computed and kept alive, not stored (results missing)
                    if (sure) {
                        uint32_t x = 0;
#pragma unroll
                        for (int q8 = 0; q8 < 4 * MM; ++q8) { if constexpr (sizeof(Cell) == 2) x ^= cellw[q8 / 2]; else if constexpr (sizeof(Cell) == 4) x ^= __builtin_bit_cast(uint32_t, cell[q8]); else x ^= (uint32_t)__builtin_bit_cast(unsigned long long, cell[q8]) ^ (uint32_t)(__builtin_bit_cast(unsigned long long, cell[q8]) >> 32); }
                        asm volatile("" : : "v"(x));
                    }
                    if (false) {
#else
                    if (sure && !(ablate & 4)) {
#endif
#pragma unroll
                        for (int m = 0; m < MM; ++m) {   // groups of 8 rows: 4 m + e / 4 of a whole unit, 4 hsel + e / 4 of a half-height one
                            Cell *const row = ubase + ((4u * m + grp0 + (e >> 2)) * LDX_UNIT_PAIRS + (e & 3) * kSlab);   // scalar
                            if constexpr (kOne >= 0) store_words2_saddr(row, lane_off_b, cellw[2 * m], cellw[2 * m + 1]);
                            else store_cells4_saddr(row, lane_off_b, cell[m * 4 + 0], cell[m * 4 + 1], cell[m * 4 + 2], cell[m * 4 + 3]);
                        }
                    }
#ifdef LDX_AB_OLD_BALLOT
                    const unsigned long long parked = __ballot(!sure);
#else
                    const unsigned long long parked = __builtin_amdgcn_ballot_w64(!sure);   // (HIP's __ballot compares an INT with 0: a v_cndmask + v_cmp per step)
#endif
                    if (parked) {   // wave-uniform
                        const uint32_t np = (uint32_t)__builtin_popcountll(parked);
                        if (qn + np > kQueueCap) {   // more than the queue holds: the fp64 epilogue redoes the unit
                            LDX_COUNT(2, 1);
                            return false;
                        }
                        if (!sure) {
                            const uint32_t pos = qn + __builtin_amdgcn_mbcnt_hi((uint32_t)(parked >> 32),
                                                                              __builtin_amdgcn_mbcnt_lo((uint32_t)parked, 0u));
                            qid[pos] = ((uint32_t)e << 8) | ln;
                            v4f *dst = reinterpret_cast<v4f *>(qcnt + (size_t)pos * 8u);
#pragma unroll
                            for (int m = 0; m < MM; ++m) dst[m] = v4f{acc[m][0][e], acc[m][1][e], acc[m][2][e], acc[m][3][e]};
                        }
                        qn += np;
                    }
                }
                }
#if defined(LDX_TUNING) && defined(LDX_STAMPS_ONLY)
                    if (my_stamps && lane == 0 && npass == 1) my_stamps[6 + 4 * kStampPasses + 16] = __builtin_amdgcn_s_memtime();
#endif
                    return true;
                };
                const bool small_n = f32_small_n((double)fc32.n);
                const bool done = any_deg ? (small_n ? steps(std::true_type{}, std::true_type{}) : steps(std::false_type{}, std::true_type{}))
                                          : (small_n ? steps(std::true_type{}, std::false_type{}) : steps(std::false_type{}, std::false_type{}));
                if (!done) return false;
                // ---- the parked steps: fp64 tier ----
                LDX_COUNT(0, 1);
                LDX_COUNT(1, qn);
                if ((ablate & 2048) != 0) qn = 0;   // tuning: skip the drain (results wrong)
                if (qn) {
                    __builtin_amdgcn_s_waitcnt(0xC07F);   // this wave's queue writes have landed (lgkmcnt(0))
                    __builtin_amdgcn_wave_barrier();
                }
                // one PAIR per lane: 4 MM lanes share a parked step (lane % (4 MM) = 4 m + tt), so the usual handful of entries
                // is one short batch at full occupancy instead of a 64-lane batch with a few busy lanes doing eight pairs each
                // (lane-derived values recomputed from an opaque copy of the lane id, as above: hoisted out of the pass loop
                // they would be spilled to scratch around the K loop)
                uint32_t ld = ln;
                asm volatile("" : "+v"(ld));
                constexpr uint32_t kPer = 4u * MM, kShift = MM == 2 ? 3u : 2u;   // lanes per parked step
                // Two phases.  First the fp64 count-domain tier on every parked pair; a pair it cannot call either (near a
                // rounding tie, Dn == 0) is DEFERRED to a list in LDS instead of being sent through the op-for-op mirror on the
                // spot: the mirror (three IEEE divisions, ~150 double-rate instructions, out of line) runs for the whole wave
                // whenever one lane needs it, and with its handful of candidates per unit spread over the batches nearly
                // every batch paid for it (50 000 x 1008: ~8 candidates and 2-3 batches per unit).  Then ONE mirror batch
                // over the compacted list.  (More than 64 deferred pairs: the surplus takes the mirror at once, as before.)
                auto pair_of = [&](uint32_t ent, uint32_t pr, uint32_t &ri2, uint32_t &cl) {
                    const uint32_t id = qid[ent];
                    const uint32_t e2 = id >> 8, l2 = id & 31u, h2 = (id >> 5) & 1u;
                    ri2 = 32u * (pr >> 2) + (e2 & 3u) + 8u * (e2 >> 2) + 4u * h2;
                    cl = 32u * (pr & 3u) + l2;
                    return qcnt[(size_t)ent * 8u + pr];
                };
                auto mirror_cell = [&](float cnt, uint32_t ri2, uint32_t cl) {
                    const uint32_t i = row0 + ri2, j = t * kSlab + cl;
#if defined(LDX_TUNING) && !defined(LDX_STAMPS_ONLY)
                    atomicAdd(&g_dbg[3], 1ull);
#endif
                    return encode_cell<Cell>(ld_pair_mirror((double)cnt / n, fa[i], fr[i], q[i], fa[j], fr[j]));
                };
                auto cell_at = [&](uint32_t ri2, uint32_t cl) -> Cell & {
                    return ubase[(size_t)(grp0 + ri2 / kGroup) * LDX_UNIT_PAIRS + cell_offset<Cell>(ri2 % kGroup, cl)];
                };
                uint32_t un = 0;   // deferred pairs (wave-uniform)
                for (uint32_t q0 = 0; q0 < qn; q0 += 64u / kPer) {   // wave-uniform
                    const uint32_t ent = q0 + (ld >> kShift), pr = ld & (kPer - 1u);
                    const bool live = ent < qn;
                    uint32_t ri2 = 0, cl = 0;
                    Cell r2[1] = {zero_cell<Cell>()};
                    bool s2[1] = {false};
                    float a2[1] = {0.0f};
                    if (live) {
                        a2[0] = pair_of(ent, pr, ri2, cl);
                        const d2 *rs = reinterpret_cast<const d2 *>(rstat + ri2 * kStat);
                        const d2 r01 = rs[0], r23 = rs[1];
                        const FastRow frk[1] = {FastRow{r01.x, r01.y, r23.x, r23.y}};
                        const d2 *cs = reinterpret_cast<const d2 *>(cstat + cl * kStat);
                        const d2 c01 = cs[0], c23 = cs[1];
                        const FastCol fcx[1] = {FastCol{c01.x, c01.y, c23.x, c23.y}};
                        if (all_ordinary) ld_multi_fast2<1, true, Cell>(a2, fk, frk, fcx, r2, s2);
                        else ld_multi_fast2<1, false, Cell>(a2, fk, frk, fcx, r2, s2);   // a monomorphic SNP / missing codes in the unit
                    }
                    const bool unsure = live && s2[0];
                    const unsigned long long um = __ballot(unsure);
                    bool deferred = false;
                    if (um) {   // wave-uniform
                        const uint32_t pos = un + __builtin_amdgcn_mbcnt_hi((uint32_t)(um >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)um, 0u));
                        deferred = unsure && pos < 64u;
                        if (deferred) ulist[pos] = (ent << 3) | pr;
                        un += (uint32_t)__builtin_popcountll(um);
                        if (__builtin_expect(un > 64u, 0)) {   // the list is full: these take the mirror now
                            if (unsure && !deferred) r2[0] = mirror_cell(a2[0], ri2, cl);
                        }
                    }
                    if (live && !deferred) cell_at(ri2, cl) = r2[0];
                }
                if (un) {   // phase 2: the deferred pairs, one mirror batch
                    __builtin_amdgcn_s_waitcnt(0xC07F);
                    __builtin_amdgcn_wave_barrier();
                    if (ld < (un < 64u ? un : 64u)) {
                        const uint32_t id = ulist[ld];
                        uint32_t ri2, cl;
                        const float cnt = pair_of(id >> 3, id & 7u, ri2, cl);
                        cell_at(ri2, cl) = mirror_cell(cnt, ri2, cl);
                    }
                }
                return true;
              } else {
                return false;
              }
            };
            // ---- ld_area: thresholded hits instead of a dense result ----
            // Pair (i, j), i > j, pos_i >= pos_j, serves two ordered pairs of the reference's loop:
            //   A: query i, opposing j -- j lies in i's window iff max(0, pos_i - flank) < pos_j     (ld_area.py:174-177)
            //   B: query j, opposing i -- iff max(0, pos_j - flank) < pos_i <= pos_j + flank  (the lower bound only bites
            //      for equal positions with flank 0)
            // D' and the fast path's r^2 do not depend on the order; the reference's r^2 does in its last bits
            // ((fa1*fr1)*fa2)*fr2 associates in argument order), so a pair that needs the mirror
            // gets it once per order.
            auto area_epilogue = [&]() {
              if constexpr (kArea && MM == 2) {
                uint64_t slot = hit_slot, slot_end = hit_slot_end;
                const double kthr = aa.k_thres;
                const bool prefilter = aa.k_thres > 2.0;
                const double kcand = aa.k_thres - 2.0;
                auto append = [&](bool keep, uint32_t qrow, uint32_t orow, ldx_ld32 v) {
                    const unsigned long long mask = __ballot(keep);
                    if (!mask) return;   // wave-uniform
                    const uint32_t cnt = __builtin_popcountll(mask);
                    if (slot + cnt > slot_end) {   // close the old batch (mark what is left invalid), open a new one
                        for (uint64_t sl = slot + lane; sl < slot_end; sl += 64u)
                            if (sl < aa.hit_cap) aa.hits[sl].query = 0xFFFFFFFFu;
                        unsigned long long base = 0;
                        if (lane == 0) base = atomicAdd(aa.n_hits, (unsigned long long)kHitBatch);
                        base = ((unsigned long long)__builtin_amdgcn_readfirstlane((uint32_t)(base >> 32)) << 32) |
                               __builtin_amdgcn_readfirstlane((uint32_t)base);
                        slot = base;
                        slot_end = base + kHitBatch;
                    }
                    if (keep) {
                        const uint64_t sl = slot + __builtin_popcountll(mask & ((1ull << lane) - 1ull));
                        if (sl < aa.hit_cap) {
                            aa.hits[sl] = ldx_hit{qrow, orow, v.r_square, v.d_prime};
                            if (aa.counts) atomicAdd(&aa.counts[qrow], 1u);   // only stored hits: what the scatter will place
                        }
                    }
                    slot += cnt;
                };
                // Float32 screen of a whole step (r^2 thresholds): hits are rare, and the fp64 prefilter below still costs every
                // (step, column tile) three 16-byte LDS reads per lane and ~7 double-rate instructions per pair.  With this lane's
                // four columns held in registers as {ah, al, c = sqrt(k) / s2} and the step's two rows read as {a, s1} (s = 10 /
                // sqrt(a r)), a pair is within reach of the threshold iff |Dn| s1 >= c, i.e. 10^4 r^2 = (Dn s1 s2)^2 >= k: 5
                // single-rate instructions per pair (Dn exact through the split column count, f32_split_a; c a hair low and k 2.5
                // units under the threshold: float32's ~1e-6 is < 0.01 unit); a step in which no lane is within reach --
                // whatever the pair's window -- is skipped before any of its fp64 operands are read, and only column tiles
                // with a lane within reach go on to the fp64 prefilter.  The fp64 prefilter and the full epilogue decide
                // everything else, so hits are unchanged.  (A count of 0 -- r^2 is the int 0, never a hit under a positive
                // threshold -- is out of reach by construction: s1 = 0 for such a row, c = +inf for such a column.)
                const bool screen = kBandF32 && prefilter && aa.measure == LDX_MEASURE_RSQ;
                float sc_ah[4], sc_al[4], sc_s[4];
                const float sc_n = aa.f32.n;
                const bool sc_diag = row0 < (t + 1u) * kSlab;      // the unit has cells on or above the diagonal (wave-uniform)
                const uint32_t t_col0 = t * kSlab + l32;            // this lane's column of the tile's first 32
                if constexpr (kBandF32) {
                    if (screen) {
#pragma unroll
                        for (int tt = 0; tt < 4; ++tt) {
                            const v4f v = *reinterpret_cast<const v4f *>(ctab32 + (32u * tt + l32) * 4u);
                            sc_ah[tt] = v.x;
                            sc_al[tt] = v.y;
                            sc_s[tt] = v.z;   // the column's share of the threshold
                        }
                    }
                }
                // Candidates are COLLECTED, not evaluated where they are found: a pair that the (fp64) prefilter cannot rule out --
                // valid cells only: the upper triangle of a diagonal unit mirrors real hits -- goes into the wave's LDS queue
                // (step, row tile, column tile, lane + the accumulator word), and the full epilogue runs over the queue with ONE
                // PAIR PER LANE.  Hits cluster near the diagonal: evaluated in place, every step of a tile's first pass ran the
                // full fp64 epilogue for two or three of its column tiles, each time for 128 pairs of which a handful mattered
                // (round 4 stamps: 97k cycles per such pass against 11k for the others, 40 % of the kernel's work).
                uint32_t aqn = 0;   // queued candidates (wave-uniform)
                auto drain = [&]() {
                    if (!aqn) return;
                    LDX_COUNT(0, 1);     // tuning builds: drains, candidates evaluated
                    LDX_COUNT(1, aqn);
                    __builtin_amdgcn_s_waitcnt(0xC07F);   // this wave's queue writes have landed
                    __builtin_amdgcn_wave_barrier();
                    for (uint32_t q0 = 0; q0 < aqn; q0 += 64u) {   // wave-uniform
                        const uint32_t ent = q0 + lane;
                        const bool live = ent < aqn;
                        const uint32_t id = live ? aq_id[ent] : 0u;
                        const uint32_t e2 = id >> 9, m2 = (id >> 8) & 1u, tt2 = (id >> 6) & 3u, l2 = id & 63u;
                        const uint32_t ri2 = 32u * m2 + (e2 & 3u) + 8u * (e2 >> 2) + 4u * (l2 >> 5), cl = 32u * tt2 + (l2 & 31u);
                        const uint32_t i = row0 + ri2, j = t * kSlab + cl;
                        accel_t a1[1] = {__builtin_bit_cast(accel_t, live ? aq_cnt[ent] : 0u)};
                        const d2 *rs = reinterpret_cast<const d2 *>(rstat + ri2 * kStat);
                        const d2 r01 = rs[0], r23 = rs[1], r45 = rs[2];
                        const FastRow fr1[1] = {FastRow{r01.x, r01.y, r23.x, r23.y}};
                        const double pi = r45.x, qi = r45.y;
                        const d2 *cs = reinterpret_cast<const d2 *>(cstat + cl * kStat);
                        const d2 c01 = cs[0], c23 = cs[1], c45 = cs[2];
                        const FastCol fc1[1] = {FastCol{c01.x, c01.y, c23.x, c23.y}};
                        const double pj = c45.x, qj = c45.y;
                        ldx_ld32 r1[1];
                        bool s1[1];
                        ld_multi_fast2<1, false, ldx_ld32>(a1, fk, fr1, fc1, r1, s1);
                        const bool valid = live && (i > j) && (i < n_snps);
                        double low = pi - aa.flank;
                        low = low < 0.0 ? 0.0 : low;
                        const bool in_a = valid && qi != 0.0 && low < pj;                 // A: query i, opposing j
                        double lowj = pj - aa.flank;
                        lowj = lowj < 0.0 ? 0.0 : lowj;
                        const bool in_b = valid && qj != 0.0 && lowj < pi && pi <= pj + aa.flank;   // B: query j, opposing i
                        ldx_ld32 ra = r1[0], rb = r1[0];
                        if (__builtin_expect(__any(s1[0] && (in_a || in_b)), 0)) {
                            if (s1[0] && (in_a || in_b)) {
                                const double f11 = (double)count_of(a1[0]) / n;
                                ra = encode_cell<ldx_ld32>(ld_pair_mirror(f11, fa[i], fr[i], q[i], fa[j], fr[j]));
                                rb = encode_cell<ldx_ld32>(ld_pair_mirror(f11, fa[j], fr[j], q[j], fa[i], fr[i]));
                            }
                        }
                        // rounded value * 10^4 back as an integer (exact for values < 1024; -0.0f = int 0 -> 0; the
                        // escape NaN of a value >= 1024 counts as +inf: it passes every threshold the band accepts)
                        const float va = aa.measure == LDX_MEASURE_RSQ ? ra.r_square : ra.d_prime;
                        const float vb = aa.measure == LDX_MEASURE_RSQ ? rb.r_square : rb.d_prime;
                        const double ka = va != va ? __builtin_inf() : __builtin_rint((double)va * 1e4);
                        const double kb = vb != vb ? __builtin_inf() : __builtin_rint((double)vb * 1e4);
                        append(in_a && ka >= kthr, i, j, ra);                              // ld_area.py:248
                        append(in_b && kb >= kthr, j, i, rb);
                    }
                    aqn = 0;
                    __builtin_amdgcn_wave_barrier();   // (the next pushes overwrite entries this loop has read)
                };
#pragma unroll 1
                for (int e = 0; e < 16; ++e) {
                    uint32_t tt_live = 0xFu;   // column tiles of this step that the screen could not rule out (wave-uniform)
                    if constexpr (kBandF32) {
                        if (screen) {
                            float ymx[4] = {-1.0f, -1.0f, -1.0f, -1.0f};
#pragma unroll
                            for (int m = 0; m < 2; ++m) {
                                typedef float v2f __attribute__((ext_vector_type(2)));
                                const uint32_t rim = 32u * m + (e & 3) + 8u * (e >> 2) + 4u * half;
                                const v2f rv = *reinterpret_cast<const v2f *>(rtab32 + rim * 4u);
                                float dn[4], t[4];
#pragma unroll
                                for (int tt = 0; tt < 4; ++tt) dn[tt] = __builtin_fmaf(acc[m][tt][e], sc_n, -(rv.x * sc_ah[tt]));
#pragma unroll
                                for (int tt = 0; tt < 4; ++tt) dn[tt] = __builtin_fmaf(-rv.x, sc_al[tt], dn[tt]);
#pragma unroll
                                for (int tt = 0; tt < 4; ++tt) t[tt] = __builtin_fmaf(__builtin_fabsf(dn[tt]), rv.y, -sc_s[tt]);   // >= 0: within reach
                                if (sc_diag) {   // a unit on the diagonal: cells with row <= column mirror real hits (and i == j is r^2 = 1)
#pragma unroll
                                    for (int tt = 0; tt < 4; ++tt) t[tt] = row0 + rim > t_col0 + 32u * tt ? t[tt] : -1.0f;
                                }
#pragma unroll
                                for (int tt = 0; tt < 4; ++tt) ymx[tt] = __builtin_fmaxf(ymx[tt], t[tt]);
                            }
                            if (!__builtin_amdgcn_ballot_w64(__builtin_fmaxf(__builtin_fmaxf(ymx[0], ymx[1]), __builtin_fmaxf(ymx[2], ymx[3])) >= 0.0f)) continue;   // wave-uniform
                            tt_live = 0u;   // (steps with candidates are rare: four more ballots only here)
#pragma unroll
                            for (int tt = 0; tt < 4; ++tt) tt_live |= __any(ymx[tt] >= 0.0f) ? 1u << tt : 0u;
                        }
                    }
                    uint32_t ri[2];
                    double ras[2], rqs[2], rra[2], rrr[2];   // the rows' prefilter operands: 1e4 a, 1e-4 / (a r), 1 / a, 1 / r
#pragma unroll
                    for (int m = 0; m < 2; ++m) {
                        ri[m] = 32u * m + (e & 3) + 8u * (e >> 2) + 4u * half;
                        const d2 *rs = reinterpret_cast<const d2 *>(rstat + ri[m] * kStat);
                        const d2 r01 = rs[0], r23 = rs[1];
                        ras[m] = r01.x;
                        rra[m] = r01.y;
                        rrr[m] = r23.x;
                        rqs[m] = r23.y;
                    }
#pragma unroll
                    for (int tt = 0; tt < 4; ++tt) {
                        if (!((tt_live >> tt) & 1u)) continue;   // wave-uniform
                        const d2 *cs = reinterpret_cast<const d2 *>(cstat + (32u * tt + l32) * kStat);
                        const d2 c01 = cs[0], c23 = cs[1];
                        const uint32_t j = t * kSlab + 32u * tt + l32;
                        // Prefilter: hits are rare (r^2 >= 0.8: ~6e-4 of the pairs), so first price the thresholded
                        // measure alone -- 6 VALU for r^2 * 10^4, 15 for D' * 10^4, exact to ~1e-11 -- and queue only
                        // the pairs that come within 2 units of the threshold (the two orders of a pair
                        // and the reference's own rounding differ by far less).  A NaN (a count of 0: int-0 results)
                        // never passes; thresholds <= 2e-4 switch the prefilter off: every valid pair is queued.
                        bool cand[2];
                        // ONE register-indexed read per accumulator, pinned: hipcc (ROCm 7.2) turned a second `acc[m][tt][e]`
                        // -- the one stored into the queue below -- into a plain read of element 0 of the tile
                        // (ds_write_b32 of the tile's first register, no index mode around it; found with tools/gpu_area_diff.py)
                        accel_t a8[2];
#pragma unroll
                        for (int m = 0; m < 2; ++m) {
                            a8[m] = acc[m][tt][e];
                            asm volatile("" : "+v"(a8[m]));
                        }
#pragma unroll
                        for (int m = 0; m < 2; ++m) {
                            const uint32_t i = row0 + ri[m];
                            cand[m] = (i > j) && (i < n_snps);
                            if (prefilter) {
                                const double dn4 = __builtin_fma((double)a8[m], fk.nsc, -(ras[m] * c01.x));
                                double y;
                                if (aa.measure == LDX_MEASURE_RSQ) {
                                    y = (dn4 * (rqs[m] * c23.y)) * dn4;
                                } else {
                                    const bool neg = dn4 < 0.0;
                                    const double x = neg ? c01.y : c23.x, yy = neg ? c23.x : c01.y;
                                    y = __builtin_fabs(dn4) * max_raw(rra[m] * x, rrr[m] * yy);
                                }
                                cand[m] = cand[m] && (y >= kcand);
                            }
                        }
                        if (!__any(cand[0] || cand[1])) continue;   // wave-uniform: nothing near the threshold in these 128 pairs
                        if (aqn + 128u > kAreaQueue) drain();       // wave-uniform: room for this column tile's worst case
#pragma unroll
                        for (int m = 0; m < 2; ++m) {
                            const unsigned long long mask = __ballot(cand[m]);
                            if (cand[m]) {
                                const uint32_t pos = aqn + __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
                                aq_id[pos] = ((uint32_t)e << 9) | ((uint32_t)m << 8) | ((uint32_t)tt << 6) | lane;
                                aq_cnt[pos] = __builtin_bit_cast(uint32_t, a8[m]);
                            }
                            aqn += (uint32_t)__builtin_popcountll(mask);
                        }
                    }
                }
                drain();
                hit_slot = slot;
                hit_slot_end = slot_end;
              }
            };
            if constexpr (kArea) {
                area_epilogue();
                if (tid == 0) tickets[parity] = next_ticket;
                if (!(ablate & 16)) __builtin_amdgcn_s_setprio(0);
#ifdef LDX_TUNING
                LDX_STAMP(3);
                ++npass;
                if (my_stamps && lane == 0) {
                    my_stamps[3] = npass;
                    my_stamps[4] = __builtin_amdgcn_s_memrealtime();
                    my_stamps[5] = __builtin_amdgcn_s_memtime();
                }
#endif
                return;
            }
            // `inside`: the unit lies wholly below the diagonal, inside the panel and inside [u_begin, u_end) (no validity
            // tests); `clean`: and all of its 64 rows and 128 columns are ordinary SNPs (no degenerate handling either)
            const bool inside = !kRaw && row0 >= (t + 1u) * kSlab && row0 + 32u * MM <= n_snps && (t + 1u) * kSlab <= n_snps &&
                                vv * 8u >= u_begin && vv * 8u + 8u <= u_end;
            const bool all_ordinary = rows_ordinary && (cols_odd[0] | cols_odd[1]) == 0u;
            const bool clean = inside && all_ordinary;
            if constexpr (kF32Tier) {
                // The fp32 tier takes every `inside` unit: rows / columns of SNPs that are not ordinary park their lane-steps
                // (ldx_common.h, f32_row) and the drain runs the general fp64 variant for such a unit.  A unit that parks more
                // than the queue holds (four or more such SNPs) goes through the fp64 epilogue whole.
                if (inside && !(ablate & 512) && epilogue_f32(all_ordinary)) {
                } else if (clean && !(ablate & 512)) {
                    epilogue(std::true_type{});
                } else {
                    epilogue(std::false_type{});
                }
            } else {
                if (clean && !(ablate & 512)) epilogue(std::true_type{});
                else epilogue(std::false_type{});
            }
            if (tid == 0) tickets[parity] = next_ticket;
            if (!(ablate & 16)) __builtin_amdgcn_s_setprio(0);
#ifdef LDX_TUNING
            LDX_STAMP(3);
            ++npass;
            if (my_stamps && lane == 0) {
                my_stamps[3] = npass;
                my_stamps[4] = __builtin_amdgcn_s_memrealtime();
                my_stamps[5] = __builtin_amdgcn_s_memtime();
            }
#endif
        };
        if constexpr (kArea) {
            pass_body(std::integral_constant<int, 2>{});
        } else {
#ifdef LDX_MM1
            pass_body(std::integral_constant<int, 1>{});
#else
            if (short_pass) pass_body(std::integral_constant<int, 1>{});
            else pass_body(std::integral_constant<int, 2>{});
#endif
        }
    }
    if (kArea)   // the unused slots of this wave's last batch
        for (uint64_t sl = hit_slot + lane; sl < hit_slot_end; sl += 64u)
            if (sl < aa.hit_cap) aa.hits[sl].query = 0xFFFFFFFFu;
}

template <bool kRaw, bool kN11, bool kFp4, typename Cell>
static int launch_mfma(const void *alt, const double *fa, const double *fr, const double *q, uint32_t n_snps,
                       uint32_t n_hap, uint64_t unit_begin, uint64_t unit_end, Cell *out, ldx_ld64 *out_raw,
                       uint32_t *out_n11, uint32_t *sched, hipStream_t s)
{
    const uint32_t nch = n_chunks(n_hap);
    if ((uint64_t)n_slabs(n_snps) * nch * kSlab * 16u >= (1ull << 32)) {   // the K loop addresses the plane with 32-bit lane offsets
        set_error("ld_triangle on the matrix pipe: a bit plane of 4 GiB or more (%u SNPs x %u haplotypes)", n_snps, n_hap);
        return kNoMatrixPath;   // LDX_PATH_AUTO: the popcount kernel; an explicit matrix-pipe path: LDX_E_UNSUPPORTED
    }
    const size_t lds = mfma_lds_bytes(kStatRows, kFp4 && !kRaw && !kN11);
    if (lds > 64u * 1024u) {   // above 64 KiB the dynamic LDS size needs the opt-in attribute: once per device
        static std::atomic<uint64_t> opted{0};   // one bit per device ordinal, per instantiation
        int dev = 0;
        LDX_HIP(hipGetDevice(&dev));
        if (dev < 0 || dev >= 64 || !((opted.load(std::memory_order_relaxed) >> dev) & 1u)) {
            LDX_HIP(hipFuncSetAttribute((const void *)triangle_mfma_kernel<kRaw, kN11, false, kFp4, Cell>,
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            if (dev >= 0 && dev < 64) opted.fetch_or(1ull << dev, std::memory_order_relaxed);
        }
    }
    const int cus = device_cus();
    // the range of passes that intersect [unit_begin, unit_end)
    const uint32_t ns = n_slabs(n_snps);
    const uint64_t G64 = (uint64_t)ns * 2u;
    auto base64 = [&](uint64_t t) { return t * G64 - t * (t - 1u); };
    auto tile_of = [&](uint64_t v) {
        uint32_t lo = 0, hi = ns;
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) / 2;
            if (base64(mid) <= v) lo = mid; else hi = mid;
        }
        return lo;
    };
    const uint64_t v_begin = unit_begin / 8u, v_end = (unit_end + 7u) / 8u;
    const uint32_t t0 = tile_of(v_begin), t1 = tile_of(v_end - 1u);
    const uint32_t p_begin = mfma_pass_base(t0, ns) + (uint32_t)((v_begin - base64(t0)) / kMfmaWaves);
    const uint32_t p_end = mfma_pass_base(t1, ns) + (uint32_t)((v_end - base64(t1) + kMfmaWaves - 1u) / kMfmaWaves);
    // The last passes go out as two half-height tickets each (32-row tiles: ~0.55 of a pass time for half the work),
    // so that the launch does not end with whole passes on a few workgroups while the rest of the chip idles.
    const uint64_t slots = (uint64_t)cus * kWgPerCu;   // persistent: two workgroups (8 waves) per CU
    const uint32_t n_pass = p_end - p_begin;
    // Measured at 10 000 x 5008 (1580 passes on 512 workgroups): 128 halved passes -3 %, 256..384 -1 %, all +13 %.
    // A launch of at most half a round (tiny panels) halves every pass: -20 % at 1000-2000 SNPs.
    uint32_t n_short = 2ull * n_pass <= slots ? n_pass : (n_pass > slots ? (uint32_t)(slots / 4u) : 0u);
    const int forced = g_forced_short.load(std::memory_order_relaxed);   // tests / tuning: ldx_debug_force_short_passes
    if (forced >= 0) n_short = (uint32_t)forced < n_pass ? (uint32_t)forced : n_pass;
#ifdef LDX_MM1
    n_short = n_pass;
#endif
    uint64_t grid = slots;
    if (grid > (uint64_t)n_pass + n_short) grid = (uint64_t)n_pass + n_short;
    if (grid < 1) grid = 1;
    AreaArgs tri_args{};
    tri_args.f32 = f32_const((double)n_hap);
    int ablate = 0;
    unsigned long long *stamps = nullptr;
#ifdef LDX_TUNING
    ablate = getenv("LDX_ABLATE") ? atoi(getenv("LDX_ABLATE")) : 0;
    const char *stamp_file = getenv("LDX_STAMPS");
    const size_t stamp_words = (size_t)grid * kMfmaWaves * kStampStride;
    if (stamp_file) {
        LDX_HIP(hipMalloc(&stamps, stamp_words * 8));
        LDX_HIP(hipMemsetAsync(stamps, 0, stamp_words * 8, s));
    }
#endif
    triangle_mfma_kernel<kRaw, kN11, false, kFp4, Cell><<<(uint32_t)grid, kMfmaThreads, lds, s>>>(
        (const uint4 *)alt, fa, fr, q, n_snps, n_slabs(n_snps), nch, (double)n_hap, 1.0 / (double)n_hap, unit_begin,
        unit_end, out, out_raw, out_n11, p_begin, p_end, n_short, sched, ablate, stamps, tri_args);
    LDX_HIP(hipGetLastError());
#ifdef LDX_TUNING
    if (stamps) {   // tuning only: synchronous; the file holds the stamps of the LAST launch
        unsigned long long *h = (unsigned long long *)malloc(stamp_words * 8);
        LDX_HIP(hipStreamSynchronize(s));
        LDX_HIP(hipMemcpy(h, stamps, stamp_words * 8, hipMemcpyDeviceToHost));
        if (FILE *f = fopen(stamp_file, "wb")) {
            const unsigned long long hdr[4] = {grid, (unsigned long long)kMfmaWaves, kStampStride, kStampPasses};
            fwrite(hdr, 8, 4, f);
            fwrite(h, 8, stamp_words, f);
            fclose(f);
        }
        free(h);
        LDX_HIP(hipFree(stamps));
    }
#endif
    return LDX_OK;
}

size_t triangle_mfma_workspace_bytes() { return kTriWorkspaceBytes; }

int triangle_mfma(const void *alt, const double *fa, const double *fr, const double *q, uint32_t n_snps, uint32_t n_hap,
                  uint64_t unit_begin, uint64_t unit_end, int out_format, void *out, ldx_ld64 *out_raw, uint32_t *out_n11,
                  bool fp4, void *workspace, hipStream_t s)
{
    uint32_t *const sched = (uint32_t *)workspace;   // null: round-robin passes (no counters)
#define LDX_GO(R, N, CELL)                                                                                          \
    return fp4 ? launch_mfma<R, N, true, CELL>(alt, fa, fr, q, n_snps, n_hap, unit_begin, unit_end, (CELL *)out,    \
                                               out_raw, out_n11, sched, s)                                          \
               : launch_mfma<R, N, false, CELL>(alt, fa, fr, q, n_snps, n_hap, unit_begin, unit_end, (CELL *)out,   \
                                                out_raw, out_n11, sched, s)
    if (out_format == LDX_OUT_K16_RSQ || out_format == LDX_OUT_K16_DPRIME) {   // one measure, 2-byte cells: the FP4 kernel only
        if (!fp4) {
            set_error("the one-measure cell formats run on the FP4 or the popcount kernel, not on the int8 matrix kernel");
            return LDX_E_UNSUPPORTED;
        }
        if (out_format == LDX_OUT_K16_RSQ)
            return launch_mfma<false, false, true, ldx_k16r>(alt, fa, fr, q, n_snps, n_hap, unit_begin, unit_end, (ldx_k16r *)out,
                                                            nullptr, nullptr, sched, s);
        return launch_mfma<false, false, true, ldx_k16d>(alt, fa, fr, q, n_snps, n_hap, unit_begin, unit_end, (ldx_k16d *)out,
                                                        nullptr, nullptr, sched, s);
    }
    if (out_format == LDX_OUT_K16) {   // no unrounded output beside the 4-byte cells (ldx_triangle_ex_dev checks)
        if (out_n11) LDX_GO(false, true, ldx_k16);
        LDX_GO(false, false, ldx_k16);
    }
    if (out_raw && out_n11) LDX_GO(true, true, ldx_ld32);
    if (out_raw) LDX_GO(true, false, ldx_ld32);
    if (out_n11) LDX_GO(false, true, ldx_ld32);
    LDX_GO(false, false, ldx_ld32);
#undef LDX_GO
}

}  // namespace ldx

// ---- ld_area on the matrix pipe (banded use of the kernel above) --------------------------------------------
namespace ldx {

__global__ void area_mask_kernel(const uint32_t *__restrict__ queries, uint32_t n_query, uint8_t *__restrict__ is_query)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n_query) is_query[queries[k]] = 1;
}

// single workgroup: per j-tile the range of 64-row groups that can hold a hit, then the exclusive scan of the tiles' pass
// counts.  A pair (row i, column j), i > j, serves query i and query j; with the queries inside rows [qmin, qmax] (the
// ascending list's ends: a rank's share of a sharded scan is one contiguous range) a tile whose columns lie outside that
// range only needs the row groups inside it, and a tile whose columns intersect it needs its whole band.
__global__ void __launch_bounds__(1024) area_band_plan_kernel(const int64_t *__restrict__ pos, uint32_t n_snps, uint32_t T,
                                                              int64_t flank, const uint32_t *__restrict__ queries,
                                                              uint32_t n_query, uint32_t *__restrict__ g_begin,
                                                              uint32_t *__restrict__ g_end, uint32_t *__restrict__ pass_base,
                                                              unsigned long long *__restrict__ n_hits, uint32_t *__restrict__ order,
                                                              uint32_t *__restrict__ first_base, uint32_t *__restrict__ sched)
{
    const uint32_t qmin = queries[0], qmax = queries[n_query - 1u];
    __shared__ unsigned long long carry;
    __shared__ unsigned long long wsum[16];
    // the first position of every tile, in LDS (panels up to 4096 tiles = 524 288 SNPs): a tile's search for the end of its
    // band then takes ~log2(T) LDS steps + 7 dependent global loads instead of ~17 (the kernel is one workgroup in front
    // of the band kernel: 17 -> ~9 us at 100 000 SNPs)
    constexpr uint32_t kCoarse = 4096;
    __shared__ int64_t tile_pos[kCoarse];
    const bool coarse = T <= kCoarse;
    if (coarse)
        for (uint32_t k = threadIdx.x; k < T; k += 1024u) tile_pos[k] = pos[k * kSlab];
    if (threadIdx.x == 0) { carry = 0; pass_base[0] = 0; first_base[0] = 0; store_agent(n_hits, 0ull); }   // (the slot counter of the scan that follows: no memset node)
    // the band kernel's ticket counters (its own words of the workspace: sched[0], sched[1], one per XCD at 2 + 32 x)
    if (threadIdx.x < 256u) store_agent(&sched[threadIdx.x], 0u);
    block_sync();
    for (uint32_t t0 = 0; t0 < T; t0 += 1024u) {
        const uint32_t t = t0 + threadIdx.x;
        uint32_t cnt = 0;
        if (t < T) {
            const uint32_t jlast = ((t + 1u) * kSlab < n_snps ? (t + 1u) * kSlab : n_snps) - 1u;
            const int64_t lim = pos[jlast] + flank;            // rows with pos <= lim can pair with a column of the tile
            uint32_t lo = jlast + 1u, hi = n_snps;             // first row index with pos > lim
            if (coarse) {   // smallest tile k > t whose first position exceeds lim: the answer lies in ((k - 1) 128, k 128]
                uint32_t klo = t + 1u, khi = T;
                while (klo < khi) { const uint32_t m = (klo + khi) / 2; if (tile_pos[m] > lim) khi = m; else klo = m + 1u; }
                if (klo < T) hi = klo * kSlab;                  // pos[klo * 128] > lim
                if (klo > t + 1u) lo = (klo - 1u) * kSlab + 1u;    // pos[(klo - 1) * 128] <= lim
            }
            while (lo < hi) { const uint32_t m = (lo + hi) / 2; if (pos[m] > lim) hi = m; else lo = m + 1u; }
            uint32_t ge = (lo + kRows64 - 1u) / kRows64;   // lo rows -> groups
            uint32_t gb = 2u * t;
            if (t * kSlab > qmax || jlast < qmin) {   // no query among the tile's columns: only rows that are queries matter
                const uint32_t qb = qmin / kRows64, qe = qmax / kRows64 + 1u;
                gb = gb > qb ? gb : qb;
                ge = ge < qe ? ge : qe;
            }
            if (ge < gb) ge = gb;
            g_begin[t] = gb;
            g_end[t] = ge;
            cnt = (ge - gb + kMfmaWaves - 1u) / kMfmaWaves;
        }
        // two prefix sums in one 64-bit scan: passes (low word) and tiles that have at least one pass (high word)
        unsigned long long x = (unsigned long long)cnt | ((unsigned long long)(cnt != 0u) << 32);
        const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const unsigned long long y = __shfl_up(x, off); if (lane >= (uint32_t)off) x += y; }
        if (lane == 63) wsum[wv] = x;
        block_sync();
        unsigned long long pre = 0;
        for (uint32_t k = 0; k < wv; ++k) pre += wsum[k];
        const unsigned long long incl = carry + pre + x;
        if (t < T) {
            pass_base[t + 1u] = (uint32_t)incl;
            first_base[t + 1u] = (uint32_t)(incl >> 32);
        }
        block_sync();
        if (threadIdx.x == 1023) carry = incl;
        block_sync();
    }
    // Ticket order (ticket -> pass).  The kernel hands the passes out per XCD: eight contiguous ranges of TILES, walked in
    // position order for the L2s.  A tile's FIRST pass holds its diagonal units -- where the hits are, i.e. where the full
    // fp64 epilogue runs (stamps: epilogue ~38k cycles against 11k for the other passes) -- and in plain tile order a range
    // ended with such passes while the rest of the chip idled.  So a tile's first pass is handed out EARLY: kLead tiles
    // ahead of the tile whose other passes are being handed out (one sweep per range: the formula is at the loop below);
    // round 4's order -- all first passes of a range, then all the others: two sweeps -- is kept under LDX_AB_FIRST_FIRST.
    if (!order) return;   // (a panel whose full triangle has more than kOrderCap passes: plain tile order)
    __threadfence_block();
    block_sync();
    for (uint32_t t = threadIdx.x; t < T; t += 1024u) {
        uint32_t a = 0, b = T;
        for (uint32_t xr = 0; xr < 8u; ++xr) {
            const uint32_t lo = (uint32_t)((uint64_t)T * xr / 8u), hi = (uint32_t)((uint64_t)T * (xr + 1u) / 8u);
            if (t >= lo && t < hi) { a = lo; b = hi; }
        }
        const uint32_t pb = pass_base[t], cnt = pass_base[t + 1u] - pb;
        if (!cnt) continue;
        const uint32_t r0 = pass_base[a];
#ifdef LDX_AB_FIRST_FIRST   // round 4's order: all first passes of the range, then all the others (each range swept twice)
        const uint32_t nfirst = first_base[b] - first_base[a], kf = first_base[t] - first_base[a];
        order[r0 + kf] = kAreaDecoded | (t << 12);
        const uint32_t lt = (pb - r0) - kf;   // passes other than first ones before this tile, inside the range
        for (uint32_t i = 1; i < cnt; ++i) order[r0 + nfirst + lt + (i - 1u)] = kAreaDecoded | (t << 12) | i;
#else
        // Round 5: ONE sweep per range.  A tile's first pass is still handed out early -- kLead tiles ahead of the tile whose
        // other passes are being handed out, i.e. while the rows of ITS diagonal are inside the band that is streaming
        // through the XCD's L2 anyway (a +-flank window is ~kLead tiles high at the bench's geometry) -- and the range still
        // ends with short items (the last kLead tiles' other passes).  Sequence: the first passes of tiles [a, a + kLead);
        // then for c = a, a + 1, ...: the first pass of tile c + kLead, the other passes of tile c.  With F(t) / L(t) = first /
        // other passes of the range before tile t:  first pass of t at F(t) + L(max(t - kLead, a)); other pass i of t at
        // F(min(t + kLead + 1, b)) + L(t) + i - 1.
        constexpr uint32_t kLead = 8;
        auto F = [&](uint32_t tt) { return first_base[tt] - first_base[a]; };
        auto L = [&](uint32_t tt) { return (pass_base[tt] - r0) - F(tt); };
        // (entries carry the tile and the pass inside it, kAreaDecoded: the band kernel need not search pass_base for them)
        order[r0 + F(t) + L(t >= a + kLead ? t - kLead : a)] = kAreaDecoded | (t << 12);
        const uint32_t ahead = t + kLead + 1u < b ? t + kLead + 1u : b;
        for (uint32_t i = 1; i < cnt; ++i) order[r0 + F(ahead) + L(t) + (i - 1u)] = kAreaDecoded | (t << 12) | i;
#endif
    }
}

// room for the band's ticket order: one word per pass of the FULL triangle (what a band can need at most), for panels up
// to ~512 000 SNPs; beyond that the band keeps plain tile order
constexpr size_t kOrderCap = 4u << 20;
constexpr size_t kAreaSchedWords = 256;   // the band's ticket counters inside the workspace: [0], [1], [2 + 32 x] for x < 8
static size_t area_order_entries(uint32_t n_snps)
{
    const uint32_t T = n_slabs(n_snps);
    const size_t worst = mfma_pass_base(T, T);
    return (worst <= kOrderCap && T < 4096u) ? worst : 0u;   // (T < 4096: an order entry packs tile and pass-in-tile in 12 bits each)
}

size_t area_mfma_workspace_bytes(uint32_t n_snps)
{
    const size_t T = n_slabs(n_snps);
    return ((size_t)n_snps + 255u) / 256u * 256u + 2u * (((T + 1u) * 4u + 255u) / 256u * 256u) + 2u * ((T * 4u + 255u) / 256u * 256u) +
           (area_order_entries(n_snps) * 4u + 255u) / 256u * 256u + kAreaSchedWords * 4u;
}

int area_mfma(const void *alt, const double *fa, const double *fr, const double *q, uint32_t n_snps, uint32_t n_hap,
              const int64_t *positions, const uint32_t *queries, uint32_t n_query, int64_t flank, int measure, double thres,
              ldx_hit *hits, uint64_t hit_cap, uint64_t *n_hits, uint32_t *query_counts, void *workspace, bool fp4, hipStream_t s)
{
    const uint32_t T = n_slabs(n_snps), nch = n_chunks(n_hap);
    if ((uint64_t)T * nch * kSlab * 16u >= (1ull << 32)) {   // the K loop addresses the plane with 32-bit lane offsets
        set_error("ld_area on the matrix pipe: a bit plane of 4 GiB or more (%u SNPs x %u haplotypes)", n_snps, n_hap);
        return kNoMatrixPath;
    }
    char *w = (char *)workspace;
    uint8_t *is_query = (uint8_t *)w;
    w += ((size_t)n_snps + 255u) / 256u * 256u;
    uint32_t *pass_base = (uint32_t *)w;
    w += (((size_t)T + 1u) * 4u + 255u) / 256u * 256u;
    uint32_t *g_end = (uint32_t *)w;
    w += ((size_t)T * 4u + 255u) / 256u * 256u;
    uint32_t *g_begin = (uint32_t *)w;
    w += ((size_t)T * 4u + 255u) / 256u * 256u;
    uint32_t *first_base = (uint32_t *)w;   // [T + 1] tiles with at least one pass before tile t
    w += (((size_t)T + 1u) * 4u + 255u) / 256u * 256u;
    uint32_t *order = area_order_entries(n_snps) ? (uint32_t *)w : nullptr;
    w += (area_order_entries(n_snps) * 4u + 255u) / 256u * 256u;
    // The band's ticket counters: 256 words of THIS call's workspace, zeroed by the plan kernel in front of the band kernel
    // (round 5, ADVICE r04: they used to be the (device, stream) slot of g_sched, and every ld_area plan graph -- captured
    // on torch's one shared capture stream -- had the same slot baked in: two plans replayed on two streams at once shared
    // their counters).  Concurrent scans need distinct workspaces anyway.
    uint32_t *sched = (uint32_t *)w;
    if (n_query == n_snps) {   // ascending distinct rows: every SNP is a query -- no mask at all
        is_query = nullptr;
    } else {
        LDX_HIP(hipMemsetAsync(is_query, 0, n_snps, s));
        area_mask_kernel<<<(n_query + 255u) / 256u, 256, 0, s>>>(queries, n_query, is_query);
        LDX_HIP(hipGetLastError());
    }
    area_band_plan_kernel<<<1, 1024, 0, s>>>(positions, n_snps, T, flank, queries, n_query, g_begin, g_end, pass_base,
                                             (unsigned long long *)n_hits, order, first_base, sched);
    LDX_HIP(hipGetLastError());
    const size_t lds = mfma_lds_bytes(kRows64, false, true);
    {   // above 64 KiB the dynamic LDS size needs the opt-in attribute: once per device
        static std::atomic<uint64_t> opted{0};
        int dev = 0;
        LDX_HIP(hipGetDevice(&dev));
        if (dev < 0 || dev >= 64 || !((opted.load(std::memory_order_relaxed) >> dev) & 1u)) {
            LDX_HIP(hipFuncSetAttribute((const void *)triangle_mfma_kernel<false, false, true, true>,
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            LDX_HIP(hipFuncSetAttribute((const void *)triangle_mfma_kernel<false, false, true, false>,
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            if (dev >= 0 && dev < 64) opted.fetch_or(1ull << dev, std::memory_order_relaxed);
        }
    }
    const int cus = device_cus();
    AreaArgs aa{};
    aa.f32 = f32_const((double)n_hap);
    aa.pos = positions;
    aa.is_query = is_query;
    aa.pass_base = pass_base;
    aa.g_begin = g_begin;
    aa.g_end = g_end;
    aa.order = order;
    aa.hits = hits;
    aa.counts = query_counts;
    aa.n_hits = (unsigned long long *)n_hits;
    aa.hit_cap = hit_cap;
    aa.flank = (double)flank;
    aa.k_thres = thres_to_k(thres);
    aa.measure = measure;
    const uint64_t units = ldx_triangle_units(n_snps) / 8u;   // 64-row units of the full triangle
    unsigned long long *stamps = nullptr;
#ifdef LDX_TUNING   // in-kernel stamps of the band (env LDX_STAMPS=file), as in launch_mfma
    const char *stamp_file = getenv("LDX_STAMPS");
    const size_t stamp_words = (size_t)cus * 2u * kMfmaWaves * kStampStride;
    if (stamp_file) {
        LDX_HIP(hipMalloc(&stamps, stamp_words * 8));
        LDX_HIP(hipMemsetAsync(stamps, 0, stamp_words * 8, s));
    }
#endif
    if (fp4)
        triangle_mfma_kernel<false, false, true, true><<<(uint32_t)cus * 2u, kMfmaThreads, lds, s>>>(
            (const uint4 *)alt, fa, fr, q, n_snps, T, nch, (double)n_hap, 1.0 / (double)n_hap, 0, units * 8u,
            (ldx_ld32 *)nullptr, nullptr, nullptr, 0u, 0u, 0u, sched, 0, stamps, aa);
    else
        triangle_mfma_kernel<false, false, true, false><<<(uint32_t)cus * 2u, kMfmaThreads, lds, s>>>(
            (const uint4 *)alt, fa, fr, q, n_snps, T, nch, (double)n_hap, 1.0 / (double)n_hap, 0, units * 8u,
            (ldx_ld32 *)nullptr, nullptr, nullptr, 0u, 0u, 0u, sched, 0, nullptr, aa);
    LDX_HIP(hipGetLastError());
#ifdef LDX_TUNING
    if (stamps) {   // tuning only: synchronous; the file holds the stamps of the LAST launch
        unsigned long long *h = (unsigned long long *)malloc(stamp_words * 8);
        LDX_HIP(hipStreamSynchronize(s));
        LDX_HIP(hipMemcpy(h, stamps, stamp_words * 8, hipMemcpyDeviceToHost));
        if (FILE *f = fopen(stamp_file, "wb")) {
            const unsigned long long hdr[4] = {(unsigned long long)cus * 2u, (unsigned long long)kMfmaWaves, kStampStride, kStampPasses};
            fwrite(hdr, 8, 4, f);
            fwrite(h, 8, stamp_words, f);
            fclose(f);
        }
        free(h);
        LDX_HIP(hipFree(stamps));
    }
#endif
    return LDX_OK;
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

extern "C" int ldx_debug_force_short_passes(int n_short)
{
    ldx::g_forced_short.store(n_short < 0 ? -1 : n_short, std::memory_order_relaxed);
    return LDX_OK;
}

extern "C" int ldx_debug_counters(uint64_t out[8], int reset)
{
    LDX_REQUIRE(out, "null pointer");
    LDX_HIP(hipDeviceSynchronize());
    LDX_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(ldx::g_dbg), 8 * sizeof(uint64_t)));
    if (reset) {
        const uint64_t zero[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        LDX_HIP(hipMemcpyToSymbol(HIP_SYMBOL(ldx::g_dbg), zero, sizeof(zero)));
    }
    return LDX_OK;
}
