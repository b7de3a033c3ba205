// Shared definitions for the ldx HIP sources (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include <type_traits>

#include "../../include/ldx.h"

namespace ldx {

constexpr uint32_t kSlab = LDX_SLAB_ROWS;     // 128 SNP rows per slab / j-tile
constexpr uint32_t kGroup = LDX_GROUP_ROWS;   // 8 SNP rows per wave unit
constexpr uint32_t kGroupsPerSlab = kSlab / kGroup;   // 16

void set_error(const char *fmt, ...);

// Workgroup barrier that first drains THIS wave's LDS traffic.  Symptom it removes: about once in 10^6 barriers a
// ds_write issued just before the barrier (the pass ticket, by wave 0) was not yet visible to the ds_read another wave
// issued right behind it -- that wave then redid an old pass and left its own undone (tools/gpu_soak.py).
// This is an EMPIRICAL fix: with it the symptom has not recurred in 6 100+ soak launches; the mechanism is unproven.
// (profiles/r03/syncthreads_isa_evidence.txt shows that hipcc puts s_waitcnt lgkmcnt(3), not 0, in front of the
// top-of-pass s_barrier of triangle_mfma_kernel; a non-zero count normally still covers the older ds_write, so that
// listing is context, not a root cause.)  The explicit wait costs nothing measurable.  Every barrier in these sources
// goes through here.
__device__ __forceinline__ void block_sync()
{
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();
}

// A word that OTHER launches touch with atomics is only ever written with atomics: agent-scope stores go through to the
// memory side, where the eight XCDs' atomics meet -- a plain store stays in the storing XCD's L2 until a release writes it
// back.  Round 5 found the ticket counters re-armed with plain stores at the end of a launch still EXHAUSTED for the next
// launch's atomicAdd: inside a HIP graph whose consecutive kernel nodes share no buffer argument that either writes (two
// ld_triangle launches into alternating result buffers; the counters live in a __device__ array no argument tracking
// sees) the runtime chains the nodes without the cache write-back a stream gives -- every launch after the first drew no
// tickets beyond its static ones and returned a third of the triangle (tools/gpu_streams_dbg.py, profiles/r05/).
template <typename T>
__device__ __forceinline__ void store_agent(T *p, T v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Smallest integer k >= 0 with (double)k / 1e4 >= thres: "rounded value >= thres" (ld_area.py:248,
// ld_triangle.py:224) becomes the exact integer test k >= thres_to_k(thres).
double thres_to_k(double thres);

// internal return code of the two matrix-pipe entries below: the matrix kernel cannot take this panel (a bit plane of 4 GiB
// or more: its K loop addresses the plane with 32-bit lane offsets).  LDX_PATH_AUTO callers fall back to the popcount
// kernels (identical results), explicit paths report LDX_E_UNSUPPORTED (the message is set).
constexpr int kNoMatrixPath = -1000;

// ld_triangle on the matrix cores (ldx_mfma.hip); same contract as ldx_triangle_ex_dev after argument checks.
// workspace: triangle_mfma_workspace_bytes() zeroed bytes holding the pass scheduler's ticket counters, or null (round-robin)
size_t triangle_mfma_workspace_bytes();
int triangle_mfma(const void *alt, const double *fa, const double *fr, const double *q, uint32_t n_snps, uint32_t n_hap,
                  uint64_t unit_begin, uint64_t unit_end, int out_format, void *out, ldx_ld64 *out_raw, uint32_t *out_n11,
                  bool fp4, void *workspace, hipStream_t s);

// ld_area on the matrix pipe (ldx_mfma.hip): all (query, opposing) pairs inside the +-flank band through the MFMA
// kernel; same hit contract as the popcount scan of ldx_area.hip
size_t area_mfma_workspace_bytes(uint32_t n_snps);
int area_mfma(const void *alt, const double *fa, const double *fr, const double *q, uint32_t n_snps, uint32_t n_hap,
              const int64_t *positions, const uint32_t *queries, uint32_t n_query, int64_t flank, int measure, double thres,
              ldx_hit *hits, uint64_t hit_cap, uint64_t *n_hits, uint32_t *query_counts, void *workspace, bool fp4,
              hipStream_t s);

#define LDX_HIP(call)                                                                       \
    do {                                                                                    \
        hipError_t e_ = (call);                                                             \
        if (e_ != hipSuccess) {                                                             \
            ::ldx::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, \
                             __LINE__);                                                     \
            return LDX_E_HIP;                                                               \
        }                                                                                   \
    } while (0)

#define LDX_REQUIRE(cond, msg)                                     \
    do {                                                           \
        if (!(cond)) {                                             \
            ::ldx::set_error("%s: %s", __func__, msg);             \
            return LDX_E_ARG;                                      \
        }                                                          \
    } while (0)

__host__ __device__ inline uint32_t n_slabs(uint32_t n_snps) { return (n_snps + kSlab - 1) / kSlab; }
// chunks of 128 haplotypes per row, allocated in PAIRS (256 haplotypes): the FP4 matrix kernel consumes two chunks
// per K-block (one per lane half), so a row always holds an even number of chunks; pad chunks are zero bits
__host__ __device__ inline uint32_t n_chunks(uint32_t n_hap) { return (n_hap + 255u) / 256u * 2u; }

// first unit of j-tile t: t*G - 8*t*(t-1), G = groups in the padded panel
__host__ __device__ inline uint64_t tile_base(uint64_t t, uint64_t G) { return t * G - 8u * t * (t - 1u); }

// ---- the epilogue: calc_ld.py:33-97 mirrored op for op in fp64 (compile with -ffp-contract=off) ----
// Inputs: f11 = n11/n, and the per-SNP frequencies fa = a/n, fr = r/n, q1 = fa1*fr1.
// var_1 is the row / query, var_2 the column / opposing variant.
struct LdRaw {
    double rsq, dprime;
    uint32_t flags;
};

__device__ inline LdRaw ld_epilogue(double f11, double fa1, double fr1, double q1, double fa2, double fr2)
{
    LdRaw o;
    const double p = fa1 * fa2;
    const double d = f11 - p;                       // calc_ld.py:50 (product rounded, then the difference)
    const double m1 = fa1 * fr2, m2 = fr1 * fa2;    // :64-65
    const double dmax = m1 < m2 ? m1 : m2;
    const double m3 = fr1 * fr2;                    // :71-72  max(-p, -m3) == -min(p, m3)
    const double dmin = -(p < m3 ? p : m3);
    const double bound = d >= 0.0 ? dmax : dmin;
    o.flags = 0;
    if (bound == 0.0) {                             // :68-69, :75-76  ZeroDivisionError -> int 0
        o.dprime = 0.0;
        o.flags |= LDX_FLAG_DPRIME_INT0;
    } else {
        o.dprime = d / bound;
    }
    if (o.dprime != 0.0) {                          // :86-88  (q1*fa2)*fr2 == ((fa1*fr1)*fa2)*fr2
        o.rsq = (d * d) / ((q1 * fa2) * fr2);
    } else {                                        // :89-90
        o.rsq = 0.0;
        o.flags |= LDX_FLAG_RSQ_INT0;
    }
    return o;
}

// Python round(x, 4) for finite x >= 0 as the integer k with result == k / 10^4 (calc_ld.py:94-97).
// x*1e4 == y + e exactly; see oracle/ld_oracle.py:round4 for why deciding on frac(y), then e, is exact.
__device__ inline double round4_k(double x)
{
    const double y = x * 1e4;
    const double e = __builtin_fma(x, 1e4, -y);
    double k = __builtin_floor(y);
    const double f = y - k;
    const bool odd = (k - 2.0 * __builtin_floor(k * 0.5)) != 0.0;
    const bool up = (f > 0.5) || (f == 0.5 && (e > 0.0 || (e == 0.0 && odd)));
    return up ? k + 1.0 : k;
}

// ---- result cells.  A rounded result is the integer k = round(x, 4) * 10^4 (a double: without missing codes k <= 10^4,
// with them D' and r^2 have no upper bound) plus the int-0 mark.  Two cell formats carry it (include/ldx.h):
//   ldx_ld32: two float32 nearest to k / 10^4, -0.0f = int 0; identifies k while k < kBig32 (value < 1024, float32 ulp
//             below 10^-4); larger values are stored as the quiet NaN LDX_LD32_BIG ("fetch the exact value": ldx_ld_pairs_dev)
//   ldx_k16:  two uint16: k in bits 0..14, bit 15 = int 0; k >= 32767 is stored as the escape LDX_K16_BIG (0x7FFF)
struct LdK {
    double kr, kd;      // k of r_square, d_prime
    uint32_t flags;     // LDX_FLAG_RSQ_INT0 | LDX_FLAG_DPRIME_INT0
};
constexpr double kBig32 = 1.024e7;

template <typename Cell>
__device__ __forceinline__ Cell encode_cell(double kr, double kd, bool r_int0, bool d_int0);

// the one-measure cells of include/ldx.h (ldx_k16one) as two types, so that the kernels know at compile time which value
// to compute: the r_square half / the d_prime half of ldx_k16
struct ldx_k16r { uint16_t value; };
struct ldx_k16d { uint16_t value; };
template <typename Cell> struct cell_measure { static constexpr int value = -1; };                  // both values
template <> struct cell_measure<ldx_k16r> { static constexpr int value = LDX_MEASURE_RSQ; };
template <> struct cell_measure<ldx_k16d> { static constexpr int value = LDX_MEASURE_DPRIME; };

template <>
__device__ __forceinline__ ldx_ld32 encode_cell<ldx_ld32>(double kr, double kd, bool r_int0, bool d_int0)
{
    ldx_ld32 o;
    o.r_square = r_int0 ? -0.0f : (kr < kBig32 ? (float)(kr * 1e-4) : __uint_as_float(LDX_LD32_BIG_BITS));
    o.d_prime = d_int0 ? -0.0f : (kd < kBig32 ? (float)(kd * 1e-4) : __uint_as_float(LDX_LD32_BIG_BITS));
    return o;
}

template <>
__device__ __forceinline__ ldx_k16 encode_cell<ldx_k16>(double kr, double kd, bool r_int0, bool d_int0)
{
    ldx_k16 o;
    o.r_square = r_int0 ? (uint16_t)LDX_K16_INT0 : (kr < 32767.0 ? (uint16_t)(uint32_t)kr : (uint16_t)LDX_K16_BIG);
    o.d_prime = d_int0 ? (uint16_t)LDX_K16_INT0 : (kd < 32767.0 ? (uint16_t)(uint32_t)kd : (uint16_t)LDX_K16_BIG);
    return o;
}

template <>
__device__ __forceinline__ ldx_k16r encode_cell<ldx_k16r>(double kr, double, bool r_int0, bool)
{
    return ldx_k16r{r_int0 ? (uint16_t)LDX_K16_INT0 : (kr < 32767.0 ? (uint16_t)(uint32_t)kr : (uint16_t)LDX_K16_BIG)};
}

template <>
__device__ __forceinline__ ldx_k16d encode_cell<ldx_k16d>(double, double kd, bool, bool d_int0)
{
    return ldx_k16d{d_int0 ? (uint16_t)LDX_K16_INT0 : (kd < 32767.0 ? (uint16_t)(uint32_t)kd : (uint16_t)LDX_K16_BIG)};
}

// element of cell (row % 8, column % 128) inside its unit, in the order of the cell format (include/ldx.h)
template <typename Cell>
__host__ __device__ __forceinline__ uint32_t cell_offset(uint32_t r8, uint32_t c)
{
    if constexpr (sizeof(Cell) <= 4) return LDX_CELL_OFFSET4(r8, c);
    else return LDX_CELL_OFFSET8(r8, c);
}

template <typename Cell>
__device__ __forceinline__ Cell zero_cell();   // a cell outside the triangle (row <= col, pad rows): the template's 0
template <>
__device__ __forceinline__ ldx_ld32 zero_cell<ldx_ld32>() { return ldx_ld32{0.0f, 0.0f}; }
template <>
__device__ __forceinline__ ldx_k16 zero_cell<ldx_k16>() { return ldx_k16{0, 0}; }
template <>
__device__ __forceinline__ ldx_k16r zero_cell<ldx_k16r>() { return ldx_k16r{0}; }
template <>
__device__ __forceinline__ ldx_k16d zero_cell<ldx_k16d>() { return ldx_k16d{0}; }

template <typename Cell>
__device__ __forceinline__ Cell encode_cell(const LdK &k)
{
    return encode_cell<Cell>(k.kr, k.kd, (k.flags & LDX_FLAG_RSQ_INT0) != 0, (k.flags & LDX_FLAG_DPRIME_INT0) != 0);
}

__device__ inline LdK round_pair(const LdRaw &r)
{
    return LdK{round4_k(r.rsq), round4_k(r.dprime), r.flags};
}

// The mirror as one out-of-line call: the fallback of the fast epilogues (rare) must not be inlined 16-128 times.
static __device__ __noinline__ LdK ld_pair_mirror(double f11, double fa1, double fr1, double q1, double fa2, double fr2)
{
    return round_pair(ld_epilogue(f11, fa1, fr1, q1, fa2, fr2));
}

// host: true iff div_by_n(c, n, 1/n) == c / n for every integer c in [0, n] (always, by Markstein's theorem;
// checked anyway because the parity claim rests on it)
bool check_recip(uint32_t n);

// n11 / n exactly as one IEEE division (calc_ld.py:33) in three operations: q = c*rn, r = c - q*n (exact,
// fma), q' = q + r*rn.  With rn = RN(1/n) this is the correctly rounded quotient (Markstein); the host checks
// it against real divisions for every c in [0, n] before the first launch with a given n (ldx_check_recip).
__device__ __forceinline__ double div_by_n(double c, double n, double rn)
{
    const double q = c * rn;
    const double r = __builtin_fma(-q, n, c);
    return __builtin_fma(r, rn, q);
}

// ---- fast epilogue for the 8-byte/pair output: same k = round(x, 4) * 10^4 and same int-0 marks as the
// mirror above, at roughly half the instructions.
//   * d, the sign branch and `bound` are computed exactly as the mirror does (they decide the int-0 marks
//     and which bound applies);
//   * the two quotients D' = d / bound and r^2 = d*d / den share ONE reciprocal 1/(bound*den) (v_rcp_f64 + two
//     Newton steps): relative error ~1e-15 instead of correctly rounded;
//   * y = x*1e4 is rounded with rint; the mirror's value differs from ours by < 1e-8 in y, so both round to
//     the same integer unless y is within 1e-6 of a half-integer (or absurdly large): only then `slow` is
//     set and the caller recomputes that pair with ld_pair_mirror.  Exact ties (e.g. D' = 27/32) always
//     take the slow path, so the reference's tie behaviour (decided by its own rounding errors) is kept.
__device__ __forceinline__ LdK ld_pair_fast(double f11, double fa1, double fr1, double q1, double fa2,
                                            double fr2, bool &slow)
{
    const double p = fa1 * fa2;
    const double d = f11 - p;
    const double dmax = __builtin_fmin(fa1 * fr2, fr1 * fa2);   // no NaNs here: fmin == the mirror's compare-select
    const double dmin = __builtin_fmin(p, fr1 * fr2);
    const double bound = d >= 0.0 ? dmax : -dmin;            // signed like the mirror's
    const double den = (q1 * fa2) * fr2;
    // degenerate (a monomorphic variant): bound == 0 -> t == 0 -> r = inf -> y = NaN; NaN compares false
    // below and the two results are replaced by the int-0 mark at the end, so no select is needed here.
    // bound != 0 implies all four frequencies != 0, hence den != 0.
    const bool degenerate = bound == 0.0;
    const double t = bound * den;
    double r = __builtin_amdgcn_rcp(t);
    r = __builtin_fma(r, __builtin_fma(-t, r, 1.0), r);
    r = __builtin_fma(r, __builtin_fma(-t, r, 1.0), r);
    const double r4 = r * 1e4;
    const double yd = d * (den * r4);                        // D' * 10^4   (den*r = 1/bound, sign included)
    const double yr = (d * d) * (bound * r4);                // r^2 * 10^4  (bound*r = 1/den > 0)
    const double kd = __builtin_rint(yd), kr = __builtin_rint(yr);
    const bool sure = __builtin_fabs(yd - kd) < 0.499999 && __builtin_fabs(yr - kr) < 0.499999 &&
                      __builtin_fmax(yd, yr) < 1e9;
    slow = !(sure || degenerate);
    LdK o;
    o.kr = kr;
    o.kd = kd;
    // d_prime == 0 <=> d == 0 when bound != 0
    o.flags = (degenerate ? LDX_FLAG_DPRIME_INT0 : 0u) | ((degenerate || d == 0.0) ? LDX_FLAG_RSQ_INT0 : 0u);
    return o;
}

// ---- count-domain fast epilogue (used by the fused pair kernels for the 8-byte/pair output) ---------------
// Same contract as ld_pair_fast -- k = round(x, 4) * 10^4 and the int-0 marks of the mirror, or `slow` -- but
// computed from the EXACT integer  Dn = n*n11 - a1*a2  (d = Dn / n^2) and per-SNP reciprocals, so a pair costs
// ~33 VALU instructions with no division, no reciprocal instruction and no Newton step:
//     D'  * 10^4 = 10^4 |Dn| / B,   B = Dn >= 0 ? min(a1 r2, r1 a2) : min(a1 a2, r1 r2)   (calc_ld.py:63-76)
//                                   1/B = max of the two products of per-SNP reciprocals
//     r^2 * 10^4 = 10^4 Dn^2 / (a1 r1 a2 r2)                                              (calc_ld.py:86-88)
// Why this may replace the op-for-op mirror: the reference's own value differs from the exact rational one by
// its rounding errors, dominated by the cancellation in d = f11 - fa1*fa2 (|delta d| <= 4.5e-16: f11 and the
// product are each rounded below 1).  In units of y = value * 10^4 that is  delta_d * n^2 * 10^4 / B  for D' and
// 2 * delta_d * n^2 * y_r / |Dn|  for r^2; every other rounding (ours and the reference's) is < 1e-8 for
// y < 1e7.  So when y is farther from a half-integer than 1e-6 + that bound, both round to the same k.  Pairs
// that are not `sure` -- exact ties such as D' = 27/32, tiny |Dn| with tiny B, Dn == 0 (the reference's d may be
// a rounding residue instead of 0 and that decides its int-0 mark), y >= 1e7 -- are recomputed by the caller
// with ld_pair_mirror.  Degenerate pairs (a count of 0: both of the reference's bounds are 0, calc_ld.py:68-69,
// 75-76, 89-90) have 1/B = inf and get the int-0 marks directly.
struct FastRow {   // per var_1 (row / query):  1e4 * a,  1/a,  1/r,  1e-4 / (a r)
    double a_s, ra, rr, rq_s;
};
struct FastCol {   // per var_2 (column / opposing):  a,  1/a,  1/r,  1 / (a r)
    double a, ra, rr, rq;
};
struct FastConst {   // per launch
    double nsc;        // 1e4 * n / count_scale   (count_scale = 8 on the MFMA path: its accumulators hold 8 * n11)
    double ncd, ncr;   // minus the margins per unit of 1/B and of |z| (see fast_const)
};

__host__ __device__ inline FastConst fast_const(double n, double count_scale)
{
    FastConst c;
    c.nsc = 1e4 * n / count_scale;   // exact: count_scale is 1 or 8 and n < 2^32
    c.ncd = -6e-12 * n * n;          // |.| >= 1e4 * 4.5e-16 * n^2, with a third to spare
    c.ncr = 2.0 * c.ncd;
    return c;
}

// v_max_f64 as is: __builtin_fmax adds a canonicalising v_max x, x per operand (the operands here are never sNaN)
__device__ __forceinline__ double max_raw(double a, double b)
{
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// SNP classes of the fast epilogue tiers (round 6: VERDICT r05 item 2 -- panels that are not all complete and polymorphic):
//   kSnpOrdinary    polymorphic (a > 0, r > 0) with FEW missing codes: m = n - a - r <= r / 8 (m = 0: rounds 1-5's "ordinary").
//                   Every bound the fast tiers use still holds: with missing codes D' and r^2 are no longer <= 1, but
//                       |Dn| <= min(a1, a2) (n - max(a1, a2))   for Dn > 0   (n11 <= min(a1, a2))
//                       |Dn| <= min(a1 a2, (r1 + m1)(r2 + m2))  for Dn < 0   (n11 >= a1 + a2 - n)
//                   give  D', r^2 <= (1 + m1 / r1)(1 + m2 / r2) <= (9/8)^2 < 1.27  in both sign branches (B = the smaller
//                   product of calc_ld.py:63-76 in each case), so y = 10^4 value < 12 700 fits the 15-bit cell and the
//                   kClean variants' dropped guards (y < 10^7, k < 32767, no degenerate operand); a r >= (n - 1) / 2 replaces
//                   a r >= n - 1 in the r^2 share of the reference's own error (f32_const doubles that term and more);
//   kSnpDegenerate  a == 0 or r == 0 (whatever is missing): calc_ld.py returns the int 0 for BOTH values against ANY other
//                   SNP, whatever n11 is (:66-69 / :73-76: the bound is 0 or -0.0 -> ZeroDivisionError -> d_prime = 0 ->
//                   :89-90 r_square = 0) -- and the alt/alt count of such a SNP with any SNP X is 0 (a == 0) or at most a_X,
//                   which lets the fp32 tier keep such rows / columns on its common path (f32_row below);
//   kSnpOdd         everything else (polymorphic with many missing codes): parks, as every non-ordinary SNP did before.
constexpr int kSnpOrdinary = 0, kSnpDegenerate = 1, kSnpOdd = 2;
__device__ __forceinline__ int snp_class(double fa, double fr, double n)
{
    const double a = __builtin_rint(fa * n), r = __builtin_rint(fr * n);
#ifdef LDX_AB_R5CLASS   // tuning: rounds 1-5's rule -- ordinary iff polymorphic and complete, everything else parks
    return (a > 0.0 && r > 0.0 && a + r == n) ? kSnpOrdinary : kSnpOdd;
#endif
    if (a == 0.0 || r == 0.0) return kSnpDegenerate;
    return 8.0 * (n - a - r) <= r ? kSnpOrdinary : kSnpOdd;
}
__device__ __forceinline__ bool fast_ordinary(double fa, double fr, double n) { return snp_class(fa, fr, n) == kSnpOrdinary; }

__device__ __forceinline__ FastRow fast_row(double fa, double fr, double n)
{
    const double a = __builtin_rint(fa * n), r = __builtin_rint(fr * n);   // the counts back from a/n, r/n: exact
    FastRow o;
    o.a_s = 1e4 * a;
    o.ra = 1.0 / a;   // inf for a count of 0
    o.rr = 1.0 / r;
    o.rq_s = 1e-4 * (o.ra * o.rr);
    return o;
}

__device__ __forceinline__ FastCol fast_col(double fa, double fr, double n)
{
    const double a = __builtin_rint(fa * n), r = __builtin_rint(fr * n);
    FastCol o;
    o.a = a;
    o.ra = 1.0 / a;
    o.rr = 1.0 / r;
    o.rq = o.ra * o.rr;
    return o;
}

// W pairs at once, stage by stage: ld_pair_fast2 for the operand sets (r[k], c[k]), k < W, with the W dependent
// chains INTERLEAVED by construction.  Left to itself hipcc schedules each pair's ~36 instructions back to back
// (register pressure beside 128 live accumulators), every fp64 instruction then waits for its predecessor's
// result, and a lone epilogue wave ran at ~10 cycles per instruction instead of the 4-5 it can issue.  The
// sched_barriers pin "all W pairs, one stage" as the instruction order.
#ifdef LDX_NOSTAGE   // tuning: leave the order to the compiler
#define LDX_STAGE(body) _Pragma("unroll") for (int t_ = 0; t_ < W; ++t_) { body; }
#else
#define LDX_STAGE(body)                                  \
    _Pragma("unroll") for (int t_ = 0; t_ < W; ++t_) { body; } \
    __builtin_amdgcn_sched_barrier(0);
#endif

// kClean: the caller has established that no operand is degenerate (every count > 0) and that no SNP has missing
// codes (a + r == n, hence D' <= 1 and r^2 <= 1 up to rounding): the inf test, the int-0 selects and the y < 1e7
// guard are dropped.
// T = int (the int8 matrix kernel's accumulators, 8 * n11) or float (the FP4 kernel's, n11: exact integers < 2^24);
// Cell = ldx_ld32 or ldx_k16 (the last stage encodes k for it)
template <int W, bool kClean, typename Cell, typename T>
__device__ __forceinline__ void ld_multi_fast2(const T (&cnt_scaled)[W], const FastConst &k, const FastRow (&r)[W],
                                               const FastCol (&c)[W], Cell (&out)[W], bool (&slow)[W])
{
    double dn4[W], x[W], y[W], inv[W], w[W], z[W], yd[W], yr[W], kd[W], kr[W], hd[W], hr[W], ed[W], er[W], mx[W];
    bool ok[W], deg[W];
    __builtin_amdgcn_sched_barrier(0);
    LDX_STAGE(dn4[t_] = (double)cnt_scaled[t_]; w[t_] = r[t_].a_s * c[t_].a)
    LDX_STAGE(dn4[t_] = __builtin_fma(dn4[t_], k.nsc, -w[t_]); w[t_] = r[t_].rq_s * c[t_].rq)   // 1e4 * Dn, exact
    LDX_STAGE(const bool neg = dn4[t_] < 0.0; x[t_] = neg ? c[t_].ra : c[t_].rr; y[t_] = neg ? c[t_].rr : c[t_].ra)
    LDX_STAGE(x[t_] = r[t_].ra * x[t_]; y[t_] = r[t_].rr * y[t_]; z[t_] = dn4[t_] * w[t_])
    LDX_STAGE(inv[t_] = max_raw(x[t_], y[t_]); yr[t_] = z[t_] * dn4[t_])                       // 1 / B;  r^2 * 10^4
    LDX_STAGE(yd[t_] = __builtin_fabs(dn4[t_]) * inv[t_]; kr[t_] = __builtin_rint(yr[t_]);     // D' * 10^4
              hr[t_] = __builtin_fma(__builtin_fabs(z[t_]), k.ncr, 0.499999))
    LDX_STAGE(kd[t_] = __builtin_rint(yd[t_]); hd[t_] = __builtin_fma(inv[t_], k.ncd, 0.499999); er[t_] = yr[t_] - kr[t_];
              if (!kClean) mx[t_] = max_raw(yd[t_], yr[t_]))
    constexpr bool kF32 = std::is_same<Cell, ldx_ld32>::value;
    LDX_STAGE(ed[t_] = yd[t_] - kd[t_]; ok[t_] = (__builtin_fabs(er[t_]) < hr[t_]) & (dn4[t_] != 0.0);
              if (!kClean) { ok[t_] = ok[t_] & (mx[t_] < 1e7); deg[t_] = inv[t_] == __builtin_inf(); }
              if (kF32) kr[t_] = kr[t_] * 1e-4)
    LDX_STAGE(ok[t_] = ok[t_] & (__builtin_fabs(ed[t_]) < hd[t_]); if (kF32) kd[t_] = kd[t_] * 1e-4)
    if constexpr (cell_measure<Cell>::value >= 0) {   // one value: its k in 15 bits; the other value's margin still counts (a pair
        // near a tie of EITHER value takes the caller's slow path: correct, and this tier is not where the time goes)
        LDX_STAGE(const uint32_t u = (uint32_t)(cell_measure<Cell>::value == LDX_MEASURE_RSQ ? kr[t_] : kd[t_]);
                  if (kClean) { out[t_].value = (uint16_t)u; slow[t_] = !ok[t_]; }
                  else { out[t_].value = deg[t_] ? (uint16_t)LDX_K16_INT0 : (uint16_t)u;
                         slow[t_] = !((ok[t_] & (u < 32767u)) | deg[t_]); })
    } else if constexpr (kF32) {
        LDX_STAGE(const float vr = (float)kr[t_]; const float vd = (float)kd[t_];   // float32 nearest to k / 10^4 (k < 10^7 here)
                  if (kClean) { out[t_].r_square = vr; out[t_].d_prime = vd; slow[t_] = !ok[t_]; }
                  else { out[t_].r_square = deg[t_] ? -0.0f : vr; out[t_].d_prime = deg[t_] ? -0.0f : vd;
                         slow[t_] = !(ok[t_] | deg[t_]); })
    } else {   // k itself: 15 bits each; a sure k >= 32767 (only with missing codes) goes to the caller's slow path, whose
               // encoder writes the escape
        LDX_STAGE(const uint32_t ur = (uint32_t)kr[t_]; const uint32_t ud = (uint32_t)kd[t_];
                  if (kClean) { out[t_].r_square = (uint16_t)ur; out[t_].d_prime = (uint16_t)ud; slow[t_] = !ok[t_]; }
                  else { const bool fits = (ur | ud) < 32767u;
                         out[t_].r_square = deg[t_] ? (uint16_t)LDX_K16_INT0 : (uint16_t)ur;
                         out[t_].d_prime = deg[t_] ? (uint16_t)LDX_K16_INT0 : (uint16_t)ud;
                         slow[t_] = !((ok[t_] & fits) | deg[t_]); })
    }
}

// ---- fp32 first tier (the FP4 matrix kernel's epilogue for units of ordinary SNPs) ------------------------------------
// The same two quantities, y_r = 10^4 r^2 and y_d = 10^4 D', in float32 from the exact integer Dn -- 24 single-rate
// VALU instructions per pair instead of ~33 double-rate ones, short enough dependent chains to interleave four pairs
// -- with a margin test that is wide enough for float32: a lane whose 8 pairs of a step are not ALL provably rounded
// like the reference hands that step to the fp64 tier above (ld_multi_fast2), through a per-wave queue in LDS, so
// the common path never branches per pair.  Valid for ordinary SNPs on both sides (snp_class: polymorphic, at most r / 8
// missing codes: then r^2, D' < 1.27, B >= 1 and a r >= (n - 1) / 2, which the error bounds below use); degenerate SNPs ride
// along with forced cells, every other SNP parks (f32_row).
// Round 4 built and measured a leaner form of this arithmetic (integer Dn from accumulators that start at 2^23, floor /
// fract instead of the magic-number rounding, per-step instead of per-value margins: 22 instead of 28 VALU per pair, 25 %
// faster stand-alone) -- and it was NOT faster inside the kernel (it parks 0.81 % instead of 0.59 % of the lane-steps, and
// the step was bound by its store instructions, not its arithmetic): tools/probes/epi.hip, profiles/r04/fp32_tier_*.log.
//
// Error budget (u = 2^-24; every table entry is a double-precision value rounded once to float32):
//   Dn = n c - a1 a2 is computed exactly while |Dn| < 2^24 (n <= 4096: p = a1 a2 is exact; n > 4096, the kernel's form since
//        round 4: the column count split a2 = ah + al, f32_split_a below, p = a1 ah exact, fma(c, n, -p) and fma(-a1, al, .)
//        exact; the error-free product p = fl(a1 a2), e = a1 a2 - p by fma, (fl(n c - p)) - e, is kept as the third
//        instantiation); beyond 2^24 two roundings: (1 + 2u);
//   y_r = ((Dn s1) s2)^2, s = 10 / sqrt(a r): rel. error <= 13u;   y_d = |Dn| max(ra1s x, rr1s y): rel. error <= 6u;
//   the reference's own deviation from the exact rational value, in y units: <= 6e-12 n^2 / B <= 6e-12 n^2 for D'
//   and <= 1.2e-11 n^2 |Dn| / (a1 r1 a2 r2) <= 1.2e-11 n^2 / (n - 1) for r^2 (the fp64 tier's bounds with B >= 1 and
//   |Dn| <= sqrt(a1 r1 a2 r2), a r >= n - 1).
// A pair is sure when  |y - rint(y)| + eta y < 1/2 - c0  for both quantities and Dn != 0 (y_d > 0): then the float32
// value, the exact value and the reference's double all round to the same integer k.  rint(y) is taken by adding
// 2^23 (the sum's low mantissa bits ARE k, which is what the 4-byte cell stores).
struct F32Row {   // per var_1 (row): a, 1e4 / a, 1e4 / r, 10 / sqrt(a r)
    float a, ra_s, rr_s, s;
};
struct F32Col {   // per var_2 (column): a, 1 / a, 1 / r, 10 / sqrt(a r)
    float a, ra, rr, s;
};
struct F32Const {
    float n;       // haplotypes
    float tol;     // 1/2 - c0: the margin threshold
};
constexpr float kEtaR = 14.0f * 5.9604645e-8f, kEtaD = 7.0f * 5.9604645e-8f;
constexpr float kMagic = 8388608.0f;   // 2^23

__host__ __device__ inline F32Const f32_const(double n)
{
    F32Const c;
    c.n = (float)n;
    // (r^2 share: 1.2e-11 n^2 |Dn| / (a1 r1 a2 r2) with |Dn| <= 1.13 sqrt(a1 r1 a2 r2) and a r >= (n - 1) / 2 for ordinary SNPs
    // with a few missing codes (snp_class): <= 2.8e-11 n^2 / (n - 1); 4e-11 for slack)
    const double c0 = 6e-12 * n * n + 4e-11 * n * n / (n > 2.0 ? n - 1.0 : 1.0) + 2e-6;
    c.tol = (float)(0.5 - c0);   // rounds to nearest: the 2e-6 covers that and the last-place effects of the test itself
    return c;
}

// from the fp64 tier's per-SNP operands (a, 1/a, 1/r as doubles: errors ~1e-16, far below float32's u).
// kSnpOdd gets all-zero entries: every y_d it takes part in is then exactly 0, the step's  min y_d > 0  test fails and the
// lane parks the step for the fp64 tier -- the GENERAL variant of it, which knows degenerate operands.  So one such SNP
// costs its own row / column of cells the slow path, not the whole unit.
// kSnpDegenerate (round 6) stays on the common path.  Its cells are the int-0 code whatever the arithmetic says, so the
// step loop FORCES them (one v_cndmask per cell under a scalar lane mask, only in units that hold such a SNP:
// epilogue_f32, kDeg) and the table entries only have to keep the margin test quiet: a fake count A (1 for a == 0, n + 1
// for r == 0) makes  Dn = n c - A1 A2  a NEGATIVE NON-ZERO number against every other entry -- c is 0 for a SNP without ALT
// alleles and at most a_X for one without REF alleles, so Dn = -A_X resp. Dn <= a_X n - (n + 1) A_X = -a_X (ordinary X),
// -(n + 1), -(2n + 1); where |Dn| exceeds 2^24 the two fma round, which cannot reach zero or change the sign -- the
// sign picks x = ra, y = rr;  ra = eps, rr = 0, s = 2^-20  give tiny positive  y_r = (Dn s1 s2)^2 < 0.01  and  y_d  (eps_r =
// 2^-10 against an ordinary column: y_d = a_X eps_r fl(1 / a_X); eps_c = 2^-24 against an ordinary row: 10^4 eps_c; both:
// <= (2n + 2) 2^-34): both round to 0 with a margin near 0, min y_d > 0 (and, in the one-measure r^2 variant, |Dn s1 s2| > 0)
// holds, nothing parks.  s = 2^-20 is also what MARKS such an entry (an ordinary s is >= 20 / n, kSnpOdd's is 0).  (30 % monomorphic rows at 50 000 x 1008 -- a
// sub-panel of the ALL-panel variants -- sent EVERY unit through the fp64 epilogue before: bench.py, other_workloads.)
constexpr float kDegS = 0x1p-20f;   // the `s` entry of a degenerate SNP: marks it, and keeps Dn s1 s2 non-zero
__device__ __forceinline__ F32Row f32_row(double a, double ra, double rr, int cls, double n)
{
    if (cls == kSnpOdd) return F32Row{0.0f, 0.0f, 0.0f, 0.0f};
    if (cls == kSnpDegenerate) return F32Row{a == 0.0 ? 1.0f : (float)(n + 1.0), 0x1p-10f, 0.0f, kDegS};
    return F32Row{(float)a, (float)(1e4 * ra), (float)(1e4 * rr), (float)(10.0 * __builtin_sqrt(ra * rr))};
}

__device__ __forceinline__ F32Col f32_col(double a, double ra, double rr, int cls, double n)
{
    if (cls == kSnpOdd) return F32Col{0.0f, 0.0f, 0.0f, 0.0f};
    if (cls == kSnpDegenerate) return F32Col{a == 0.0 ? 1.0f : (float)(n + 1.0), 0x1p-24f, 0.0f, kDegS};
    return F32Col{(float)a, (float)ra, (float)rr, (float)(10.0 * __builtin_sqrt(ra * rr))};
}
// a table entry of a kSnpDegenerate SNP
__device__ __forceinline__ bool f32_entry_degenerate(float, float s) { return s == kDegS; }

// dst = lane's mask bit ? forced : keep, the mask a wave-uniform 64-bit value (SALU-made: s_or_b64 of a row and a column mask).
// __builtin_amdgcn_inverse_ballot_w64 hands the scalar mask to the compiler AS a lane mask: it emits  s_or_b64 vcc, .. ;
// v_cndmask_b32 dst, keep, forced, vcc  -- one vector instruction -- and keeps the hazard bookkeeping.  (The first form of
// this, an inline-asm  v_cndmask_b32_e64 dst, keep, forced, s[mask]  right behind the s_or_b64, returned ZERO in lanes 12-15
// / 28-31 / 44-47 / 60-63 of the first select behind a ds_read-written destination in some steps: found by
// tools/gpu_diff6.py; a plain C++  (mask >> lane) & 1  costs four vector instructions per cell.)
__device__ __forceinline__ uint32_t select_lanes(uint32_t keep, uint32_t forced, uint64_t mask)
{
    return __builtin_amdgcn_inverse_ballot_w64(mask) ? forced : keep;
}

// float32 nearest to k / 10^4 for an integer-valued float k < 2^15: quotient by the reciprocal plus one exact
// residual correction (checked exhaustively for 0 <= k < 32768 in tests/test_abi_and_host.py against exact rationals)
__device__ __forceinline__ float f32_k_to_value(float k)
{
    const float c4 = 1e-4f;
    const float q = k * c4;
    const float r = __builtin_fmaf(-q, 1e4f, k);
    return __builtin_fmaf(r, c4, q);
}

// W pairs, stage by stage.  cnt: n11 as floats; out: the encoded cells (valid where the lane turns out sure);
// wmax / ymin accumulate the margin quantity and the smallest y_d over the pairs of a step (the caller tests
// wmax < tol and ymin > 0 once per step).
// kSmallN (n <= 4096: f32_small_n): a1 a2 <= 2^24 is exact in float32, so the product needs no error term and
// Dn = fl(n c - p) is exact in one fma -- two instructions per pair less (of 24).
__host__ __device__ inline bool f32_small_n(double n) { return n <= 4096.0; }

// The column count split for n > 4096 (f32_split_a): a2 = ah + al with al = a2 mod 16, so that a1 * ah has at most 24
// significant bits (a1 < 2^14: LDX_MAX_HAPS) and is EXACT in float32; then q = fma(c, n, -a1 ah) = Dn + a1 al is an integer
// below 2^24 (exact) and Dn = fma(-a1, al, q) too: three instructions where the error-free product needs four.  c.a
// holds ah and `al` the remainder when kSplit; beyond |Dn| >= 2^24 (n > 8192) the two fma round once each, which the
// error budget above allows for.
__device__ __forceinline__ void f32_split_a(float a, float &ah, float &al)
{
    ah = (float)((uint32_t)a & ~15u);
    al = a - ah;
}

template <int W, typename Cell, bool kSmallN = false, bool kSplit = false>
__device__ __forceinline__ void ld_multi_f32(const float (&cnt)[W], const F32Const &k, const F32Row (&r)[W],
                                             const F32Col (&c)[W], Cell (&out)[W], float &wmax, float &ymin,
                                             const float *al = nullptr)
{
    float p[W], e[W], dn[W], t[W], yr[W], x[W], y[W], yd[W], ar[W], ad[W], kr[W], kd[W], fr_[W], fd[W];
    __builtin_amdgcn_sched_barrier(0);
    LDX_STAGE(p[t_] = r[t_].a * c[t_].a)
    if constexpr (kSmallN) {
        LDX_STAGE(dn[t_] = __builtin_fmaf(cnt[t_], k.n, -p[t_]))                    // Dn, exact: p is
    } else if constexpr (kSplit) {
        LDX_STAGE(dn[t_] = __builtin_fmaf(cnt[t_], k.n, -p[t_]))                    // Dn + a1 al, exact (p = a1 ah is)
        LDX_STAGE(dn[t_] = __builtin_fmaf(-r[t_].a, al[t_], dn[t_]))                // Dn, exact below 2^24
    } else {
        LDX_STAGE(e[t_] = __builtin_fmaf(r[t_].a, c[t_].a, -p[t_]); dn[t_] = __builtin_fmaf(cnt[t_], k.n, -p[t_]))
        LDX_STAGE(dn[t_] = dn[t_] - e[t_])                                          // Dn, exact below 2^24
    }
    LDX_STAGE(t[t_] = dn[t_] * r[t_].s; const bool neg = dn[t_] < 0.0f;
              x[t_] = neg ? c[t_].ra : c[t_].rr; y[t_] = neg ? c[t_].rr : c[t_].ra)
    LDX_STAGE(t[t_] = t[t_] * c[t_].s; x[t_] = r[t_].ra_s * x[t_]; y[t_] = r[t_].rr_s * y[t_])
    LDX_STAGE(yr[t_] = t[t_] * t[t_]; x[t_] = __builtin_fmaxf(x[t_], y[t_]))       // 1e4 r^2;  1e4 / B
    LDX_STAGE(yd[t_] = __builtin_fabsf(dn[t_]) * x[t_]; ar[t_] = yr[t_] + kMagic)   // 1e4 D'
    LDX_STAGE(ad[t_] = yd[t_] + kMagic; kr[t_] = ar[t_] - kMagic)
    LDX_STAGE(kd[t_] = ad[t_] - kMagic; fr_[t_] = yr[t_] - kr[t_])
    LDX_STAGE(fd[t_] = yd[t_] - kd[t_]; fr_[t_] = __builtin_fmaf(yr[t_], kEtaR, __builtin_fabsf(fr_[t_])))
    LDX_STAGE(fd[t_] = __builtin_fmaf(yd[t_], kEtaD, __builtin_fabsf(fd[t_])))
    LDX_STAGE(wmax = __builtin_fmaxf(__builtin_fmaxf(wmax, fr_[t_]), fd[t_]); ymin = __builtin_fminf(ymin, yd[t_]))   // one v_max3 per pair
    if constexpr (std::is_same<Cell, ldx_k16>::value) {   // the low mantissa bits of 2^23 + k are k: one byte permute per cell
        LDX_STAGE(out[t_] = __builtin_bit_cast(ldx_k16, __builtin_amdgcn_perm(__float_as_uint(ad[t_]), __float_as_uint(ar[t_]),
                                                                             0x05040100u)))
    } else {
        LDX_STAGE(out[t_].r_square = f32_k_to_value(kr[t_]); out[t_].d_prime = f32_k_to_value(kd[t_]))
    }
}
// The same tier for ONE measure (the 2-byte cells): the other value's products, rounding and margin are not computed --
// r^2 alone: 12 instructions per pair (11 at n <= 4096), D' alone: 14 (13) -- against 24.5.  bits[] receive the float32 bit
// patterns of 2^23 + k: the cell is their low 16 bits (the caller packs two cells per byte permute).  ymin tracks what keeps
// Dn == 0 and kSnpOdd SNPs (all-zero entries) parking: |Dn s1 s2| for r^2, y_d for D'.
template <int W, int kMeasure, bool kSmallN = false, bool kSplit = false>
__device__ __forceinline__ void ld_multi_f32_one(const float (&cnt)[W], const F32Const &k, const F32Row (&r)[W],
                                                 const F32Col (&c)[W], uint32_t (&bits)[W], float &wmax, float &ymin,
                                                 const float *al = nullptr)
{
    float p[W], e[W], dn[W], t[W], yv[W], x[W], y[W], av[W], kv[W], fv[W];
    __builtin_amdgcn_sched_barrier(0);
    LDX_STAGE(p[t_] = r[t_].a * c[t_].a)
    if constexpr (kSmallN) {
        LDX_STAGE(dn[t_] = __builtin_fmaf(cnt[t_], k.n, -p[t_]))
    } else if constexpr (kSplit) {
        LDX_STAGE(dn[t_] = __builtin_fmaf(cnt[t_], k.n, -p[t_]))
        LDX_STAGE(dn[t_] = __builtin_fmaf(-r[t_].a, al[t_], dn[t_]))
    } else {
        LDX_STAGE(e[t_] = __builtin_fmaf(r[t_].a, c[t_].a, -p[t_]); dn[t_] = __builtin_fmaf(cnt[t_], k.n, -p[t_]))
        LDX_STAGE(dn[t_] = dn[t_] - e[t_])
    }
    if constexpr (kMeasure == LDX_MEASURE_RSQ) {
        LDX_STAGE(t[t_] = dn[t_] * r[t_].s)
        LDX_STAGE(t[t_] = t[t_] * c[t_].s)
        LDX_STAGE(yv[t_] = t[t_] * t[t_])                                              // 1e4 r^2
        LDX_STAGE(av[t_] = yv[t_] + kMagic; ymin = __builtin_fminf(ymin, __builtin_fabsf(t[t_])))   // 0: Dn == 0, or a kSnpOdd SNP
        LDX_STAGE(kv[t_] = av[t_] - kMagic)
        LDX_STAGE(fv[t_] = yv[t_] - kv[t_])
        LDX_STAGE(fv[t_] = __builtin_fmaf(yv[t_], kEtaR, __builtin_fabsf(fv[t_])))
    } else {
        LDX_STAGE(const bool neg = dn[t_] < 0.0f; x[t_] = neg ? c[t_].ra : c[t_].rr; y[t_] = neg ? c[t_].rr : c[t_].ra)
        LDX_STAGE(x[t_] = r[t_].ra_s * x[t_]; y[t_] = r[t_].rr_s * y[t_])
        LDX_STAGE(x[t_] = __builtin_fmaxf(x[t_], y[t_]))                               // 1e4 / B
        LDX_STAGE(yv[t_] = __builtin_fabsf(dn[t_]) * x[t_])                            // 1e4 D'
        LDX_STAGE(av[t_] = yv[t_] + kMagic; ymin = __builtin_fminf(ymin, yv[t_]))
        LDX_STAGE(kv[t_] = av[t_] - kMagic)
        LDX_STAGE(fv[t_] = yv[t_] - kv[t_])
        LDX_STAGE(fv[t_] = __builtin_fmaf(yv[t_], kEtaD, __builtin_fabsf(fv[t_])))
    }
    LDX_STAGE(wmax = __builtin_fmaxf(wmax, fv[t_]); bits[t_] = __float_as_uint(av[t_]))
}
#undef LDX_STAGE

// Compute units of the current device, cached per device ordinal: hipGetDeviceProperties costs tens of
// microseconds per call, which an eager launch of a 0.25 ms kernel would feel.
inline int device_cus()
{
    static int cached[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (cached[dev] > 0) return cached[dev];
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) return 256;
    cached[dev] = cus;   // a race writes the same value twice
    return cus;
}

}  // namespace ldx
