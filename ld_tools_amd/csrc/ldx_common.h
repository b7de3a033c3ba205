// Shared definitions for the ldx HIP sources (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/ldx.h"

namespace ldx {

constexpr uint32_t kSlab = LDX_SLAB_ROWS;     // 128 SNP rows per slab / j-tile
constexpr uint32_t kGroup = LDX_GROUP_ROWS;   // 8 SNP rows per wave unit
constexpr uint32_t kGroupsPerSlab = kSlab / kGroup;   // 16

void set_error(const char *fmt, ...);

// Smallest integer k >= 0 with (double)k / 1e4 >= thres: "rounded value >= thres" (ld_area.py:248,
// ld_triangle.py:224) becomes the exact integer test k >= thres_to_k(thres).
double thres_to_k(double thres);

// ld_triangle on the matrix cores (ldx_mfma.hip); same contract as ldx_triangle_dev after argument checks
int triangle_mfma(const void *alt, const double *fa, const double *fr, const double *q, uint32_t n_snps, uint32_t n_hap,
                  uint64_t unit_begin, uint64_t unit_end, ldx_ld32 *out, ldx_ld64 *out_raw, uint32_t *out_n11,
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
__host__ __device__ inline uint32_t n_chunks(uint32_t n_hap) { return (n_hap + 127u) / 128u; }

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

// float32 nearest to k/10^4; an int-0 result carries the sign bit (-0.0f).
__device__ inline float encode32(double k, bool int0)
{
    float v = (float)(k * 1e-4);
    return int0 ? -0.0f : v;
}

__device__ inline ldx_ld32 round_pair(const LdRaw &r)
{
    ldx_ld32 o;
    o.r_square = encode32(round4_k(r.rsq), (r.flags & LDX_FLAG_RSQ_INT0) != 0);
    o.d_prime = encode32(round4_k(r.dprime), (r.flags & LDX_FLAG_DPRIME_INT0) != 0);
    return o;
}

// The mirror as one out-of-line call: the fallback of ld_pair_fast (rare) must not be inlined 16-128 times.
static __device__ __noinline__ ldx_ld32 ld_pair_mirror(double f11, double fa1, double fr1, double q1, double fa2,
                                                       double fr2)
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
__device__ __forceinline__ ldx_ld32 ld_pair_fast(double f11, double fa1, double fr1, double q1, double fa2,
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
    ldx_ld32 o;
    const float vd = (float)(kd * 1e-4), vr = (float)(kr * 1e-4);   // float32 nearest to k / 10^4
    o.d_prime = degenerate ? -0.0f : vd;
    o.r_square = (degenerate || d == 0.0) ? -0.0f : vr;      // d_prime == 0 <=> d == 0 when bound != 0
    return o;
}

}  // namespace ldx
