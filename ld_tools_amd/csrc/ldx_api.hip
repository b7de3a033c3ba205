// Library-level entry points: version, errors, geometry helpers, host-pointer conveniences.
#include <math.h>
#include <stdarg.h>
#include <string.h>

#include <vector>

#include "ldx_common.h"

namespace ldx {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

double thres_to_k(double thres)
{
    if (!(thres > 0.0)) return 0.0;          // every k >= 0 passes (also for NaN-free negative thresholds)
    double k = ceil(thres * 1e4) - 2.0;
    if (k < 0.0) k = 0.0;
    while (k / 1e4 < thres) k += 1.0;        // at most a few steps
    return k;
}

bool check_recip(uint32_t n)
{
    static thread_local uint32_t last_ok = 0;
    if (n == last_ok) return true;
    const double nn = (double)n, rn = 1.0 / nn;
    for (uint32_t c = 0; c <= n; ++c) {
        const double cc = (double)c;
        const double q = cc * rn;
        const double r = fma(-q, nn, cc);
        if (fma(r, rn, q) != cc / nn) return false;
    }
    last_ok = n;
    return true;
}

}  // namespace ldx

using namespace ldx;

extern "C" int ldx_version(void) { return LDX_VERSION; }

extern "C" const char *ldx_last_error(void) { return g_err; }

extern "C" int ldx_device_count(void)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        set_error("hipGetDeviceCount failed: %s", hipGetErrorString(e));
        return LDX_E_HIP;
    }
    return n;
}

extern "C" int ldx_device_arch(int device, char *buf, size_t buflen)
{
    LDX_REQUIRE(buf && buflen > 0, "null buffer");
    hipDeviceProp_t prop;
    LDX_HIP(hipGetDeviceProperties(&prop, device));
    strncpy(buf, prop.gcnArchName, buflen - 1);
    buf[buflen - 1] = 0;
    if (char *colon = strchr(buf, ':')) *colon = 0;   // "gfx950:sramecc+:xnack-" -> "gfx950"
    return LDX_OK;
}

extern "C" uint32_t ldx_n_slabs(uint32_t n_snps) { return ldx::n_slabs(n_snps); }
extern "C" uint32_t ldx_n_chunks(uint32_t n_hap) { return ldx::n_chunks(n_hap); }
extern "C" uint32_t ldx_padded_snps(uint32_t n_snps) { return ldx::n_slabs(n_snps) * kSlab; }

extern "C" size_t ldx_plane_bytes(uint32_t n_snps, uint32_t n_hap)
{
    return (size_t)ldx::n_slabs(n_snps) * ldx::n_chunks(n_hap) * kSlab * 16u;
}

extern "C" uint64_t ldx_triangle_units(uint32_t n_snps)
{
    const uint64_t T = ldx::n_slabs(n_snps);
    return tile_base(T, T * kGroupsPerSlab);
}

extern "C" uint64_t ldx_triangle_tile_base(uint32_t n_snps, uint32_t tile)
{
    const uint64_t T = ldx::n_slabs(n_snps);
    return tile_base(tile, T * kGroupsPerSlab);
}

extern "C" uint64_t ldx_triangle_unit_of(uint32_t n_snps, uint32_t row, uint32_t col)
{
    const uint64_t T = ldx::n_slabs(n_snps);
    const uint64_t t = col / kSlab, g = row / kGroup;
    return tile_base(t, T * kGroupsPerSlab) + (g - t * kGroupsPerSlab);
}

// ---- calc_ld for one pair, host pointers (used by the backend/calc_ld.py drop-in) ----
namespace {
struct DevBuf {
    void *p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    int alloc(size_t n) { LDX_HIP(hipMalloc(&p, n)); return LDX_OK; }
};
}  // namespace

extern "C" int ldx_calc_ld_host(const int8_t *g1, uint32_t h1, const int8_t *g2, uint32_t h2,
                                uint32_t counts[6], ldx_ld64 *raw, ldx_ld64 *rounded, double freq4[2],
                                uint8_t *flags)
{
    LDX_REQUIRE(g1 && g2 && counts, "null pointer");
    LDX_REQUIRE(h1 >= 1 && h2 >= 1, "empty genotype vector (the reference raises ZeroDivisionError, calc_ld.py:33)");
    const uint32_t hmax = h1 > h2 ? h1 : h2, n = h1 < h2 ? h1 : h2;
    if (hmax > LDX_MAX_HAPS) {
        set_error("ldx_calc_ld_host: %u haplotypes > LDX_MAX_HAPS %u", hmax, LDX_MAX_HAPS);
        return LDX_E_UNSUPPORTED;
    }
    // a 2-row panel of width hmax; the shorter vector is padded with code 2 (neither plane), so the AND over
    // the full width equals the zipped-prefix count of calc_ld.py:30-32 while a/r cover the full vectors (:37-40)
    const size_t ld = ((size_t)hmax + 15u) & ~(size_t)15u;
    std::vector<int8_t> codes(2 * ld, (int8_t)2);
    memcpy(codes.data(), g1, h1);
    memcpy(codes.data() + ld, g2, h2);
    const size_t pb = ldx_plane_bytes(2, hmax);
    const uint32_t npad = ldx_padded_snps(2);
    DevBuf dcodes, dalt, dref, dcnt, dn11, dsix, draw, drnd, dfl, dfreq;
    int rc;
    if ((rc = dcodes.alloc(codes.size())) || (rc = dalt.alloc(pb)) || (rc = dref.alloc(pb)) ||
        (rc = dcnt.alloc(2 * npad * sizeof(uint32_t))) || (rc = dn11.alloc(4 * sizeof(uint32_t))) ||
        (rc = dsix.alloc(5 * sizeof(uint32_t))) || (rc = draw.alloc(sizeof(ldx_ld64))) ||
        (rc = drnd.alloc(sizeof(ldx_ld32))) || (rc = dfl.alloc(16)) || (rc = dfreq.alloc(npad * sizeof(double))))
        return rc;
    uint32_t *acnt = (uint32_t *)dcnt.p, *rcnt = acnt + npad;
    LDX_HIP(hipMemcpy(dcodes.p, codes.data(), codes.size(), hipMemcpyHostToDevice));
    if ((rc = ldx_pack_codes_dev((const int8_t *)dcodes.p, 2, hmax, ld, dalt.p, dref.p, acnt, rcnt, nullptr))) return rc;
    if ((rc = ldx_pair_counts_dev(dalt.p, 2, dalt.p, 2, hmax, (uint32_t *)dn11.p, 2, nullptr))) return rc;
    if ((rc = ldx_alt_freq4_dev(acnt, 2, n, (double *)dfreq.p, nullptr))) return rc;
    uint32_t hn11[4], ha[2], hr[2];
    LDX_HIP(hipMemcpy(hn11, dn11.p, sizeof(hn11), hipMemcpyDeviceToHost));
    LDX_HIP(hipMemcpy(ha, acnt, sizeof(ha), hipMemcpyDeviceToHost));
    LDX_HIP(hipMemcpy(hr, rcnt, sizeof(hr), hipMemcpyDeviceToHost));
    counts[0] = n; counts[1] = hn11[1]; counts[2] = ha[0]; counts[3] = hr[0]; counts[4] = ha[1]; counts[5] = hr[1];
    // the epilogue runs on the device from the six integers (n is the zipped length)
    uint32_t *six = (uint32_t *)dsix.p;
    LDX_HIP(hipMemcpy(six, counts + 1, 5 * sizeof(uint32_t), hipMemcpyHostToDevice));
    if ((rc = ldx_ld_from_counts_dev(n, 1, six, six + 1, six + 2, six + 3, six + 4, (ldx_ld64 *)draw.p,
                                     (ldx_ld32 *)drnd.p, (uint8_t *)dfl.p, nullptr)))
        return rc;
    ldx_ld64 hraw;
    ldx_ld32 hrnd;
    uint8_t hfl;
    LDX_HIP(hipMemcpy(&hraw, draw.p, sizeof(hraw), hipMemcpyDeviceToHost));
    LDX_HIP(hipMemcpy(&hrnd, drnd.p, sizeof(hrnd), hipMemcpyDeviceToHost));
    LDX_HIP(hipMemcpy(&hfl, dfl.p, 1, hipMemcpyDeviceToHost));
    if (raw) *raw = hraw;
    if (flags) *flags = hfl;
    if (rounded) {
        // the float32 cell is the one nearest to k/10^4: hand back k/10^4 itself as a double
        rounded->r_square = rint((double)hrnd.r_square * 1e4) / 1e4;
        rounded->d_prime = rint((double)hrnd.d_prime * 1e4) / 1e4;
    }
    if (freq4) LDX_HIP(hipMemcpy(freq4, dfreq.p, 2 * sizeof(double), hipMemcpyDeviceToHost));
    return LDX_OK;
}
