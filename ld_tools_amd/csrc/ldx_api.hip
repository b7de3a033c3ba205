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

extern "C" uint64_t ldx_triangle_cell_index(uint32_t n_snps, uint32_t row, uint32_t col, int out_format)
{
    return ldx_triangle_unit_of(n_snps, row, col) * LDX_UNIT_PAIRS + LDX_CELL_OFFSET(out_format, row % kGroup, col % kSlab);
}

// ---- calc_ld for one pair, host pointers (used by the backend/calc_ld.py drop-in; ld_lite.py:143) ----
namespace ldx {

struct CalcLdRecord {       // the 80 bytes that travel back (static_assert below)
    uint32_t counts[6];     // n, n11, a1, r1, a2, r2
    uint32_t flags;
    uint32_t pad;
    double raw[2];          // r_square, d_prime unrounded
    double k[2];            // round(x, 4) * 10^4
    double freq4[2];        // round(fa1, 4), round(fa2, 4)
};
static_assert(sizeof(CalcLdRecord) == 80, "record layout");

// One workgroup: the five counts of calc_ld.py:30-40 straight from the two code vectors (zip semantics: n11 over the
// common prefix, allele counts over each full vector), then the op-for-op epilogue with real divisions and the exact
// 4-decimal rounding on one lane.
__global__ void __launch_bounds__(1024) calc_ld_pair_kernel(const int8_t *__restrict__ g1, uint32_t h1,
                                                            const int8_t *__restrict__ g2, uint32_t h2,
                                                            CalcLdRecord *__restrict__ rec)
{
    __shared__ uint32_t part[16][5];
    const uint32_t hmax = h1 > h2 ? h1 : h2, hmin = h1 < h2 ? h1 : h2;
    uint32_t c[5] = {0, 0, 0, 0, 0};   // n11, a1, r1, a2, r2
    for (uint32_t h = threadIdx.x; h < hmax; h += blockDim.x) {
        const int x = h < h1 ? g1[h] : 2, y = h < h2 ? g2[h] : 2;
        c[0] += (h < hmin) & (x == 1) & (y == 1);
        c[1] += x == 1;
        c[2] += x == 0;
        c[3] += y == 1;
        c[4] += y == 0;
    }
#pragma unroll
    for (int k = 0; k < 5; ++k) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) c[k] += __shfl_xor(c[k], off);
        if ((threadIdx.x & 63u) == 0) part[threadIdx.x >> 6][k] = c[k];
    }
    block_sync();
    if (threadIdx.x == 0) {
        uint32_t t[5] = {0, 0, 0, 0, 0};
        for (uint32_t w = 0; w < blockDim.x / 64u; ++w)
            for (int k = 0; k < 5; ++k) t[k] += part[w][k];
        const double n = (double)hmin;
        const double fa1 = (double)t[1] / n, fr1 = (double)t[2] / n, fa2 = (double)t[3] / n, fr2 = (double)t[4] / n;
        const LdRaw lr = ld_epilogue((double)t[0] / n, fa1, fr1, fa1 * fr1, fa2, fr2);
        CalcLdRecord r;
        r.counts[0] = hmin;
        for (int k = 0; k < 5; ++k) r.counts[k + 1] = t[k];
        r.flags = lr.flags;
        r.pad = 0;
        r.raw[0] = lr.rsq;
        r.raw[1] = lr.dprime;
        r.k[0] = round4_k(lr.rsq);
        r.k[1] = round4_k(lr.dprime);
        r.freq4[0] = round4_k(fa1) / 1e4;
        r.freq4[1] = round4_k(fa2) / 1e4;
        *rec = r;
    }
}

// device scratch of the calling host thread (grown on demand, released when the thread ends)
struct CalcLdScratch {
    int dev = -1;
    void *codes = nullptr;
    size_t cap = 0;
    CalcLdRecord *rec = nullptr;
    hipStream_t stream = nullptr;
    void release()
    {
        if (codes) (void)hipFree(codes);
        if (rec) (void)hipFree(rec);
        if (stream) (void)hipStreamDestroy(stream);
        codes = nullptr;
        rec = nullptr;
        stream = nullptr;
        cap = 0;
    }
    ~CalcLdScratch() { release(); }
};

}  // namespace ldx

extern "C" int ldx_calc_ld_host(const int8_t *g1, uint32_t h1, const int8_t *g2, uint32_t h2,
                                uint32_t counts[6], ldx_ld64 *raw, ldx_ld64 *rounded, double freq4[2],
                                uint8_t *flags)
{
    LDX_REQUIRE(g1 && g2 && counts, "null pointer");
    LDX_REQUIRE(h1 >= 1 && h2 >= 1, "empty genotype vector (the reference raises ZeroDivisionError, calc_ld.py:33)");
    static thread_local CalcLdScratch sc;
    int dev = 0;
    LDX_HIP(hipGetDevice(&dev));
    if (sc.dev != dev) {   // first call of this thread, or the thread switched devices
        sc.release();
        sc.dev = dev;
    }
    const size_t need = ((size_t)h1 + 15u) / 16u * 16u + h2;
    if (!sc.rec) {
        LDX_HIP(hipMalloc((void **)&sc.rec, sizeof(CalcLdRecord)));
        LDX_HIP(hipStreamCreateWithFlags(&sc.stream, hipStreamNonBlocking));
    }
    if (need > sc.cap) {
        if (sc.codes) LDX_HIP(hipFree(sc.codes));
        sc.codes = nullptr;
        sc.cap = 0;
        const size_t cap = need < 65536 ? 65536 : need * 2;
        LDX_HIP(hipMalloc(&sc.codes, cap));
        sc.cap = cap;
    }
    int8_t *d1 = (int8_t *)sc.codes, *d2 = d1 + ((size_t)h1 + 15u) / 16u * 16u;
    LDX_HIP(hipMemcpyAsync(d1, g1, h1, hipMemcpyHostToDevice, sc.stream));
    LDX_HIP(hipMemcpyAsync(d2, g2, h2, hipMemcpyHostToDevice, sc.stream));
    calc_ld_pair_kernel<<<1, 1024, 0, sc.stream>>>(d1, h1, d2, h2, sc.rec);
    LDX_HIP(hipGetLastError());
    CalcLdRecord r;
    LDX_HIP(hipMemcpyAsync(&r, sc.rec, sizeof(r), hipMemcpyDeviceToHost, sc.stream));
    LDX_HIP(hipStreamSynchronize(sc.stream));
    for (int k = 0; k < 6; ++k) counts[k] = r.counts[k];
    if (raw) *raw = ldx_ld64{r.raw[0], r.raw[1]};
    if (flags) *flags = (uint8_t)r.flags;
    if (rounded) {   // k / 10^4 in double IS Python's round(x, 4): the double nearest to the decimal k * 10^-4
        rounded->r_square = r.k[0] / 1e4;
        rounded->d_prime = r.k[1] / 1e4;
    }
    if (freq4) {
        freq4[0] = r.freq4[0];
        freq4[1] = r.freq4[1];
    }
    return LDX_OK;
}
