// What bounds the fp32 epilogue tier's step loop?  The loop of triangle_mfma_kernel's epilogue_f32 (ldx_mfma.hip), stand-alone:
// 128 accumulator registers per wave, row operands from LDS every step, column operands in registers, dynamic accumulator
// index, eight 4-byte non-temporal stores per step behind the once-per-step margin test.  One 512-thread workgroup per CU:
// waves 0-3 (one per SIMD) run role A, waves 4-7 (their SIMD partners) role B; a role is E = the step loop, M = back-to-back
// FP4 MFMAs (a K loop's matrix work, nothing else), or idle.  Reported: shader cycles per 16-step unit for the E waves
// (and per MFMA for the M waves), by variant of the arithmetic:
//   0  the product's tier (error-free float product for Dn, magic-number rounding, per-value margins: 28 VALU per pair)
//   1  round 4's CANDIDATE (integer Dn from accumulators that start at 2^23, fract / floor, per-step trackers: ~22 VALU per
//      pair; namespace cand4 below -- built into the kernel, measured, not kept: DESIGN.md section 7)
//   2  the candidate's arithmetic, no stores    3  stores only (no arithmetic: the cells are the accumulators' bits)
//   4  the candidate's arithmetic, accumulators read with STATIC indices (16 steps unrolled: no s_set_gpr_idx / v_mov) -- NOT
//      run: with real data (stores executed) this instantiation raised a memory fault at address 0 that was not tracked down
//      (round 3 measured the unrolled step loop inside the kernel: 3.5 % slower)
//   5  the product's tier with ONE 16-byte store per row (the cell order of include/ldx.h) instead of four 4-byte ones
//   6  the candidate's arithmetic with 16-byte stores
// Build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -I../../ epi.hip -o epi
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>

#include "../../ld_tools_amd/csrc/ldx_common.h"

namespace ldx { void set_error(const char *, ...) {} }
using namespace ldx;

typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v8i __attribute__((ext_vector_type(8)));

// ---- round 4's candidate arithmetic (see the header comment); the product's tier is ldx::ld_multi_f32 of ldx_common.h ----
namespace cand4 {
#define CSTAGE(body) _Pragma("unroll") for (int t_ = 0; t_ < W; ++t_) { body; } __builtin_amdgcn_sched_barrier(0);
struct F32Row {   // per var_1 (row): a (kSmallN: as a float; integer mode: the int32 bit pattern), 1e4 / a, 1e4 / r, 10 / sqrt(a r)
    float a, ra_s, rr_s, s;
};
struct F32Col {   // per var_2 (column): a (kSmallN: as a float; integer mode: the bit pattern of -a), 1 / a, 1 / r, 10 / sqrt(a r)
    float a, ra, rr, s;
};
struct F32Const {
    float n;       // haplotypes
    float c0;      // the margin every value needs on top of eta * y
    int n_i;       // haplotypes as an integer (integer mode)
    int mode;      // F32_SMALL_N, F32_INT or F32_OFF
};
enum { F32_OFF = 0, F32_SMALL_N = 1, F32_INT = 2 };
constexpr float kEtaR = 14.0f * 5.9604645e-8f, kEtaD = 7.0f * 5.9604645e-8f;
constexpr float kMagic = 8388608.0f;   // 2^23: where the accumulators of an integer-mode launch start

__host__ __device__ inline int f32_mode(double n) { return n <= 4096.0 ? F32_SMALL_N : (n <= 32768.0 ? F32_INT : F32_OFF); }
__host__ __device__ inline bool f32_small_n(double n) { return f32_mode(n) == F32_SMALL_N; }
// what the accumulators of a launch with the fp32 tier start at (and every other reader of them subtracts)
__host__ __device__ inline float f32_acc_bias(double n) { return f32_mode(n) == F32_INT ? kMagic : 0.0f; }

__host__ __device__ inline F32Const f32_const(double n)
{
    F32Const c;
    c.n = (float)n;
    c.c0 = (float)(6e-12 * n * n + 1.2e-11 * n * n / (n > 2.0 ? n - 1.0 : 1.0) + 2e-6);
    c.n_i = (int)n;
    c.mode = f32_mode(n);
    return c;
}

// from the fp64 tier's per-SNP operands (a, 1/a, 1/r as doubles: errors ~1e-16, far below float32's u).
// A SNP that is not ordinary (monomorphic, or with missing codes: a + r < n) gets all-zero reciprocals: every y_d' it
// takes part in is then exactly 1/2, the step's  min y_d' > 1/2  test fails and the lane parks the step for the fp64
// tier -- the GENERAL variant of it, which knows degenerate operands.  So one such SNP costs its own row / column of
// cells the slow path, not the whole unit (round 3: one monomorphic SNP among a tile's 128 columns sent every unit of
// the tile through the fp64 epilogue -- 13 % of the units of the 50 000 x 1008 bench panel for 0.07 % such SNPs).
template <bool kInt>
__device__ __forceinline__ F32Row f32_row(double a, double ra, double rr, bool ordinary)
{
    if (!ordinary) return F32Row{kInt ? __int_as_float(0) : 0.0f, 0.0f, 0.0f, 0.0f};
    const float af = kInt ? __int_as_float((int)__builtin_rint(a)) : (float)a;
    return F32Row{af, (float)(1e4 * ra), (float)(1e4 * rr), (float)(10.0 * __builtin_sqrt(ra * rr))};
}

template <bool kInt>
__device__ __forceinline__ F32Col f32_col(double a, double ra, double rr, bool ordinary)
{
    if (!ordinary) return F32Col{kInt ? __int_as_float(0) : 0.0f, 0.0f, 0.0f, 0.0f};
    const float af = kInt ? __int_as_float(-(int)__builtin_rint(a)) : (float)a;
    return F32Col{af, (float)ra, (float)rr, (float)(10.0 * __builtin_sqrt(ra * rr))};
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

// what a lane accumulates over the pairs of a step
struct F32Track {
    float fminr, fmaxr, fmind, fmaxd, ymaxr, ymaxd, ymin;
};
__device__ __forceinline__ F32Track f32_track_init() { return F32Track{1.0f, 0.0f, 1.0f, 0.0f, 0.0f, 0.0f, 2.0f}; }
__device__ __forceinline__ bool f32_sure(const F32Track &t, const F32Const &k)
{
    const float dr = __builtin_fmaf(t.ymaxr, kEtaR, k.c0), dd = __builtin_fmaf(t.ymaxd, kEtaD, k.c0);
    const float mr = __builtin_fminf(t.fminr, 1.0f - t.fmaxr), md = __builtin_fminf(t.fmind, 1.0f - t.fmaxd);
    return (mr > dr) & (md > dd) & (t.ymin > 0.5f);
}

// W pairs, stage by stage.  cnt: the accumulators as they are (kSmallN: n11 as floats; integer mode: 2^23 + n11); out:
// the encoded cells (valid where the lane turns out sure); trk accumulates over the pairs of a step (the caller tests
// f32_sure once per step).
template <int W, typename Cell, bool kSmallN = false>
__device__ __forceinline__ void ld_multi_f32(const float (&cnt)[W], const F32Const &k, const F32Row (&r)[W],
                                             const F32Col (&c)[W], Cell (&out)[W], F32Track &trk)
{
    float p[W], dn[W], t[W], yr[W], x[W], y[W], yd[W], fr_[W], fd[W];
    int di[W];
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (kSmallN) {
        CSTAGE(p[t_] = r[t_].a * c[t_].a)
        CSTAGE(dn[t_] = __builtin_fmaf(cnt[t_], k.n, -p[t_]))                    // Dn, exact
    } else {
        CSTAGE(di[t_] = __mul24(__float_as_int(r[t_].a), __float_as_int(c[t_].a)))     // -a1 a2
        CSTAGE(di[t_] = __mul24(__float_as_int(cnt[t_]), k.n_i) + di[t_])              // v_mad_i32_i24: n c - a1 a2
        CSTAGE(dn[t_] = (float)di[t_])                                            // exact below 2^24
    }
    CSTAGE(t[t_] = dn[t_] * r[t_].s; const bool neg = dn[t_] < 0.0f;
              x[t_] = neg ? c[t_].ra : c[t_].rr; y[t_] = neg ? c[t_].rr : c[t_].ra)
    CSTAGE(t[t_] = t[t_] * c[t_].s; x[t_] = r[t_].ra_s * x[t_]; y[t_] = r[t_].rr_s * y[t_])
    CSTAGE(yr[t_] = __builtin_fmaf(t[t_], t[t_], 0.5f); x[t_] = __builtin_fmaxf(x[t_], y[t_]))   // 1e4 r^2 + 1/2;  1e4 / B
    CSTAGE(yd[t_] = __builtin_fmaf(__builtin_fabsf(dn[t_]), x[t_], 0.5f); fr_[t_] = __builtin_amdgcn_fractf(yr[t_]))   // 1e4 D' + 1/2
    CSTAGE(fd[t_] = __builtin_amdgcn_fractf(yd[t_]))
    // trackers: two values per v_min3 / v_max3
    {
        constexpr int kPairs = W / 2;
#pragma unroll
        for (int h = 0; h < kPairs; ++h) {
            trk.fminr = __builtin_fminf(__builtin_fminf(trk.fminr, fr_[2 * h]), fr_[2 * h + 1]);
            trk.fmaxr = __builtin_fmaxf(__builtin_fmaxf(trk.fmaxr, fr_[2 * h]), fr_[2 * h + 1]);
            trk.fmind = __builtin_fminf(__builtin_fminf(trk.fmind, fd[2 * h]), fd[2 * h + 1]);
            trk.fmaxd = __builtin_fmaxf(__builtin_fmaxf(trk.fmaxd, fd[2 * h]), fd[2 * h + 1]);
            trk.ymaxr = __builtin_fmaxf(__builtin_fmaxf(trk.ymaxr, yr[2 * h]), yr[2 * h + 1]);
            trk.ymaxd = __builtin_fmaxf(__builtin_fmaxf(trk.ymaxd, yd[2 * h]), yd[2 * h + 1]);
            trk.ymin = __builtin_fminf(__builtin_fminf(trk.ymin, yd[2 * h]), yd[2 * h + 1]);
        }
        if constexpr (W & 1) {
            trk.fminr = __builtin_fminf(trk.fminr, fr_[W - 1]);
            trk.fmaxr = __builtin_fmaxf(trk.fmaxr, fr_[W - 1]);
            trk.fmind = __builtin_fminf(trk.fmind, fd[W - 1]);
            trk.fmaxd = __builtin_fmaxf(trk.fmaxd, fd[W - 1]);
            trk.ymaxr = __builtin_fmaxf(trk.ymaxr, yr[W - 1]);
            trk.ymaxd = __builtin_fmaxf(trk.ymaxd, yd[W - 1]);
            trk.ymin = __builtin_fminf(trk.ymin, yd[W - 1]);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (std::is_same<Cell, ldx_k16>::value) {   // k = floor(y'): two conversions and one shift-or per cell
        CSTAGE(const uint32_t ur = (uint32_t)yr[t_]; const uint32_t ud = (uint32_t)yd[t_];
                  out[t_] = __builtin_bit_cast(ldx_k16, (ud << 16) | ur))
    } else {
        CSTAGE(out[t_].r_square = f32_k_to_value(yr[t_] - fr_[t_]); out[t_].d_prime = f32_k_to_value(yd[t_] - fd[t_]))
    }
}
#undef CSTAGE
}  // namespace cand4

typedef unsigned v4u __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store4_saddr(ldx_k16 *sbase, uint32_t voff_bytes, ldx_k16 c0, ldx_k16 c1, ldx_k16 c2, ldx_k16 c3)
{
    const v4u v = {__builtin_bit_cast(uint32_t, c0), __builtin_bit_cast(uint32_t, c1), __builtin_bit_cast(uint32_t, c2),
                   __builtin_bit_cast(uint32_t, c3)};
    asm volatile("global_store_dwordx4 %0, %1, %2 nt" : : "v"(voff_bytes), "v"(v), "s"(sbase) : "memory");
}

template <int kOffset>
__device__ __forceinline__ void store_saddr(ldx_k16 *sbase, uint32_t voff_bytes, ldx_k16 v)
{
    asm volatile("global_store_dword %0, %1, %2 offset:%3 nt" : : "v"(voff_bytes), "v"(__builtin_bit_cast(uint32_t, v)), "s"(sbase), "n"(kOffset) : "memory");
}

enum { ROLE_IDLE = 0, ROLE_E = 1, ROLE_M = 2 };

template <int VAR>
__global__ void __launch_bounds__(512) k(const float *__restrict__ counts, const float *__restrict__ rowtab,
                                         const float *__restrict__ coltab, ldx_k16 *__restrict__ out, int units, int n_mfma,
                                         int role_a, int role_b, cand4::F32Const fc, ldx::F32Const fo, unsigned long long *cyc,
                                         unsigned *parked)
{
    __shared__ float rt[8][64 * 4];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int role = wave < 4 ? role_a : role_b;
    unsigned long long t0 = 0, t1 = 0;
    if (role == ROLE_E) {
        // this wave's 64 rows -> LDS, its lane's 4 columns -> registers, the unit's counts -> 128 accumulators
        for (int k4 = 0; k4 < 4; ++k4) rt[wave][lane * 4 + k4] = rowtab[lane * 4 + k4];
        constexpr bool kProd = VAR == 0 || VAR == 5;      // the product's arithmetic; else the candidate's
        constexpr bool kWide = VAR == 5 || VAR == 6;      // 16-byte stores (the cell order of include/ldx.h)
        ldx::F32Col cols[4];
        cand4::F32Col ccols[4];
        const uint32_t l32 = lane & 31u, half = lane >> 5;
        for (int tt = 0; tt < 4; ++tt) {
            const float *c = coltab + (32 * tt + l32) * 4;
            cols[tt] = ldx::F32Col{c[0], c[1], c[2], c[3]};
            ccols[tt] = cand4::F32Col{c[0], c[1], c[2], c[3]};
        }
        v16f acc[2][4];
        for (int m = 0; m < 2; ++m)
            for (int tt = 0; tt < 4; ++tt)
                for (int e = 0; e < 16; ++e) {
                    const uint32_t row = 32 * m + (e & 3) + 8 * (e >> 2) + 4 * half, col = 32 * tt + l32;
                    acc[m][tt][e] = counts[row * 128 + col] + (kProd ? 0.0f : cand4::kMagic)   /* the candidate always runs its integer mode here */;
                }
        __syncthreads();
        const float *const rtw = rt[wave] + half * 16u;
        ldx_k16 *const wbase = out + (size_t)(blockIdx.x * 8 + wave) * 8192u;
        const uint32_t lane_off_b = (half * 4u * 128u + (kWide ? 4u : 1u) * l32) * 4u;
        unsigned np = 0;
        t0 = __builtin_amdgcn_s_memtime();
        for (int u = 0; u < units; ++u) {
            auto step = [&](int e, auto static_c) {
                v4f rows[2];
#pragma unroll
                for (int m = 0; m < 2; ++m) rows[m] = *reinterpret_cast<const v4f *>(rtw + (32u * m + (e & 3) + 8u * (e >> 2)) * 4u);
                ldx_k16 cell[8];
                float wmax = 0.0f, ymin = 1.0f;
                cand4::F32Track trk = cand4::f32_track_init();
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    float c4[4];
                    ldx::F32Row r4[4];
                    cand4::F32Row cr4[4];
                    ldx_k16 o4[4];
#pragma unroll
                    for (int tt = 0; tt < 4; ++tt) {
                        c4[tt] = acc[g][tt][e];
                        r4[tt] = ldx::F32Row{rows[g].x, rows[g].y, rows[g].z, rows[g].w};
                        cr4[tt] = cand4::F32Row{rows[g].x, rows[g].y, rows[g].z, rows[g].w};
                    }
                    if constexpr (kProd) ldx::ld_multi_f32<4, ldx_k16, false>(c4, fo, r4, cols, o4, wmax, ymin);
                    else if constexpr (VAR == 3) {
#pragma unroll
                        for (int tt = 0; tt < 4; ++tt) o4[tt] = __builtin_bit_cast(ldx_k16, c4[tt]);
                    } else cand4::ld_multi_f32<4, ldx_k16, false>(c4, fc, cr4, ccols, o4, trk);
#pragma unroll
                    for (int tt = 0; tt < 4; ++tt) cell[g * 4 + tt] = o4[tt];
                }
                const bool sure = kProd ? ((wmax < fo.tol) & (ymin > 0.0f)) : (VAR == 3 ? true : cand4::f32_sure(trk, fc));
                if (sure && VAR != 2) {
#pragma unroll
                    for (int m = 0; m < 2; ++m) {
                        ldx_k16 *const row = wbase + ((4u * m + (e >> 2)) * 1024u + (e & 3) * 128u);
                        if constexpr (kWide) {
                            store4_saddr(row, lane_off_b, cell[m * 4 + 0], cell[m * 4 + 1], cell[m * 4 + 2], cell[m * 4 + 3]);
                            continue;
                        }
                        store_saddr<0>(row, lane_off_b, cell[m * 4 + 0]);
                        store_saddr<128>(row, lane_off_b, cell[m * 4 + 1]);
                        store_saddr<256>(row, lane_off_b, cell[m * 4 + 2]);
                        store_saddr<384>(row, lane_off_b, cell[m * 4 + 3]);
                    }
                }
                if (VAR == 2) {   // keep the arithmetic alive
                    uint32_t x = 0;
#pragma unroll
                    for (int q = 0; q < 8; ++q) x ^= __builtin_bit_cast(uint32_t, cell[q]);
                    asm volatile("" : : "v"(x));
                }
                const unsigned long long pk = __ballot(!sure);
                if (pk) np += (unsigned)__builtin_popcountll(pk);
            };
            if constexpr (VAR == 4) {
#pragma unroll
                for (int e = 0; e < 16; ++e) step(e, 0);
            } else {
#pragma unroll 1
                for (int e = 0; e < 16; ++e) step(e, 0);
            }
        }
        t1 = __builtin_amdgcn_s_memtime();
        if (lane == 0) atomicAdd(parked, np);
    } else if (role == ROLE_M) {
        v16f acc[8];
        for (int i = 0; i < 8; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0;
        v4i a = {0x11111111, 0x22222222, 0x11111111, 0x22222222}, b = {0x44444444, 0x22222222, 0x44444444, 0x22222222};
        __syncthreads();
        t0 = __builtin_amdgcn_s_memtime();
        for (int it = 0; it < n_mfma; it += 8) {
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_mfma_f32_32x32x64_f8f6f4 %0, %1, %2, %0 cbsz:4 blgp:4" : "+v"(acc[i]) : "v"(a), "v"(b));
        }
        asm volatile("s_nop 15\n s_nop 15");
        t1 = __builtin_amdgcn_s_memtime();
        float s = 0;
        for (int i = 0; i < 8; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
        if (s == 12345.0f) out[0] = ldx_k16{1, 1};
    } else {
        __syncthreads();
    }
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int VAR>
void run(const char *name, int role_a, int role_b, int n, const float *d_counts, const float *d_rt, const float *d_ct, ldx_k16 *d_out)
{
    const int blocks = 256, units = 60, n_mfma = 60 * 640;
    unsigned long long *cyc; unsigned *parked;
    static unsigned long long h[256 * 8];
    (void)hipMalloc(&cyc, blocks * 8 * 8); (void)hipMalloc(&parked, 4);
    (void)hipMemset(parked, 0, 4);
    const cand4::F32Const fc = cand4::f32_const((double)n);
    const ldx::F32Const fo = ldx::f32_const((double)n);
    for (int rep = 0; rep < 2; ++rep) {
        (void)hipMemset(parked, 0, 4);
        k<VAR><<<blocks, 512>>>(d_counts, d_rt, d_ct, d_out, units, n_mfma, role_a, role_b, fc, fo, cyc, parked);
    }
    const hipError_t err = hipDeviceSynchronize();
    if (err != hipSuccess) { printf("var %d %s: %s\n", VAR, name, hipGetErrorString(err)); exit(1); }
    unsigned hp = 0;
    (void)hipMemcpy(h, cyc, blocks * 8 * 8, hipMemcpyDeviceToHost);
    (void)hipMemcpy(&hp, parked, 4, hipMemcpyDeviceToHost);
    double ea = 0, eb = 0;
    for (int b = 0; b < blocks; ++b)
        for (int w = 0; w < 8; ++w) (w < 4 ? ea : eb) += (double)h[b * 8 + w];
    ea /= blocks * 4; eb /= blocks * 4;
    auto show = [&](int role, double c) {
        if (role == ROLE_E) printf("  E: %8.0f cycles/unit (%6.1f per step)", c / units, c / units / 16);
        else if (role == ROLE_M) printf("  M: %6.1f cycles/MFMA", c / n_mfma);
        else printf("  idle");
    };
    printf("var %d %-34s A", VAR, name); show(role_a, ea); printf("   B"); show(role_b, eb);
    const double lane_steps = blocks * 4.0 * ((role_a == ROLE_E) + (role_b == ROLE_E)) * units * 16 * 64;
    printf("   parked lane-steps %.3f %%\n", lane_steps > 0 ? 100.0 * hp / lane_steps : 0.0);
    (void)hipFree(cyc); (void)hipFree(parked);
}

int main(int argc, char **argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 5008;
    const unsigned long mask = argc > 2 ? strtoul(argv[2], nullptr, 0) : ~0ul;   // bit k = the k-th run below (debugging aid)
    setvbuf(stdout, nullptr, _IONBF, 0);
    int run_no = 0;
#define RUN(V, ...) do { if (mask >> run_no++ & 1ul) run<V>(__VA_ARGS__); } while (0)
    // a plausible unit: 64 rows, 128 columns of ordinary SNPs; counts near independence plus noise
    static float counts[64 * 128], rt[64 * 4], ct[128 * 4];
    static double ra[64], ca[128];
    srand(7);
    for (int i = 0; i < 64; ++i) ra[i] = 1 + rand() % (n - 1);
    for (int j = 0; j < 128; ++j) ca[j] = 1 + rand() % (n - 1);
    const bool int_mode = true;   // the candidate's tables: integer mode at every n (its arithmetic cost is what is measured)
    for (int i = 0; i < 64; ++i) {
        const double a = ra[i], r = n - a;
        rt[i * 4 + 0] = int_mode ? __builtin_bit_cast(float, (int)a) : (float)a;
        rt[i * 4 + 1] = (float)(1e4 / a); rt[i * 4 + 2] = (float)(1e4 / r); rt[i * 4 + 3] = (float)(10.0 / sqrt(a * r));
    }
    for (int j = 0; j < 128; ++j) {
        const double a = ca[j], r = n - a;
        ct[j * 4 + 0] = int_mode ? __builtin_bit_cast(float, -(int)a) : (float)a;
        ct[j * 4 + 1] = (float)(1.0 / a); ct[j * 4 + 2] = (float)(1.0 / r); ct[j * 4 + 3] = (float)(10.0 / sqrt(a * r));
    }
    for (int i = 0; i < 64; ++i)
        for (int j = 0; j < 128; ++j) {
            const double lo = fmax(0.0, ra[i] + ca[j] - n), hi = fmin(ra[i], ca[j]);
            double c = floor(ra[i] * ca[j] / n + (rand() % 2001 - 1000) * 1e-3 * sqrt((double)n) * 0.3);
            counts[i * 128 + j] = (float)fmin(hi, fmax(lo, c));
        }
    // variant 0 reads float a's: a second pair of tables
    static float rt0[64 * 4], ct0[128 * 4];
    for (int i = 0; i < 64 * 4; ++i) rt0[i] = rt[i];
    for (int j = 0; j < 128 * 4; ++j) ct0[j] = ct[j];
    for (int i = 0; i < 64; ++i) rt0[i * 4] = (float)ra[i];
    for (int j = 0; j < 128; ++j) ct0[j * 4] = (float)ca[j];
    float *d_counts, *d_rt, *d_ct, *d_rt0, *d_ct0; ldx_k16 *d_out;
    (void)hipMalloc(&d_counts, sizeof(counts)); (void)hipMalloc(&d_rt, sizeof(rt)); (void)hipMalloc(&d_ct, sizeof(ct));
    (void)hipMalloc(&d_rt0, sizeof(rt)); (void)hipMalloc(&d_ct0, sizeof(ct));
    (void)hipMalloc(&d_out, (size_t)256 * 8 * 8192 * 4);
    (void)hipMemcpy(d_counts, counts, sizeof(counts), hipMemcpyHostToDevice);
    (void)hipMemcpy(d_rt, rt, sizeof(rt), hipMemcpyHostToDevice); (void)hipMemcpy(d_ct, ct, sizeof(ct), hipMemcpyHostToDevice);
    (void)hipMemcpy(d_rt0, rt0, sizeof(rt0), hipMemcpyHostToDevice); (void)hipMemcpy(d_ct0, ct0, sizeof(ct0), hipMemcpyHostToDevice);
    printf("n = %d (%s mode)\n", n, int_mode ? "integer" : "small-n");
    RUN(0, "product tier, alone", ROLE_E, ROLE_IDLE, n, d_counts, d_rt0, d_ct0, d_out);
    RUN(0, "product tier, both waves", ROLE_E, ROLE_E, n, d_counts, d_rt0, d_ct0, d_out);
    RUN(0, "product tier beside MFMAs", ROLE_E, ROLE_M, n, d_counts, d_rt0, d_ct0, d_out);
    RUN(1, "candidate, alone", ROLE_E, ROLE_IDLE, n, d_counts, d_rt, d_ct, d_out);
    RUN(1, "candidate, both waves", ROLE_E, ROLE_E, n, d_counts, d_rt, d_ct, d_out);
    RUN(1, "candidate beside MFMAs", ROLE_E, ROLE_M, n, d_counts, d_rt, d_ct, d_out);
    RUN(2, "candidate, no stores, alone", ROLE_E, ROLE_IDLE, n, d_counts, d_rt, d_ct, d_out);
    RUN(2, "candidate, no stores, both", ROLE_E, ROLE_E, n, d_counts, d_rt, d_ct, d_out);
    RUN(2, "candidate, no stores, beside MFMAs", ROLE_E, ROLE_M, n, d_counts, d_rt, d_ct, d_out);
    RUN(3, "stores only, alone", ROLE_E, ROLE_IDLE, n, d_counts, d_rt, d_ct, d_out);
    RUN(3, "stores only, both", ROLE_E, ROLE_E, n, d_counts, d_rt, d_ct, d_out);
    RUN(5, "product tier, 16-byte stores, alone", ROLE_E, ROLE_IDLE, n, d_counts, d_rt0, d_ct0, d_out);
    RUN(5, "product tier, 16-byte stores, both", ROLE_E, ROLE_E, n, d_counts, d_rt0, d_ct0, d_out);
    RUN(5, "product tier, 16-B stores, beside MFMAs", ROLE_E, ROLE_M, n, d_counts, d_rt0, d_ct0, d_out);
    RUN(6, "candidate, 16-byte stores, alone", ROLE_E, ROLE_IDLE, n, d_counts, d_rt, d_ct, d_out);
    RUN(6, "candidate, 16-byte stores, both", ROLE_E, ROLE_E, n, d_counts, d_rt, d_ct, d_out);
    RUN(1, "MFMAs alone (A idle)", ROLE_IDLE, ROLE_M, n, d_counts, d_rt, d_ct, d_out);
    return 0;
}
