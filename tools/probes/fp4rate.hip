// FP4 (E2M1) operands on the block-scaled matrix instruction v_mfma_scale_f32_32x32x64_f8f6f4 as a {0,1} co-occurrence
// counter: (1) exactness and fragment layout on random bit rows, expanded exactly as the kernel would
// (A nibble = one bit at position p in {0,1,2} = 0.5 / 1 / 2, B nibble = the bit at position 2 - p, product 1);
// (2) cycles per instruction and wall-clock rate against v_mfma_i32_32x32x32_i8, 1 and 2 waves per SIMD.
// Build: hipcc -O3 --offload-arch=gfx950 fp4rate.hip -o fp4rate
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef float v16f __attribute__((ext_vector_type(16)));

// 32 haplotype bits -> 4 registers of 8 FP4 nibbles.  Register v holds haplotypes {v, v+4, ..., v+28}; the K order
// inside an instruction is free as long as both operands use the same one.
__device__ __host__ inline void expand_a(uint32_t w, uint32_t out[4])   // as expand32_a4 in csrc/ldx_mfma.hip
{
    const uint32_t t = w >> 2;
    out[0] = w & 0x11111111u;   // haplotypes 0, 4, ...: 0.5
    out[1] = w & 0x22222222u;   // 1, 5, ...: 1
    out[2] = t & 0x11111111u;   // 2, 6, ...: 0.5
    out[3] = t & 0x22222222u;   // 3, 7, ...: 1
}
__device__ __host__ inline void expand_b(uint32_t w, uint32_t out[4])   // as expand32_b4
{
    const uint32_t t = w >> 2;
    out[0] = (w << 2) & 0x44444444u;   // 2
    out[1] = w & 0x22222222u;          // 1
    out[2] = w & 0x44444444u;          // 2
    out[3] = t & 0x22222222u;          // 1
}

template <int SCALE>
__device__ inline v16f mfma_fp4(v4i a, v4i b, v16f c)
{
    const v8i av = {a.x, a.y, a.z, a.w, 0, 0, 0, 0}, bv = {b.x, b.y, b.z, b.w, 0, 0, 0, 0};
    return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, c, 4, 4, 0, SCALE, 0, SCALE);
}

// rows: [32][2] words for A (row r, half h = haplotypes 32h..32h+31 of the 64-haplotype step), same for B
template <int SCALE>
__global__ void check_kernel(const uint32_t *arow, const uint32_t *brow, float *out)
{
    const uint32_t lane = threadIdx.x, l32 = lane & 31u, half = lane >> 5;
    uint32_t ea[4], eb[4];
    expand_a(arow[l32 * 2 + half], ea);
    expand_b(brow[l32 * 2 + half], eb);
    v16f c = {};
    c = mfma_fp4<SCALE>(v4i{(int)ea[0], (int)ea[1], (int)ea[2], (int)ea[3]},
                        v4i{(int)eb[0], (int)eb[1], (int)eb[2], (int)eb[3]}, c);
    for (int e = 0; e < 16; ++e) {
        const uint32_t i = (e & 3) + 8u * (e >> 2) + 4u * half, j = l32;   // the documented 32x32 C/D map
        out[i * 32 + j] = c[e];
    }
}

#define ITERS 512
template <int KIND>   // 0: i8 32x32x32, 1: fp4 scaled (scale 127), 2: fp4 scale operand 0
__global__ void __launch_bounds__(256) rate_kernel(float *sink, unsigned long long *cyc, const uint32_t *seed)
{
    v16f accf[8];
    v16i acci[8];
    for (int i = 0; i < 8; ++i)
        for (int e = 0; e < 16; ++e) { accf[i][e] = 0.f; acci[i][e] = 0; }
    uint32_t ea[2][4], eb[4][4];
    for (int m = 0; m < 2; ++m) expand_a(seed[(threadIdx.x + 64 * m + blockIdx.x) & 1023], ea[m]);
    for (int t = 0; t < 4; ++t) expand_b(seed[(threadIdx.x * 7 + 131 * t + blockIdx.x) & 1023], eb[t]);
    v4i a[2], b[4];
    for (int m = 0; m < 2; ++m) a[m] = v4i{(int)ea[m][0], (int)ea[m][1], (int)ea[m][2], (int)ea[m][3]};
    for (int t = 0; t < 4; ++t) b[t] = v4i{(int)eb[t][0], (int)eb[t][1], (int)eb[t][2], (int)eb[t][3]};
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (KIND == 0) acci[i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[i >> 2], b[i & 3], acci[i], 0, 0, 0);
            if (KIND == 1) accf[i] = mfma_fp4<127>(a[i >> 2], b[i & 3], accf[i]);
            if (KIND == 2) accf[i] = mfma_fp4<0>(a[i >> 2], b[i & 3], accf[i]);
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 8; ++i)
        for (int e = 0; e < 16; ++e) s += accf[i][e] + (float)acci[i][e];
    sink[blockIdx.x * 256 + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int KIND>
static void rate(const char *name, int wps, const uint32_t *dseed)
{
    const int blocks = 256 * wps;
    float *sink;
    unsigned long long *cyc;
    static unsigned long long h[8192];
    (void)hipMalloc(&sink, blocks * 256 * 4);
    (void)hipMalloc(&cyc, blocks * 32);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    for (int w = 0; w < 20; ++w) rate_kernel<KIND><<<blocks, 256>>>(sink, cyc, dseed);
    (void)hipEventRecord(e0);
    const int reps = 50;
    for (int w = 0; w < reps; ++w) rate_kernel<KIND><<<blocks, 256>>>(sink, cyc, dseed);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipMemcpy(h, cyc, blocks * 32, hipMemcpyDeviceToHost);
    double tot = 0;
    for (int i = 0; i < blocks * 4; ++i) tot += (double)h[i];
    const double n_mfma = (double)blocks * 4 * ITERS * 8 * reps;
    const double macs = KIND == 0 ? 32768.0 : 65536.0;
    printf("%-28s waves/SIMD=%d : %.1f cycles per MFMA per wave; wall %.3f ms -> %.0f TOP/s (2 ops per MAC)\n", name, wps,
           tot / (blocks * 4) / (ITERS * 8), ms / reps, n_mfma * macs * 2 / (ms * 1e-3) / 1e12);
    (void)hipFree(sink);
    (void)hipFree(cyc);
}

template <int SCALE>
static int check(const char *name)
{
    uint32_t ha[64], hb[64];
    srand(12345);
    for (int i = 0; i < 64; ++i) { ha[i] = (uint32_t)rand() ^ ((uint32_t)rand() << 16); hb[i] = (uint32_t)rand() ^ ((uint32_t)rand() << 16); }
    ha[0] = ha[1] = 0xFFFFFFFFu; hb[0] = hb[1] = 0xFFFFFFFFu;   // a full row pair: count 64
    uint32_t *da, *db;
    float *dout, hout[1024];
    (void)hipMalloc(&da, 256); (void)hipMalloc(&db, 256); (void)hipMalloc(&dout, 4096);
    (void)hipMemcpy(da, ha, 256, hipMemcpyHostToDevice);
    (void)hipMemcpy(db, hb, 256, hipMemcpyHostToDevice);
    check_kernel<SCALE><<<1, 64>>>(da, db, dout);
    (void)hipMemcpy(hout, dout, 4096, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 32; ++i)
        for (int j = 0; j < 32; ++j) {
            const int want = __builtin_popcount(ha[2 * i] & hb[2 * j]) + __builtin_popcount(ha[2 * i + 1] & hb[2 * j + 1]);
            if (hout[i * 32 + j] != (float)want) {
                if (bad < 5) printf("  mismatch (%d,%d): got %g want %d\n", i, j, hout[i * 32 + j], want);
                ++bad;
            }
        }
    printf("%s: %d of 1024 cells wrong (cell (0,0) = %g, expected 64)\n", name, bad, hout[0]);
    return bad;
}

int main()
{
    int bad = check<127>("fp4 counts, scale operands 127 (x1)");
    check<0>("fp4 counts, scale operands 0");
    uint32_t hs[1024], *ds;
    for (int i = 0; i < 1024; ++i) hs[i] = (uint32_t)rand() ^ ((uint32_t)rand() << 16);
    (void)hipMalloc(&ds, 4096);
    (void)hipMemcpy(ds, hs, 4096, hipMemcpyHostToDevice);
    for (int wps = 1; wps <= 2; ++wps) {
        rate<0>("v_mfma_i32_32x32x32_i8", wps, ds);
        rate<1>("v_mfma_scale_f32_32x32x64 fp4", wps, ds);
        rate<2>("  same, scale operands 0", wps, ds);
    }
    return bad ? 1 : 0;
}
