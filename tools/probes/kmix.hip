// Does the matrix pipe keep its rate when the MFMA wave also issues VALU / LDS work (like the K loop) and its SIMD
// partner streams VALU (like the epilogue)?  One 512-thread workgroup per CU: waves 0-3 run FP4 MFMAs with VPM
// independent v_and_b32 and LPM ds_read_b128 per MFMA in between; waves 4-7 stream v_fma_f32 on 8 chains.
// Build: hipcc -O3 --offload-arch=gfx950 kmix.hip -o kmix
#include <hip/hip_runtime.h>
#include <stdio.h>
#ifndef WIDTH
#define WIDTH 128
#endif
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

template <int VPM, int LPM>
__global__ void __launch_bounds__(512) k(int *sink, unsigned long long *cyc, int n_mfma, int n_valu, const v4i *gbuf)
{
    __shared__ v4i lds[1024];
    const int wave = threadIdx.x >> 6;
    lds[threadIdx.x] = v4i{(int)threadIdx.x, 1, 2, 3};
    lds[threadIdx.x + 512] = v4i{(int)threadIdx.x, 4, 5, 6};
    unsigned long long t0 = 0, t1 = 0;
    int s = 0;
    if (wave < 4) {
        v16i acc[8];
        for (int i = 0; i < 8; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0;
        v4i a = {(int)threadIdx.x, 1, 2, 3}, b = {5, (int)threadIdx.x, 7, 8};
        unsigned xi[8];
        for (int c = 0; c < 8; ++c) xi[c] = threadIdx.x + c;
        v4i l = {0, 0, 0, 0};
        v4i cur[4] = {l, l, l, l};
#ifdef GLOADS
        v4i gcur[GLOADS > 0 ? GLOADS : 1];
        for (int r = 0; r < GLOADS; ++r) gcur[r] = l;
#endif
        __syncthreads();
        t0 = __builtin_amdgcn_s_memtime();
        for (int it = 0; it < n_mfma; it += 8) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                asm volatile("v_mfma_f32_32x32x64_f8f6f4 %0, %1, %2, %0 cbsz:4 blgp:4" : "+v"(acc[i]) : "v"(a), "v"(b));
#pragma unroll
                for (int v = 0; v < VPM; ++v) asm volatile("v_and_b32 %0, %0, %1" : "+v"(xi[(i + v) & 7]) : "v"(xi[(i + v + 1) & 7]));
#ifdef GLOADS     // GLOADS global_load_dwordx4 per group of 8 MFMAs (L1 / L2 hits), consumed by the next group
                if (i == 0) {
#pragma unroll
                    for (int r = 0; r < GLOADS; ++r) { l += gcur[r]; }
#pragma unroll
                    for (int r = 0; r < GLOADS; ++r) gcur[r] = gbuf[(threadIdx.x + r * 64 + it * 8 + blockIdx.x * 512) & 4095];
                }
#endif
#ifdef PREFETCH   // like the K loop: the four reads of a group of 8 MFMAs are consumed by the NEXT group
                if (LPM && i == 0) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) { l += cur[r]; }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
#if WIDTH == 128
                        cur[r] = lds[(threadIdx.x + r * 64 + it) & 1023];
#elif WIDTH == 64
                        const long long t = reinterpret_cast<const long long *>(lds)[(threadIdx.x + r * 64 + it) & 2047];
                        cur[r].x = (int)t; cur[r].y = (int)(t >> 32);
#else
                        cur[r].x = reinterpret_cast<const int *>(lds)[(threadIdx.x + r * 64 + it) & 4095];
#endif
                    }
                }
#else
                if (LPM && (i % (8 / (LPM > 8 ? 8 : LPM))) == 0) {
                    v4i t = lds[(threadIdx.x + i * 64) & 1023];
                    asm volatile("" : "+v"(t));
                    l += t;
                }
#endif
            }
        }
        asm volatile("s_nop 15\n s_nop 15");
        t1 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < 8; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
        for (int c = 0; c < 8; ++c) s += (int)xi[c];
        s += l.x + l.y;
    } else {
        float xf[8];
        for (int c = 0; c < 8; ++c) xf[c] = 1.25f + threadIdx.x * 1e-3f + c;
        const float a = 1.0000001f, b = 0.3f;
        __syncthreads();
        t0 = __builtin_amdgcn_s_memtime();
        for (int it = 0; it < n_valu; it += 8) {
#pragma unroll
            for (int c = 0; c < 8; ++c) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(xf[c]) : "v"(a), "v"(b));
        }
        t1 = __builtin_amdgcn_s_memtime();
        for (int c = 0; c < 8; ++c) s += (int)xf[c];
    }
    sink[blockIdx.x * 512 + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int VPM, int LPM>
void run(int n_mfma, int n_valu)
{
    const int blocks = 256;
    int *sink; unsigned long long *cyc; static unsigned long long h[256 * 8];
    v4i *gbuf; (void)hipMalloc(&gbuf, 4096 * 16); (void)hipMemset(gbuf, 1, 4096 * 16);
    (void)hipMalloc(&sink, blocks * 512 * 4); (void)hipMalloc(&cyc, blocks * 64);
    for (int rep = 0; rep < 2; ++rep) k<VPM, LPM><<<blocks, 512>>>(sink, cyc, n_mfma, n_valu, gbuf);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(h, cyc, blocks * 64, hipMemcpyDeviceToHost);
    double m = 0, v = 0;
    for (int b = 0; b < blocks; ++b) {
        for (int w = 0; w < 4; ++w) m += (double)h[b * 8 + w];
        for (int w = 4; w < 8; ++w) v += (double)h[b * 8 + w];
    }
    printf("MFMA wave with %d v_and + %d ds_read_b128 per 8 MFMAs x8 | partner VALU %6d : %6.1f cycles per MFMA, %5.2f cycles per partner VALU\n",
           VPM * 8, LPM, n_valu, n_mfma ? m / (blocks * 4) / n_mfma : 0.0, n_valu ? v / (blocks * 4) / n_valu : 0.0);
    (void)hipFree(sink); (void)hipFree(cyc);
}

int main()
{
#ifdef GLOADS
    printf("global_load_dwordx4 per 8 MFMAs: %d, ds_read width %d bits\n", GLOADS, WIDTH);
#endif
    run<2, 0>(2048, 0);
    run<4, 0>(2048, 0);
    run<4, 0>(2048, 10240);
    run<2, 4>(2048, 0);
    return 0;
}
