// Do a K-loop wave's MFMAs and another wave's VALU work overlap on one SIMD?  One 512-thread workgroup per CU:
// waves 0-3 (one per SIMD) issue back-to-back v_mfma_i32_32x32x32_i8 on 8 accumulators, waves 4-7 (their SIMD
// partners) a stream of one VALU opcode on 8 independent chains.  Cycles per instruction for each role, alone and
// together.  Build: hipcc -O3 --offload-arch=gfx950 overlap.hip -o overlap
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef int v8i __attribute__((ext_vector_type(8)));

template <int OP>
__global__ void __launch_bounds__(512) k(int *sink, unsigned long long *cyc, int n_mfma, int n_valu, unsigned long long *simd_id)
{
    const int wave = threadIdx.x >> 6;
    unsigned long long t0 = 0, t1 = 0;
    int s = 0;
    if (wave < 4) {
        v16i acc[8];
        for (int i = 0; i < 8; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0;
        v4i a = {(int)threadIdx.x, 1, 2, 3}, b = {5, (int)threadIdx.x, 7, 8};
#ifdef FP4_MFMA
        v8i a8 = {0x11111111, 0x22222222, 0x11111111, 0x22222222, 0, 0, 0, 0}, b8 = {0x44444444, 0x22222222, 0x44444444, 0x22222222, 0, 0, 0, 0};
#endif
        __syncthreads();
        t0 = __builtin_amdgcn_s_memtime();
        for (int it = 0; it < n_mfma; it += 8) {
#pragma unroll
#ifdef FP4_MFMA
            for (int i = 0; i < 8; ++i) asm volatile("v_mfma_f32_32x32x64_f8f6f4 %0, %1, %2, %0 cbsz:4 blgp:4" : "+v"(acc[i]) : "v"(a), "v"(b));
#else
            for (int i = 0; i < 8; ++i) asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
#endif
        }
        asm volatile("s_nop 15\n s_nop 15");
        t1 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < 8; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
    } else {
        double x[8];
        float xf[8];
        unsigned xi[8];
        for (int c = 0; c < 8; ++c) { x[c] = 1.25 + threadIdx.x * 1e-3 + c; xf[c] = (float)x[c]; xi[c] = threadIdx.x + c; }
        const double a = 1.0000001, b = 0.3;
        __syncthreads();
        t0 = __builtin_amdgcn_s_memtime();
        for (int it = 0; it < n_valu; it += 8) {
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                if (OP == 0) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x[c]) : "v"(a), "v"(b));
                if (OP == 1) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(xf[c]) : "v"((float)a), "v"((float)b));
                if (OP == 2) asm volatile("v_and_b32 %0, %0, %1" : "+v"(xi[c]) : "v"(xi[(c + 1) & 7]));
                if (OP == 3) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x[c]) : "v"(a), "v"(b));
                if (OP == 4) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x[c]) : "v"(a));
                if (OP == 5) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(xf[c]) : "v"(x[c]));
                if (OP == 6) asm volatile("v_rndne_f64 %0, %0" : "+v"(x[c]));
                if (OP == 7) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(xi[c]) : "v"(xi[(c + 1) & 7]), "v"(0x05010400u));
                if (OP == 8) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(xi[c]) : "v"(xi[(c + 1) & 7]));
            }
        }
        t1 = __builtin_amdgcn_s_memtime();
        for (int c = 0; c < 8; ++c) s += (int)x[c] + (int)xf[c] + (int)xi[c];
    }
    sink[blockIdx.x * 512 + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) {
        cyc[blockIdx.x * 8 + wave] = t1 - t0;
        simd_id[blockIdx.x * 8 + wave] = (__builtin_amdgcn_s_getreg(63492) >> 4) & 3;   // HW_ID.SIMD_ID
    }
}

template <int OP>
void run(const char *name, int n_mfma, int n_valu)
{
    const int blocks = 256;
    int *sink; unsigned long long *cyc, *sid; static unsigned long long h[256 * 8], hs[256 * 8];
    (void)hipMalloc(&sink, blocks * 512 * 4); (void)hipMalloc(&cyc, blocks * 64); (void)hipMalloc(&sid, blocks * 64);
    for (int rep = 0; rep < 2; ++rep) k<OP><<<blocks, 512>>>(sink, cyc, n_mfma, n_valu, sid);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(h, cyc, blocks * 64, hipMemcpyDeviceToHost);
    (void)hipMemcpy(hs, sid, blocks * 64, hipMemcpyDeviceToHost);
    double m = 0, v = 0; int paired = 0;
    for (int b = 0; b < blocks; ++b) {
        for (int w = 0; w < 4; ++w) m += (double)h[b * 8 + w];
        for (int w = 4; w < 8; ++w) v += (double)h[b * 8 + w];
        for (int w = 0; w < 4; ++w) for (int u = 4; u < 8; ++u) paired += hs[b * 8 + w] == hs[b * 8 + u];
    }
    printf("%-14s mfma=%6d valu=%6d : %6.1f cycles per MFMA, %5.2f cycles per VALU   (wave durations %8.0f / %8.0f; SIMD partners found %d of %d)\n", name,
           n_mfma, n_valu, n_mfma ? m / (blocks * 4) / n_mfma : 0.0, n_valu ? v / (blocks * 4) / n_valu : 0.0,
           m / (blocks * 4), v / (blocks * 4), paired, blocks * 4);
    (void)hipFree(sink); (void)hipFree(cyc); (void)hipFree(sid);
}

int main()
{
    run<0>("mfma alone", 2048, 0);
#define BOTH(OP, NAME, NV) run<OP>(NAME " alone", 0, NV); run<OP>(NAME " + mfma", 2048, NV);
    BOTH(0, "v_fma_f64", 12288)
    BOTH(4, "v_mul_f64", 12288)
    BOTH(6, "v_rndne_f64", 12288)
    BOTH(5, "v_cvt_f32_f64", 8192)
    BOTH(1, "v_fma_f32", 12288)
    BOTH(3, "v_pk_fma_f32", 12288)
    BOTH(2, "v_and_b32", 12288)
    BOTH(7, "v_perm_b32", 12288)
    BOTH(8, "v_mul_u32_u24", 12288)
    return 0;
}
