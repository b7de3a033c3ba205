// Cycles per v_mfma_i32_32x32x32_i8 with the accumulators in AGPRs vs ArchVGPRs, bare and with VALU fillers.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
#define ITERS 256

template <int ACC_A, int FILL, int FRESH>
__global__ void __launch_bounds__(256) k(int *sink, unsigned long long *cyc)
{
    v16i acc[8];
    for (int i = 0; i < 8; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0;
    v4i a[2] = {{(int)threadIdx.x, 1, 2, 3}, {5, (int)threadIdx.x, 7, 8}};
    v4i b[4] = {{1, 1, (int)threadIdx.x, 1}, {2, 2, 2, (int)threadIdx.x}, {3, 1, 3, 1}, {4, 1, 1, 4}};
    unsigned x[8];
    for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * (i + 3);
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (ACC_A) asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a[i >> 2]), "v"(b[i & 3]));
            else       asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a[i >> 2]), "v"(b[i & 3]));
#pragma unroll
            for (int f = 0; f < FILL; ++f) {
                if (FRESH && f < 4)   // the filler writes the NEXT mfma's operands, like the bit expansion does
                    asm volatile("v_and_b32 %0, %1, %2" : "=v"(a[((i + 1) & 7) >> 2][f]) : "v"(x[f]), "v"(x[(f + i) & 7]));
                else
                    asm volatile("v_and_b32 %0, %0, %1" : "+v"(x[(f + i) & 7]) : "v"(x[(f + i + 1) & 7]));
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    int s = 0;
    for (int i = 0; i < 8; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
    for (int i = 0; i < 8; ++i) s += x[i];
    sink[blockIdx.x * 256 + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int ACC_A, int FILL, int FRESH>
void run(int wps)
{
    const int blocks = 256 * wps;
    int *sink; unsigned long long *cyc; static unsigned long long h[4096];
    (void)hipMalloc(&sink, blocks * 256 * 4); (void)hipMalloc(&cyc, blocks * 32);
    k<ACC_A, FILL, FRESH><<<blocks, 256>>>(sink, cyc);
    k<ACC_A, FILL, FRESH><<<blocks, 256>>>(sink, cyc);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(h, cyc, blocks * 32, hipMemcpyDeviceToHost);
    double tot = 0; for (int i = 0; i < blocks * 4; ++i) tot += (double)h[i];
    printf("acc in %s  fillers/mfma=%d fresh_operands=%d waves/SIMD=%d : %.1f cycles per MFMA (per wave)\n", ACC_A ? "AGPR" : "VGPR", FILL, FRESH, wps,
           tot / (blocks * 4) / (ITERS * 8));
    (void)hipFree(sink); (void)hipFree(cyc);
}

int main()
{
    run<1, 0, 0>(1); run<0, 0, 0>(1); run<1, 0, 0>(2); run<0, 0, 0>(2);
    run<1, 4, 0>(1); run<0, 4, 0>(1); run<1, 4, 1>(1); run<0, 4, 1>(1);
    run<1, 5, 1>(1); run<0, 5, 1>(1); run<1, 6, 0>(1); run<0, 6, 0>(1);
    run<1, 4, 1>(2); run<0, 4, 1>(2);
    return 0;
}
