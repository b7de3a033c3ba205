// How fast does ONE wave issue VALU work?  A single wave per SIMD (256-thread workgroups, one per CU) runs a stream of
// v_fma_f32 organised as C independent dependent-chains (C = 1..16), or a mix of VALU and SALU / LDS instructions.
// cycles per instruction = s_memtime delta / instructions.  If the figure falls with C the wave is bound by the
// latency of dependent instructions; if it stays, by its issue rate.  Second part: 2, 3, 4 waves per SIMD (more
// workgroups per CU) running the same 8-chain stream: cycles per instruction per wave and per SIMD.
// Build: hipcc -O3 --offload-arch=gfx950 issue.hip -o issue
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int C, int MIX>
__global__ void __launch_bounds__(256) k(float *sink, unsigned long long *cyc, int n)
{
    float x[16];
    for (int c = 0; c < 16; ++c) x[c] = 1.25f + threadIdx.x * 1e-3f + c;
    const float a = 1.0000001f, b = 0.3f;
    int s0 = __builtin_amdgcn_readfirstlane(blockIdx.x);
    __shared__ float lds[256];
    lds[threadIdx.x] = x[0];
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < n; it += 16) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[j % C]) : "v"(a), "v"(b));
            if (MIX == 1) asm volatile("s_add_i32 %0, %0, 1" : "+s"(s0));                 // one SALU per VALU
            if (MIX == 2 && (j & 3) == 0) asm volatile("s_add_i32 %0, %0, 1" : "+s"(s0));   // one SALU per four VALU
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int c = 0; c < 16; ++c) s += x[c];
    sink[blockIdx.x * 256 + threadIdx.x] = s + s0;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int C, int MIX>
void run(const char *name, int wgs_per_cu)
{
    const int blocks = 256 * wgs_per_cu, n = 16384;
    float *sink; unsigned long long *cyc; static unsigned long long h[256 * 8 * 4];
    (void)hipMalloc(&sink, blocks * 256 * 4); (void)hipMalloc(&cyc, blocks * 32);
    for (int rep = 0; rep < 2; ++rep) k<C, MIX><<<blocks, 256>>>(sink, cyc, n);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(h, cyc, blocks * 32, hipMemcpyDeviceToHost);
    double m = 0;
    for (int i = 0; i < blocks * 4; ++i) m += (double)h[i];
    const double per = m / (blocks * 4) / n;
    printf("%-34s waves/SIMD=%d : %5.2f cycles per VALU per wave, %5.2f per SIMD\n", name, wgs_per_cu, per, per / wgs_per_cu);
    (void)hipFree(sink); (void)hipFree(cyc);
}

int main()
{
    run<1, 0>("v_fma_f32, 1 chain", 1);
    run<2, 0>("v_fma_f32, 2 chains", 1);
    run<4, 0>("v_fma_f32, 4 chains", 1);
    run<8, 0>("v_fma_f32, 8 chains", 1);
    run<16, 0>("v_fma_f32, 16 chains", 1);
    run<8, 1>("8 chains + 1 SALU per VALU", 1);
    run<8, 2>("8 chains + 1 SALU per 4 VALU", 1);
    run<8, 0>("v_fma_f32, 8 chains", 2);
    run<8, 0>("v_fma_f32, 8 chains", 3);
    run<8, 0>("v_fma_f32, 8 chains", 4);
    run<8, 0>("v_fma_f32, 8 chains", 8);
    run<1, 0>("v_fma_f32, 1 chain", 4);
    return 0;
}
