// Issue-cost probe for the VALU instructions of the LD epilogue (gfx950).  One wave per SIMD, 8 independent
// chains per op, cycles per instruction from s_memtime.  Build: hipcc -O3 --offload-arch=gfx950 oprate.hip -o oprate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CHAINS 8
#define ITERS 512

template <int OP>
__global__ void __launch_bounds__(256) k(double *sink, unsigned long long *cyc, double seed)
{
    double x[CHAINS];
    float xf[CHAINS];
    unsigned xi[CHAINS];
    for (int c = 0; c < CHAINS; ++c) {
        x[c] = seed + threadIdx.x * 1e-3 + c;
        xf[c] = (float)x[c];
        xi[c] = threadIdx.x + c;
    }
    const double a = seed * 1.0000001, b = seed * 0.3;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) {
            if (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(xf[c]) : "v"((float)a), "v"((float)b));
            if (OP == 1) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x[c]) : "v"(a), "v"(b));
            if (OP == 2) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x[c]) : "v"(a));
            if (OP == 3) asm volatile("v_add_f64 %0, %0, %1" : "+v"(x[c]) : "v"(b));
            if (OP == 4) asm volatile("v_rcp_f64 %0, %0" : "+v"(x[c]));
            if (OP == 5) asm volatile("v_rndne_f64 %0, %0" : "+v"(x[c]));
            if (OP == 6) asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(x[c]) : "v"(xi[c]));
            if (OP == 7) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(xf[c]) : "v"(x[c]));
            if (OP == 8) asm volatile("v_min_f64 %0, %0, %1" : "+v"(x[c]) : "v"(a));
            if (OP == 9) asm volatile("v_cmp_lt_f64 vcc, %0, %1\n v_cndmask_b32 %2, %2, %3, vcc" : : "v"(x[c]), "v"(a), "v"(xi[c]), "v"(xi[(c + 1) % CHAINS]) : "vcc");
            if (OP == 10) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(xi[c]) : "v"(xi[(c + 1) % CHAINS]) : "vcc");
            if (OP == 11) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(xi[c]) : "v"(xi[(c + 1) % CHAINS]));
            if (OP == 12) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(xi[c]) : "v"(xi[(c + 1) % CHAINS]));
            if (OP == 13) asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(x[c]) : "v"(xi[c]));
            if (OP == 14) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(xi[c]) : "v"(xi[(c + 1) % CHAINS]), "v"(0x05010400u));
            if (OP == 15) asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(xi[c]) : "v"(x[c]));
            if (OP == 16) asm volatile("v_max_f64 %0, |%0|, |%1|" : "+v"(x[c]) : "v"(a));
            if (OP == 17) asm volatile("v_cvt_f32_u32 %0, %1" : "=v"(xf[c]) : "v"(xi[c]));
            if (OP == 18) asm volatile("v_rcp_f32 %0, %0" : "+v"(xf[c]));
            if (OP == 19) asm volatile("v_bfe_u32 %0, %0, 4, 4" : "+v"(xi[c]));
            if (OP == 20) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(xi[c]) : "v"(xi[(c + 1) % CHAINS]), "v"(xi[(c + 2) % CHAINS]));
            if (OP == 21) asm volatile("v_cmp_class_f64 vcc, %0, %1" : : "v"(x[c]), "v"(0x204u) : "vcc");
            if (OP == 22) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(x[c]) : "v"(a));
            if (OP == 23) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x[c]) : "v"(a), "v"(b));
            if (OP == 24) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(xi[c]) : "v"(xi[(c + 1) % CHAINS]), "v"(xi[(c + 2) % CHAINS]));
            if (OP == 25) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(x[c]) : "v"(xf[c]));
            if (OP == 26) asm volatile("v_ldexp_f64 %0, %0, %1" : "+v"(x[c]) : "v"(xi[c]));
            if (OP == 27) asm volatile("v_floor_f64 %0, %0" : "+v"(x[c]));
            if (OP == 28) asm volatile("v_fract_f64 %0, %0" : "+v"(x[c]));
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
    for (int c = 0; c < CHAINS; ++c) s += x[c] + xf[c] + xi[c];
    sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int OP>
void run(const char *name, int waves_per_simd)
{
    const int blocks = 256 * waves_per_simd;
    double *sink;
    unsigned long long *cyc, h[256 * 4 * 4];
    hipMalloc(&sink, blocks * 256 * 8);
    hipMalloc(&cyc, blocks * 4 * 8);
    k<OP><<<blocks, 256>>>(sink, cyc, 1.25);
    k<OP><<<blocks, 256>>>(sink, cyc, 1.25);
    hipDeviceSynchronize();
    hipMemcpy(h, cyc, blocks * 4 * 8, hipMemcpyDeviceToHost);
    double tot = 0;
    for (int i = 0; i < blocks * 4; ++i) tot += (double)h[i];
    printf("%-22s waves/SIMD=%d  cycles per wave-instruction: %.2f\n", name, waves_per_simd,
           tot / (blocks * 4) / (ITERS * CHAINS) / (OP == 9 ? 2 : 1) * 1.0);
    hipFree(sink);
    hipFree(cyc);
}

int main()
{
#define R(op, name) run<op>(name, 1); run<op>(name, 2);
    R(0, "v_fma_f32") R(1, "v_fma_f64") R(2, "v_mul_f64") R(3, "v_add_f64") R(4, "v_rcp_f64") R(5, "v_rndne_f64")
    R(6, "v_cvt_f64_u32") R(7, "v_cvt_f32_f64") R(8, "v_min_f64") R(9, "v_cmp_f64+cndmask(/2)") R(10, "v_cndmask_b32")
    R(11, "v_mul_u32_u24") R(12, "v_mul_lo_u32") R(13, "v_cvt_f64_i32") R(14, "v_perm_b32") R(15, "v_cvt_i32_f64")
    R(16, "v_max_f64 |a|,|b|") R(17, "v_cvt_f32_u32") R(18, "v_rcp_f32") R(19, "v_bfe_u32") R(20, "v_and_or_b32")
    R(21, "v_cmp_class_f64") R(22, "v_pk_mul_f32") R(23, "v_pk_fma_f32") R(24, "v_mad_u32_u24") R(25, "v_cvt_f64_f32")
    R(26, "v_ldexp_f64") R(27, "v_floor_f64") R(28, "v_fract_f64")
    return 0;
}
