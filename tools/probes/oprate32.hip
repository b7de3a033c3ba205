// Throughput of the fp32 epilogue tier's VALU instructions (and of candidates for it) per SIMD at 1, 2 and 4 waves per
// SIMD: a single wave issues at most one instruction per ~4 cycles whatever the instruction, so what an instruction costs
// the SIMD's VALU shows only when several waves compete.  8 independent chains per op, cycles from s_memtime.
// Build: hipcc -O3 --offload-arch=gfx950 oprate32.hip -o oprate32
#include <hip/hip_runtime.h>
#include <stdio.h>

#define CHAINS 8
#define ITERS 1024

template <int OP>
__global__ void __launch_bounds__(256) k(float *sink, unsigned long long *cyc, float seed)
{
    float xf[CHAINS];
    double xd[CHAINS];
    unsigned xi[CHAINS], yi[CHAINS];
    const unsigned ca = threadIdx.x * 3u + 1u, cb = threadIdx.x * 5u + 2u;

    for (int c = 0; c < CHAINS; ++c) {
        yi[c] = threadIdx.x * 7u + c;
        xf[c] = seed + threadIdx.x * 1e-3f + c;
        xd[c] = xf[c];
        xi[c] = threadIdx.x + c;
    }
    const float a = seed * 1.0000001f, b = seed * 0.3f;
    const double ad = a;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) {
            const int d = (c + 1) % CHAINS, e = (c + 2) % CHAINS;
            if (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(xf[c]) : "v"(a), "v"(b));
            if (OP == 1) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(xf[c]) : "v"(a));
            if (OP == 2) asm volatile("v_add_f32 %0, 0x4b000000, %0" : "+v"(xf[c]));
            if (OP == 3) asm volatile("v_max_f32 %0, %0, %1" : "+v"(xf[c]) : "v"(xf[d]));
            if (OP == 4) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(xf[c]) : "v"(xf[d]), "v"(xf[e]));
            if (OP == 5) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(xf[c]) : "v"(xf[d]), "v"(xf[e]));
            if (OP == 6) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(xi[c]) : "v"(xi[d]) : "vcc");
            if (OP == 7) asm volatile("v_cmp_gt_f32 vcc, 0, %0" : : "v"(xf[c]) : "vcc");
            if (OP == 8) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(xi[c]) : "v"(xi[d]), "v"(0x05040100u));
            if (OP == 9) asm volatile("v_cvt_f32_i32 %0, %1" : "=v"(xf[c]) : "v"(xi[c]));
            if (OP == 10) asm volatile("v_cvt_u32_f32 %0, %1" : "=v"(xi[c]) : "v"(xf[c]));
            if (OP == 11) asm volatile("v_fract_f32 %0, %0" : "+v"(xf[c]));
            if (OP == 12) asm volatile("v_mul_i32_i24 %0, %0, %1" : "+v"(xi[c]) : "v"(xi[d]));
            if (OP == 13) asm volatile("v_mad_i32_i24 %0, %0, %1, %2" : "+v"(xi[c]) : "v"(xi[d]), "v"(xi[e]));
            if (OP == 14) asm volatile("v_lshl_or_b32 %0, %0, 16, %1" : "+v"(xi[c]) : "v"(xi[d]));
            if (OP == 15) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(xd[c]) : "v"(ad));
            if (OP == 16) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(xd[c]) : "v"(ad));
            if (OP == 17) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(xd[c]) : "v"(ad));
            if (OP == 18) asm volatile("v_fma_f32 %0, %0, %1, |%2|" : "+v"(xf[c]) : "v"(a), "v"(xf[d]));
            if (OP == 19) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(xf[c]) : "v"(xf[d]));
            if (OP == 20) asm volatile("v_rndne_f32 %0, %0" : "+v"(xf[c]));
            if (OP == 21) asm volatile("v_mov_b32 %0, %1" : "=v"(xi[c]) : "v"(xi[d]));
            if (OP == 22) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(xd[c]) : "v"(ad));
            if (OP == 23) asm volatile("v_and_b32 %0, %0, %1" : "+v"(xi[c]) : "v"(xi[d]));
            // the sign-select of the tier, as the product does it (compare + two selects on VCC) and without VCC
            if (OP == 24) asm volatile("v_cmp_gt_f32 vcc, 0, %2\n v_cndmask_b32 %0, %3, %4, vcc\n v_cndmask_b32 %1, %4, %3, vcc"
                                       : "=v"(xi[c]), "=v"(yi[c]) : "v"(xf[c]), "v"(ca), "v"(cb) : "vcc");
            if (OP == 25) asm volatile("v_ashrrev_i32 %0, 31, %2\n v_bfi_b32 %1, %0, %3, %4\n v_bfi_b32 %0, %0, %4, %3"
                                       : "=&v"(xi[c]), "=&v"(yi[c]) : "v"(xf[c]), "v"(ca), "v"(cb));
            if (OP == 26) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(xi[c]) : "v"(ca), "v"(cb) : "vcc");
            if (OP == 27) asm volatile("v_bfi_b32 %0, %1, %2, %3" : "=v"(xi[c]) : "v"(yi[c]), "v"(ca), "v"(cb));
            if (OP == 28) asm volatile("v_ashrrev_i32 %0, 31, %1" : "=v"(xi[c]) : "v"(xf[c]));
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int c = 0; c < CHAINS; ++c) s += xf[c] + (float)xd[c] + xi[c] + yi[c];
    sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int OP>
void run(const char *name)
{
    printf("%-22s", name);
    for (int w : {1, 2, 4}) {
        const int blocks = 256 * w;
        float *sink;
        unsigned long long *cyc;
        static unsigned long long h[256 * 4 * 4];
        (void)hipMalloc(&sink, blocks * 256 * 4);
        (void)hipMalloc(&cyc, blocks * 4 * 8);
        k<OP><<<blocks, 256>>>(sink, cyc, 1.25f);
        k<OP><<<blocks, 256>>>(sink, cyc, 1.25f);
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(h, cyc, blocks * 4 * 8, hipMemcpyDeviceToHost);
        double tot = 0;
        for (int i = 0; i < blocks * 4; ++i) tot += (double)h[i];
        const double per_wave = tot / (blocks * 4) / (ITERS * CHAINS);
        printf("  %d/SIMD: %5.2f per wave = %5.2f per SIMD", w, per_wave, per_wave / w);
        (void)hipFree(sink);
        (void)hipFree(cyc);
    }
    printf("\n");
}

int main()
{
    setvbuf(stdout, nullptr, _IONBF, 0);
    printf("cycles per wave-instruction (loop overhead included: ~0.3 per instruction)\n");
    run<0>("v_fma_f32"); run<1>("v_mul_f32"); run<2>("v_add_f32 literal"); run<19>("v_sub_f32"); run<3>("v_max_f32");
    run<4>("v_max3_f32"); run<5>("v_min3_f32"); run<7>("v_cmp_gt_f32"); run<8>("v_perm_b32");
    run<18>("v_fma_f32 |src2|"); run<21>("v_mov_b32"); run<23>("v_and_b32");
    run<9>("v_cvt_f32_i32"); run<10>("v_cvt_u32_f32"); run<11>("v_fract_f32"); run<20>("v_rndne_f32");
    run<12>("v_mul_i32_i24"); run<13>("v_mad_i32_i24"); run<14>("v_lshl_or_b32");
    run<26>("v_cndmask_b32 (vcc)"); run<27>("v_bfi_b32"); run<28>("v_ashrrev_i32");
    run<24>("cmp + 2 cndmask (3 ins)"); run<25>("ashr + 2 bfi (3 ins)");
    run<15>("v_pk_mul_f32"); run<16>("v_pk_fma_f32"); run<17>("v_pk_add_f32"); run<22>("v_fma_f64");
    return 0;
}
