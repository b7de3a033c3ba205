// Cycles per "K step" (8 x v_mfma_i32_32x32x32_i8 + its side work) for a lone wave per SIMD: which ingredient of
// the triangle kernel's K loop stretches a step beyond the 256 cycles of its MFMAs?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <type_traits>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
#define ITERS 200

// NV = VALU fillers per step, LIT = fillers use 32-bit literals, NR = ds_read_b128 per step, NW = ds_write_b128 per step,
// DEP = the MFMAs consume the registers the ds_reads of the PREVIOUS step produced (with s_waitcnt)
template <int NV, int LIT, int NR, int NW, int DEP, int MF, int PLACE, int SELF, int WIDTH, int DRAIN>
__global__ void __launch_bounds__(256) k(int *sink, unsigned long long *cyc)
{
    __shared__ v4i lds[2048];
    v16i acc[8];
    for (int i = 0; i < 8; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0;
    v4i a[2] = {{(int)threadIdx.x, 1, 2, 3}, {5, (int)threadIdx.x, 7, 8}};
    v4i b[2][4];
    for (int j = 0; j < 4; ++j) b[0][j] = b[1][j] = v4i{j, 1, (int)threadIdx.x, j};
    unsigned x[8];
    for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * (i + 3);
    for (int i = threadIdx.x; i < 2048; i += 256) lds[i] = v4i{i, i, i, i};
    __syncthreads();
    const unsigned lane = threadIdx.x & 63;
    const unsigned raddr = (lane & 31) * 144 + (lane >> 5) * 16;      // the kernel's conflict-free fragment pattern
    const unsigned waddr = (threadIdx.x >> 1) * 144 + (threadIdx.x & 1) * 64;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    auto step = [&, raddr, waddr](auto curc) {
        constexpr int cur = decltype(curc)::value, nxt = cur ^ 1;
        if (PLACE == 0) {
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                if (WIDTH == 16) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(b[nxt][r & 3]) : "v"(raddr), "n"(0) );
                if (WIDTH == 4) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(b[nxt][r & 3][0]) : "v"(raddr), "n"(0) );
            }
            if (DRAIN) asm volatile("s_waitcnt lgkmcnt(0)");
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (DEP && i == 0) asm volatile("s_waitcnt lgkmcnt(%0)" : : "n"(NR > 15 ? 15 : NR));   // previous step's reads done
            if (MF) asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a[i >> 2]), "v"(b[DEP ? cur : 0][i & 3]));
            if (PLACE == 1 && i < NR) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(b[nxt][i & 3]) : "v"(raddr), "n"(0) );
#pragma unroll
            for (int f = 0; f < (NV + 7 - i) / 8; ++f) {
                if (SELF) asm volatile("v_and_b32 %0, %0, %1" : "+v"(x[(f + i) & 7]) : "v"(x[(f + i + 1) & 7]));
                else if (LIT) asm volatile("v_and_b32 %0, 0x8040201, %1" : "=v"(a[(i >> 2) ^ 1][f & 3]) : "v"(x[(f + i) & 7]));
                else     asm volatile("v_and_b32 %0, %1, %2" : "=v"(a[(i >> 2) ^ 1][f & 3]) : "v"(x[f & 7]), "v"(x[(f + i) & 7]));
            }
            if (i < NW) asm volatile("ds_write_b128 %0, %1 offset:18432" : : "v"(waddr), "v"(a[0]));
        }
    };
    for (int it = 0; it < ITERS; it += 2) {
        step(std::integral_constant<int, 0>{});
        step(std::integral_constant<int, 1>{});
    }
    asm volatile("s_waitcnt lgkmcnt(0)");
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    int s = 0;
    for (int i = 0; i < 8; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
    for (int i = 0; i < 8; ++i) s += x[i];
    for (int j = 0; j < 4; ++j) s += b[0][j][0] + b[1][j][1];
    sink[blockIdx.x * 256 + threadIdx.x] = s;
    if (lane == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int NV, int LIT, int NR, int NW, int DEP, int MF = 1, int PLACE = 0, int SELF = 0, int WIDTH = 16, int DRAIN = 0>
void run(int wps = 1)
{
    const int blocks = 256 * wps;
    int *sink; unsigned long long *cyc; static unsigned long long h[4096];
    (void)hipMalloc(&sink, blocks * 256 * 4); (void)hipMalloc(&cyc, blocks * 32);
    k<NV, LIT, NR, NW, DEP, MF, PLACE, SELF, WIDTH, DRAIN><<<blocks, 256>>>(sink, cyc);
    k<NV, LIT, NR, NW, DEP, MF, PLACE, SELF, WIDTH, DRAIN><<<blocks, 256>>>(sink, cyc);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(h, cyc, blocks * 32, hipMemcpyDeviceToHost);
    double tot = 0; for (int i = 0; i < blocks * 4; ++i) tot += (double)h[i];
    printf("VALU=%2d literal=%d ds_read=%d ds_write=%d consume_reads=%d mfma=%d reads_interleaved=%d fillers_private=%d width=%d drain=%d waves/SIMD=%d : %.0f cycles per step\n", NV, LIT, NR, NW, DEP, MF, PLACE, SELF, WIDTH, DRAIN, wps,
           tot / (blocks * 4) / ITERS);
    (void)hipFree(sink); (void)hipFree(cyc);
}

int main()
{
    run<0, 0, 0, 0, 0>(); run<20, 0, 0, 0, 0>(); run<30, 0, 0, 0, 0>(); run<0, 0, 4, 0, 1>(); run<8, 0, 4, 0, 1>(); run<20, 0, 4, 0, 1>();
    run<30, 0, 4, 0, 1>(); run<30, 0, 4, 2, 1>(); run<40, 0, 4, 2, 1>(); run<20, 0, 4, 0, 1, 1, 1>(); run<20, 0, 4, 0, 1, 1, 0, 1>();
    run<20, 0, 4, 0, 0, 0>(); run<20, 0, 16, 0, 0, 0>(); run<0, 0, 16, 0, 0, 0>();
    run<20, 0, 4, 0, 1>(2); run<30, 0, 4, 2, 1>(2); run<0, 0, 0, 0, 0>(2);
    return 0;
}
