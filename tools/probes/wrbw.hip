// What write bandwidth does HBM give a result stream shaped like ld_triangle's?  Persistent grid (2 workgroups of 256 threads
// per CU, as the kernel), every wave owns consecutive 32 KiB regions of a large buffer (one 64 x 128 unit of 4-byte cells) and
// writes each in one of these ways:
//   0  the kernel's: 16 steps x 8 global_store_dword ... nt (a half-wave = 128 contiguous bytes, the halves 2 KiB apart)
//   1  the same without nt                    2  the same with sc1 (write-through)
//   3  dwordx4 per lane, nt: a wave instruction = 1 KiB contiguous, 32 per region
//   4  dwordx4 per lane, plain
//   5  dwordx2 per lane, nt: a half-wave = 256 contiguous bytes (two adjacent column tiles per store), 64 per region
// `gap` = s_sleep units between steps (the epilogue's arithmetic: ~1000-2500 cycles per step) to see the effect of trickling.
// Build: hipcc -O3 --offload-arch=gfx950 wrbw.hip -o wrbw
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

template <int MODE>
__global__ void __launch_bounds__(256, 2) k(uint32_t *__restrict__ out, size_t n_regions, int gap)
{
    const uint32_t lane = threadIdx.x & 63u, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const size_t nw = (size_t)gridDim.x * 4u, w0 = (size_t)blockIdx.x * 4u + wave;
    const uint32_t l32 = lane & 31u, half = lane >> 5;
    for (size_t r = w0; r < n_regions; r += nw) {
        uint32_t *base = out + r * 8192u;
        const uint32_t v = (uint32_t)r + lane;
        if (MODE <= 2) {
            const uint32_t off = (half * 512u + l32) * 4u;   // bytes: rows e and e + 4 of a group of 8
            for (int e = 0; e < 16; ++e) {
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    uint32_t *row = base + ((4u * m + (e >> 2)) * 1024u + (e & 3) * 128u);
#define ST(o)                                                                                                        \
    if (MODE == 0) asm volatile("global_store_dword %0, %1, %2 offset:" #o " nt" : : "v"(off), "v"(v), "s"(row) : "memory");    \
    else if (MODE == 1) asm volatile("global_store_dword %0, %1, %2 offset:" #o : : "v"(off), "v"(v), "s"(row) : "memory");    \
    else asm volatile("global_store_dword %0, %1, %2 offset:" #o " sc1" : : "v"(off), "v"(v), "s"(row) : "memory");
                    ST(0) ST(128) ST(256) ST(384)
#undef ST
                }
                if (gap) for (int g = 0; g < gap; ++g) __builtin_amdgcn_s_sleep(8);
            }
        } else if (MODE == 3 || MODE == 4) {
            typedef unsigned u4 __attribute__((ext_vector_type(4))); const u4 vv = {v, v, v, v};
            for (int s = 0; s < 32; ++s) {
                u4 *p = reinterpret_cast<u4 *>(base + s * 256u) + lane;
                if (MODE == 3) __builtin_nontemporal_store(vv, p); else *p = vv;
                if (gap && (s & 1)) for (int g = 0; g < gap; ++g) __builtin_amdgcn_s_sleep(8);
            }
        } else {
            typedef unsigned u2 __attribute__((ext_vector_type(2))); const u2 vv = {v, v};
            for (int s = 0; s < 64; ++s) {
                u2 *p = reinterpret_cast<u2 *>(base + s * 128u) + lane;
                __builtin_nontemporal_store(vv, p);
                if (gap && (s & 3) == 3) for (int g = 0; g < gap; ++g) __builtin_amdgcn_s_sleep(8);
            }
        }
    }
}

template <int MODE>
void run(const char *name, uint32_t *buf, size_t n_regions, int gap)
{
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    k<MODE><<<512, 256>>>(buf, n_regions, gap);
    (void)hipEventRecord(a);
    for (int r = 0; r < 3; ++r) k<MODE><<<512, 256>>>(buf, n_regions, gap);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, a, b);
    ms /= 3;
    printf("mode %d %-44s gap %3d : %7.3f ms  %6.2f TB/s\n", MODE, name, gap, ms, n_regions * 32768.0 / (ms * 1e-3) / 1e12);
}

int main()
{
    const size_t bytes = (size_t)5 << 30, n_regions = bytes / 32768;
    uint32_t *buf;
    if (hipMalloc(&buf, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
    (void)hipMemset(buf, 0, bytes);
    for (int gap : {0, 4, 10}) {
        run<0>("dword nt, kernel pattern", buf, n_regions, gap);
        run<1>("dword plain, kernel pattern", buf, n_regions, gap);
        run<2>("dword sc1, kernel pattern", buf, n_regions, gap);
        run<3>("dwordx4 nt, 1 KiB per instruction", buf, n_regions, gap);
        run<4>("dwordx4 plain", buf, n_regions, gap);
        run<5>("dwordx2 nt, 256 B per half-wave", buf, n_regions, gap);
    }
    return 0;
}
