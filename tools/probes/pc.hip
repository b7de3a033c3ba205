// Producer / consumer mock of the triangle kernel: can ONE matrix-pipe wave per SIMD that does nothing
// but the K loop be kept fed, with the j-tile image produced and the pairs' epilogue run by OTHER waves of the workgroup?
//
// One workgroup per CU, 256 * WPS threads (WPS = waves per SIMD: 2, 3 or 4; all waves of a kernel share one register
// allocation, 512 / WPS).  Waves 0-3 are PRODUCERS (one per SIMD): MA x NB accumulator tiles of 32 x 32, FP4 MFMAs, A
// operands expanded from bits in registers, B fragments from an LDS image ring; at the end of a unit the counts go to an
// LDS hand-off buffer as uint16 (the accumulators start at 2^23, so the low 16 bits of the float ARE the count:
// ds_write_b16, no conversion).  The other 4 * (WPS - 1) waves are HELPERS: the first NLH of them also expand the j-tile's
// bits into the image ring (flag hand-shake in LDS, no workgroup barrier anywhere in the loop), all of them (or, with DED,
// only those without image duty) claim strips of finished units from an LDS queue and run the REAL fp32 epilogue
// (ld_multi_f32, ldx_common.h) + one 16-byte non-temporal store per lane and step.
// Data are random bits in the product's tiled layout; results are not checked (a structure / rate probe).
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize pc.hip -o pc
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "../../ld_tools_amd/csrc/ldx_common.h"

using namespace ldx;
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v8i __attribute__((ext_vector_type(8)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));
typedef unsigned v2u __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));

__device__ __forceinline__ v4i expand32_a4(uint32_t w)
{
    const uint32_t t = w >> 2;
    return v4i{(int)(w & 0x11111111u), (int)(w & 0x22222222u), (int)(t & 0x11111111u), (int)(t & 0x22222222u)};
}
__device__ __forceinline__ v4i expand32_b4(uint32_t w)
{
    const uint32_t t = w >> 2;
    return v4i{(int)((w << 2) & 0x44444444u), (int)(w & 0x22222222u), (int)((w) & 0x44444444u), (int)(t & 0x22222222u)};
}
__device__ unsigned long long g_oob[4];   // out-of-range accesses caught (the probe must never fault)
__device__ __forceinline__ void gload16(v4u &dst, const uint4 *base, size_t idx, size_t n)
{
    if (idx >= n) { atomicAdd(&g_oob[0], 1ull); idx = 0; }
    const v4u *p = reinterpret_cast<const v4u *>(base + idx);
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(p));
}
__device__ __forceinline__ void gstore16_nt(uint4 *base, size_t idx, size_t n, v4u v)
{
    if (idx >= n) { atomicAdd(&g_oob[1], 1ull); return; }
    uint4 *p = base + idx;
    asm volatile("global_store_dwordx4 %0, %1, off nt" : : "v"(p), "v"(v) : "memory");
}

// wait until all but the k youngest vector-memory operations of this wave are done (k is wave-uniform, any value)
__device__ __forceinline__ void wait_vm(uint32_t k)
{
    switch (k < 8u ? k : 8u) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    }
}

typedef __attribute__((address_space(3))) uint32_t lds_u32;
__device__ __forceinline__ uint32_t lds_peek(const uint32_t *p)
{
    const uint32_t v = *(const volatile lds_u32 *)p;   // ds_read_b32 (a generic pointer would make it a flat load: vmcnt AND lgkmcnt)
    return __builtin_amdgcn_readfirstlane(v);
}
// spin until *p >= target (wrap-safe); returns the number of polls that failed
__device__ __forceinline__ uint32_t wait_ge(const uint32_t *p, uint32_t target)
{
    uint32_t spins = 0;
    while (__builtin_expect((int)(lds_peek(p) - target) < 0, 0)) {   // the fast path falls through: a lone wave pays for every taken branch
        __builtin_amdgcn_s_sleep(1);
        ++spins;
    }
    asm volatile("" ::: "memory");
    return spins;
}
__device__ __forceinline__ void lds_signal_add(uint32_t *p, uint32_t v, uint32_t lane)
{
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's LDS traffic before the flag
    if (lane == 0) __hip_atomic_fetch_add((lds_u32 *)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

struct Stats {   // per wave
    unsigned long long cycles, spin_img, spin_ho, tasks, idle, imgs, t_k, t_ho;
};

// LDS-DMA: 64 lanes x 16 bytes from per-lane global addresses to LDS at lds_dst + 16 * lane (lds_dst wave-uniform).
// M0 is written in the same statement that reads it (cdna_hip_programming.md, inline-asm notes).
__device__ __forceinline__ void glds16(const uint4 *base, size_t idx, size_t n, uint32_t lds_dst)
{
    if (idx >= n) { atomicAdd(&g_oob[0], 1ull); idx = 0; }
    const uint4 *gsrc = base + idx;
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ uint32_t lds_addr(const void *p) { return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void *)p; }

// Roles: waves 0-3 producers (P), waves 4 .. 4+NL-1 loaders (L), the rest epilogue waves (E).
// Ring slot (one K-block of 256 haplotypes): [A raw bits: 4 P x 2 halves x UROWS x 16 B][B image: 4 steps x 2 halves x ROWS x 16 B][B raw: 2 x ROWS x 16 B]
template <int MA, int NB, int WPS, int NL, int RING, int S, int D, int PV = 0, int UNR = 1, int NPC = 1, bool AREG = false>
__global__ void __launch_bounds__(256 * WPS, WPS)
pc_kernel(const uint4 *__restrict__ alt, size_t alt_n, uint32_t nchunks, uint32_t n_rows, uint4 *__restrict__ out, size_t out_n, uint32_t npass,
          Stats *__restrict__ stats, F32Const fc, int epi_scale, int mode)
{
    constexpr uint32_t ROWS = NB * 32u, UROWS = MA * 32u;
    constexpr uint32_t NP = 4u * NPC, IMGROWS = NPC * ROWS;   // producers: 4 row groups x NPC column groups (NPC = 2: two per SIMD)
    // AREG: the producers load their own A bits (global -> registers, three blocks deep); the ring then holds the image only
    constexpr uint32_t ARAW = AREG ? 0u : 4u * 2u * UROWS * 16u, IMG = 4u * 2u * IMGROWS * 16u, SLOT = ARAW + IMG;
    constexpr uint32_t HO = UROWS * ROWS * 2u;
    constexpr uint32_t LPR = ROWS / 4u, RPS = 64u / LPR;
    constexpr uint32_t STRIP = 16u, NSTRIP = UROWS / STRIP, STEPS = STRIP / RPS;
    constexpr uint32_t AITEMS = 4u * UROWS * 2u, ADMA = AREG ? 0u : AITEMS / 64u / NL, ADMA1 = ADMA ? ADMA : 1u, BITEMS = IMGROWS * 2u, BPARTS = BITEMS / 64u;
    constexpr uint32_t BPL = BPARTS >= (uint32_t)NL ? BPARTS / NL : 1u;   // B wave-loads (and expansions) per loader that has B duty
    static_assert(AITEMS % (64u * NL) == 0 && BPARTS % BPL == 0, "whole wave-loads per loader");
    static_assert(!AREG || UNR == 3, "the A ring is three blocks deep");
    static_assert(D <= RING - 2, "loader lead: a producer asks for block q + 1 before it releases block q");
    extern __shared__ uint4 lds[];
    unsigned char *ring = reinterpret_cast<unsigned char *>(lds);
    unsigned char *ho = ring + RING * SLOT;
    float *ctab = reinterpret_cast<float *>(ho + NP * S * HO);     // [IMGROWS][4]
    float *rtab = ctab + IMGROWS * 4u;                             // [NP][S][UROWS][4]
    uint32_t *flag = reinterpret_cast<uint32_t *>(rtab + NP * S * UROWS * 4u);
    uint32_t *ready = flag, *free_ = flag + RING, *pub = flag + 2 * RING, *claim = pub + NP, *ho_done = claim + NP;   // [NP * S]

    const uint32_t tid = threadIdx.x, lane = tid & 63u, l32 = lane & 31u, half = lane >> 5;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t nblocks = nchunks / 2u, total = npass * nblocks;
    for (uint32_t k = tid; k < 2u * RING + 2u * NP + NP * S; k += blockDim.x) flag[k] = 0u;
    for (uint32_t k = tid; k < IMGROWS; k += blockDim.x)
        *reinterpret_cast<v4f *>(ctab + k * 4u) = v4f{2000.0f + k, 1.0f / (2000.0f + k), 1.0f / (3008.0f - k), 10.0f / sqrtf((2000.0f + k) * (3008.0f - k))};
    block_sync();
    const unsigned long long t_start = __builtin_amdgcn_s_memtime();
    unsigned long long spin_img = 0, spin_ho = 0, n_tasks = 0, n_idle = 0, n_imgs = 0, t_k = 0, t_ho = 0;
    auto unit_row0 = [&](uint32_t unit, uint32_t rgx) { return (((blockIdx.x * npass + unit) * 4u + rgx) * UROWS) % n_rows; };

    if (wave < NP) {
        // ------------------------------------------------------------------ producer --------------------------------
        if (!(mode & 16)) __builtin_amdgcn_s_setprio(2);
        const uint32_t p = wave, rg = p & 3u, cg = p >> 2;
        v16f acc[MA][NB];
        v4u aw[MA], awn[MA];
        v4u ar[3][MA];   // AREG: ring slot k holds the A words of the blocks q with q % 3 == k
        v4i af0[MA], af1[MA], bf0[NB], bf1[NB];
        size_t pa_base[MA];
        uint32_t lu = 0, lc = 0;   // AREG: load cursor (unit, block)
        auto pa_bases = [&]() {
#pragma unroll
            for (int m = 0; m < MA; ++m) {
                const uint32_t row0 = unit_row0(lu, rg) + 32u * m;
                pa_base[m] = ((size_t)(row0 / kSlab) * nchunks + half) * kSlab + (row0 % kSlab) + l32;
                if (pa_base[m] + (size_t)(nblocks - 1u) * 2u * kSlab >= alt_n) { atomicAdd(&g_oob[0], 1ull); pa_base[m] = 0; }
            }
        };
        auto issue_a = [&](v4u (&r)[MA]) {
            const size_t off = (size_t)lc * (2u * kSlab);
#pragma unroll
            for (int m = 0; m < MA; ++m) {
                const uint4 *pp = alt + pa_base[m] + off;
                asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r[m]) : "v"(pp));
            }
            if (lu + 1u < npass || lc + 1u < nblocks) {
                if (++lc == nblocks) { lc = 0; ++lu; pa_bases(); }
            }
        };
        auto touch = [&](v4u (&r)[MA]) {
#pragma unroll
            for (int m = 0; m < MA; ++m) asm volatile("" : "+v"(r[m]));
        };
        if constexpr ((PV & 6) != 0) {
#pragma unroll
            for (int m = 0; m < MA; ++m) { aw[m] = v4u{lane, 1u, 2u, 3u}; awn[m] = aw[m]; }
#pragma unroll
            for (int tt = 0; tt < NB; ++tt) { bf0[tt] = v4i{(int)lane, 1, 2, 3}; bf1[tt] = bf0[tt]; }
        }
        auto read_bf = [&](v4i (&bf)[NB], const unsigned char *buf, int w) {
#pragma unroll
            for (int tt = 0; tt < NB; ++tt) {
                if constexpr (PV & 2) asm volatile("" : "+v"(bf[tt]));
                else bf[tt] = *reinterpret_cast<const v4i *>(buf + (((uint32_t)w * 2u + half) * IMGROWS + cg * ROWS + 32u * tt + l32) * 16u);
            }
        };
        auto read_a = [&](v4u (&a)[MA], const unsigned char *slot) {
#pragma unroll
            for (int m = 0; m < MA; ++m) {
                if constexpr (PV & 4) asm volatile("" : "+v"(a[m]));
                else a[m] = *reinterpret_cast<const v4u *>(slot + ((rg * 2u + half) * UROWS + 32u * m + l32) * 16u);
            }
        };
        auto expand_a = [&](uint32_t x) {
            if constexpr (PV & 1) return v4i{(int)x, 0x11111111, 0x22222222, 0x11111111};
            else return expand32_a4(x);
        };
        auto mma = [&](const v4i (&af)[MA], const v4i (&bf)[NB]) {
#pragma unroll
            for (int m = 0; m < MA; ++m)
#pragma unroll
                for (int tt = 0; tt < NB; ++tt) {
                    const v8i a8 = {af[m].x, af[m].y, af[m].z, af[m].w, 0, 0, 0, 0};
                    const v8i b8 = {bf[tt].x, bf[tt].y, bf[tt].z, bf[tt].w, 0, 0, 0, 0};
                    acc[m][tt] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, acc[m][tt], 4, 4, 0, 0, 0, 0);
                }
        };
        auto interleave = [&](auto nread) {   // LDS reads first, then {1 MFMA, up to 3 VALU} groups
            __builtin_amdgcn_sched_group_barrier(0x100, decltype(nread)::value, 0);
#pragma unroll
            for (int k = 0; k < MA * NB; ++k) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
            }
        };
        if constexpr (AREG) {
            pa_bases();
            issue_a(ar[0]);
            issue_a(ar[1]);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            touch(ar[0]);
            touch(ar[1]);
        }
        if (!(mode & 1)) spin_img += wait_ge(&ready[0], NL);
        if constexpr (!AREG) read_a(aw, ring);
        read_bf(bf0, ring + ARAW, 0);
#pragma unroll
        for (int m = 0; m < MA; ++m) af0[m] = expand_a(AREG ? ar[0][m].x : aw[m].x);
        uint32_t u = 0, c = 0, islot = 0, igen = 0;
        unsigned long long tk0 = __builtin_amdgcn_s_memtime();
        auto init_acc = [&]() {
#pragma unroll
            for (int m = 0; m < MA; ++m)
#pragma unroll
                for (int tt = 0; tt < NB; ++tt)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[m][tt][e] = kMagic;
        };
        init_acc();
#define PC_BLOCK(q, CURV)                                                                                        \
        {                                                                                                         \
            constexpr int CUR = (CURV) % 3, NXT = (CUR + 1) % 3, PRV = (CUR + 2) % 3; \
            if constexpr (AREG) issue_a(ar[PRV]); \
            const unsigned char *sl = ring + islot * SLOT, *rd = sl + ARAW; \
            const uint32_t nslot = islot + 1u == (uint32_t)RING ? 0u : islot + 1u, ngen = nslot == 0u ? igen + 1u : igen; \
            const unsigned char *nsl = ring + nslot * SLOT; \
            const uint32_t early = *(const volatile lds_u32 *)&ready[nslot];   /* the next block's flag, read two steps before it is needed */ \
            read_bf(bf1, rd, 1); \
_Pragma("unroll") \
            for (int m = 0; m < MA; ++m) af1[m] = expand_a(AREG ? ar[CUR][m].y : aw[m].y); \
            mma(af0, bf0); \
            interleave(std::integral_constant<int, NB>{}); \
            __builtin_amdgcn_sched_barrier(0); \
            read_bf(bf0, rd, 2); \
_Pragma("unroll") \
            for (int m = 0; m < MA; ++m) af0[m] = expand_a(AREG ? ar[CUR][m].z : aw[m].z); \
            mma(af1, bf1); \
            interleave(std::integral_constant<int, NB>{}); \
            __builtin_amdgcn_sched_barrier(0); \
            if (q + 1u < total && !(mode & 1) && \
                __builtin_expect((int)(__builtin_amdgcn_readfirstlane(early) - NL * (ngen + 1u)) < 0, 0)) \
                spin_img += 1u + wait_ge(&ready[nslot], NL * (ngen + 1u)); \
            if constexpr (!AREG) read_a(awn, nsl); \
            read_bf(bf1, rd, 3); \
_Pragma("unroll") \
            for (int m = 0; m < MA; ++m) af1[m] = expand_a(AREG ? ar[CUR][m].w : aw[m].w); \
            mma(af0, bf0); \
            interleave(std::integral_constant<int, AREG ? NB : NB + MA>{}); \
            __builtin_amdgcn_sched_barrier(0); \
            if constexpr (AREG) { \
                asm volatile("s_waitcnt vmcnt(%0)" : : "n"(MA) : "memory"); \
                touch(ar[NXT]); \
            } \
            read_bf(bf0, nsl + ARAW, 0); \
_Pragma("unroll") \
            for (int m = 0; m < MA; ++m) af0[m] = expand_a(AREG ? ar[NXT][m].x : awn[m].x); \
            mma(af1, bf1); \
            interleave(std::integral_constant<int, NB>{}); \
            __builtin_amdgcn_sched_barrier(0); \
            if constexpr (!AREG) { \
_Pragma("unroll") \
                for (int m = 0; m < MA; ++m) aw[m] = awn[m]; \
            } \
            if (!(PV & 8) && lane == 0) __hip_atomic_fetch_add((lds_u32 *)&free_[islot], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); \
            islot = nslot; \
            igen = ngen; \
            if (__builtin_expect(++c == nblocks, 0)) { \
                c = 0; \
                const unsigned long long h0 = __builtin_amdgcn_s_memtime(); \
                t_k += h0 - tk0; \
                const uint32_t slot = u % S; \
                if (!(mode & 2)) spin_ho += wait_ge(&ho_done[p * S + slot], NSTRIP * (u / S)); \
                unsigned char *hb = ho + (p * S + slot) * HO; \
                if (!(mode & 8)) \
_Pragma("unroll") \
                for (int m = 0; m < MA; ++m) \
_Pragma("unroll") \
                    for (int tt = 0; tt < NB; ++tt) \
_Pragma("unroll") \
                        for (int e = 0; e < 16; ++e) { \
                            const uint32_t row = 32u * m + (e & 3) + 8u * (e >> 2) + 4u * half, col = 32u * tt + l32; \
                            *reinterpret_cast<uint16_t *>(hb + (row * ROWS + col) * 2u) = (uint16_t)__float_as_uint(acc[m][tt][e]); \
                        } \
                for (uint32_t r = lane; r < UROWS; r += 64u) { \
                    const float a = 1500.0f + (float)r; \
                    *reinterpret_cast<v4f *>(rtab + ((p * S + slot) * UROWS + r) * 4u) = v4f{a, 6.5f, 2.9f, 0.0045f}; \
                } \
                lds_signal_add(&pub[p], NSTRIP, lane); \
                ++u; \
                init_acc(); \
                tk0 = __builtin_amdgcn_s_memtime(); \
                t_ho += tk0 - h0; \
            } \
        }
#pragma unroll 1
        for (uint32_t q = 0; q < total; q += UNR) {   // total is a multiple of UNR (host)
            PC_BLOCK(q, 0)
            if constexpr (UNR > 1) PC_BLOCK(q + 1, 1)
            if constexpr (UNR > 2) PC_BLOCK(q + 2, 2)
            if constexpr (UNR > 3) PC_BLOCK(q + 3, 3)
        }
        if constexpr (AREG) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            touch(ar[0]);
            touch(ar[1]);
            touch(ar[2]);
        }
    } else if (wave < NP + NL) {
        // ------------------------------------------------------------------ loader ----------------------------------
        // Per block and loader: ADMA wave-items of A bits (global -> registers -> LDS as they are) and BPL wave-items of B bits
        // (global -> registers -> expanded -> image).  The loads run two blocks ahead in REGISTERS, whatever the state of
        // the ring: only the LDS writes wait for a free slot, so the ring hand-shake sees LDS latency, not memory latency.
        // Plain C++ loads in a loop unrolled by two with static names: hipcc counts them itself.
        if (!(mode & 16)) __builtin_amdgcn_s_setprio(3);
        const uint32_t w = wave - NP;
        const bool b_duty = w * BPL < BPARTS;
        size_t abase[ADMA1], bbase[BPL];
        uint32_t lunit = 0, lcc = 0;   // load cursor
        auto unit_bases = [&]() {
#pragma unroll
            for (uint32_t k = 0; k < ADMA; ++k) {
                const uint32_t a = (w * ADMA + k) * 64u + lane;
                const uint32_t pp = a / (UROWS * 2u), rem = a % (UROWS * 2u), hf = rem / UROWS, row = unit_row0(lunit, pp) + rem % UROWS;
                abase[k] = ((size_t)(row / kSlab) * nchunks + hf) * kSlab + (row % kSlab);
                if (abase[k] + (size_t)(nblocks - 1u) * 2u * kSlab >= alt_n) { atomicAdd(&g_oob[0], 1ull); abase[k] = 0; }
            }
#pragma unroll
            for (uint32_t k = 0; k < BPL; ++k) {
                const uint32_t bitem = (w * BPL + k) * 64u + lane, bhalf = (bitem / IMGROWS) & 1u, brow = bitem % IMGROWS;
                const uint32_t jrow = ((blockIdx.x * 7u + lunit) * IMGROWS) % n_rows + brow;
                bbase[k] = ((size_t)(jrow / kSlab) * nchunks + bhalf) * kSlab + (jrow % kSlab);
                if (bbase[k] + (size_t)(nblocks - 1u) * 2u * kSlab >= alt_n) { if (b_duty) atomicAdd(&g_oob[0], 1ull); bbase[k] = 0; }
            }
        };
        unit_bases();
        // The loads are asm statements with static register homes (r0 / r1, loop unrolled by two) and hand-counted
        // s_waitcnt vmcnt: left to hipcc, the same loop copies the loaded registers at the back edge and waits for every
        // load right behind its issue (a full memory latency per block).  A loader issues nothing else that vmcnt counts.
        auto gl = [&](v4u &dst, size_t idx) {
            const uint4 *p = alt + idx;
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(p));
        };
        uint32_t islot = 0, igen = 0;
        auto body = [&](auto bd) {
            constexpr bool kB = decltype(bd)::value;
            constexpr uint32_t NLOAD = ADMA + (kB ? BPL : 0u);
            v4u a0[ADMA1], a1[ADMA1], b0[BPL], b1[BPL];
            auto load = [&](v4u (&ra)[ADMA1], v4u (&rb)[BPL]) {   // the block at the load cursor (past the end: the last block again)
                const size_t off = (size_t)lcc * (2u * kSlab);
#pragma unroll
                for (uint32_t k = 0; k < ADMA; ++k) gl(ra[k], abase[k] + off);
                if (kB) {
#pragma unroll
                    for (uint32_t k = 0; k < BPL; ++k) gl(rb[k], bbase[k] + off);
                }
                if (lunit + 1u < npass || lcc + 1u < nblocks) {
                    if (++lcc == nblocks) { lcc = 0; ++lunit; unit_bases(); }
                }
            };
            auto store = [&](v4u (&ra)[ADMA1], v4u (&rb)[BPL]) {
                const unsigned long long f0 = __builtin_amdgcn_s_memtime();
                n_idle += wait_ge(&free_[islot], NP * igen);
                const unsigned long long f1 = __builtin_amdgcn_s_memtime();
                spin_img += f1 - f0;   // cycles waiting for a free slot
                asm volatile("s_waitcnt vmcnt(%0)" : : "n"(NLOAD) : "memory");   // all but the other register set's loads
#pragma unroll
                for (uint32_t k = 0; k < ADMA; ++k) asm volatile("" : "+v"(ra[k]));
                if (kB) {
#pragma unroll
                    for (uint32_t k = 0; k < BPL; ++k) asm volatile("" : "+v"(rb[k]));
                }
                const unsigned long long f2 = __builtin_amdgcn_s_memtime();
                spin_ho += f2 - f1;   // cycles waiting for the loads
                unsigned char *sl = ring + islot * SLOT;
#pragma unroll
                for (uint32_t k = 0; k < ADMA; ++k) *reinterpret_cast<v4u *>(sl + ((w * ADMA + k) * 64u + lane) * 16u) = ra[k];
                if (kB) {
#pragma unroll
                    for (uint32_t k = 0; k < BPL; ++k) {
                        const uint32_t bitem = (w * BPL + k) * 64u + lane, bhalf = bitem / IMGROWS, brow = bitem % IMGROWS;
#pragma unroll
                        for (int s4 = 0; s4 < 4; ++s4)
                            *reinterpret_cast<v4i *>(sl + ARAW + (((uint32_t)s4 * 2u + bhalf) * IMGROWS + brow) * 16u) = expand32_b4(rb[k][s4]);
                    }
                }
                lds_signal_add(&ready[islot], 1u, lane);
                if (++islot == (uint32_t)RING) { islot = 0; ++igen; }
                t_k += __builtin_amdgcn_s_memtime() - f2;   // expansion + LDS writes + signal
                ++n_imgs;
            };
            load(a0, b0);
            load(a1, b1);
#pragma unroll 1
            for (uint32_t i = 0; i < total; i += 2) {   // total is even (host)
                store(a0, b0);
                load(a0, b0);
                store(a1, b1);
                load(a1, b1);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (uint32_t k = 0; k < ADMA; ++k) asm volatile("" : "+v"(a0[k]), "+v"(a1[k]));
            if (kB) {
#pragma unroll
                for (uint32_t k = 0; k < BPL; ++k) asm volatile("" : "+v"(b0[k]), "+v"(b1[k]));
            }
        };
        if (!(mode & 4)) { if (b_duty) body(std::true_type{}); else body(std::false_type{}); }
    } else {
        // ------------------------------------------------------------------ epilogue wave ---------------------------
        const uint32_t h = wave - NP - NL;
        if (mode & 32) __builtin_amdgcn_s_setprio(3);
        const uint32_t strips_per_p = npass * NSTRIP;
        uint32_t rot = h % NP, done_p = 0;
        const uint32_t col4 = lane % LPR, rsub = lane / LPR;
        while (done_p != (1u << NP) - 1u && !(mode & 2)) {
            uint32_t got = 0xFFFFFFFFu, gp = 0;
            for (uint32_t k = 0; k < NP && got == 0xFFFFFFFFu; ++k) {
                const uint32_t pp = (rot + k) % NP;
                if (done_p & (1u << pp)) continue;
                const uint32_t cl = lds_peek(&claim[pp]);
                if (cl >= strips_per_p) { done_p |= 1u << pp; continue; }
                const uint32_t pb = lds_peek(&pub[pp]);
                if (cl < pb) {
                    uint32_t old = 0xFFFFFFFFu;
                    if (lane == 0) old = atomicCAS(&claim[pp], cl, cl + 1u);   // ds_cmpst_rtn_b32
                    old = __builtin_amdgcn_readfirstlane(old);
                    if (old == cl) { got = cl; gp = pp; }
                }
            }
            if (got == 0xFFFFFFFFu) {
                __builtin_amdgcn_s_sleep(2);
                ++n_idle;
                continue;
            }
            asm volatile("" ::: "memory");
            rot = (gp + 1u) % NP;
            const uint32_t uu = got / NSTRIP, ss = got % NSTRIP, slot = uu % S;
            const unsigned char *hb = ho + (gp * S + slot) * HO;
            const float *rt = rtab + (gp * S + slot) * UROWS * 4u;
            F32Col cols[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const v4f v = *reinterpret_cast<const v4f *>(ctab + ((gp >> 2) * ROWS + col4 * 4u + k) * 4u);
                cols[k] = F32Col{v.x, v.y, v.z, v.w};
            }
            const uint32_t urow0 = (((blockIdx.x * npass + uu) * NP + gp) * UROWS);   // output rows of the unit
            v2u c2n = *reinterpret_cast<const v2u *>(hb + ((ss * STRIP + rsub) * ROWS + col4 * 4u) * 2u);
            v4f rvn = *reinterpret_cast<const v4f *>(rt + (ss * STRIP + rsub) * 4u);
#pragma unroll 1
            for (uint32_t st = 0; st < STEPS; ++st) {
                const uint32_t row = ss * STRIP + st * RPS + rsub;
                const v2u c2 = c2n;
                const v4f rv = rvn;
                if (st + 1u < STEPS) {   // the next step's operands: their LDS latency hides behind this step's arithmetic
                    c2n = *reinterpret_cast<const v2u *>(hb + ((row + RPS) * ROWS + col4 * 4u) * 2u);
                    rvn = *reinterpret_cast<const v4f *>(rt + (row + RPS) * 4u);
                }
                float c4[4] = {(float)(c2.x & 0xFFFFu), (float)(c2.x >> 16), (float)(c2.y & 0xFFFFu), (float)(c2.y >> 16)};
                F32Row r4[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) r4[k] = F32Row{rv.x, rv.y, rv.z, rv.w};
                ldx_k16 o4[4];
                float wmax = 0.0f, ymin = 1.0f;
                for (int rep = 0; rep < epi_scale; ++rep) {   // epi_scale = 1: the real mix; 0 / 2: what the epilogue arithmetic costs
                    ld_multi_f32<4, ldx_k16, false>(c4, fc, r4, cols, o4, wmax, ymin);
                    if (epi_scale > 1) c4[0] += wmax;
                }
                const bool sure = (wmax < fc.tol) & (ymin > -1.0f);
                if (sure) {
                    v4u cell;
#pragma unroll
                    for (int k = 0; k < 4; ++k) cell[k] = epi_scale ? __builtin_bit_cast(uint32_t, o4[k]) : __float_as_uint(c4[k]);
                    gstore16_nt(out, ((size_t)(urow0 + row) * ROWS + col4 * 4u) / 4u, out_n, cell);
                }
            }
            lds_signal_add(&ho_done[gp * S + slot], 1u, lane);
            ++n_tasks;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (lane == 0) {
        Stats s;
        s.cycles = __builtin_amdgcn_s_memtime() - t_start;
        s.spin_img = spin_img;
        s.spin_ho = spin_ho;
        s.tasks = n_tasks;
        s.idle = n_idle;
        s.imgs = n_imgs;
        s.t_k = t_k;
        s.t_ho = t_ho;
        stats[blockIdx.x * (4 * WPS) + wave] = s;
    }
}

template <int MA, int NB, int WPS, int NL, int RING, int S, int D, int PV = 0, int UNR = 1, int NPC = 1, bool AREG = false>
static void run(const char *name, const uint4 *alt, size_t alt_n, uint32_t nchunks, uint32_t n_rows, uint4 *out, size_t out_n, uint32_t npass, int epi_scale,
                int mode = 0, int blocks = 256)
{
    constexpr uint32_t ROWS = NB * 32u, UROWS = MA * 32u;
    constexpr uint32_t NP = 4u * NPC, IMGROWS = NPC * ROWS;
    const size_t lds = RING * ((AREG ? 0u : 4u * 2u * UROWS * 16u) + 4u * 2u * IMGROWS * 16u) + NP * S * (UROWS * ROWS * 2u) + IMGROWS * 16u +
                       NP * S * UROWS * 16u + (2 * RING + 2 * NP + NP * S) * 4u;
    auto kern = pc_kernel<MA, NB, WPS, NL, RING, S, D, PV, UNR, NPC, AREG>;
    if (lds > 163840u || hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
        printf("%s: cannot set %zu bytes of LDS\n", name, lds);
        return;
    }
    hipFuncAttributes fa;
    (void)hipFuncGetAttributes(&fa, (const void *)kern);
    Stats *st;
    const int waves = 4 * WPS;
    if (hipMalloc(&st, sizeof(Stats) * blocks * waves) != hipSuccess) { printf("hipMalloc failed\n"); exit(1); }
    const F32Const fc = f32_const(5008.0);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        (void)hipEventRecord(e0);
        kern<<<blocks, 256 * WPS, lds>>>(alt, alt_n, nchunks, n_rows, out, out_n, npass, st, fc, epi_scale, mode);
        (void)hipEventRecord(e1);
        if (hipEventSynchronize(e1) != hipSuccess) { printf("%s: launch failed: %s\n", name, hipGetErrorString(hipGetLastError())); return; }
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    std::vector<Stats> h(blocks * waves);
    (void)hipMemcpy(h.data(), st, sizeof(Stats) * blocks * waves, hipMemcpyDeviceToHost);
    double pc = 0, pk = 0, ph = 0, psi = 0, psh = 0, et = 0, ei = 0, li = 0, lf = 0, lv = 0, le = 0, lc = 0;
    for (int b = 0; b < blocks; ++b)
        for (int w = 0; w < waves; ++w) {
            const Stats &s = h[b * waves + w];
            if (w < (int)NP) { pc += s.cycles; pk += s.t_k; ph += s.t_ho; psi += s.spin_img; psh += s.spin_ho; }
            else if (w < (int)NP + NL) { li += s.idle; lf += s.spin_img; lv += s.spin_ho; le += s.t_k; lc += s.cycles; }
            else { et += s.tasks; ei += s.idle; }
        }
    const double np = blocks * (double)NP, ne = blocks * (waves - (double)NP - NL), nl = blocks * (double)NL;
    const double pairs = (double)blocks * npass * (double)NP * UROWS * ROWS;
    const double nblocks = nchunks / 2.0, mfma_per_p = npass * nblocks * 4.0 * MA * NB;
    const double hap = nchunks * 128.0;
    if (mode | PV | (UNR - 1) | (NPC - 1)) printf("[mode %2d PV %2d unroll %d, %d P/SIMD%s] ", mode, PV, UNR, NPC, AREG ? ", A in regs" : "");
    printf("%-22s vgpr=%3d lds=%6zu %8.3f ms %6.2fe11 pairs/s frac=%.3f | P %7.0f cyc/unit (K %7.0f, hand-off %5.0f), %5.1f cyc/MFMA, "
           "spins/unit: ring %.1f ho %.1f | L cyc/block %.0f: free-wait %.0f load-wait %.0f write+signal %.0f | E %.1f tasks, %.0f idle polls per wave\n",
           name, fa.numRegs, lds, best, pairs / (best * 1e-3) / 1e11, pairs * 2.0 * hap / (best * 1e-3) / 1e16, pc / np / npass, pk / np / npass,
           ph / np / npass, pc / np / mfma_per_p, psi / np / npass, psh / np / npass, lc / nl / (npass * nblocks), lf / nl / (npass * nblocks), lv / nl / (npass * nblocks), le / nl / (npass * nblocks), et / ne, ei / ne);
    unsigned long long oob[4] = {0, 0, 0, 0};
    (void)hipMemcpyFromSymbol(oob, HIP_SYMBOL(g_oob), sizeof(oob));
    if (oob[0] | oob[1]) printf("   !!! out-of-range accesses caught: %llu loads, %llu stores\n", oob[0], oob[1]);
    fflush(stdout);
    (void)hipFree(st);
}

int main(int argc, char **argv)
{
    const uint32_t n_rows = 9984, nchunks = 40;   // 5120 haplotypes (5008 padded), 78 slabs
    const uint32_t npass = argc > 1 ? (uint32_t)atoi(argv[1]) : 12;
    const size_t alt_bytes = (size_t)(n_rows / 128) * nchunks * 128 * 16;
    std::vector<uint32_t> host(alt_bytes / 4);
    uint64_t x = 88172645463325252ull;
    for (auto &w : host) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; w = (uint32_t)(x >> 16) & (uint32_t)(x >> 40); }   // ~25 % ones
    uint4 *alt, *out;
    setvbuf(stdout, nullptr, _IONBF, 0);
    if (hipMalloc(&alt, alt_bytes) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
    (void)hipMemcpy(alt, host.data(), alt_bytes, hipMemcpyHostToDevice);
    const size_t out_bytes = (size_t)256 * npass * 8 * 96 * 128 * 4 + (1u << 20);
    if (hipMalloc(&out, out_bytes) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
    const size_t alt_n = alt_bytes / 16, out_n = out_bytes / 16;
    (void)hipMemset(out, 0, out_bytes);
    const unsigned mask = argc > 2 ? (unsigned)strtoul(argv[2], nullptr, 0) : 0xFFu;   // which configurations
    const int epi_lo = argc > 3 ? atoi(argv[3]) : 0;
#define run if (!((mask >> (cfg++)) & 1u)) {} else run
    if (argc > 5) {   // the producer's loop alone (mode 15), by ingredient
        int cfg = 0;
        (void)cfg;
#define PVRUN(PVV) cfg = 0; run<2, 2, 4, 4, 5, 1, 2, PVV>("64x64 4w", alt, alt_n, nchunks, n_rows, out, out_n, npass, 1, 15); \
                   cfg = 0; run<3, 2, 3, 2, 4, 1, 1, PVV>("96x64 3w", alt, alt_n, nchunks, n_rows, out, out_n, npass, 1, 15); \
                   cfg = 0; run<2, 4, 2, 4, 3, 1, 1, PVV>("64x128 2w", alt, alt_n, nchunks, n_rows, out, out_n, npass, 1, 15);
        PVRUN(0) PVRUN(8) PVRUN(9) PVRUN(10) PVRUN(12) PVRUN(14) PVRUN(15)
#undef PVRUN
#define PVRUN(PVV, U) cfg = 0; run<2, 2, 4, 4, 5, 1, 2, PVV, U>("64x64 4w", alt, alt_n, nchunks, n_rows, out, out_n, npass, 1, 15); \
                   cfg = 0; run<3, 2, 3, 2, 4, 1, 1, PVV, U>("96x64 3w", alt, alt_n, nchunks, n_rows, out, out_n, npass, 1, 15); \
                   cfg = 0; run<2, 4, 2, 4, 3, 1, 1, PVV, U>("64x128 2w", alt, alt_n, nchunks, n_rows, out, out_n, npass, 1, 15);
        PVRUN(0, 2) PVRUN(0, 4) PVRUN(15, 2) PVRUN(15, 4)
        return 0;
    }
    if (argc > 4) {   // ablations of two configurations: mode bits (see the kernel)
        int cfg = 0;
        (void)cfg;
        for (int mode : {0, 1, 2, 3, 5, 7, 15}) { cfg = 0;
            run<2, 2, 4, 4, 5, 1, 2>("64x64 4w L4 ring5 D2", alt, alt_n, nchunks, n_rows, out, out_n, npass, 1, mode); }
        for (int mode : {0, 1, 2, 3, 5, 7, 15}) { cfg = 0;
            run<3, 2, 3, 2, 4, 1, 1>("96x64 3w L2 ring4 D1", alt, alt_n, nchunks, n_rows, out, out_n, npass, 1, mode); }
        return 0;
    }
    for (int epi = 1; epi >= epi_lo; --epi) {
        int cfg = 0;
        printf("---- epilogue scale %d, %u passes per CU\n", epi, npass);
        //  MA NB WPS NL RING S  D  PV UNR NPC AREG
        run<2, 2, 4, 2, 5, 1, 1, 0, 3, 2, true>("2P+2H 64x64 L2 ring5", alt, alt_n, nchunks, n_rows, out, out_n, npass, epi);
        run<2, 2, 4, 2, 5, 1, 1, 0, 3, 2, true>("same, no priorities", alt, alt_n, nchunks, n_rows, out, out_n, npass, epi, 16);
        run<2, 2, 4, 2, 5, 1, 1, 0, 3, 2, true>("same, E prio 3, others 0", alt, alt_n, nchunks, n_rows, out, out_n, npass, epi, 48);
        run<2, 2, 4, 2, 5, 1, 1, 0, 3, 2, true>("same, E 3, P 2, L 3", alt, alt_n, nchunks, n_rows, out, out_n, npass, epi, 32);
        run<2, 2, 4, 2, 3, 1, 1, 0, 2, 2, false>("2P+2H A via LDS ring3, no prio", alt, alt_n, nchunks, n_rows, out, out_n, npass, epi, 16);
    }
#undef run
    return 0;
}
