// Deterministic synthetic genotype panels, integer-only so that host (numpy,
// ld_tools_amd/synth.py) and device produce identical codes.
//   key(seed, i, h) = splitmix64-finaliser(seed ^ i*0x9E3779B97F4A7C15 ^ h*0xBF58476D1CE4E5B9)
//   fresh(i, h)     = key(seed + 2, i, h) < thresholds[i]            (per-SNP ALT probability * 2^64)
//   copy(i, h)      = i % block_len != 0 && key(seed + 1, i, h) < rho_thr
//   g[i][h]         = copy ? g[i-1][h] : fresh(i, h)                 (LD blocks of block_len SNPs)
//   code            = key(seed + 3, i, h) < miss_thr ? 2 : g[i][h]   (code 2 = neither allele)
// Round 6 (panels that are not all "ordinary": what a sub-panel of the ALL-panel variants looks like):
//   miss only in rows with key(seed + 6, i, 2^64-1) < miss_rows_thr         (miss_rows_thr = 2^64-1: every row, as before)
//   a row with key(seed + 4, i, 2^64-1) < mono_thr is MONOMORPHIC: every code 0 -- or every code 1 when the low three
//   bits of key(seed + 5, i, 2^64-1) are 0 (one in eight) -- whatever the chain says (the chain itself goes on underneath)
#include "ldx_common.h"

namespace ldx {

__host__ __device__ inline uint64_t mix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

__host__ __device__ inline uint64_t key64(uint64_t seed, uint64_t i, uint64_t h)
{
    return mix64(seed ^ (i * 0x9E3779B97F4A7C15ull) ^ (h * 0xBF58476D1CE4E5B9ull));
}

// one thread per (LD block, haplotype): walks the block's SNPs in order
__global__ void synth_codes_kernel(int8_t *__restrict__ codes, uint32_t n_snps, uint32_t n_hap, size_t ld,
                                   uint64_t seed, const uint64_t *__restrict__ thr, uint64_t rho_thr,
                                   uint32_t block_len, uint64_t miss_thr, uint32_t snp_offset, uint64_t mono_thr,
                                   uint64_t miss_rows_thr)
{
    const uint32_t h = blockIdx.x * blockDim.x + threadIdx.x;
    if (h >= n_hap) return;
    // local rows [r0, r1) of this launch belong to global LD block blockIdx.y (global SNP = snp_offset + local)
    const uint64_t gb = (uint64_t)(snp_offset / block_len + blockIdx.y) * block_len;   // first global SNP of block
    int g = 0;
    for (uint32_t k = 0; k < block_len; ++k) {
        const uint64_t gi = gb + k;
        const bool copy = (k != 0) && key64(seed + 1, gi, h) < rho_thr;
        // rows of the block that precede this shard's first SNP still steer the chain, so thr covers whole
        // blocks: thr[k] belongs to global SNP (snp_offset / block_len) * block_len + k
        const uint64_t th = thr[gi - (uint64_t)(snp_offset / block_len) * block_len];
        if (!copy) g = key64(seed + 2, gi, h) < th;
        if (gi >= snp_offset && gi - snp_offset < n_snps) {
            const bool miss_row = miss_rows_thr == ~0ull || key64(seed + 6, gi, ~0ull) < miss_rows_thr;
            const bool miss = miss_row && key64(seed + 3, gi, h) < miss_thr;
            int8_t code = miss ? (int8_t)2 : (int8_t)g;
            if (mono_thr && key64(seed + 4, gi, ~0ull) < mono_thr) code = (key64(seed + 5, gi, ~0ull) & 7u) == 0u ? (int8_t)1 : (int8_t)0;
            codes[(size_t)(gi - snp_offset) * ld + h] = code;
        }
    }
}

}  // namespace ldx

using namespace ldx;

// thresholds are indexed from the first SNP of the LD block containing snp_offset:
//   thresholds[k] belongs to global SNP (snp_offset / block_len) * block_len + k
extern "C" int ldx_synth_codes_ex_dev(int8_t *codes, uint32_t n_snps, uint32_t n_hap, size_t ld_codes, uint64_t seed,
                                      const uint64_t *thresholds, uint64_t rho_thr, uint32_t block_len,
                                      uint64_t miss_thr, uint32_t snp_offset, uint64_t mono_thr, uint64_t miss_rows_thr,
                                      void *stream)
{
    LDX_REQUIRE(codes && thresholds, "null pointer");
    LDX_REQUIRE(n_snps > 0 && n_hap > 0 && ld_codes >= n_hap && block_len >= 1, "bad shape");
    const uint32_t first_block = snp_offset / block_len;
    const uint32_t last_block = (snp_offset + n_snps - 1) / block_len;
    synth_codes_kernel<<<dim3((n_hap + 255u) / 256u, last_block - first_block + 1), 256, 0, (hipStream_t)stream>>>(
        codes, n_snps, n_hap, ld_codes, seed, thresholds, rho_thr, block_len, miss_thr, snp_offset, mono_thr, miss_rows_thr);
    LDX_HIP(hipGetLastError());
    return LDX_OK;
}

extern "C" int ldx_synth_codes_dev(int8_t *codes, uint32_t n_snps, uint32_t n_hap, size_t ld_codes, uint64_t seed,
                                   const uint64_t *thresholds, uint64_t rho_thr, uint32_t block_len,
                                   uint64_t miss_thr, uint32_t snp_offset, void *stream)
{
    return ldx_synth_codes_ex_dev(codes, n_snps, n_hap, ld_codes, seed, thresholds, rho_thr, block_len, miss_thr, snp_offset,
                                  0ull, ~0ull, stream);
}
