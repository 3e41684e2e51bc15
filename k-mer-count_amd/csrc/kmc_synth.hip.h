// kmc_synth.hip.h -- seeded, size-parameterised re-creation of the input distribution of the
// reference's generator, /root/reference/random_fasta_generator.py:5-15: a pool of `pool`
// uniform-random ACGT lines of `line_len` bases (:5-8); record i (1-based) has header
// ">dummy_sequence_{i:03d} {i}th record" (:11-12) and `lines_per_record` lines, each drawn
// uniformly from the pool (:13-15).  The reference script is unseeded, fixed at 200 records and
// uses Python's Mersenne Twister; this generator uses a counter-based splitmix64 so that any
// record range comes out identical on the host and on the device.
#pragma once
#include <stdint.h>
#ifdef __HIPCC__
#include <hip/hip_runtime.h>
#define KMC_HD __host__ __device__
#else
#define KMC_HD
#endif

KMC_HD inline uint64_t kmc_synth_mix(uint64_t seed, uint64_t stream, uint64_t ctr) {
    uint64_t z = seed + stream * 0xD6E8FEB86659FD93ull;
    z += (ctr + 1) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
// base x of pool line p
KMC_HD inline uint8_t kmc_synth_pool_base(uint64_t seed, uint32_t line_len, uint32_t p, uint32_t x) {
    return (uint8_t)"ACGT"[kmc_synth_mix(seed, 0, (uint64_t)p * line_len + x) >> 62];
}
// pool line chosen for global line index gl = record*lines_per_record + j
KMC_HD inline uint32_t kmc_synth_choice(uint64_t seed, uint32_t pool, uint64_t gl) {
    return (uint32_t)(((kmc_synth_mix(seed, 1, gl) >> 32) * (uint64_t)pool) >> 32);
}
// pool == 0: every line is fresh random
KMC_HD inline uint8_t kmc_synth_fresh_base(uint64_t seed, uint32_t line_len, uint64_t gl, uint32_t x) {
    return (uint8_t)"ACGT"[kmc_synth_mix(seed, 2, gl * line_len + x) >> 62];
}

#ifdef __HIPCC__
// one thread per 16 output bases; pool_bases = pool*line_len bytes in device memory (pool > 0)
__global__ void kmc_synth_kernel(uint64_t seed, uint32_t pool, uint32_t line_len, uint32_t lines_per_record,
                                 uint64_t first_record, uint64_t n_bases, const uint8_t* __restrict__ pool_bases,
                                 uint8_t* __restrict__ out) {
    const uint64_t n16 = (n_bases + 15) / 16;
    for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n16; t += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t p = t * 16;
        uint64_t gl = p / line_len + first_record * lines_per_record;  // global line index
        uint32_t x = (uint32_t)(p % line_len);
        uint32_t pl = pool ? kmc_synth_choice(seed, pool, gl) : 0;
        uint8_t buf[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            buf[i] = pool ? pool_bases[(uint64_t)pl * line_len + x] : kmc_synth_fresh_base(seed, line_len, gl, x);
            if (++x == line_len) {
                x = 0;
                ++gl;
                if (pool) pl = kmc_synth_choice(seed, pool, gl);
            }
        }
        if (p + 16 <= n_bases) {
            *reinterpret_cast<uint4*>(out + p) = *reinterpret_cast<const uint4*>(buf);
        } else {
            for (int i = 0; p + i < n_bases; ++i) out[p + i] = buf[i];
        }
    }
}

__global__ void kmc_synth_offsets_kernel(uint64_t n_records, uint64_t read_len, uint64_t* __restrict__ offsets) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i <= n_records; i += (uint64_t)gridDim.x * blockDim.x)
        offsets[i] = i * read_len;
}
#endif
