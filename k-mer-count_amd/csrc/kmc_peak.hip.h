// kmc_peak.hip.h -- the measured streaming-read peak of this GPU ("measure an achievable-peak
// read with a plain dwordx4 copy/reduce kernel so the fraction can be quoted against both nominal and
// measured peak").  Nothing of the product path runs through here: bench.py launches it over the resident
// batch (the bytes the walk kernel streams: the input shape of random_fasta_generator.py:5-15) and quotes
// roofline.frac_of_measured beside the nominal fraction.
//
// The kernel is as plain as a read can be: every lane issues UNROLL independent non-temporal 16-byte loads per
// trip (a wave-instruction = 1 KiB contiguous), xors them into four registers, and one lane per wave adds the
// wave's word to the result with a single atomic at the end.  Grid shapes: the walk kernel's own (one
// 1024-thread workgroup per CU) and the usual memory-bound shape (8 workgroups of 256 threads per CU).
#pragma once
#include "kmc_device.hip.h"

template <int UNROLL>
__global__ __launch_bounds__(1024)
void kmc_read_peak_kernel(const uint8_t* __restrict__ buf, u64 n16, unsigned long long* __restrict__ out) {
    typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
    const u32x4_t* const p = reinterpret_cast<const u32x4_t*>(buf);
    const u64 nthreads = (u64)gridDim.x * blockDim.x;
    const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    u32x4_t acc = {0u, 0u, 0u, 0u};
    // trip g: pieces [g * nthreads * UNROLL, ...): lane i of the grid takes i + u * nthreads -- every wave-load is 1 KiB contiguous
    u64 i = t;
    for (; i + (u64)(UNROLL - 1) * nthreads < n16; i += nthreads * UNROLL) {
        u32x4_t v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) v[u] = __builtin_nontemporal_load(p + i + (u64)u * nthreads);
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) acc ^= v[u];
    }
    for (; i < n16; i += nthreads) acc ^= __builtin_nontemporal_load(p + i);
    u64 w = ((u64)(acc.x ^ acc.z) << 32) | (u64)(acc.y ^ acc.w);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) w ^= __shfl_xor(w, o);
    if ((threadIdx.x & 63) == 0) atomicXor(out, (unsigned long long)w);
}
