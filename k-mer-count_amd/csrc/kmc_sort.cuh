// kmc_sort.cuh -- KMC_ALGO_SORT: sort-based counting for high-cardinality input.
//
// When almost every k-mer is new, a hash-table update per occurrence is bound by scattered global
// atomics (measured ~1.3 G k-mers/s).  The sort path does what the reference does --
// materialise every window, sort, run-length (k-mer-count/src/main.rs:78-79,87 + `uniq -c`) -- the
// MI355X way: the extraction front end of the stream kernel (kmc_stream.cuh, SINK == 1) writes ONE
// packed key per base position with coalesced 128-byte-per-lane stores (all-ones where no valid
// window ends), a device LSD radix sort orders them, and the kernels below collapse equal
// neighbours into (key, count) runs.  Everything is sequential HBM traffic; 288 GB of HBM is what
// makes "materialise everything" affordable (16 B x 2 buffers per base position in flight).
// The runs stay as sorted arrays next to the hash table and are merged by kmc_finalize.
#pragma once
#include "kmc_device.cuh"

// flags[i] = 1 where a new run starts (key differs from its left neighbour)
template <int KW>
__global__ void kmc_run_flags_kernel(const u64* __restrict__ hi, const u64* __restrict__ lo, u64 n, u32* __restrict__ flags) {
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) {
        bool head = i == 0 || lo[i] != lo[i - 1];
        if (KW == 2 && !head) head = hi[i] != hi[i - 1];
        flags[i] = head ? 1u : 0u;
    }
}

// pos = exclusive scan of flags.  Heads write their key and their position.
template <int KW>
__global__ void kmc_run_heads_kernel(const u64* __restrict__ hi, const u64* __restrict__ lo, u64 n, const u32* __restrict__ flags,
                                     const u32* __restrict__ pos, u64* __restrict__ out_hi, u64* __restrict__ out_lo, u64* __restrict__ head) {
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) {
        if (flags[i]) {
            const u32 r = pos[i];
            out_lo[r] = lo[i];
            if (KW == 2) out_hi[r] = hi[i];
            head[r] = i;
        }
    }
}

// run length = distance to the next head
__global__ void kmc_run_lengths_kernel(const u64* __restrict__ head, u64 n_runs, u64 n, u64* __restrict__ out_cnt) {
    for (u64 r = (u64)blockIdx.x * blockDim.x + threadIdx.x; r < n_runs; r += (u64)gridDim.x * blockDim.x)
        out_cnt[r] = (r + 1 < n_runs ? head[r + 1] : n) - head[r];
}

// reduce-by-key over sorted (key, count) pairs: every element adds its count to its run's total
__global__ void kmc_run_sums_kernel(const u32* __restrict__ flags, const u32* __restrict__ pos, const u64* __restrict__ cnt_in, u64 n,
                                    u64* __restrict__ out_cnt) {
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) {
        const u32 r = pos[i] + flags[i] - 1u;  // inclusive number of heads up to i, minus one
        atomicAdd((unsigned long long*)&out_cnt[r], cnt_in[i]);
    }
}
