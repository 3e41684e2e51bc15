// kmc_lr.cuh -- KMC_MODE_LR: the computation the reference actually performs,
// k-mer-count/src/main.rs:58-81 (== test.py:22-38): for every record, every chunk size
// s in 80..=140 and every window start i with i + s <= len:  key = seq[i..i+27] ++ seq[i+s-27..i+s]
// (27 + gap + 27).  The reference materialises each key as a String and sorts (main.rs:78-79,87);
// here each occurrence is a 108-bit key {hi,lo} added to the global two-word table, which
// kmc_finalize compacts and sorts, so the expanded output is byte-identical to main.rs:88-90.
//
// One thread per window start.  L is packed once; R slides one base per chunk size, so a thread
// reads 27 + 87 bases for its 61 keys.  This mode is 61 table updates per base and inherently
// high-cardinality (1.08 M distinct keys on the 80 kB fixture), i.e. bound by global atomics, not
// by HBM streaming (15.8 G keys/s on generator-style input); it exists for reference parity, not
// for the roofline run.  The host feeds ranges of window starts sized by the launch planner.
// A non-ACGT byte aborts the reference (main.rs:23); here it raises error bit 4 (KMC_ERR_ALPHABET).
#pragma once
#include "kmc_device.cuh"

#define KMC_LR_L 27
#define KMC_LR_R 27
#define KMC_LR_SMIN 80
#define KMC_LR_SMAX 140

__device__ __forceinline__ int kmc_code_of(uint8_t b) { return b == 'A' ? 0 : b == 'C' ? 1 : b == 'G' ? 2 : b == 'T' ? 3 : -1; }

// window starts [p_begin, p_end) of the batch (the host feeds ranges sized by the launch planner)
__global__ void kmc_lr_kernel(const uint8_t* __restrict__ bases, u64 n_bases, const u64* __restrict__ offsets, u64 n_reads,
                              u64 p_begin, u64 p_end, GTable g) {
    u64 nk = 0;
    __shared__ u64 s_first;  // read containing the block's first position of this sweep
    for (u64 p0 = p_begin + (u64)blockIdx.x * blockDim.x; p0 < p_end; p0 += (u64)gridDim.x * blockDim.x) {
        const u64 p = p0 + threadIdx.x;
        // read containing position p: last r with offsets[r] <= p.  One binary search per block (for
        // p0), then every thread walks forward from there: its read is at most a few reads further on.
        __syncthreads();
        if (threadIdx.x == 0) {
            u64 lo_i = 0, hi_i = n_reads;  // invariant: offsets[lo_i] <= p0 < offsets[hi_i]
            while (hi_i - lo_i > 1) {
                u64 mid = (lo_i + hi_i) >> 1;
                if (offsets[mid] <= p0) lo_i = mid; else hi_i = mid;
            }
            s_first = lo_i;
        }
        __syncthreads();
        if (p >= p_end) continue;
        u64 lo_i = s_first;
        while (offsets[lo_i + 1] <= p) lo_i++;  // (empty reads are skipped too; offsets[n_reads] == n_bases > p ends it)
        const u64 end = offsets[lo_i + 1];
        if (p + KMC_LR_SMIN > end) continue;  // main.rs:73-75: r_end > seq.len() -> break
        bool bad = false;
        u64 L = 0;
        for (int i = 0; i < KMC_LR_L; ++i) {
            int c = kmc_code_of(bases[p + i]);
            bad |= c < 0;
            L = (L << 2) | (u64)(c & 3);
        }
        const u64 rmask = (1ull << (2 * KMC_LR_R)) - 1;
        u64 R = 0;
        for (int i = KMC_LR_SMIN - KMC_LR_R; i < KMC_LR_SMIN - 1; ++i) {
            int c = kmc_code_of(bases[p + i]);
            bad |= c < 0;
            R = (R << 2) | (u64)(c & 3);
        }
        for (int s = KMC_LR_SMIN; s <= KMC_LR_SMAX && p + s <= end; ++s) {
            int c = kmc_code_of(bases[p + s - 1]);
            bad |= c < 0;
            R = ((R << 2) | (u64)(c & 3)) & rmask;
            if (bad) break;
            // key = L (54 bits) ++ R (54 bits), MSB first
            const u64 hi = L >> (64 - 2 * KMC_LR_R);
            const u64 lo = (L << (2 * KMC_LR_R)) | R;
            gtable_add<2>(g, hi, lo, 1);
            nk++;
        }
        if (bad) atomicOr((unsigned long long*)&g.counters[KMC_CTR_ERR], 4ull);
    }
    nk = wave_sum_u64(nk);
    if ((threadIdx.x & 63) == 0 && nk) atomicAdd((unsigned long long*)&g.counters[KMC_CTR_KMERS], nk);
}

static inline int kmc_lr_launch(hipStream_t st, int n_cu, const uint8_t* d_bases, const u64* d_offsets, u64 n_reads, u64 n_bases,
                                u64 p_begin, u64 p_end, GTable g) {
    u64 blocks = (p_end - p_begin + 255) / 256;
    u64 cap = (u64)n_cu * 16;
    int grid = (int)(blocks < cap ? blocks : cap);
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(kmc_lr_kernel, dim3(grid), dim3(256), 0, st, d_bases, n_bases, d_offsets, n_reads, p_begin, p_end, g);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}
