// kmc_lr.hip.h -- KMC_MODE_LR: the computation the reference actually performs,
// k-mer-count/src/main.rs:58-81 (== test.py:22-38): for every record, every chunk size
// s in 80..=140 and every window start i with i + s <= len:  key = seq[i..i+27] ++ seq[i+s-27..i+s]
// (27 + gap + 27).  The reference materialises each key as a String and sorts (main.rs:78-79,87);
// here each occurrence is a 108-bit key {hi,lo} (2 bits per base, MSB first), the keys are sorted and
// run-length counted on the device (kmc_msd.hip.h), and the expanded output is byte-identical to
// main.rs:88-90.
//
// The first version made one 128-bit global-atomic table insert per occurrence (one thread per window
// start, 61 inserts each: 7.8 G keys/s); the mode is inherently high-cardinality (1.08 M distinct keys
// among 3.55 M occurrences on the 80 kB fixture), so it now goes the reference's own way: form every
// key, sort, run-length (below).
// A non-ACGT byte aborts the reference (main.rs:23); here it raises error bit 4 (KMC_ERR_ALPHABET).
#pragma once
#include "kmc_device.hip.h"

#define KMC_LR_L 27
#define KMC_LR_R 27
#define KMC_LR_SMIN 80
#define KMC_LR_SMAX 140

__device__ __forceinline__ int kmc_code_of(uint8_t b) { return b == 'A' ? 0 : b == 'C' ? 1 : b == 'G' ? 2 : b == 'T' ? 3 : -1; }

// ---- LR mode as a throughput path: extraction for the sort pipeline -----------------------------------
// This kernel only FORMS the keys -- what main.rs:76-79 does with slices and a String -- and the hand-written
// radix sort + run-length (kmc_msd.hip.h) does the grouping, as main.rs:87 does with sort().
//
// One thread per (window start, chunk size) pair, so stores are fully coalesced: key q = 61 * (p - p_begin)
// + (s - 80).  A workgroup of 256 threads covers KMC_LRX_POS = 256 consecutive window starts (15,616 keys):
// their bases (plus the 140 that follow) are packed once into LDS at 2 bits per base, MSB first, and every
// key is two funnel shifts out of that stream (no per-base loop).  Pairs that do not exist (read too short
// for this size, window start in the last 79 bases of a read) get the all-ones filler the sort drops.
// (The first version covered 64 window starts with 1024 threads and had thread 0 find the first read by
// binary search before anything else could start: 25 k workgroups of mostly latency, 4.4 ms for 71 M keys
// = 260 GB/s of stores.  Now every thread finds the read of its own window start.)
// A non-ACGT byte inside an emitted chunk raises error bit 4 (main.rs:23).
#define KMC_LRX_POS 256
#define KMC_LRX_THREADS 256
#define KMC_LRX_NS (KMC_LR_SMAX - KMC_LR_SMIN + 1)   // 61 chunk sizes
#define KMC_LRX_SPAN (KMC_LRX_POS + KMC_LR_SMAX)     // bases a workgroup looks at
#define KMC_LRX_WORDS ((KMC_LRX_SPAN + 15) / 16 + 3)

// 64-bit window of a big-endian 2-bit stream (w[i] holds bases 16i.., first base in the top bits)
// starting at base `a`: the 32 bases a .. a+31
__device__ __forceinline__ u64 lrx_window(const u32* w, u32 a) {
    const u32 i = a >> 4, o = 2 * (a & 15);
    const u64 hi = ((u64)w[i] << 32) | w[i + 1];
    const u64 x = hi << o;
    return o ? (x | ((u64)w[i + 2] >> (32 - o))) : x;
}
// one bit per base (bit set = not ACGT), 32 bases per word, first base in the top bit: bits a .. a+31
__device__ __forceinline__ u32 lrx_badwin(const u32* b, u32 a) {
    const u32 i = a >> 5, o = a & 31;
    return o ? ((b[i] << o) | (b[i + 1] >> (32 - o))) : b[i];
}

__global__ __launch_bounds__(KMC_LRX_THREADS)
void kmc_lr_extract_kernel(const uint8_t* __restrict__ bases, u64 n_bases, const u64* __restrict__ offsets, u64 n_reads,
                           u64 p_begin, u64 p_end, u64* __restrict__ out_hi, u64* __restrict__ out_lo, u64* __restrict__ counters) {
    __shared__ u32 w[KMC_LRX_WORDS];          // 2-bit codes
    __shared__ u32 bad[KMC_LRX_WORDS / 2 + 2]; // 1 bit per base
    __shared__ u64 rend[KMC_LRX_POS];          // end of the read a window start lies in
    const u32 tid = threadIdx.x;
    const u64 P0 = p_begin + (u64)blockIdx.x * KMC_LRX_POS;
    if (P0 >= p_end) return;
    const u32 npos = (u32)min((u64)KMC_LRX_POS, p_end - P0);
    for (u32 i = tid; i < KMC_LRX_WORDS; i += KMC_LRX_THREADS) w[i] = 0;
    for (u32 i = tid; i < KMC_LRX_WORDS / 2 + 2; i += KMC_LRX_THREADS) bad[i] = 0;
    __syncthreads();
    // pack the span: base P0 + i
    for (u32 i = tid; i < KMC_LRX_SPAN; i += KMC_LRX_THREADS) {
        const u64 p = P0 + i;
        if (p < n_bases) {
            const uint8_t c = bases[p];
            const int code = kmc_code_of(c);
            if (code < 0) atomicOr(&bad[i >> 5], 0x80000000u >> (i & 31));
            else if (code) atomicOr(&w[i >> 4], (u32)code << (30 - 2 * (i & 15)));
        }
    }
    if (tid < npos) {  // the read of my window start: last r with offsets[r] <= p  (offsets[n_reads] == n_bases > p)
        const u64 p = P0 + tid;
        u64 lo_i = 0, hi_i = n_reads;
        while (hi_i - lo_i > 1) {
            const u64 mid = (lo_i + hi_i) >> 1;
            if (offsets[mid] <= p) lo_i = mid; else hi_i = mid;
        }
        rend[tid] = offsets[lo_i + 1];
    }
    __syncthreads();
    const u32 n_keys = npos * KMC_LRX_NS;
    u64 nk = 0;
    bool saw_bad = false;
    for (u32 q = tid; q < n_keys; q += KMC_LRX_THREADS) {
        const u32 t = q / KMC_LRX_NS, j = q - t * KMC_LRX_NS, sz = KMC_LR_SMIN + j;
        u64 hi = ~0ull, lo = ~0ull;
        if (P0 + t + sz <= rend[t]) {  // main.rs:73-75
            const u32 a = t, b = t + sz - KMC_LR_R;
            const u32 mbits = 0xFFFFFFE0u;  // the first 27 of 32
            if ((lrx_badwin(bad, a) | lrx_badwin(bad, b)) & mbits) saw_bad = true;
            else {
                const u64 L = lrx_window(w, a) >> (64 - 2 * KMC_LR_L);
                const u64 R = lrx_window(w, b) >> (64 - 2 * KMC_LR_R);
                hi = L >> (64 - 2 * KMC_LR_R);
                lo = (L << (2 * KMC_LR_R)) | R;
                nk++;
            }
        }
        const u64 o = (P0 - p_begin) * KMC_LRX_NS + q;
        out_hi[o] = hi;
        out_lo[o] = lo;
    }
    if (saw_bad) atomicOr((unsigned long long*)&counters[KMC_CTR_ERR], 4ull);
    nk = wave_sum_u64(nk);
    if ((tid & 63) == 0 && nk) atomicAdd((unsigned long long*)&counters[KMC_CTR_KMERS], nk);
}
