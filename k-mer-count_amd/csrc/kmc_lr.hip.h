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

// ---- LR mode as a throughput path: keys as pairs of 27-mer RANKS ----------------------------------------
// A key is (27-mer at p, 27-mer at p + d) with d = s - 27 in 53..=113, so every key of a batch is a pair of
// 27-mers of the SAME batch, and a batch of n bases has at most n distinct 27-mers.  Therefore:
//   1. kmc_lr_mer_kernel<0> the 27-mer at every base position (54 bits, one word); the MSD sort + run-length
//                           (kmc_msd.hip.h) turns the n of them into the sorted list of DISTINCT 27-mers of
//                           the batch: the dictionary, dict[rank] = 27-mer;
//   2. kmc_lr_mer_kernel<1> rank[q] = position of the 27-mer at q in the dictionary (binary search).  Rank
//                           order = string order (2 bits per base, MSB first);
//   3. kmc_lr_pair_kernel   every key of main.rs:76-79 as ONE word (rank[p] << B) | rank[p + d], B = bits of
//                           the number of distinct 27-mers: 2B <= 2 * log2(n) bits instead of 108, spread
//                           evenly over their range -- what the MSD sort + run-length likes best: two levels
//                           of one-word keys for 71 M keys where the 108-bit keys took five levels of two-word
//                           keys (all 61 keys of a window start share their first 54 bits);
//   4. kmc_lr_compose_kernel  (key, count) pairs of the sorted run back to 108-bit keys through the dictionary.
// The order of the pairs of ranks is the order of the pairs of 27-mers, i.e. of the reference's strings
// (main.rs:87), so the result is the same sorted (key, count) run as before; sort() + "equal lines" of the
// reference = MSD sort + run-length here, on keys a quarter of the size.
// A non-ACGT byte inside an emitted chunk raises error bit 4 (main.rs:23): its 27-mers get no rank.
#define KMC_LRX_POS 256
#define KMC_LRX_THREADS 256
#define KMC_LRX_NS (KMC_LR_SMAX - KMC_LR_SMIN + 1)   // 61 chunk sizes
#define KMC_LRX_DMIN (KMC_LR_SMIN - KMC_LR_R)        // 53: distance of the right 27-mer, smallest
#define KMC_LRX_DMAX (KMC_LR_SMAX - KMC_LR_R)        // 113: largest
#define KMC_LRX_SPAN (KMC_LRX_POS + 32)              // bases a workgroup of kmc_lr_mer_kernel looks at
#define KMC_LRX_WORDS ((KMC_LRX_SPAN + 15) / 16 + 3)
#define KMC_LR_NORANK 0xFFFFFFFFu

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

// 1. / 2. positions [q0, q1), M = the 27-mer at q (all-ones filler where fewer than 27 bases follow or one of
//    them is not ACGT).  MODE 0: out_lo[q - q0] = M, the keys of the dictionary sort (one word, 54 bits).
//    MODE 1: rank[q - q0] = index of M in the sorted dictionary (binary search, log2(n_dict) probes of an
//    array that sits in L2), KMC_LR_NORANK for the filler.  A workgroup packs its 256 + 26 bases into LDS at
//    2 bits per base once; every 27-mer is one funnel shift out of it.
template <int MODE>
__global__ __launch_bounds__(KMC_LRX_THREADS)
void kmc_lr_mer_kernel(const uint8_t* __restrict__ bases, u64 n_bases, u64 q0, u64 q1, u64* __restrict__ out_lo,
                       const u64* __restrict__ dict, u32 n_dict, u32* __restrict__ rank) {
    __shared__ u32 w[KMC_LRX_WORDS];
    __shared__ u32 bad[KMC_LRX_WORDS / 2 + 2];
    const u32 tid = threadIdx.x;
    const u64 Q0 = q0 + (u64)blockIdx.x * KMC_LRX_POS;
    if (Q0 >= q1) return;
    for (u32 i = tid; i < KMC_LRX_WORDS; i += KMC_LRX_THREADS) w[i] = 0;
    for (u32 i = tid; i < KMC_LRX_WORDS / 2 + 2; i += KMC_LRX_THREADS) bad[i] = 0;
    __syncthreads();
    for (u32 i = tid; i < KMC_LRX_SPAN; i += KMC_LRX_THREADS) {
        const u64 p = Q0 + i;
        if (p < n_bases) {
            const int code = kmc_code_of(bases[p]);
            if (code < 0) atomicOr(&bad[i >> 5], 0x80000000u >> (i & 31));
            else if (code) atomicOr(&w[i >> 4], (u32)code << (30 - 2 * (i & 15)));
        } else atomicOr(&bad[i >> 5], 0x80000000u >> (i & 31));
    }
    __syncthreads();
    const u64 q = Q0 + tid;
    if (q >= q1) return;
    const bool ok = !(lrx_badwin(bad, tid) & 0xFFFFFFE0u);  // the first 27 of 32
    const u64 M = lrx_window(w, tid) >> (64 - 2 * KMC_LR_L);
    if (MODE == 0) { out_lo[q - q0] = ok ? M : ~0ull; return; }
    u32 r = KMC_LR_NORANK;
    if (ok) {
        u32 lo_i = 0, hi_i = n_dict;  // last i with dict[i] <= M (M is in the dictionary)
        while (hi_i - lo_i > 1) {
            const u32 mid = (lo_i + hi_i) >> 1;
            if (dict[mid] <= M) lo_i = mid; else hi_i = mid;
        }
        r = lo_i;
    }
    rank[q - q0] = r;
}

// 3. A workgroup forms the keys of 256 consecutive window starts, size by size: its store j writes the keys of size
//    80 + j of all its starts (2 KB contiguous), key index = 61 * (P0 - p_begin) + j * npos + t.  (Round 2 laid a start's
//    61 keys side by side -- 61 consecutive keys with the same left rank, i.e. the same level-0 digit: the sort's histogram
//    and scatter kernels spent their time in same-address LDS atomics, 343 + 569 us for 71 M keys.  The sort does not care
//    where a key starts out.)  Pairs that do not exist (read too short for this size, window start in the last 79 bases
//    of a read) get the all-ones filler the sort drops.  rank[] is indexed from q0.
//    The read a start lies in: ONE 64-ary search per workgroup for the read of its first start, then the following read
//    ends from LDS (round 2: a 12-probe binary search over the offsets per thread, before the first store was issued).
__global__ __launch_bounds__(KMC_LRX_THREADS)
void kmc_lr_pair_kernel(const u64* __restrict__ offsets, u64 n_reads, u64 p_begin, u64 p_end, u64 q0, u64 nq, const u32* __restrict__ rank, int B,
                        u64* __restrict__ out_lo, u64* __restrict__ counters) {
    __shared__ u64 rend[KMC_LRX_POS];   // end of the read a window start lies in
    __shared__ u64 roff[KMC_LRX_POS + 1];   // read ends behind the workgroup's first start
    __shared__ u32 rk[KMC_LRX_POS + KMC_LRX_DMAX + 1];
    __shared__ u64 s_first;
    const u32 tid = threadIdx.x, lane = tid & 63;
    const u64 P0 = p_begin + (u64)blockIdx.x * KMC_LRX_POS;
    if (P0 >= p_end) return;
    const u32 npos = (u32)min((u64)KMC_LRX_POS, p_end - P0);
    for (u32 i = tid; i < KMC_LRX_POS + KMC_LRX_DMAX + 1; i += KMC_LRX_THREADS) rk[i] = (P0 + i - q0 < nq) ? rank[P0 + i - q0] : KMC_LR_NORANK;
    if (tid < 64) {   // wave 0: last r with offsets[r] <= P0  (offsets[0] = 0 <= P0 < offsets[n_reads])
        u64 lo_i = 0, hi_i = n_reads;
        while (hi_i - lo_i > 1) {
            const u64 step = (hi_i - lo_i + 63) / 64;
            const u64 idx = lo_i + (u64)lane * step;
            const bool le = idx < hi_i && offsets[idx] <= P0;
            const u32 cnt = (u32)__popcll(__builtin_amdgcn_ballot_w64(le));   // (lane 0 always: offsets[lo_i] <= P0)
            lo_i += (u64)(cnt - 1) * step;
            hi_i = min(lo_i + step, hi_i);
        }
        if (lane == 0) s_first = lo_i;
    }
    __syncthreads();
    const u64 rf = s_first;
    for (u32 i = tid; i < KMC_LRX_POS + 1; i += KMC_LRX_THREADS) roff[i] = rf + 1 + i <= n_reads ? offsets[rf + 1 + i] : ~0ull;
    __syncthreads();
    if (tid < npos) {
        const u64 p = P0 + tid;
        u64 e;
        if (roff[KMC_LRX_POS] > p) {   // first read end > p among the staged ones (sorted; roff[0] > P0)
            u32 lo_i = 0, hi_i = KMC_LRX_POS;   // roff[hi_i] > p
            if (roff[0] > p) hi_i = 0;
            while (hi_i - lo_i > 1) {
                const u32 mid = (lo_i + hi_i) >> 1;
                if (roff[mid] > p) hi_i = mid; else lo_i = mid;
            }
            e = roff[hi_i];
        } else {   // more than 256 reads end within these 256 positions (empty reads): the general search
            u64 lo_i = rf, hi_i = n_reads;
            while (hi_i - lo_i > 1) {
                const u64 mid = (lo_i + hi_i) >> 1;
                if (offsets[mid] <= p) lo_i = mid; else hi_i = mid;
            }
            e = offsets[lo_i + 1];
        }
        rend[tid] = e;
    }
    __syncthreads();
    bool saw_bad = false;
    if (tid < npos) {
        const u64 pe = rend[tid], p = P0 + tid;
        const u32 a = rk[tid];
        u64* const o = out_lo + (P0 - p_begin) * KMC_LRX_NS + tid;
#pragma unroll 4
        for (u32 j = 0; j < KMC_LRX_NS; ++j) {
            const u32 sz = KMC_LR_SMIN + j;
            u64 key = ~0ull;
            if (p + sz <= pe) {  // main.rs:73-75
                const u32 b = rk[tid + sz - KMC_LR_R];
                if (a == KMC_LR_NORANK || b == KMC_LR_NORANK) saw_bad = true;
                else key = ((u64)a << B) | b;
            }
            o[(size_t)j * npos] = key;
        }
    }
    if (saw_bad) atomicOr((unsigned long long*)&counters[KMC_CTR_ERR], 4ull);
}
// The keys of a batch are counted by their sort (its number of valid keys), added to the ctx counter behind it.  (The pair
// kernel used to add them up itself: one same-address atomic per wave, 25 k of them on the 71 M-key benchmark -- they, not
// its 568 MB of stores, were most of its 0.3 ms.)
__global__ void kmc_lr_addcount_kernel(const u32* __restrict__ n_valid, u64* __restrict__ counters) {
    if (blockIdx.x == 0 && threadIdx.x == 0) counters[KMC_CTR_KMERS] += *n_valid;
}

// 4. the sorted run's one-word keys back to {hi, lo} = L (54 bits) ++ R (54 bits)
__global__ void kmc_lr_compose_kernel(u64* __restrict__ r_hi, u64* __restrict__ r_lo, u64 n, const u64* __restrict__ dict, int B) {
    const u64 mask = (1ull << B) - 1ull;
    for (u64 i = blockIdx.x * (u64)blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) {
        const u64 key = r_lo[i];
        const u64 L = dict[key >> B], R = dict[key & mask];
        r_hi[i] = L >> (64 - 2 * KMC_LR_R);
        r_lo[i] = (L << (2 * KMC_LR_R)) | R;
    }
}
