// kmc_msd.hip.h -- hand-written MSD radix sort + run-length for packed k-mer keys: the GPU form of the
// reference's grouping step, bucket_sort / radix_sort / sort() + "count the repeated lines"
// (k-mer-count/src/main.rs:9-40,84,87), for inputs where almost every key is new (KMC_ALGO_SORT, the
// reference's own LR mode) and for ordering large count tables.
//
// The reference sorts 54-character strings one character position at a time from the END
// (main.rs:34-40: 53 stable bucket passes over everything).  Here keys are 2 bits per base, MSB
// first, so unsigned order = string order, and the sort goes from the FRONT, ten bits (five bases)
// per pass: two passes bring 2^30 keys down to buckets that fit LDS, so a key crosses HBM a few times
// instead of once per digit:
//
//   level l   every ACTIVE segment (a span of the key array whose keys share their first 10*l bits) is
//             partitioned by its next 10 bits: kmc_msd_hist_kernel counts per range of 65536 keys (level 0 of the
//             k-mer path gets its rows from the extraction kernel instead: kmc_extract.hip.h),
//             kmc_msd_scan_*_kernel turn the counts into destinations and classify the 1024 children,
//             kmc_msd_scatter_kernel moves the keys into the other buffer, tile by tile through LDS so
//             that every child receives contiguous runs.  Children of at most
//             KMC_MSD_LEAF keys, children whose keys are all equal and children with no bits left are
//             TERMINAL; the rest are the next level's active segments.  Positions are shared by both
//             buffers, so position order is key order at every level.
//   leaves    one workgroup per terminal segment: an LDS pass into 512 sub-buckets of the leaf's own key range,
//             every key then counts the smaller keys of its sub-bucket (three or four keys; sub-buckets of more
//             than 32 keys are split again and rank-sorted by a wave), then run-length: (key, count) pairs.
//   output    the terminals are ordered by position with a bitmap + popcount scan; a leaf writes its pairs into the
//             run at its own positions, which IS the dense sorted run when no key repeats; otherwise the pair counts
//             are scanned and the pairs gathered (work proportional to the distinct keys).
//
// Invalid positions (all-ones filler from the extraction kernel) fall into a 1025th bucket at level 0
// and are dropped.  Keys may carry a 64-bit weight (an existing count): the run-length then sums
// weights -- that is how count tables and earlier runs are merged and ordered.
#pragma once
#include <type_traits>
#include "kmc_device.hip.h"

#define KMC_MSD_RANGE 65536   // keys per histogram / scatter workgroup (a histogram row of 1025 counters per range): what the extraction
                              // kernel's rows cover; sorts that build their own rows take `rsz` = msd_range_for(n) instead
// keys per range of a sort of n keys: about two thousand ranges, i.e. workgroups per pass, at least (a 1.7 M-key sort -- the
// LR mode's dictionary -- ran its passes on 27 workgroups with 65536-key ranges: 90 us for a 14 MB histogram pass)
static inline unsigned msd_range_for(unsigned long long n) {
    unsigned r = 4096;
    while (r < KMC_MSD_RANGE && (unsigned long long)r * 2048 < n) r <<= 1;
    return r;
}
#define KMC_MSD_BITS 10       // digit width of a level (the last level of a key may be narrower)
#define KMC_MSD_ND (1 << KMC_MSD_BITS)
#define KMC_MSD_NB (KMC_MSD_ND + 1)   // digit bins + "invalid position"
#define KMC_MSD_THREADS 256
#define KMC_MSD_LEAF1 2048    // leaf capacity, one-word keys  (two LDS images of 16 KB: four leaves per CU in flight)
#ifndef KMC_MSD_LEAF2
#define KMC_MSD_LEAF2 2048    // leaf capacity, two-word keys without weights
#endif
#define KMC_MSD_LEAF2W 1024   // leaf capacity, two-word keys with weights
#ifndef KMC_MSD_THREAD_SORT
#define KMC_MSD_THREAD_SORT 32  // sub-buckets up to this size are insertion-sorted by one thread
#endif
#ifndef KMC_MSD_LEAF_LOGNSB
#define KMC_MSD_LEAF_LOGNSB 9
#endif
#define KMC_MSD_LEAF_NSB (1 << KMC_MSD_LEAF_LOGNSB)   // sub-buckets of a leaf: three or four keys each, so the per-thread insertion sorts
                                // (a chain of dependent LDS round trips per move) stay a handful of moves long

// hib = key bits still unsorted in the segment: its keys agree above bit hib, the next digit is bits
// [hib - w, hib) with w = min(hib, 10).  (Per segment, not per level: a segment whose keys turn out to
// share a long prefix -- the junction k-mers of repetitive input, the 61 LR keys of one window start --
// jumps over that prefix instead of taking one level per ten bits of it: when a level finds all keys
// of a segment in ONE digit, the segment is not moved; it is queued again, where it lies, with hib at the
// highest bit in which its smallest and largest key differ.)
struct MsdSeg { u32 begin, len, hib, parity; };  // parity: which of the two key buffers holds the segment
__device__ __forceinline__ int msd_seg_shift(u32 hib) { return hib > KMC_MSD_BITS ? (int)hib - KMC_MSD_BITS : 0; }
__device__ __forceinline__ u32 msd_seg_mask(u32 hib) { return hib >= KMC_MSD_BITS ? (u32)KMC_MSD_ND - 1u : (1u << hib) - 1u; }
// kind 0: leaf (sort in LDS); kind 1: all keys equal (one pair, key = first element); kind 2 | bits << 8: a span of ANY length
// whose keys differ in their lowest `bits` <= KMC_MSD_CNT_BITS bits only: counted in an LDS histogram (kmc_msd_count_kernel)
struct MsdTerm { u32 begin, len, kind, parity; };
#define KMC_MSD_CNT_BITS 14   // 2^14 u32 counters = 64 KB of LDS: two such workgroups per CU

// device-side bookkeeping of one sort (all counters of a level are zeroed by the host before use)
struct MsdCtl {
    u32 n_next;      // active segments appended for the next level
    u32 n_term;      // terminals so far
    u32 n_valid;     // keys that are not filler (written at level 0)
    u32 n_ranges;    // ranges of the current level (written by kmc_msd_ranges_kernel)
    u32 overflow;    // a list ran out of room (host checks)
    u32 n_pairs;     // total (key, count) pairs (written by the terminal scan, when one is needed)
    u32 scan_total;  // total of the last kmc_scan_* call
    u32 pad;
    unsigned long long w_total;  // sum of all weights (weighted sorts: the merged table's total count)
    u32 n_cnt;       // kind-2 terminals made so far (kmc_msd_scan_kernel)
    u32 n_cnt2;      // ... listed by kmc_msd_order_kernel
    u32 n_dups[64];  // keys that repeat an earlier key of their terminal (terminal t adds to slot t % 64: one word would
                     // take 400 k same-address atomics on heavily repeated keys): pairs = valid keys - their sum
};

// Where the histograms live: counter (range r, digit d).  Blocks of 64 ranges, inside a block digit-major: the 64
// ranges' counters of one digit are 256 contiguous bytes, so the column scans of kmc_msd_scan_a_kernel (a wave walks
// ONE digit over its segment's ranges, 64 at a time) read and write whole lines.  Row-major hist[r][1025] made every
// lane of those scans touch its own line: 0.8 ms per GB of keys at level 0 for 57 MB of counters.  The price is paid
// by the writers (a range stores its 1025 counters 256 bytes apart) where it is hidden: fire-and-forget stores
// behind kernels that move gigabytes.  A buffer for R ranges holds msd_hist_words(R) counters.
__host__ __device__ __forceinline__ size_t msd_hist_idx(size_t r, u32 d) { return ((r >> 6) * KMC_MSD_NB + d) * 64 + (r & 63); }
static inline size_t msd_hist_words(size_t n_ranges) { return ((n_ranges + 63) / 64) * 64 * (size_t)KMC_MSD_NB; }

template <int KW>
__device__ __forceinline__ bool msd_is_filler(u64 hi, u64 lo, int kb) {
    if (KW == 1) return kb < 64 && (lo >> kb) != 0;
    return (hi >> (kb - 64)) != 0;  // kb in [64, 126]
}
// bits [shift, shift + 32) of the key (the caller masks)
template <int KW>
__device__ __forceinline__ u32 msd_bits(u64 hi, u64 lo, int shift) {
    if (KW == 1) return (u32)(lo >> shift);
    if (shift >= 64) return (u32)(hi >> (shift - 64));
    if (shift == 0) return (u32)lo;
    return (u32)((lo >> shift) | (hi << (64 - shift)));
}

// ---- exclusive scan of a u32 sequence, three small kernels (block sums, scan of the sums, final) ----
// MODE 0: the values themselves; MODE 1: popcount of 64-bit words (bitmap rank)
#define KMC_SCAN_PER_BLOCK 2048
template <int MODE>
__device__ __forceinline__ u32 scan_value(const void* src, u32 i) {
    if (MODE == 0) return reinterpret_cast<const u32*>(src)[i];
    return (u32)__popcll(reinterpret_cast<const unsigned long long*>(src)[i]);
}
template <int MODE>
__global__ __launch_bounds__(256)
void kmc_scan_sums_kernel(const void* __restrict__ src, u32 n, u32* __restrict__ bsum) {
    __shared__ u32 ws[4];
    const u32 tid = threadIdx.x, i0 = blockIdx.x * KMC_SCAN_PER_BLOCK + tid * 8;
    u32 s = 0;
#pragma unroll
    for (int e = 0; e < 8; ++e) if (i0 + e < n) s += scan_value<MODE>(src, i0 + e);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if ((tid & 63) == 0) ws[tid >> 6] = s;
    __syncthreads();
    if (tid == 0) bsum[blockIdx.x] = ws[0] + ws[1] + ws[2] + ws[3];
}
// in-place exclusive scan of up to a few hundred thousand block sums; total -> *total (one workgroup)
__global__ __launch_bounds__(1024)
void kmc_scan_top_kernel(u32* __restrict__ bsum, u32 nb, u32* __restrict__ total) {
    __shared__ u32 part[1024];
    const u32 tid = threadIdx.x;
    const u32 per = (nb + 1023) / 1024;
    const u32 a = min(tid * per, nb), b = min(a + per, nb);
    u32 s = 0;
    for (u32 i = a; i < b; ++i) s += bsum[i];
    part[tid] = s;
    __syncthreads();
    for (u32 o = 1; o < 1024; o <<= 1) {
        u32 v = tid >= o ? part[tid - o] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    u32 run = tid ? part[tid - 1] : 0;
    for (u32 i = a; i < b; ++i) { const u32 v = bsum[i]; bsum[i] = run; run += v; }
    if (tid == 1023) *total = part[1023];
}
template <int MODE>
__global__ __launch_bounds__(256)
void kmc_scan_final_kernel(const void* __restrict__ src, u32 n, const u32* __restrict__ bbase, u32* __restrict__ out) {
    __shared__ u32 ws[4];
    const u32 tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, i0 = blockIdx.x * KMC_SCAN_PER_BLOCK + tid * 8;
    u32 v[8], s = 0;
#pragma unroll
    for (int e = 0; e < 8; ++e) { v[e] = (i0 + e < n) ? scan_value<MODE>(src, i0 + e) : 0u; s += v[e]; }
    u32 inc = s;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const u32 t = __shfl_up(inc, o); if ((int)lane >= o) inc += t; }
    if (lane == 63) ws[wv] = inc;
    __syncthreads();
    u32 run = bbase[blockIdx.x] + inc - s;
    for (u32 w = 0; w < wv; ++w) run += ws[w];
#pragma unroll
    for (int e = 0; e < 8; ++e) { if (i0 + e < n) out[i0 + e] = run; run += v[e]; }
}

// first[i] = number of ranges of segments 0..i-1; ctl->n_ranges = total.  One workgroup.
__global__ __launch_bounds__(1024)
void kmc_msd_ranges_kernel(const MsdSeg* __restrict__ seg, u32 n_seg, u32 rsz, u32* __restrict__ first, MsdCtl* ctl) {
    __shared__ u32 part[1024];
    const u32 tid = threadIdx.x;
    const u32 per = (n_seg + 1023) / 1024;
    const u32 a = min(tid * per, n_seg), b = min(a + per, n_seg);
    u32 s = 0;
    for (u32 i = a; i < b; ++i) s += (seg[i].len + rsz - 1) / rsz;
    part[tid] = s;
    __syncthreads();
    for (u32 o = 1; o < 1024; o <<= 1) {
        u32 v = tid >= o ? part[tid - o] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    u32 run = tid ? part[tid - 1] : 0;
    for (u32 i = a; i < b; ++i) { first[i] = run; run += (seg[i].len + rsz - 1) / rsz; }
    if (tid == 1023) { first[n_seg] = part[1023]; ctl->n_ranges = part[1023]; }
}

// segment of range r: last s with first[s] <= r
__device__ __forceinline__ u32 msd_seg_of(const u32* __restrict__ first, u32 n_seg, u32 r) {
    u32 lo = 0, hi = n_seg;
    while (hi - lo > 1) {
        const u32 mid = (lo + hi) >> 1;
        if (first[mid] <= r) lo = mid; else hi = mid;
    }
    return lo;
}

// per range: digit histogram (msd_hist_idx) and min / max key (all-equal segments end here)
template <int KW>
__global__ __launch_bounds__(KMC_MSD_THREADS)
void kmc_msd_hist_kernel(const u64* __restrict__ hi0, const u64* __restrict__ lo0, const u64* __restrict__ hi1, const u64* __restrict__ lo1,
                         const MsdSeg* __restrict__ seg, u32 n_seg,
                         const u32* __restrict__ first, u32 rsz, int kb, int level0,
                         u32* __restrict__ hist, u64* __restrict__ rmin, u64* __restrict__ rmax, const MsdCtl* __restrict__ ctl) {
    __shared__ u32 h[4][KMC_MSD_NB + 3];
    __shared__ u64 smin[4][2], smax[4][2];
    const u32 r = blockIdx.x, tid = threadIdx.x, wv = tid >> 6;
    if (r >= ctl->n_ranges) return;  // (the grid is the host's upper bound)
    for (u32 i = tid; i < 4 * (KMC_MSD_NB + 3); i += KMC_MSD_THREADS) (&h[0][0])[i] = 0;
    __syncthreads();
    const u32 s = msd_seg_of(first, n_seg, r);
    const u32 idx = r - first[s];
    const u32 b = seg[s].begin + idx * rsz;
    const u32 n = min(rsz, seg[s].len - idx * rsz);
    const int shift = msd_seg_shift(seg[s].hib);
    const u32 mask = msd_seg_mask(seg[s].hib);
    const u64* const khi = seg[s].parity ? hi1 : hi0;
    const u64* const klo = seg[s].parity ? lo1 : lo0;
    u64 mnh = ~0ull, mnl = ~0ull, mxh = 0, mxl = 0;
    for (u32 i = tid; i < n; i += KMC_MSD_THREADS) {
        const u64 lo = klo[b + i], hi = KW == 2 ? khi[b + i] : 0ull;
        u32 d;
        if (level0 && msd_is_filler<KW>(hi, lo, kb)) d = KMC_MSD_ND;
        else {
            d = msd_bits<KW>(hi, lo, shift) & mask;
            if (key_less(hi, lo, mnh, mnl)) { mnh = hi; mnl = lo; }
            if (key_less(mxh, mxl, hi, lo)) { mxh = hi; mxl = lo; }
        }
        atomicAdd(&h[wv][d], 1u);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const u64 oh = __shfl_xor(mnh, o), ol = __shfl_xor(mnl, o);
        if (key_less(oh, ol, mnh, mnl)) { mnh = oh; mnl = ol; }
        const u64 ph = __shfl_xor(mxh, o), pl = __shfl_xor(mxl, o);
        if (key_less(mxh, mxl, ph, pl)) { mxh = ph; mxl = pl; }
    }
    if ((tid & 63) == 0) { smin[wv][0] = mnh; smin[wv][1] = mnl; smax[wv][0] = mxh; smax[wv][1] = mxl; }
    __syncthreads();
    for (u32 d = tid; d < KMC_MSD_NB; d += KMC_MSD_THREADS) hist[msd_hist_idx(r, d)] = h[0][d] + h[1][d] + h[2][d] + h[3][d];
    if (tid == 0) {
        for (int w = 1; w < 4; ++w) {
            if (key_less(smin[w][0], smin[w][1], mnh, mnl)) { mnh = smin[w][0]; mnl = smin[w][1]; }
            if (key_less(mxh, mxl, smax[w][0], smax[w][1])) { mxh = smax[w][0]; mxl = smax[w][1]; }
        }
        rmin[2 * (size_t)r] = mnh; rmin[2 * (size_t)r + 1] = mnl;
        rmax[2 * (size_t)r] = mxh; rmax[2 * (size_t)r + 1] = mxl;
    }
}

// Scan, part A: for every (segment, digit) the exclusive running offsets over the segment's ranges (in
// place in hist) and the digit's total (stot[s][NB]).  One WAVE scans one digit's column 64 ranges at a
// time; the grid is n_seg x S workgroups of four waves, wave w of workgroup j of a segment takes the
// digits j*4 + w + 4*S*i.  (Level 0 is ONE segment of tens of thousands of ranges, which a single
// workgroup would walk for milliseconds: S = 64 there; S = 1 when there are many segments.)
__global__ __launch_bounds__(KMC_MSD_THREADS)
void kmc_msd_scan_a_kernel(u32 n_seg, u32 S, const u32* __restrict__ first, u32* __restrict__ hist, u32* __restrict__ stot) {
    const u32 s = blockIdx.x / S, j = blockIdx.x % S, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (s >= n_seg) return;
    const u32 r0 = first[s], r1 = first[s + 1];
    if (r1 - r0 == 1) return;  // one range: its histogram IS the total (kmc_msd_scan_kernel takes it from there)
    for (u32 d = j * 4 + wv; d < KMC_MSD_NB; d += 4 * S) {
        u32 run = 0;
        // four blocks of 64 ranges per trip: their loads are in flight together (one load per trip left the single wave
        // of a column waiting out a memory round trip 217 times at level 0 of a 1 GB sort: 0.66 ms for 57 MB)
        for (u32 base = r0; base < r1; base += 256) {
            u32 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const u32 r = base + 64u * u + lane;
                v[u] = r < r1 ? hist[msd_hist_idx(r, d)] : 0u;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const u32 r = base + 64u * u + lane;
                u32 inc = v[u];
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) { const u32 t = __shfl_up(inc, o); if ((int)lane >= o) inc += t; }
                if (r < r1) hist[msd_hist_idx(r, d)] = run + inc - v[u];
                run += __shfl(inc, 63);
            }
        }
        if (lane == 0) stot[(size_t)s * KMC_MSD_NB + d] = run;
    }
}

// Scan, part B: one workgroup (one thread per digit) per active segment: child destinations
// (cbase[s][NB]) from the digit totals, and the classification of the children.
//   seg_skip[s] = 1: every key of the segment is equal -- it becomes a terminal as it stands (in the
//   SOURCE buffer) and its ranges are not scattered.
__global__ __launch_bounds__(KMC_MSD_ND)
void kmc_msd_scan_kernel(const MsdSeg* __restrict__ seg, u32 n_seg, const u32* __restrict__ first, u32* __restrict__ hist, const u32* __restrict__ stot,
                         const u64* __restrict__ rmin, const u64* __restrict__ rmax, u32* __restrict__ cbase, u32* __restrict__ seg_skip,
                         int level0, int wide, u32 leaf_cap, u32 cnt_bits,
                         MsdSeg* __restrict__ next, u32 next_cap, MsdTerm* __restrict__ term, u32 term_cap,
                         unsigned long long* __restrict__ bitmap, MsdCtl* ctl) {
    constexpr int NWV = KMC_MSD_ND / 64;
    __shared__ u32 tot[KMC_MSD_NB + 7];
    __shared__ u32 cb_s[KMC_MSD_ND];
    __shared__ u32 wsum[NWV];
    __shared__ u64 smm[NWV][4];
    const u32 s = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const u32 r0 = first[s], r1 = first[s + 1];
    // are all keys of the segment equal?  (fold the ranges' min / max)
    {
        u64 mnh = ~0ull, mnl = ~0ull, mxh = 0, mxl = 0;
        for (u32 r = r0 + tid; r < r1; r += KMC_MSD_ND) {
            const u64 ah = rmin[2 * (size_t)r], al = rmin[2 * (size_t)r + 1], bh = rmax[2 * (size_t)r], bl = rmax[2 * (size_t)r + 1];
            if (key_less(ah, al, mnh, mnl)) { mnh = ah; mnl = al; }
            if (key_less(mxh, mxl, bh, bl)) { mxh = bh; mxl = bl; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const u64 oh = __shfl_xor(mnh, o), ol = __shfl_xor(mnl, o);
            if (key_less(oh, ol, mnh, mnl)) { mnh = oh; mnl = ol; }
            const u64 ph = __shfl_xor(mxh, o), pl = __shfl_xor(mxl, o);
            if (key_less(mxh, mxl, ph, pl)) { mxh = ph; mxl = pl; }
        }
        if (lane == 0) { smm[wv][0] = mnh; smm[wv][1] = mnl; smm[wv][2] = mxh; smm[wv][3] = mxl; }
    }
    if (r1 - r0 == 1) {  // a segment of one range (most segments below level 0): no column scan was run for it
        for (u32 d = tid; d < KMC_MSD_NB; d += KMC_MSD_ND) { tot[d] = hist[msd_hist_idx(r0, d)]; hist[msd_hist_idx(r0, d)] = 0; }
    } else {
        for (u32 d = tid; d < KMC_MSD_NB; d += KMC_MSD_ND) tot[d] = stot[(size_t)s * KMC_MSD_NB + d];
    }
    __syncthreads();
    bool all_equal;
    int top_diff = -1;  // highest bit in which two keys of the segment differ
    {
        u64 mnh = smm[0][0], mnl = smm[0][1], mxh = smm[0][2], mxl = smm[0][3];
        for (int w = 1; w < NWV; ++w) {
            if (key_less(smm[w][0], smm[w][1], mnh, mnl)) { mnh = smm[w][0]; mnl = smm[w][1]; }
            if (key_less(mxh, mxl, smm[w][2], smm[w][3])) { mxh = smm[w][2]; mxl = smm[w][3]; }
        }
        all_equal = mnh == mxh && mnl == mxl;
        const u64 xh = mnh ^ mxh, xl = mnl ^ mxl;
        if (xh) top_diff = 127 - __clzll((long long)xh);
        else if (xl) top_diff = 63 - __clzll((long long)xl);
    }
    const int shift = msd_seg_shift(seg[s].hib);
    const bool last_level = shift == 0;
    const u32 src_parity = seg[s].parity;
    const u32 n_filler = level0 ? tot[KMC_MSD_ND] : 0;
    const bool equal = all_equal && seg[s].len > n_filler && (!level0 || n_filler == 0);
    // exclusive scan of tot[0..ND) -> child begin (the filler bucket is dropped)
    const u32 mine = tot[tid];
    u32 inc = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const u32 v = __shfl_up(inc, o); if ((int)lane >= o) inc += v; }
    if (lane == 63) wsum[wv] = inc;
    __syncthreads();
    u32 wbase = 0;
    for (u32 w = 0; w < wv; ++w) wbase += wsum[w];
    const u32 cb = seg[s].begin + wbase + inc - mine;
    cbase[(size_t)s * KMC_MSD_NB + tid] = cb;
    cb_s[tid] = cb;
    __syncthreads();
    // What the segment turns into.  Wave 0 decides, the other fifteen are done (a workgroup that waits for
    // one thread holds a thousand thread slots: with 400 k segments per level that tripled the kernel).
    // The 1024 totals sit in wave 0's registers, sixteen per lane; the walk over them is wave-uniform
    // (v_readlane, everything in scalar registers) and skips empty digits by ballot; the lists go to LDS
    // and all 64 lanes write them out.  (The first version had thread 0 read every total from LDS, store
    // every list entry to global memory and set its bitmap bit itself, in two passes: 120-280 us per
    // level even for a single segment -- a fifth of the LR mode's sort.)
    //   - all keys equal: a terminal as it stands (source buffer), one pair
    //   - every key in ONE digit (they agree above bit top_diff, which lies below this digit): the segment
    //     stays where it is and is queued again at the first bit that splits it
    //   - otherwise its children, in position order: large ones go on to the next level (or, with no bits
    //     left, are one pair each); runs of consecutive small ones are merged into leaves of at most
    //     leaf_cap keys (a leaf sorts whatever keys it holds, so it need not be a single child)
    //   `wide` (the host sets it while a level has a handful of segments -- level 0 is ONE): every wave walks its own
    //   64 digits and keeps its own lists (leaves are then not merged across a multiple of 64 digits).  A lone wave
    //   issues an instruction every four or five cycles: the walk over 1024 occupied digits took 83-98 us per level-0
    //   scan, as much as the LR mode's whole dictionary histogram.
    __shared__ u32 l_tb[KMC_MSD_ND], l_tl[KMC_MSD_ND], l_tk[KMC_MSD_ND], l_nb[KMC_MSD_ND], l_nl[KMC_MSD_ND];
    const bool nomove = !equal && !last_level && top_diff >= 0 && top_diff < shift && n_filler == 0;
    if (wv != 0 && (!wide || equal || nomove)) return;
    const u32 lb = wide ? wv * 64u : 0u;   // this wave's part of the lists
    if (tid == 0) {
        cbase[(size_t)s * KMC_MSD_NB + KMC_MSD_ND] = 0;
        seg_skip[s] = (equal || nomove) ? 1u : 0u;
        if (level0) ctl->n_valid = seg[s].len - n_filler;
    }
    u32 nt = 0, nn = 0, ncnt = 0;
    u32 q_hib = (u32)shift, q_par = src_parity ^ 1u, t_par = src_parity ^ 1u;
    if (equal) {
        if (lane == 0) { l_tb[0] = seg[s].begin; l_tl[0] = seg[s].len; l_tk[0] = 1u; }
        nt = 1; t_par = src_parity;
    } else if (nomove && cnt_bits && (u32)(top_diff + 1) <= cnt_bits && seg[s].len > leaf_cap) {
        // its keys differ in a few low bits only: counted where it lies, however long it is
        if (lane == 0) { l_tb[0] = seg[s].begin; l_tl[0] = seg[s].len; l_tk[0] = 2u | ((u32)(top_diff + 1) << 8); }
        nt = 1; ncnt = 1; t_par = src_parity;
    } else if (nomove) {
        if (lane == 0) { l_nb[0] = seg[s].begin; l_nl[0] = seg[s].len; }
        nn = 1; q_hib = (u32)(top_diff + 1); q_par = src_parity;
    } else {
        u32 gb = 0, gl = 0;
        auto flush = [&]() { if (lane == 0) { l_tb[lb + nt] = gb; l_tl[lb + nt] = gl; l_tk[lb + nt] = gl == 1 ? 1u : 0u; } ++nt; gl = 0; };
        const int jb = wide ? (int)wv : 0, je = wide ? (int)wv + 1 : NWV;
        for (int j = jb; j < je; ++j) {
            const u32 Tj = tot[j * 64 + lane], Cj = cb_s[j * 64 + lane];
            unsigned long long live = __builtin_amdgcn_ballot_w64(Tj != 0);
            while (live) {
                const int bl = (int)__builtin_ctzll(live);
                live &= live - 1;
                const u32 m = (u32)__builtin_amdgcn_readlane((int)Tj, bl);
                const u32 cbv = (u32)__builtin_amdgcn_readlane((int)Cj, bl);
                if (m > leaf_cap) {
                    if (gl) flush();
                    if (last_level) { if (lane == 0) { l_tb[lb + nt] = cbv; l_tl[lb + nt] = m; l_tk[lb + nt] = 1u; } ++nt; }
                    else if (cnt_bits && (u32)shift <= cnt_bits) {   // few bits left, many keys: no further level, an LDS histogram
                        if (lane == 0) { l_tb[lb + nt] = cbv; l_tl[lb + nt] = m; l_tk[lb + nt] = 2u | ((u32)shift << 8); }
                        ++nt; ++ncnt;
                    }
                    else { if (lane == 0) { l_nb[lb + nn] = cbv; l_nl[lb + nn] = m; } ++nn; }
                } else {
                    if (gl && gl + m > leaf_cap) flush();
                    if (!gl) gb = cbv;
                    gl += m;
                }
            }
        }
        if (gl) flush();
    }
    u32 t_base = 0, n_base = 0;  // ONE returning atomic per list (one per terminal cost 1.6 ms per level)
    if (lane == 0) {
        if (nt) { t_base = atomicAdd(&ctl->n_term, nt); if (t_base + nt > term_cap) atomicOr(&ctl->overflow, 1u); }
        if (nn) { n_base = atomicAdd(&ctl->n_next, nn); if (n_base + nn > next_cap) atomicOr(&ctl->overflow, 2u); }
        if (ncnt) atomicAdd(&ctl->n_cnt, ncnt);
    }
    t_base = (u32)__builtin_amdgcn_readfirstlane((int)t_base);
    n_base = (u32)__builtin_amdgcn_readfirstlane((int)n_base);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    for (u32 i = lane; i < nt; i += 64) {
        if (t_base + i >= term_cap) break;
        const u32 bg = l_tb[lb + i];
        term[t_base + i] = MsdTerm{bg, l_tl[lb + i], l_tk[lb + i], t_par};
        atomicOr(&bitmap[bg >> 6], 1ull << (bg & 63));
    }
    for (u32 i = lane; i < nn; i += 64) {
        if (n_base + i >= next_cap) break;
        next[n_base + i] = MsdSeg{l_nb[lb + i], l_nl[lb + i], q_hib, q_par};
    }
}

#ifdef KMC_LEAF_STAMPS
// diagnostic build only (tools/leaf_stamps.py): cycles per phase of the leaf kernel, summed over all workgroups (thread 0's clock)
__device__ unsigned long long kmc_leaf_stamps[16];
#define LEAF_STAMP(i) do { __syncthreads(); if (threadIdx.x == 0) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); atomicAdd(&kmc_leaf_stamps[i], now_ - t_prev_); t_prev_ = __builtin_amdgcn_s_memtime(); } } while (0)
#else
#define LEAF_STAMP(i) do { } while (0)
#endif
// Per range: move every key (and weight) to its child's span in the other buffer.  A tile of keys is
// grouped by digit in LDS first (rank within the tile from one returning LDS add per key), then
// written out in that order: consecutive lanes write consecutive addresses of one child, so a child
// receives a contiguous run per tile instead of single 8-byte stores (the first version scattered
// straight from registers: 1.4 TB/s).
template <int KW, bool WEIGHTS> struct MsdScatterLds {
    static constexpr int WORDS = KW + (WEIGHTS ? 1 : 0);
    // 128 KB of keys per tile, one workgroup per CU: a child receives 16 keys = 128 contiguous bytes per tile
    // (tiles of 64 KB, two workgroups per CU: 64-byte fragments; same-box A/B on 831 M keys 35.9 -> 33.6 ms)
    static constexpr int TILE = WORDS == 1 ? 16384 : (WORDS == 2 ? 8192 : 4096);
    u64 lo[TILE];
    u64 hi[KW == 2 ? TILE : 1];
    u64 w[WEIGHTS ? TILE : 1];
    u32 cnt[KMC_MSD_NB + 3];   // keys of the tile per digit, then their exclusive prefix
    u32 gdst[KMC_MSD_NB + 3];  // global index of the tile's first key of the digit, minus its LDS index
    u32 cur[KMC_MSD_NB + 3];   // the range's running destination per digit
    u32 wsum[16];
};
template <int KW, bool WEIGHTS>
__global__ __launch_bounds__(1024)
void kmc_msd_scatter_kernel(u64* __restrict__ hi0, u64* __restrict__ lo0, u64* __restrict__ w0,
                            u64* __restrict__ hi1, u64* __restrict__ lo1, u64* __restrict__ w1,
                            const MsdSeg* __restrict__ seg, u32 n_seg, const u32* __restrict__ first, u32 rsz,
                            const u32* __restrict__ hist, const u32* __restrict__ cbase, const u32* __restrict__ seg_skip,
                            int kb, int level0, const MsdCtl* __restrict__ ctl) {
    extern __shared__ __align__(16) unsigned char msd_smem[];
    typedef MsdScatterLds<KW, WEIGHTS> LT;
    LT& L = *reinterpret_cast<LT*>(msd_smem);
    constexpr int TILE = LT::TILE, PER = TILE / 1024;
    const u32 r = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (r >= ctl->n_ranges) return;
    const u32 s = msd_seg_of(first, n_seg, r);
    if (seg_skip[s]) return;  // (block-uniform)
    const int shift = msd_seg_shift(seg[s].hib);
    const u32 mask = msd_seg_mask(seg[s].hib);
    const bool sp = seg[s].parity != 0;  // from the segment's buffer into the other one
    const u64* const khi = sp ? hi1 : hi0;
    const u64* const klo = sp ? lo1 : lo0;
    const u64* const kw = sp ? w1 : w0;
    u64* const ohi = sp ? hi0 : hi1;
    u64* const olo = sp ? lo0 : lo1;
    u64* const ow = sp ? w0 : w1;
    for (u32 d = tid; d < KMC_MSD_NB; d += 1024) L.cur[d] = cbase[(size_t)s * KMC_MSD_NB + d] + hist[msd_hist_idx(r, d)];
    const u32 idx = r - first[s];
    const u32 b = seg[s].begin + idx * rsz;
    const u32 n = min(rsz, seg[s].len - idx * rsz);
    // the keys of the first tile; every later tile is loaded while the one before it goes through LDS.
    // A tile comes in as ALIGNED PAIRS (16 bytes per lane and load, 1 KiB per wave instruction): register slot s of the
    // workgroup (thread t: slots 2 (t + 1024 e) + {0, 1}) holds key s + off of the tile, off = parity of the tile's first
    // index; with an odd first index the tile's first key goes to slot tn - 1, which is free then.  Which key sits in which
    // slot is irrelevant (ranks within the tile come from LDS atomics).  Half a pair may be a neighbouring range's key or
    // lie up to 8 bytes past the array (inside the allocation's slack): loaded, never used.
    static_assert(PER % 2 == 0, "keys per thread must be even");
    typedef unsigned long long u64x2_t __attribute__((ext_vector_type(2)));
    const u32 off = b & 1u;   // (tiles start at multiples of TILE from b: the same parity for every tile of the range)
    u64 nlo[PER], nhi[PER], nw[PER];
    auto load_tile = [&](u32 t0, u32 tn) {
#pragma unroll
        for (int e = 0; e < PER / 2; ++e) {
            const u32 s0 = 2u * (tid + 1024u * e);
            if (s0 + off < tn) {
                const size_t g = (size_t)b + t0 + off + s0;   // even
                const u64x2_t pl = *reinterpret_cast<const u64x2_t*>(klo + g);
                nlo[2 * e] = pl.x; nlo[2 * e + 1] = pl.y;
                if (KW == 2) { const u64x2_t ph = *reinterpret_cast<const u64x2_t*>(khi + g); nhi[2 * e] = ph.x; nhi[2 * e + 1] = ph.y; }
                if (WEIGHTS) { const u64x2_t pw = *reinterpret_cast<const u64x2_t*>(kw + g); nw[2 * e] = pw.x; nw[2 * e + 1] = pw.y; }
            }
            if (off) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    if (s0 + h == tn - 1) {
                        nlo[2 * e + h] = klo[(size_t)b + t0];
                        if (KW == 2) nhi[2 * e + h] = khi[(size_t)b + t0];
                        if (WEIGHTS) nw[2 * e + h] = kw[(size_t)b + t0];
                    }
                }
            }
        }
    };
#pragma unroll
    for (int e = 0; e < PER; ++e) { nlo[e] = 0; nhi[e] = 0; nw[e] = 0; }
    load_tile(0, min((u32)TILE, n));
#ifdef KMC_LEAF_STAMPS
    unsigned long long t_prev_ = __builtin_amdgcn_s_memtime();
#endif
    for (u32 t0 = 0; t0 < n; t0 += TILE) {
        const u32 tn = min((u32)TILE, n - t0);
        for (u32 d = tid; d < KMC_MSD_NB; d += 1024) L.cnt[d] = 0;
        __syncthreads();
        LEAF_STAMP(8);    // (scatter) waiting for the stores of the tile before / tile entry
        // 1. my keys, their digits and their ranks within the tile
        u64 mlo[PER], mhi[PER], mw[PER];
        u32 md[PER];  // digit (11 bits) | rank within the tile << 11; ~0: no key  (one register per key: at 16 keys
                      // per thread a second array spilled)
#pragma unroll
        for (int e = 0; e < PER; ++e) {
            md[e] = ~0u;
            mlo[e] = nlo[e]; mhi[e] = nhi[e]; mw[e] = nw[e];
            if (2u * (tid + 1024u * (e >> 1)) + (e & 1) < tn) {   // (slot < tn: every such slot holds a key of the tile)
                const u32 d = (level0 && msd_is_filler<KW>(mhi[e], mlo[e], kb)) ? (u32)KMC_MSD_ND : (msd_bits<KW>(mhi[e], mlo[e], shift) & mask);
                md[e] = d | (atomicAdd(&L.cnt[d], 1u) << 11);
            }
        }
        if (t0 + TILE < n) load_tile(t0 + TILE, min((u32)TILE, n - t0 - TILE));  // (block-uniform) next tile's loads go out now and land during steps 2-4
        __syncthreads();
        LEAF_STAMP(9);    // (scatter) digits + rank atomics (incl. waiting for this tile's loads)
        // 2. exclusive prefix of the tile's digit counts (thread d <-> digit d; the filler bin comes last)
        {
            const u32 c = L.cnt[tid];
            u32 inc = c;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const u32 v = __shfl_up(inc, o); if ((int)lane >= o) inc += v; }
            if (lane == 63) L.wsum[wv] = inc;
            __syncthreads();
            u32 base = 0;
            for (u32 w = 0; w < wv; ++w) base += L.wsum[w];
            const u32 off = base + inc - c;
            L.cnt[tid] = off;
            L.gdst[tid] = L.cur[tid] - off;
            L.cur[tid] += c;
            if (tid == 1023) L.cnt[KMC_MSD_ND] = off + c;  // fillers sit behind every key and are not written out
        }
        __syncthreads();
        LEAF_STAMP(10);   // (scatter) prefix
        // 3. into LDS, grouped by digit
#pragma unroll
        for (int e = 0; e < PER; ++e) {
            if (md[e] != ~0u) {
                const u32 p = L.cnt[md[e] & 2047u] + (md[e] >> 11);
                L.lo[p] = mlo[e];
                if (KW == 2) L.hi[p] = mhi[e];
                if (WEIGHTS) L.w[p] = mw[e];
            }
        }
        __syncthreads();
        LEAF_STAMP(11);   // (scatter) LDS scatter
        // 4. out, in LDS order
        const u32 n_keys = L.cnt[KMC_MSD_ND];  // keys of the tile that are not filler
#pragma unroll
        for (int e = 0; e < PER; ++e) {
            const u32 p = tid + 1024u * e;
            if (p < n_keys) {
                const u64 lo = L.lo[p], hi = KW == 2 ? L.hi[p] : 0ull;
                const u32 d = msd_bits<KW>(hi, lo, shift) & mask;
                const u32 q = L.gdst[d] + p;
                olo[q] = lo;
                if (KW == 2) ohi[q] = hi;
                if (WEIGHTS) ow[q] = L.w[p];
            }
        }
        __syncthreads();
        LEAF_STAMP(12);   // (scatter) out
    }
}

// ordered[t] = the terminal whose begin has ordinal t (rank = exclusive popcount prefix of the bitmap words)
// (and the kind-2 terminals' ordinals listed in clist, for kmc_msd_count_kernel)
__global__ void kmc_msd_order_kernel(const MsdTerm* __restrict__ term, u32 n_term, const unsigned long long* __restrict__ bitmap,
                                     const u32* __restrict__ rank, MsdTerm* __restrict__ ordered, u32* __restrict__ clist, MsdCtl* ctl) {
    for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < n_term; i += gridDim.x * blockDim.x) {
        const u32 p = term[i].begin;
        const unsigned long long below = bitmap[p >> 6] & ((1ull << (p & 63)) - 1ull);
        const u32 o = rank[p >> 6] + (u32)__popcll(below);
        ordered[o] = term[i];
        if ((term[i].kind & 0xFFu) == 2u) clist[atomicAdd(&ctl->n_cnt2, 1u)] = o;
    }
}

// ---- kind-2 terminals: an LDS histogram over the bits that are left ---------------------------------------------
// One workgroup per listed terminal: every key adds one to counter [key & mask]; the non-zero counters, in order, are the
// terminal's (key, count) pairs -- written to the run at the terminal's own positions like a leaf's (there are never
// more pairs than keys).  What it replaces: the LR mode's rank pairs are 2 * log2(distinct 27-mers) bits long, 24 on the
// 4000-record benchmark, so after ONE level every segment still held 70 k keys of 14 bits: a second and a third level
// (0.79 + 0.09 ms for 71 M keys) and leaves full of repeated keys (0.88 ms) did what 64 KB of counters do in one read.
template <int KW>
__global__ __launch_bounds__(1024)
void kmc_msd_count_kernel(const u64* __restrict__ hi0, const u64* __restrict__ lo0, const u64* __restrict__ hi1, const u64* __restrict__ lo1,
                          const MsdTerm* __restrict__ term, const u32* __restrict__ clist, u32 n_list,
                          u64* __restrict__ o_hi, u64* __restrict__ o_lo, u64* __restrict__ o_cnt, u32* __restrict__ nd, MsdCtl* __restrict__ ctl) {
    __shared__ u32 cnt[1u << KMC_MSD_CNT_BITS];
    __shared__ u32 wsum[16];
    const u32 tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (blockIdx.x >= n_list) return;
    const u32 t = clist[blockIdx.x];
    const MsdTerm T = term[t];
    const u32 bits = T.kind >> 8, nbin = 1u << bits, n = T.len;
    const u64 mask = (u64)nbin - 1ull;
    const u64* const klo = (T.parity ? lo1 : lo0) + T.begin;
    for (u32 i = tid; i < nbin; i += 1024) cnt[i] = 0;
    __syncthreads();
    u32 i = tid;
    for (; i + 3 * 1024 < n; i += 4 * 1024) {   // four loads in flight per thread
        const u64 a = klo[i], b = klo[i + 1024], c = klo[i + 2048], d = klo[i + 3072];
        atomicAdd(&cnt[(u32)(a & mask)], 1u); atomicAdd(&cnt[(u32)(b & mask)], 1u);
        atomicAdd(&cnt[(u32)(c & mask)], 1u); atomicAdd(&cnt[(u32)(d & mask)], 1u);
    }
    for (; i < n; i += 1024) atomicAdd(&cnt[(u32)(klo[i] & mask)], 1u);
    const u64 plo = klo[0] & ~mask;   // what all keys share
    const u64 phi = KW == 2 ? ((T.parity ? hi1 : hi0)[T.begin]) : 0ull;
    __syncthreads();
    // thread t owns the bins [t * per, (t + 1) * per): its non-zero ones, in order, behind those of the threads before it
    const u32 per = nbin >= 1024 ? nbin / 1024 : 1u;
    const u32 b0 = tid * per;
    u32 mine = 0;
    if (b0 < nbin) for (u32 e = 0; e < per; ++e) mine += cnt[b0 + e] ? 1u : 0u;
    u32 inc = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const u32 v = __shfl_up(inc, o); if ((int)lane >= o) inc += v; }
    if (lane == 63) wsum[wv] = inc;
    __syncthreads();
    u32 base = 0, total = 0;
    for (u32 w = 0; w < 16; ++w) { if (w < wv) base += wsum[w]; total += wsum[w]; }
    u32 r = T.begin + base + inc - mine;
    if (b0 < nbin) for (u32 e = 0; e < per; ++e) {
        const u32 c = cnt[b0 + e];
        if (c) { o_lo[r] = plo | (u64)(b0 + e); if (KW == 2) o_hi[r] = phi; o_cnt[r] = c; ++r; }
    }
    if (tid == 0) {
        nd[t] = total;
        if (total != n) atomicAdd(&ctl->n_dups[t & 63u], n - total);
    }
}


// ---- leaves ------------------------------------------------------------------------------------
// One workgroup per terminal (in position order).  Result: the terminal's (key, count) pairs, sorted, written
// straight into the run o_* at the terminal's OWN position: pair r of terminal t goes to T.begin + r.  The terminals
// tile the valid positions, so when no key occurs twice -- reads that are really all distinct: the input the sort
// path exists for -- the run is dense as it stands and nothing else touches it (round 2 staged every pair in the dead
// key buffer and a gather kernel made the run dense: 5.1 ms of 28 per GB of all-distinct reads for a pure copy).  Every
// terminal with duplicates adds their number to ctl->n_dups; if that is not zero the host scans nd[] (nd[t] = the
// terminal's pair count, always written) and kmc_msd_gather_kernel moves the pairs together -- work proportional to the
// DISTINCT keys, little when keys repeat a lot (LR mode, table merges), a full copy only in between.
// (Measured and dropped: a decoupled look-back over one status word per terminal, so that every leaf writes to its
// final dense place at once.  Exact, but 11.7 instead of 7.9 ms per GB: a leaf finishes together with the ~1,800 leaves
// in flight around it, so the nearest inclusive prefix is a thousand terminals back and every workgroup ends up
// waiting for the slowest of its neighbours -- profiles/r03_sort_leaf_variants.txt.)
template <int KW, bool WEIGHTS, int CAPV, int SCRV> struct MsdLeafLds {
    static constexpr int CAP = CAPV;   // leaf capacity (KMC_MSD_LEAF1 / LEAF2 / LEAF2W; two-word sorts also run with 1024)
    // ONE image of the leaf (the keys come in through registers: with a second image a one-word leaf took
    // 36 KB and four leaves fit a CU; now five do).  A large sub-bucket is split through a small per-wave
    // scratch; what does not fit there goes through the terminal's own span of the OTHER key buffer in global
    // memory, dead until the pairs are staged there (agent-scope fences: slow, rare).
    u64 b_lo[CAP];
    u64 b_hi[KW == 2 ? CAP : 1];
    u64 b_w[WEIGHTS ? CAP : 1];   // weights (counts) of the keys
    // a wave's scratch for a large sub-bucket (8 KB per workgroup in all; larger sub-buckets use the global scratch)
    // SCRV keys of scratch per wave.  Random keys almost never need it (sub-buckets of more than 32 keys), repeated and
    // clustered keys do all the time; it decides how many leaves fit a CU: one-word keys 8 -> seven leaves (8.4 -> 6.6 ms
    // on 831 M random keys), 256 -> five (LR leaves 0.41 ms; through the global scratch 2.2 ms); two-word leaves of
    // 2048 keys 16 -> four.  The host picks by what the ctx's last sort looked like (kmc_api.hip: msd_dup_heavy).
    static constexpr int SCR = SCRV;
    u64 s_lo[4][SCR];
    u64 s_hi[KW == 2 ? 4 : 1][KW == 2 ? SCR : 1];
    u64 s_w[WEIGHTS ? 4 : 1][WEIGHTS ? SCR : 1];
    u32 cnt[KMC_MSD_LEAF_NSB], off[KMC_MSD_LEAF_NSB + 1];   // (later: the run heads' positions, 16 bits each)
    u32 big[CAPV / (KMC_MSD_THREAD_SORT + 1) + 2];        // sub-buckets too large for one thread (more than THREAD_SORT keys each)
    u32 woff[4][129];                                     // a wave's own offsets when it splits such a sub-bucket again (its counters: cnt)
    u32 nbig;
    u32 wsum[4];
    u32 bad;                  // a sub-bucket was too large for the in-wave rank sort
    u32 n_out;
    u64 sx[4][2], sy[4][2];
};

// (one-word keys without weights: seven workgroups per CU fit by LDS; the register allocation is held to that -- 72 VGPRs --
// so that the memory phases of some leaves overlap the LDS phases of others)
template <int KW, bool WEIGHTS, int CAPV, int SCRV>
__global__ __launch_bounds__(KMC_MSD_THREADS, (KW == 1 && !WEIGHTS) ? 7 : 1)
void kmc_msd_leaf_kernel(const u64* __restrict__ hi0, const u64* __restrict__ lo0, const u64* __restrict__ w0,
                         const u64* __restrict__ hi1, const u64* __restrict__ lo1, const u64* __restrict__ w1,
                         const MsdTerm* __restrict__ term, u32 n_term, int kb,
                         u64* __restrict__ s_hi0, u64* __restrict__ s_lo0, u64* __restrict__ s_w0,
                         u64* __restrict__ s_hi1, u64* __restrict__ s_lo1, u64* __restrict__ s_w1,
                         u64* __restrict__ o_hi, u64* __restrict__ o_lo, u64* __restrict__ o_cnt,
                         u32* __restrict__ nd, MsdCtl* __restrict__ ctl) {
    extern __shared__ __align__(16) unsigned char msd_smem[];
    MsdLeafLds<KW, WEIGHTS, CAPV, SCRV>& L = *reinterpret_cast<MsdLeafLds<KW, WEIGHTS, CAPV, SCRV>*>(msd_smem);
    const u32 t = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const MsdTerm T = term[t];
    const u64* khi = T.parity ? hi1 : hi0;
    const u64* klo = T.parity ? lo1 : lo0;
    const u64* kw = T.parity ? w1 : w0;
    // global scratch of a large sub-bucket: the OTHER buffers at the terminal's own positions (dead there: an ancestor's keys)
    u64* const t_lo = T.parity ? s_lo0 : s_lo1;
    u64* const t_hi = T.parity ? s_hi0 : s_hi1;
    u64* const t_w = T.parity ? s_w0 : s_w1;
    const u32 n = T.len;
#ifdef KMC_LEAF_STAMPS
    unsigned long long t_prev_ = __builtin_amdgcn_s_memtime();
#endif
    // a terminal's n_mine pairs go to its own positions of the run; what it holds fewer pairs than keys is recorded
    auto account = [&](u32 n_mine) {   // (one lane)
        nd[t] = n_mine;
        if (n_mine != n) atomicAdd(&ctl->n_dups[t & 63u], n - n_mine);
    };
    // a terminal that is ONE pair (all keys equal): key, count
    auto single_pair = [&](u64 hi, u64 lo, u64 cnt, u32 n_mine) {
        if (tid == 0) {
            account(n_mine);
            if (n_mine) {
                o_lo[T.begin] = lo; if (KW == 2) o_hi[T.begin] = hi; o_cnt[T.begin] = cnt;
                if (WEIGHTS) atomicAdd(&ctl->w_total, (unsigned long long)cnt);
            }
        }
    };
    if ((T.kind & 0xFFu) == 2u) return;   // (kmc_msd_count_kernel's)
    if (T.kind == 1) {  // all keys equal: one pair
        u64 s = n;
        if (WEIGHTS) {
            s = 0;
            for (u32 i = tid; i < n; i += KMC_MSD_THREADS) s += kw[T.begin + i];
            s = wave_sum_u64(s);
            if (lane == 0) L.sx[wv][0] = s;
            __syncthreads();
            s = L.sx[0][0] + L.sx[1][0] + L.sx[2][0] + L.sx[3][0];
        }
        single_pair(KW == 2 ? khi[T.begin] : 0ull, klo[T.begin], s, 1u);
        return;
    }
    // 1. load (n <= CAP) and find the smallest and the largest key.  Sub-bucket of a key = (key - min) >> sh,
    //    sh chosen so that max lands in bucket 255 at most: order-preserving, and even over the leaf's
    //    actual key range.  (A leaf may hold several children of its parent; the first version took the 8
    //    bits below the highest DIFFERING bit -- for a leaf that straddles a power of two, e.g. children
    //    0111111111 and 1000000000, that put all keys into two sub-buckets: 65 of the sort's 97 ms.)
    constexpr int PER = MsdLeafLds<KW, WEIGHTS, CAPV, SCRV>::CAP / KMC_MSD_THREADS;  // keys per thread, in registers
    u64 rlo[PER], rhi[KW == 2 ? PER : 1], rw[WEIGHTS ? PER : 1];
    u64 mnh = ~0ull, mnl = ~0ull, mxh = 0, mxl = 0;
    // The keys come in as ALIGNED PAIRS, 16 bytes per lane and load instruction (a wave-load = 1 KiB contiguous; 8-byte
    // loads left this kernel at 2.5 TB/s).  Register slot s of the workgroup (thread t: slots 2 (t + 256 e) + {0, 1})
    // holds key s + off of the terminal, off = T.begin & 1: with an odd begin the pairs start at the terminal's SECOND
    // key and its first key goes to slot n - 1, which is free then.  Which key sits in which slot does not matter --
    // the LDS pass below orders them.  (Half a pair may lie outside the terminal -- a neighbour's key, or up to 8
    // bytes past the last key of the array, inside the allocation's slack: loaded, never used.)
    static_assert(PER % 2 == 0, "keys per thread must be even");
    typedef unsigned long long u64x2_t __attribute__((ext_vector_type(2)));
    const u32 off = T.begin & 1u;
    bool rv[PER];   // slot holds a key
#pragma unroll
    for (int e = 0; e < PER / 2; ++e) {
        const u32 s0 = 2u * (tid + KMC_MSD_THREADS * e);   // first slot of the pair; its key is T.begin + off + s0 (an even index)
        rlo[2 * e] = rlo[2 * e + 1] = 0;
        if (KW == 2) rhi[2 * e] = rhi[2 * e + 1] = 0;
        if (WEIGHTS) rw[2 * e] = rw[2 * e + 1] = 0;
        rv[2 * e] = s0 + off < n;
        rv[2 * e + 1] = s0 + 1 + off < n;
        if (rv[2 * e]) {   // (the pair's first key is the terminal's: the pair is loaded)
            const size_t g = (size_t)T.begin + off + s0;
            const u64x2_t pl = *reinterpret_cast<const u64x2_t*>(klo + g);
            rlo[2 * e] = pl.x; rlo[2 * e + 1] = pl.y;
            if (KW == 2) { const u64x2_t ph = *reinterpret_cast<const u64x2_t*>(khi + g); rhi[2 * e] = ph.x; rhi[2 * e + 1] = ph.y; }
            if (WEIGHTS) { const u64x2_t pw = *reinterpret_cast<const u64x2_t*>(kw + g); rw[2 * e] = pw.x; rw[2 * e + 1] = pw.y; }
        }
        if (off) {   // odd begin: the terminal's first key into slot n - 1
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                if (s0 + h == n - 1) {
                    rlo[2 * e + h] = klo[T.begin];
                    if (KW == 2) rhi[2 * e + h] = khi[T.begin];
                    if (WEIGHTS) rw[2 * e + h] = kw[T.begin];
                    rv[2 * e + h] = true;
                }
            }
        }
    }
#pragma unroll
    for (int e = 0; e < PER; ++e) {
        if (rv[e]) {
            const u64 lo = rlo[e], hi = KW == 2 ? rhi[e] : 0ull;
            if (key_less(hi, lo, mnh, mnl)) { mnh = hi; mnl = lo; }
            if (key_less(mxh, mxl, hi, lo)) { mxh = hi; mxl = lo; }
        } else {
            rlo[e] = 0;
            if (KW == 2) rhi[e] = 0;
            if (WEIGHTS) rw[e] = 0;
        }
    }
    LEAF_STAMP(0);   // keys loaded (HBM latency)
    for (u32 i = tid; i < KMC_MSD_LEAF_NSB; i += KMC_MSD_THREADS) L.cnt[i] = 0;
    if (tid == 0) { L.bad = 0; L.n_out = 0; L.nbig = 0; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const u64 oh = __shfl_xor(mnh, o), ol = __shfl_xor(mnl, o);
        if (key_less(oh, ol, mnh, mnl)) { mnh = oh; mnl = ol; }
        const u64 ph = __shfl_xor(mxh, o), pl = __shfl_xor(mxl, o);
        if (key_less(mxh, mxl, ph, pl)) { mxh = ph; mxl = pl; }
    }
    if (lane == 0) { L.sx[wv][0] = mnh; L.sx[wv][1] = mnl; L.sy[wv][0] = mxh; L.sy[wv][1] = mxl; }
    __syncthreads();
    mnh = L.sx[0][0]; mnl = L.sx[0][1]; mxh = L.sy[0][0]; mxl = L.sy[0][1];
    for (int w = 1; w < 4; ++w) {
        if (key_less(L.sx[w][0], L.sx[w][1], mnh, mnl)) { mnh = L.sx[w][0]; mnl = L.sx[w][1]; }
        if (key_less(mxh, mxl, L.sy[w][0], L.sy[w][1])) { mxh = L.sy[w][0]; mxl = L.sy[w][1]; }
    }
    const u64 fh = mnh, fl = mnl;
    // range = max - min (128 bits)
    const u64 rl = mxl - mnl, rh = mxh - mnh - (mxl < mnl ? 1ull : 0ull);
    const int top = rh ? 127 - __clzll((long long)rh) : (rl ? 63 - __clzll((long long)rl) : -1);
    if (top < 0) {  // every key of the leaf is the same: one pair
        u64 sw = n;
        if (WEIGHTS) {
            __syncthreads();
            sw = 0;
#pragma unroll
            for (int e = 0; e < PER; ++e) if (rv[e]) sw += rw[e];
            sw = wave_sum_u64(sw);
            if (lane == 0) L.sx[wv][0] = sw;
            __syncthreads();
            sw = L.sx[0][0] + L.sx[1][0] + L.sx[2][0] + L.sx[3][0];
        }
        single_pair(fh, fl, sw, n ? 1u : 0u);
        return;
    }
    const int shift = top >= KMC_MSD_LEAF_LOGNSB ? top - (KMC_MSD_LEAF_LOGNSB - 1) : 0;
    // sub-bucket of a key: ((key - min) >> shift), at most NSB - 1
    auto bucket = [&](u64 hi, u64 lo) -> u32 {
        const u64 dl = lo - mnl;
        if (KW == 1) return (u32)(dl >> shift);
        const u64 dh = hi - mnh - (lo < mnl ? 1ull : 0ull);
        if (shift >= 64) return (u32)(dh >> (shift - 64));
        if (shift == 0) return (u32)dl;
        return (u32)((dl >> shift) | (dh << (64 - shift)));
    };
    LEAF_STAMP(1);   // min / max
    // 2. LDS pass: a -> b grouped by digit
    u32 rb[PER];  // sub-bucket of my keys
#pragma unroll
    for (int e = 0; e < PER; ++e) {
        rb[e] = bucket(KW == 2 ? rhi[e] : 0ull, rlo[e]) & (KMC_MSD_LEAF_NSB - 1u);
        if (rv[e]) atomicAdd(&L.cnt[rb[e]], 1u);
    }
    __syncthreads();
    {   // exclusive prefix of the sub-bucket sizes: four consecutive sub-buckets per thread
        constexpr int PT = KMC_MSD_LEAF_NSB / KMC_MSD_THREADS;
        u32 c[PT], mine = 0;
#pragma unroll
        for (int e = 0; e < PT; ++e) { c[e] = L.cnt[tid * PT + e]; mine += c[e]; }
        u32 inc = mine;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const u32 v = __shfl_up(inc, o); if ((int)lane >= o) inc += v; }
        if (lane == 63) L.wsum[wv] = inc;
        __syncthreads();
        u32 run = inc - mine;
        for (u32 w = 0; w < wv; ++w) run += L.wsum[w];
#pragma unroll
        for (int e = 0; e < PT; ++e) {
            L.off[tid * PT + e] = run;
            L.cnt[tid * PT + e] = run;  // cursor
            if (c[e] > KMC_MSD_THREAD_SORT) L.big[atomicAdd(&L.nbig, 1u)] = tid * PT + e;
            run += c[e];
        }
        if (tid == KMC_MSD_THREADS - 1) L.off[KMC_MSD_LEAF_NSB] = run;
        __syncthreads();
    }
    LEAF_STAMP(2);   // count atomics + prefix
    u32 rp[PER];  // where my keys landed in b
#pragma unroll
    for (int e = 0; e < PER; ++e) {
        rp[e] = 0;
        if (rv[e]) {
            const u32 p = atomicAdd(&L.cnt[rb[e]], 1u);
            rp[e] = p;
            L.b_lo[p] = rlo[e];
            if (KW == 2) L.b_hi[p] = rhi[e];
            if (WEIGHTS) L.b_w[p] = rw[e];
        }
    }
    __syncthreads();
    LEAF_STAMP(3);   // cursor atomics + LDS scatter
    // 3a. small sub-buckets (three or four keys on average, at most THREAD_SORT): every KEY counts the keys of its
    //     sub-bucket that sort before it -- smaller, or equal and ahead of it in b -- and that count is its place.
    //     The keys are still in their owners' registers, so the sorted image is written over b in place.  (Round 2
    //     let every THREAD insertion-sort its sub-buckets: data-dependent nested loops, 2.06 G scalar wave-instructions
    //     of exec-mask bookkeeping next to 2.47 G vector ones per GB of reads -- here the only divergence is the trip count.)
#ifdef KMC_EXP_INSERTION
    for (int e = 0; e < PER; ++e) rp[e] = ~0u;
    for (u32 d = tid * (KMC_MSD_LEAF_NSB / KMC_MSD_THREADS); d < (tid + 1) * (KMC_MSD_LEAF_NSB / KMC_MSD_THREADS); ++d) {
        const u32 o = L.off[d], m = L.off[d + 1] - o;
        if (m > 1 && m <= KMC_MSD_THREAD_SORT) {
            for (u32 i = 1; i < m; ++i) {
                const u64 lo = L.b_lo[o + i], hi = KW == 2 ? L.b_hi[o + i] : 0ull;
                u64 w = 0;
                if (WEIGHTS) w = L.b_w[o + i];
                u32 j = i;
                while (j > 0) {
                    const u64 pl = L.b_lo[o + j - 1], ph = KW == 2 ? L.b_hi[o + j - 1] : 0ull;
                    if (!key_less(hi, lo, ph, pl)) break;
                    L.b_lo[o + j] = pl;
                    if (KW == 2) L.b_hi[o + j] = ph;
                    if (WEIGHTS) L.b_w[o + j] = L.b_w[o + j - 1];
                    --j;
                }
                if (j != i) {
                    L.b_lo[o + j] = lo;
                    if (KW == 2) L.b_hi[o + j] = hi;
                    if (WEIGHTS) L.b_w[o + j] = w;
                }
            }
        }
    }
#else
#pragma unroll
    for (int e = 0; e < PER; ++e) {
        if (rv[e]) {
            const u32 o = L.off[rb[e]], m = L.off[rb[e] + 1] - o;
            if (m > 1 && m <= KMC_MSD_THREAD_SORT) {
                const u64 mlo = rlo[e], mhi = KW == 2 ? rhi[e] : 0ull;
                const u32 mp = rp[e];
                u32 rank = 0;
                for (u32 j = 0; j < m; ++j) {
                    const u64 ql = L.b_lo[o + j], qh = KW == 2 ? L.b_hi[o + j] : 0ull;
                    const bool before = key_less(qh, ql, mhi, mlo) || (qh == mhi && ql == mlo && o + j < mp);
                    rank += before ? 1u : 0u;
                }
                rp[e] = o + rank;
            } else {
                rp[e] = ~0u;   // alone, or one of a large sub-bucket (a wave sorts those below): stays where it is
            }
        } else rp[e] = ~0u;
    }
#endif
    __syncthreads();
#pragma unroll
    for (int e = 0; e < PER; ++e) {
        if (rp[e] != ~0u) {
            L.b_lo[rp[e]] = rlo[e];
            if (KW == 2) L.b_hi[rp[e]] = rhi[e];
            if (WEIGHTS) L.b_w[rp[e]] = rw[e];
        }
    }
    __syncthreads();
    LEAF_STAMP(4);   // rank count + rewrite
    // 3b. larger sub-buckets, one wave each.  Keys of real inputs cluster (families of k-mers that share
    //     all but their last few bases sit in ONE sub-bucket while the leaf's range is set by the distance
    //     between families), so a large sub-bucket is first split again by ITS OWN key range into 128
    //     parts (wave-private counters), which the lanes insertion-sort; only what is still large after
    //     that -- and not a single repeated key -- is rank-sorted all-pairs (64 keys at a time through
    //     v_readlane).  Result back in b.
    // all-pairs rank sort of m keys at s_*[0..m) into d_*[0..m)
    auto allpairs = [&](const u64* s_lo, const u64* s_hi, const u64* s_w, u64* d_lo, u64* d_hi, u64* d_w, u32 m) {
        for (u32 c = 0; c < m; c += 64) {          // the chunk whose keys get their positions
            const u32 i = c + lane;
            u64 lo = 0, hi = 0, w = 0;
            const bool have = i < m;
            if (have) { lo = s_lo[i]; if (KW == 2) hi = s_hi[i]; if (WEIGHTS) w = s_w[i]; }
            u32 rk = 0;
            for (u32 e = 0; e < m; e += 64) {      // against chunk e
                const u32 j = e + lane;
                u64 ql = ~0ull, qh = ~0ull;
                if (j < m) { ql = s_lo[j]; qh = KW == 2 ? s_hi[j] : 0ull; }
                const u32 mm = min(64u, m - e);
                for (u32 x = 0; x < mm; ++x) {
                    const u64 ol = ((u64)(u32)__builtin_amdgcn_readlane((int)(u32)(ql >> 32), (int)x) << 32) | (u32)__builtin_amdgcn_readlane((int)(u32)ql, (int)x);
                    u64 oh = 0;
                    if (KW == 2) oh = ((u64)(u32)__builtin_amdgcn_readlane((int)(u32)(qh >> 32), (int)x) << 32) | (u32)__builtin_amdgcn_readlane((int)(u32)qh, (int)x);
                    const bool less = key_less(oh, ol, hi, lo);
                    const bool eq = oh == hi && ol == lo;
                    rk += (less || (eq && (e + x) < i)) ? 1u : 0u;
                }
            }
            if (have) { d_lo[rk] = lo; if (KW == 2) d_hi[rk] = hi; if (WEIGHTS) d_w[rk] = w; }
        }
    };
    // scratch: this terminal's span of the other key buffer and of t_cnt (nothing lives there until the pairs
    // are staged at the end)
    u64* const g_lo = t_lo + T.begin;
    u64* const g_hi = KW == 2 ? t_hi + T.begin : nullptr;
    u64* const g_w = WEIGHTS ? t_w + T.begin : nullptr;
    const u32 nbig = L.nbig;
    for (u32 bi = wv; bi < nbig; bi += 4) {
        const u32 d = L.big[bi];
        const u32 o = L.off[d], m = L.off[d + 1] - o;
        // smallest / largest key of the sub-bucket
        u64 mnh = ~0ull, mnl = ~0ull, mxh = 0, mxl = 0;
        for (u32 i = lane; i < m; i += 64) {
            const u64 lo = L.b_lo[o + i], hi = KW == 2 ? L.b_hi[o + i] : 0ull;
            if (key_less(hi, lo, mnh, mnl)) { mnh = hi; mnl = lo; }
            if (key_less(mxh, mxl, hi, lo)) { mxh = hi; mxl = lo; }
        }
#pragma unroll
        for (int s2 = 32; s2 > 0; s2 >>= 1) {
            const u64 oh = __shfl_xor(mnh, s2), ol = __shfl_xor(mnl, s2);
            if (key_less(oh, ol, mnh, mnl)) { mnh = oh; mnl = ol; }
            const u64 ph = __shfl_xor(mxh, s2), pl = __shfl_xor(mxl, s2);
            if (key_less(mxh, mxl, ph, pl)) { mxh = ph; mxl = pl; }
        }
        if (mnh == mxh && mnl == mxl) continue;  // ONE key many times: sorted as it stands
        const u64 rl2 = mxl - mnl, rh2 = mxh - mnh - (mxl < mnl ? 1ull : 0ull);
        const int top2 = rh2 ? 127 - __clzll((long long)rh2) : 63 - __clzll((long long)rl2);
        const int sh2 = top2 >= 7 ? top2 - 6 : 0;
        auto bkt2 = [&](u64 hi, u64 lo) -> u32 {
            const u64 dl = lo - mnl;
            if (KW == 1) return (u32)(dl >> sh2) & 127u;
            const u64 dh = hi - mnh - (lo < mnl ? 1ull : 0ull);
            if (sh2 >= 64) return (u32)(dh >> (sh2 - 64)) & 127u;
            if (sh2 == 0) return (u32)dl & 127u;
            return (u32)((dl >> sh2) | (dh << (64 - sh2))) & 127u;
        };
        // scratch of this sub-bucket: the wave's LDS scratch (indexed 0 .. m) or global memory (indexed like b: o .. o + m);
        // xo is what turns a b index into a scratch index
        constexpr u32 SCR = (u32)MsdLeafLds<KW, WEIGHTS, CAPV, SCRV>::SCR;
        const bool use_g = m > SCR;
        const u32 xo = use_g ? 0u : o;
        u64* const x_lo = use_g ? g_lo : &L.s_lo[wv][0];
        u64* const x_hi = KW == 2 ? (use_g ? g_hi : &L.s_hi[wv][0]) : nullptr;
        u64* const x_w = WEIGHTS ? (use_g ? g_w : &L.s_w[wv][0]) : nullptr;
        auto xsync = [&]() {
            if (use_g) {   // (global memory written and read by different lanes of the wave)
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            } else {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
        };
        u32* const wc = &L.cnt[wv * 128];   // (the leaf's own cursors are dead by now)
        u32* const wo = L.woff[wv];
#pragma unroll
        for (int e = 0; e < 2; ++e) wc[lane * 2 + e] = 0;
        xsync();
        for (u32 i = lane; i < m; i += 64) atomicAdd(&wc[bkt2(KW == 2 ? L.b_hi[o + i] : 0ull, L.b_lo[o + i])], 1u);
        xsync();
        {
            u32 c4[2], mine = 0;
#pragma unroll
            for (int e = 0; e < 2; ++e) { c4[e] = wc[lane * 2 + e]; mine += c4[e]; }
            u32 inc = mine;
#pragma unroll
            for (int s2 = 1; s2 < 64; s2 <<= 1) { const u32 v = __shfl_up(inc, s2); if ((int)lane >= s2) inc += v; }
            u32 run = inc - mine;
#pragma unroll
            for (int e = 0; e < 2; ++e) { wo[lane * 2 + e] = run; wc[lane * 2 + e] = run; run += c4[e]; }
            if (lane == 63) wo[128] = run;
        }
        xsync();
        for (u32 i = lane; i < m; i += 64) {
            const u64 lo = L.b_lo[o + i], hi = KW == 2 ? L.b_hi[o + i] : 0ull;
            const u32 pp = atomicAdd(&wc[bkt2(hi, lo)], 1u);
            x_lo[(o + pp) - xo] = lo;
            if (KW == 2) x_hi[(o + pp) - xo] = hi;
            if (WEIGHTS) x_w[(o + pp) - xo] = L.b_w[o + i];
        }
        xsync();
        // every lane: its two parts, insertion sort in a; parts that are still large are left for the wave
        u32 big_mask = 0;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const u32 oo = o + wo[lane * 2 + e], mm = wo[lane * 2 + e + 1] - wo[lane * 2 + e];
            if (mm > KMC_MSD_THREAD_SORT) { big_mask |= 1u << e; continue; }
            for (u32 i = 1; i < mm; ++i) {
                const u64 lo = x_lo[(oo + i) - xo], hi = KW == 2 ? x_hi[(oo + i) - xo] : 0ull;
                u64 w = 0;
                if (WEIGHTS) w = x_w[(oo + i) - xo];
                u32 j = i;
                while (j > 0) {
                    const u64 pl = x_lo[(oo + j - 1) - xo], ph = KW == 2 ? x_hi[(oo + j - 1) - xo] : 0ull;
                    if (!key_less(hi, lo, ph, pl)) break;
                    x_lo[(oo + j) - xo] = pl;
                    if (KW == 2) x_hi[(oo + j) - xo] = ph;
                    if (WEIGHTS) x_w[(oo + j) - xo] = x_w[(oo + j - 1) - xo];
                    --j;
                }
                if (j != i) { x_lo[(oo + j) - xo] = lo; if (KW == 2) x_hi[(oo + j) - xo] = hi; if (WEIGHTS) x_w[(oo + j) - xo] = w; }
            }
        }
        xsync();
        // parts that are still large: one repeated key (nothing to do), or all-pairs a -> b -> a
        unsigned long long todo;
        while ((todo = __builtin_amdgcn_ballot_w64(big_mask != 0)) != 0) {
            const int src_lane = (int)__builtin_ctzll(todo);
            const u32 bm = (u32)__builtin_amdgcn_readlane((int)big_mask, src_lane);
            const int e = __ffs((int)bm) - 1;
            const u32 oo = o + wo[src_lane * 2 + e];
            const u32 mm = wo[src_lane * 2 + e + 1] - wo[src_lane * 2 + e];
            if ((int)lane == src_lane) big_mask &= ~(1u << e);
            const u64 plo = x_lo[(oo) - xo], phi = KW == 2 ? x_hi[(oo) - xo] : 0ull;
            bool diff = false;
            for (u32 i = lane; i < mm; i += 64) diff |= x_lo[(oo + i) - xo] != plo || (KW == 2 && x_hi[(oo + i) - xo] != phi);
            if (__builtin_amdgcn_ballot_w64(diff) == 0) continue;
            if (mm > 1024) { if (lane == 0) L.bad = 1; continue; }
            allpairs(x_lo + (oo - xo), KW == 2 ? x_hi + (oo - xo) : nullptr, WEIGHTS ? x_w + (oo - xo) : nullptr,
                     &L.b_lo[oo], KW == 2 ? &L.b_hi[oo] : nullptr, WEIGHTS ? &L.b_w[oo] : nullptr, mm);
            xsync();
            for (u32 i = lane; i < mm; i += 64) { x_lo[(oo + i) - xo] = L.b_lo[oo + i]; if (KW == 2) x_hi[(oo + i) - xo] = L.b_hi[oo + i]; if (WEIGHTS) x_w[(oo + i) - xo] = L.b_w[oo + i]; }
            xsync();
        }
        for (u32 i = lane; i < m; i += 64) {       // back into b (this wave's own sub-bucket)
            L.b_lo[o + i] = x_lo[(o + i) - xo];
            if (KW == 2) L.b_hi[o + i] = x_hi[(o + i) - xo];
            if (WEIGHTS) L.b_w[o + i] = x_w[(o + i) - xo];
        }
    }
    __syncthreads();
    if (L.bad) {
        // a sub-bucket of more than 1024 keys that differ (heavily repeated keys next to others): sort
        // the whole leaf with a bitonic network in LDS (rare), in place, padded to a power of two.
        u32 P = 1;
        while (P < n) P <<= 1;
        for (u32 i = n + tid; i < P; i += KMC_MSD_THREADS) {
            L.b_lo[i] = ~0ull;
            if (KW == 2) L.b_hi[i] = ~0ull;
            if (WEIGHTS) L.b_w[i] = 0ull;
        }
        __syncthreads();
        for (u32 kk = 2; kk <= P; kk <<= 1) {
            for (u32 jj = kk >> 1; jj > 0; jj >>= 1) {
                for (u32 i = tid; i < P; i += KMC_MSD_THREADS) {
                    const u32 ix = i ^ jj;
                    if (ix > i) {
                        const u64 al = L.b_lo[i], bl = L.b_lo[ix];
                        const u64 ah = KW == 2 ? L.b_hi[i] : 0ull, bh = KW == 2 ? L.b_hi[ix] : 0ull;
                        const bool up = (i & kk) == 0;
                        const bool sw = up ? key_less(bh, bl, ah, al) : key_less(ah, al, bh, bl);
                        if (sw) {
                            L.b_lo[i] = bl; L.b_lo[ix] = al;
                            if (KW == 2) { L.b_hi[i] = bh; L.b_hi[ix] = ah; }
                            if (WEIGHTS) { const u64 wa = L.b_w[i]; L.b_w[i] = L.b_w[ix]; L.b_w[ix] = wa; }
                        }
                    }
                }
                __syncthreads();
            }
        }
    }
    LEAF_STAMP(5);   // large sub-buckets
    // 4. run-length over the sorted image b[0..n).  Phase 1: the position of every run head, compacted
    //    (16 bits each, in the space of the sub-bucket counters and offsets, dead by now); phase 2: one thread per run -- its length is the distance
    //    to the next head (the first version let the head's thread walk its run: one thread, thousands of
    //    dependent LDS reads for a key with thousands of copies, everybody else waiting at the barrier).
    static_assert(sizeof(L.cnt) + sizeof(L.off) >= MsdLeafLds<KW, WEIGHTS, CAPV, SCRV>::CAP * sizeof(unsigned short), "run-head list does not fit");
    unsigned short* const hidx = reinterpret_cast<unsigned short*>(L.cnt);
    // Element i = tid + 256 e of the sorted image is a run head when it differs from element i - 1 (strided: conflict-free
    // LDS reads).  Heads are numbered in element order -- slab e before slab e + 1, inside a slab by thread -- with ONE
    // ballot per slab and wave and one small table of (slab, wave) totals: two barriers for the whole leaf (the first
    // version scanned slabs of 1024 elements one after the other, four barriers each).
    constexpr int NSLAB = MsdLeafLds<KW, WEIGHTS, CAPV, SCRV>::CAP / KMC_MSD_THREADS;
    u32 hmask = 0;          // bit e: my element of slab e is a run head
    u32 hpre[NSLAB];        // heads of lower lanes of my wave in slab e
    u32* const stot = L.big;   // [NSLAB][4] heads per (slab, wave) (the list of large sub-buckets is dead by now)
    static_assert(sizeof(L.big) >= (NSLAB * 4 + 1) * sizeof(u32), "head totals do not fit");
#pragma unroll
    for (int e = 0; e < NSLAB; ++e) {
        const u32 i = tid + KMC_MSD_THREADS * e;
        bool h = false;
        if (i < n) h = i == 0 || L.b_lo[i] != L.b_lo[i - 1] || (KW == 2 && L.b_hi[i] != L.b_hi[i - 1]);
        const unsigned long long m = __builtin_amdgcn_ballot_w64(h);
        hmask |= h ? (1u << e) : 0u;
        hpre[e] = __builtin_amdgcn_mbcnt_hi((u32)(m >> 32), __builtin_amdgcn_mbcnt_lo((u32)m, 0u));
        if (lane == 0) stot[e * 4 + wv] = (u32)__popcll(m);
    }
    __syncthreads();   // (every element has been compared: the counters under hidx may be overwritten)
    u32 n_heads = 0;
    {
        // exclusive prefix of the NSLAB x 4 totals by one wave (the other waves wait at the barrier), then every thread
        // picks up its NSLAB bases
        static_assert(NSLAB * 4 <= 64, "one wave scans the head totals");
        if (wv == 0) {
            const u32 v = lane < NSLAB * 4 ? stot[lane] : 0u;
            u32 inc = v;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const u32 x = __shfl_up(inc, o); if ((int)lane >= o) inc += x; }
            if (lane < NSLAB * 4) stot[lane] = inc - v;
            if (lane == NSLAB * 4 - 1) stot[NSLAB * 4] = inc;
        }
        __syncthreads();
        n_heads = stot[NSLAB * 4];
#pragma unroll
        for (int e = 0; e < NSLAB; ++e) if (hmask & (1u << e)) hidx[stot[e * 4 + wv] + hpre[e]] = (unsigned short)(tid + KMC_MSD_THREADS * e);
    }
    __syncthreads();
    LEAF_STAMP(6);   // run heads
    const u32 n_out = n_heads;
    // 5. out: straight into the run, at the terminal's own positions, two pairs per lane and store where they are aligned
    //    (run index T.begin + r even): 16-byte stores, 1 KiB per wave instruction
    if (tid == 0) account(n_out);
    const u32 base = T.begin;
    u64 wtot = 0;
    auto pair_of = [&](u32 r, u64& khi_o, u64& klo_o, u64& c_o) {
        const u32 i = hidx[r], iend = r + 1 < n_out ? hidx[r + 1] : n;
        u64 sum = iend - i;
        if (WEIGHTS) { sum = 0; for (u32 j = i; j < iend; ++j) sum += L.b_w[j]; }
        klo_o = L.b_lo[i];
        khi_o = KW == 2 ? L.b_hi[i] : 0ull;
        c_o = sum;
        wtot += sum;
    };
    const u32 lead = (base & 1u) && n_out ? 1u : 0u;   // an odd first position: that pair alone
    if (tid == 0 && lead) {
        u64 a, b, cn;
        pair_of(0, a, b, cn);
        o_lo[base] = b; if (KW == 2) o_hi[base] = a; o_cnt[base] = cn;
    }
    for (u32 r = lead + 2u * tid; r < n_out; r += 2u * KMC_MSD_THREADS) {
        u64 h0, l0, c0, h1 = 0, l1 = 0, c1 = 0;
        pair_of(r, h0, l0, c0);
        if (r + 1 < n_out) {
            pair_of(r + 1, h1, l1, c1);
            const size_t g = (size_t)base + r;   // even
            *reinterpret_cast<u64x2_t*>(o_lo + g) = u64x2_t{l0, l1};
            if (KW == 2) *reinterpret_cast<u64x2_t*>(o_hi + g) = u64x2_t{h0, h1};
            *reinterpret_cast<u64x2_t*>(o_cnt + g) = u64x2_t{c0, c1};
        } else {
            o_lo[base + r] = l0; if (KW == 2) o_hi[base + r] = h0; o_cnt[base + r] = c0;
        }
    }
    if (WEIGHTS) {
        wtot = wave_sum_u64(wtot);
        if (lane == 0 && wtot) atomicAdd(&ctl->w_total, (unsigned long long)wtot);
    }
    LEAF_STAMP(7);   // pairs out (issue)
}

// dense run: terminal t's nd[t] pairs move from its own positions of the sparse run (s_*) to base[t] of the dense one
template <int KW>
__global__ void kmc_msd_gather_kernel(const MsdTerm* __restrict__ term, u32 n_term, const u32* __restrict__ nd, const u32* __restrict__ base,
                                      const u64* __restrict__ s_hi, const u64* __restrict__ s_lo, const u64* __restrict__ s_cnt,
                                      u64* __restrict__ o_hi, u64* __restrict__ o_lo, u64* __restrict__ o_cnt) {
    // one wave per terminal
    const u32 wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63, n_waves = (gridDim.x * blockDim.x) >> 6;
    for (u32 t = wave; t < n_term; t += n_waves) {
        const u32 src = term[t].begin, dst = base[t], m = nd[t];
        for (u32 i = lane; i < m; i += 64) {
            o_lo[dst + i] = s_lo[src + i];
            if (KW == 2) o_hi[dst + i] = s_hi[src + i];
            o_cnt[dst + i] = s_cnt[src + i];
        }
    }
}
