// kmc_sklog.hip.h -- counting the walk kernel's LOGGED steps (kmc_walk.hip.h, SkLog): the (k+16)-mers of the steps
// that fell off the LDS memo, one record per step, in per-workgroup spans.  Grouping equal records is the reference's
// grouping step once more (k-mer-count/src/main.rs:84,87: sort, then equal lines), done here without a sort and
// without a global hash table:
//   1. kmc_sklog_hist_kernel +     a record goes to one of 1024 BINS by a hash of all its words: per span of the log an
//      kmc_sklog_partition_kernel  LDS histogram (bins are sized exactly from the totals), one returning global atomic per
//                                  non-empty bin to reserve the span's room there, then the records are stored, grouped
//                                  by bin through LDS (equal records meet in one bin);
//   2. kmc_sklog_consume_kernel    one workgroup per bin counts its records in an LDS hash table (4096 entries: a bin of
//                                  the plateau inputs holds a few dozen to a few thousand distinct records) and unfolds
//                                  every distinct (k+16)-mer ONCE: its 16 k-mers receive the count in the count table.
// Work per record: one coalesced read, one 16-/32-byte store, one coalesced read, one LDS table update -- against a
// read-modify-write of three random lines of a table in HBM per step before.  Records that find no room in a bin's LDS table
// (inputs with millions of distinct (k+16)-mers) are unfolded on the spot, k-mer by k-mer: always exact.
#pragma once
#include "kmc_walk.hip.h"

#define KMC_SKLOG_BINS 1024
#define KMC_SKLOG_TCAP 4096      // LDS table slots of a consume workgroup (80 / 112 KB: one workgroup per CU.  Tables of 2048 slots -- two
                                 // workgroups per CU -- were measured on the plateau inputs: no faster, 2.16 / 2.20 / 2.44 vs 2.18 / 2.20 / 2.42 ms at k=31,
                                 // 3.01 / 3.41 vs 3.09 / 3.44 at k=63: occupancy is not what bounds the kernel)

// the 16 k-mers of one (k+16)-mer {top, mid, lo}, each + cnt (what kmc_sk_unfold_kernel does per table entry)
template <int KW, bool CANON>
__device__ __forceinline__ void sklog_unfold_one(const GTable& g, int k, u64 top, u64 hi, u64 lo, u32 j, u64 cnt, u64 mask_hi, u64 mask_lo) {
    const u32 sh = 2 * j;  // drop the last j bases
    WCtx km;
    km.lo = (sh ? ((lo >> sh) | (hi << (64 - sh))) : lo) & mask_lo;
    km.hi = KW == 2 ? ((sh ? ((hi >> sh) | (top << (64 - sh))) : hi) & mask_hi) : 0ull;
    walk_gadd<KW, CANON>(g, km, k, cnt);
}

// Bins are sized EXACTLY: kmc_sklog_hist_kernel counts every span's records per bin (LDS histogram; the row is kept, the
// totals are added up with one global atomic per span and bin), and both later kernels turn the 1024 totals into bin
// offsets themselves (a 1024-entry scan per workgroup).  The first version gave every bin 1.25 x its even share: the plateau
// inputs log a few thousand DISTINCT records, each thousands of times, so the bins' loads follow those records' frequencies
// -- most bins overflowed and their records fell back to sixteen global atomics each.
// kmc_sklog_partition_kernel: one workgroup per span reserves the span's room in every bin with ONE returning global
// atomic per bin, then tile by tile (128 KB of records) groups the tile by bin in LDS and writes it out in that order, so
// that consecutive lanes store consecutive records of one bin -- runs of 128 bytes instead of single records.
// (Records stored straight from their lanes, 64 lanes into 64 different bins, took 19 ms per GB of reads for 0.45 GB of
// records; profiles/r03_sklog_partition_variants.txt.  The bin histogram built by the walk kernel itself while it logs --
// 4 KB of LDS counters, one-word keys only: the two-word kernel has none to spare -- saved 0.08-0.10 ms per GB at pools 50 and
// 100 and cost the walk kernel 1 % on the benchmark input, whose steps never reach that code: not kept.)
// A record is W words: {lo, mid} (k <= 47) or {lo, mid, top} (k >= 48: 24 bytes -- a fourth, empty word made the records a
// third larger on every pass over them).  Two-word records are 16-byte aligned and move as one dwordx4; three-word
// records are 8-byte aligned: three dwordx2 (a wave still covers 1536 contiguous bytes).
struct SkRec { u64 lo, mid, top; };
template <int W>
__device__ __forceinline__ SkRec sklog_load(const u64* __restrict__ p) {
    typedef unsigned long long u64x2_t __attribute__((ext_vector_type(2)));
    if (W == 2) { const u64x2_t a = *reinterpret_cast<const u64x2_t*>(p); return SkRec{a.x, a.y, 0ull}; }
    return SkRec{p[0], p[1], p[2]};
}
template <int W>
__device__ __forceinline__ void sklog_store(u64* __restrict__ p, const SkRec& r) {
    typedef unsigned long long u64x2_t __attribute__((ext_vector_type(2)));
    if (W == 2) { *reinterpret_cast<u64x2_t*>(p) = u64x2_t{r.lo, r.mid}; return; }
    p[0] = r.lo; p[1] = r.mid; p[2] = r.top;
}
template <int W>
__device__ __forceinline__ u32 sklog_bin_of(unsigned long long ax, unsigned long long ay, unsigned long long bx) {
    const u64 h = W == 3 ? kmc_hash_key<3>(bx, ax, ay) : kmc_hash_key<2>(ay, ax);
    return (u32)(h >> (64 - 10));
}
// exclusive prefix of the 1024 bin totals: every thread d of a 1024-thread workgroup gets off[d]; *total = their sum
__device__ __forceinline__ u32 sklog_bin_offsets(const u32* __restrict__ bin_total, u32* wsum /* 16 words of LDS */, u32 tid, u32* total) {
    const u32 lane = tid & 63, wv = tid >> 6;
    const u32 c = bin_total[tid];
    u32 inc = c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const u32 v = __shfl_up(inc, o); if ((int)lane >= o) inc += v; }
    if (lane == 63) wsum[wv] = inc;
    __syncthreads();
    u32 base = 0, all = 0;
    for (u32 x = 0; x < 16; ++x) { if (x < wv) base += wsum[x]; all += wsum[x]; }
    if (total) *total = all;
    return base + inc - c;
}

template <int W>
__global__ __launch_bounds__(1024)
void kmc_sklog_hist_kernel(const u64* __restrict__ rec, const u32* __restrict__ count, u32 cap_wg, u32* __restrict__ span_hist, u32* __restrict__ bin_total) {
    __shared__ u32 cnt[KMC_SKLOG_BINS];
    const u32 w = blockIdx.x, tid = threadIdx.x;
    const u32 n = min(count[w], cap_wg);
    const u64* const span = rec + (size_t)w * cap_wg * W;
    cnt[tid] = 0;
    __syncthreads();
    for (u32 i0 = 0; i0 < n; i0 += 1024u * 4) {
        SkRec r[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const u32 i = i0 + tid + 1024u * u;
            r[u] = SkRec{0, 0, 0};
            if (i < n) r[u] = sklog_load<W>(span + (size_t)i * W);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) if (i0 + tid + 1024u * u < n) atomicAdd(&cnt[sklog_bin_of<W>(r[u].lo, r[u].mid, r[u].top)], 1u);
    }
    __syncthreads();
    const u32 c = cnt[tid];
    span_hist[(size_t)w * KMC_SKLOG_BINS + tid] = c;
    if (c) atomicAdd(&bin_total[tid], c);
}

template <int W> struct SklogPartLds {
    typedef unsigned long long u64x2_t __attribute__((ext_vector_type(2)));
    static constexpr int TILE = W == 2 ? 8192 : 4096;   // 128 KB / 96 KB of records
    u64x2_t a[TILE];
    u64 t[W == 3 ? TILE : 1];    // third word of three-word records
    u32 cnt[KMC_SKLOG_BINS];     // records of the tile per bin, then their exclusive prefix
    u32 dst[KMC_SKLOG_BINS];     // record index (in the binned array) of the tile's first record of that bin, minus its LDS index
    u32 gpos[KMC_SKLOG_BINS];    // the span's running position in every bin (index in the binned array)
    u32 wsum[16];
};
template <int W>
__global__ __launch_bounds__(1024)
void kmc_sklog_partition_kernel(const u64* __restrict__ rec, const u32* __restrict__ count, u32 cap_wg, const u32* __restrict__ span_hist,
                                const u32* __restrict__ bin_total, u32* __restrict__ bin_cursor, u64* __restrict__ binned) {
    extern __shared__ __align__(16) unsigned char sklog_smem[];
    typedef SklogPartLds<W> LT;
    LT& L = *reinterpret_cast<LT*>(sklog_smem);
    typedef unsigned long long u64x2_t __attribute__((ext_vector_type(2)));
    constexpr int TILE = LT::TILE, PER = TILE / 1024;
    const u32 w = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const u32 n = min(count[w], cap_wg);
    if (!n) return;
    const u64* const span = rec + (size_t)w * cap_wg * W;
    {   // the span's room in every bin: bin offset + what earlier reservations took
        const u32 off = sklog_bin_offsets(bin_total, L.wsum, tid, nullptr);
        const u32 c = span_hist[(size_t)w * KMC_SKLOG_BINS + tid];
        L.gpos[tid] = off + (c ? atomicAdd(&bin_cursor[tid], c) : 0u);
    }
    for (u32 t0 = 0; t0 < n; t0 += TILE) {
        const u32 tn = min((u32)TILE, n - t0);
        __syncthreads();   // (gpos written; the previous tile read out)
        L.cnt[tid] = 0;
        __syncthreads();
        SkRec rr[PER];
        u32 dr[PER];   // bin | rank within the tile << 10
#pragma unroll
        for (int e = 0; e < PER; ++e) {
            const u32 i = tid + 1024u * e;
            dr[e] = ~0u;
            rr[e] = SkRec{0, 0, 0};
            if (i < tn) {
                rr[e] = sklog_load<W>(span + (size_t)(t0 + i) * W);
                const u32 d = sklog_bin_of<W>(rr[e].lo, rr[e].mid, rr[e].top);
                dr[e] = d | (atomicAdd(&L.cnt[d], 1u) << 10);
            }
        }
        __syncthreads();
        {   // exclusive prefix of the tile's bin counts (thread d <-> bin d)
            const u32 c = L.cnt[tid];
            u32 inc = c;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const u32 v = __shfl_up(inc, o); if ((int)lane >= o) inc += v; }
            if (lane == 63) L.wsum[wv] = inc;
            __syncthreads();
            u32 base = 0;
            for (u32 x = 0; x < wv; ++x) base += L.wsum[x];
            const u32 off = base + inc - c;
            L.cnt[tid] = off;
            L.dst[tid] = L.gpos[tid] - off;
            L.gpos[tid] += c;
        }
        __syncthreads();
#pragma unroll
        for (int e = 0; e < PER; ++e) {
            if (dr[e] != ~0u) {
                const u32 p = L.cnt[dr[e] & 1023u] + (dr[e] >> 10);
                L.a[p] = u64x2_t{rr[e].lo, rr[e].mid};
                if (W == 3) L.t[p] = rr[e].top;
            }
        }
        __syncthreads();
#pragma unroll
        for (int e = 0; e < PER; ++e) {
            const u32 p = tid + 1024u * e;
            if (p < tn) {
                const u64x2_t ra = L.a[p];
                const SkRec ro{ra.x, ra.y, W == 3 ? L.t[p] : 0ull};
                const u32 d = sklog_bin_of<W>(ro.lo, ro.mid, ro.top);
                sklog_store<W>(binned + (size_t)(L.dst[d] + p) * W, ro);
            }
        }
    }
}

template <int W> struct SklogTable {
    u64 lo[KMC_SKLOG_TCAP];
    u64 mid[KMC_SKLOG_TCAP];                 // W == 2: the claimed word (never all ones: a (k+16)-mer of k <= 47 leaves its top bits clear)
    u64 top[W == 3 ? KMC_SKLOG_TCAP : 1];    // W == 3: the claimed word
    u32 cnt[KMC_SKLOG_TCAP];
    u32 nfill;
};

template <int KW, bool CANON, int W>
__global__ __launch_bounds__(1024)
void kmc_sklog_consume_kernel(const u64* __restrict__ binned, const u32* __restrict__ bin_total, int k, GTable g) {
    extern __shared__ __align__(16) unsigned char sklog_smem[];
    SklogTable<W>& T = *reinterpret_cast<SklogTable<W>*>(sklog_smem);
    __shared__ u32 s_wsum[16], s_begin;
    const u32 d = blockIdx.x, tid = threadIdx.x;
    const u32 n = bin_total[d];
    if (!n) return;   // (block-uniform)
    {
        const u32 off = sklog_bin_offsets(bin_total, s_wsum, tid, nullptr);
        if (tid == d) s_begin = off;
        __syncthreads();
    }
    const u64* const bin = binned + (size_t)s_begin * W;
    const int kb = 2 * k;
    const u64 mask_lo = kb >= 64 ? ~0ull : ((1ull << kb) - 1);
    const u64 mask_hi = kb <= 64 ? 0ull : ((1ull << (kb - 64)) - 1);
    u64* const claim = W == 3 ? T.top : T.mid;
    for (u32 i = tid; i < KMC_SKLOG_TCAP; i += 1024) { claim[i] = KMC_EMPTY64; T.cnt[i] = 0; }
    if (tid == 0) T.nfill = 0;
    __syncthreads();
    constexpr u32 M = KMC_SKLOG_TCAP - 1;
    const u32 n_round = (n + 1023u) & ~1023u;
    // (the next record of a thread is on its way while this one goes through the LDS table: the probing loop below is a chain
    // of LDS round trips with nothing else in flight.  Four or eight records ahead instead of one: no difference)
    SkRec nr{0, 0, 0};
    if (tid < n) nr = sklog_load<W>(bin + (size_t)tid * W);
    for (u32 i = tid; i < n_round; i += 1024) {
        const bool act = i < n;
        const u64 lo = nr.lo, mid = nr.mid, top = nr.top;
        if (i + 1024 < n) nr = sklog_load<W>(bin + (size_t)(i + 1024) * W);
        const u64 cw = W == 3 ? top : mid;   // the claimed word of this record
        // slot: the record's words folded to 32 bits, ONE 32-bit multiply (as kmc_stream.hip.h's home bucket; independent of the
        // bin, which is a 64-bit mix of the same words).  The first version mixed 64 bits twice -- five 64-bit multiplies, twenty
        // quarter-rate v_mul per record; measured difference: 1-2 % of the step.
        u32 fa = (u32)lo ^ __builtin_amdgcn_alignbit((u32)(lo >> 32), (u32)(lo >> 32), 21) ^ __builtin_amdgcn_alignbit((u32)mid, (u32)mid, 27) ^
                 __builtin_amdgcn_alignbit((u32)(mid >> 32), (u32)(mid >> 32), 13);
        if (W == 3) fa ^= __builtin_amdgcn_alignbit((u32)top, (u32)top, 7) ^ __builtin_amdgcn_alignbit((u32)(top >> 32), (u32)(top >> 32), 17);
        u32 h = (((fa ^ (fa >> 15)) * 0x85EBCA6Bu) >> 20) & M;
        bool done = !act, direct = false;
        int probes = 0;
        u32 trips = 0;
        // HOT PATH, straight-line (the plateau inputs repeat every record thousands of times, so nearly every record finds
        // itself in its home slot): the slot's words are read in the order they are published in reverse -- claim word first;
        // LDS executes a wave's accesses in order, so words read behind a published claim word are the published ones --
        // and the count goes up by one on a hit, by zero otherwise: no branch, no loop.  (Before, every record went through
        // the probing loop below, whose back-edge is a wave-wide ballot: more scalar than vector instructions, PMC.)
        {
            const u64 c0 = __hip_atomic_load(&claim[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            asm volatile("" ::: "memory");   // (program order = issue order: the key words are read behind the claim word)
            const u64 l0 = __hip_atomic_load(&T.lo[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            const u64 m0 = W == 3 ? __hip_atomic_load(&T.mid[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) : mid;
            const bool hit = act && c0 == cw && c0 != KMC_EMPTY64 && c0 != KMC_LOCKED64 && l0 == lo && m0 == mid;
            atomicAdd(&T.cnt[h], hit ? 1u : 0u);
            done = done || hit;
        }
        // first sight of a record, a slot being published, a collision: the probing insert.  One loop whose only back-edge is
        // taken on a wave-uniform ballot (kmc_device.hip.h, gtable_add: why)
        while (__builtin_amdgcn_ballot_w64(!done) != 0) {
            if (!done) {
                if (probes >= 24 || ++trips > (1u << 20)) { direct = true; done = true; }
                else {
                    u64 cur = __hip_atomic_load(&claim[h], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
                    if (cur == KMC_EMPTY64) {
                        if (__hip_atomic_load(&T.nfill, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >= (u32)(KMC_SKLOG_TCAP * 7 / 8)) { direct = true; done = true; }
                        else {
                            const u64 old = atomicCAS((unsigned long long*)&claim[h], KMC_EMPTY64, KMC_LOCKED64);
                            if (old == KMC_EMPTY64) {
                                __hip_atomic_store(&T.lo[h], lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                                if (W == 3) __hip_atomic_store(&T.mid[h], mid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                                __hip_atomic_store(&claim[h], cw, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                                atomicAdd(&T.nfill, 1u);
                                atomicAdd(&T.cnt[h], 1u);
                                done = true;
                            }
                            // lost the race: the slot is LOCKED or published now; examine it next trip
                        }
                    } else if (cur == KMC_LOCKED64) {
                        // being published; examine it next trip
                    } else if (cur == cw && __hip_atomic_load(&T.lo[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == lo &&
                               (W != 3 || __hip_atomic_load(&T.mid[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == mid)) {
                        atomicAdd(&T.cnt[h], 1u);
                        done = true;
                    } else { h = (h + 1) & M; probes++; }
                }
            }
        }
        if (direct) {   // no room in the LDS table: this record's 16 k-mers at once
            for (u32 j = 0; j < 16; ++j) sklog_unfold_one<KW, CANON>(g, k, top, mid, lo, j, 1, mask_hi, mask_lo);
            atomicAdd((unsigned long long*)&g.counters[KMC_CTR_BADBASE], 16ull);
        }
    }
    __syncthreads();
    // every distinct (k+16)-mer of the bin: its 16 k-mers receive its count (item = (slot, j))
    for (u32 it = tid; it < KMC_SKLOG_TCAP * 16u; it += 1024) {
        const u32 s = it >> 4, j = it & 15u;
        if (claim[s] != KMC_EMPTY64) {
            const u64 c = T.cnt[s];
            sklog_unfold_one<KW, CANON>(g, k, W == 3 ? T.top[s] : 0ull, T.mid[s], T.lo[s], j, c, mask_hi, mask_lo);
        }
    }
}
