// kmc_sklog.hip.h -- counting the walk kernel's LOGGED steps (kmc_walk.hip.h, SkLog): the (k+16)-mers of the steps
// that fell off the LDS memo, one record per step, in per-workgroup spans.  Grouping equal records is the reference's
// grouping step once more (k-mer-count/src/main.rs:84,87: sort, then equal lines), done here without a sort and
// without a global hash table:
//   1. kmc_sklog_partition_kernel  a record goes to one of 1024 BINS by a hash of all its words: per span of the log an
//                                  LDS histogram, one returning global atomic per non-empty bin to reserve the span's
//                                  room there, then the records are stored (equal records meet in one bin);
//   2. kmc_sklog_consume_kernel    one workgroup per bin counts its records in an LDS hash table (4096 entries: a bin of
//                                  the plateau inputs holds a few dozen to a few thousand distinct records) and unfolds
//                                  every distinct (k+16)-mer ONCE: its 16 k-mers receive the count in the count table.
// Work per record: one coalesced read, one 16-/32-byte store, one coalesced read, one LDS table update -- against a
// read-modify-write of three random lines of a table in HBM per step before.  Records that find no room (a bin past its
// capacity, an LDS table that is full: inputs with millions of distinct (k+16)-mers) are unfolded on the spot, k-mer by
// k-mer: always exact.
#pragma once
#include "kmc_walk.hip.h"

#define KMC_SKLOG_BINS 1024
#define KMC_SKLOG_TCAP 4096      // LDS table slots of a consume workgroup

// the 16 k-mers of one (k+16)-mer {top, mid, lo}, each + cnt (what kmc_sk_unfold_kernel does per table entry)
template <int KW, bool CANON>
__device__ __forceinline__ void sklog_unfold_one(const GTable& g, int k, u64 top, u64 hi, u64 lo, u32 j, u64 cnt, u64 mask_hi, u64 mask_lo) {
    const u32 sh = 2 * j;  // drop the last j bases
    WCtx km;
    km.lo = (sh ? ((lo >> sh) | (hi << (64 - sh))) : lo) & mask_lo;
    km.hi = KW == 2 ? ((sh ? ((hi >> sh) | (top << (64 - sh))) : hi) & mask_hi) : 0ull;
    walk_gadd<KW, CANON>(g, km, k, cnt);
}

// One workgroup per span of the log (= per workgroup of the walk launch), two passes over its records: count them per bin
// in LDS, reserve the span's room in every bin with ONE returning global atomic per bin, then store the records.  (The first
// version reserved per slice of 4096 records: 7 M returning atomics on the same 1024 cursors per GB of reads, 12.5 ms.)
template <int KW, bool CANON, int W>
__global__ __launch_bounds__(1024)
void kmc_sklog_partition_kernel(const u64* __restrict__ rec, const u32* __restrict__ count, u32 cap_wg,
                                u64* __restrict__ bins, u32* __restrict__ bin_cursor, u32 bin_cap, int k, GTable g) {
    __shared__ u32 cnt[KMC_SKLOG_BINS], gbase[KMC_SKLOG_BINS];
    typedef unsigned long long u64x2_t __attribute__((ext_vector_type(2)));
    const u32 w = blockIdx.x, tid = threadIdx.x;
    const u32 n = min(count[w], cap_wg);
    if (!n) return;
    const u64* const span = rec + (size_t)w * cap_wg * W;
    auto bin_of = [&](u64x2_t a, u64x2_t b) -> u32 {
        const u64 h = W == 4 ? kmc_hash_key<3>(b.x, a.x, a.y) : kmc_hash_key<2>(a.y, a.x);
        return (u32)(h >> (64 - 10));
    };
    cnt[tid] = 0;
    __syncthreads();
    constexpr int U = 4;   // records per thread and trip (independent loads)
    for (u32 i0 = 0; i0 < n; i0 += 1024u * U) {
        u64x2_t a[U], b[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const u32 i = i0 + tid + 1024u * u;
            a[u] = u64x2_t{0, 0}; b[u] = u64x2_t{0, 0};
            if (i < n) { const u64x2_t* r = reinterpret_cast<const u64x2_t*>(span + (size_t)i * W); a[u] = r[0]; if (W == 4) b[u] = r[1]; }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) if (i0 + tid + 1024u * u < n) atomicAdd(&cnt[bin_of(a[u], b[u])], 1u);
    }
    __syncthreads();
    {
        const u32 c = cnt[tid];
        gbase[tid] = c ? atomicAdd(&bin_cursor[tid], c) : 0u;
        cnt[tid] = 0;   // (now the span's running position inside its reservation)
    }
    __syncthreads();
    const int kb = 2 * k;
    const u64 mask_lo = kb >= 64 ? ~0ull : ((1ull << kb) - 1);
    const u64 mask_hi = kb <= 64 ? 0ull : ((1ull << (kb - 64)) - 1);
    for (u32 i0 = 0; i0 < n; i0 += 1024u * U) {
        u64x2_t a[U], b[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const u32 i = i0 + tid + 1024u * u;
            a[u] = u64x2_t{0, 0}; b[u] = u64x2_t{0, 0};
            if (i < n) { const u64x2_t* r = reinterpret_cast<const u64x2_t*>(span + (size_t)i * W); a[u] = r[0]; if (W == 4) b[u] = r[1]; }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (i0 + tid + 1024u * u < n) {
                const u32 d = bin_of(a[u], b[u]);
                const u32 pos = gbase[d] + atomicAdd(&cnt[d], 1u);
                if (pos < bin_cap) {
                    u64x2_t* o = reinterpret_cast<u64x2_t*>(bins + ((size_t)d * bin_cap + pos) * W);
                    o[0] = a[u];
                    if (W == 4) o[1] = b[u];
                } else {
                    // a bin past its capacity (1.25 x the even share of the largest possible log: only a log dominated by a
                    // few records gets here): the record's 16 k-mers at once
                    for (u32 j = 0; j < 16; ++j) sklog_unfold_one<KW, CANON>(g, k, W == 4 ? b[u].x : 0ull, a[u].y, a[u].x, j, 1, mask_hi, mask_lo);
                }
            }
        }
    }
}

template <int W> struct SklogTable {
    u64 lo[KMC_SKLOG_TCAP];
    u64 mid[KMC_SKLOG_TCAP];                 // W == 2: the claimed word (never all ones: a (k+16)-mer of k <= 47 leaves its top bits clear)
    u64 top[W == 4 ? KMC_SKLOG_TCAP : 1];    // W == 4: the claimed word
    u32 cnt[KMC_SKLOG_TCAP];
    u32 nfill;
};

template <int KW, bool CANON, int W>
__global__ __launch_bounds__(1024)
void kmc_sklog_consume_kernel(const u64* __restrict__ bins, const u32* __restrict__ bin_cursor, u32 bin_cap, int k, GTable g) {
    extern __shared__ __align__(16) unsigned char sklog_smem[];
    SklogTable<W>& T = *reinterpret_cast<SklogTable<W>*>(sklog_smem);
    typedef unsigned long long u64x2_t __attribute__((ext_vector_type(2)));
    const u32 d = blockIdx.x, tid = threadIdx.x;
    const u32 n = min(bin_cursor[d], bin_cap);
    if (!n) return;
    const int kb = 2 * k;
    const u64 mask_lo = kb >= 64 ? ~0ull : ((1ull << kb) - 1);
    const u64 mask_hi = kb <= 64 ? 0ull : ((1ull << (kb - 64)) - 1);
    u64* const claim = W == 4 ? T.top : T.mid;
    for (u32 i = tid; i < KMC_SKLOG_TCAP; i += 1024) { claim[i] = KMC_EMPTY64; T.cnt[i] = 0; }
    if (tid == 0) T.nfill = 0;
    __syncthreads();
    constexpr u32 M = KMC_SKLOG_TCAP - 1;
    const u32 n_round = (n + 1023u) & ~1023u;
    for (u32 i = tid; i < n_round; i += 1024) {
        const bool act = i < n;
        u64 lo = 0, mid = 0, top = 0;
        if (act) {
            const u64x2_t* r = reinterpret_cast<const u64x2_t*>(bins + ((size_t)d * bin_cap + i) * W);
            const u64x2_t a = r[0];
            lo = a.x; mid = a.y;
            if (W == 4) top = r[1].x;
        }
        const u64 cw = W == 4 ? top : mid;   // the claimed word of this record
        u32 h = (u32)(kmc_mix64(lo ^ kmc_mix64(mid + 0x9E3779B97F4A7C15ull) ^ (top * 0xD6E8FEB86659FD93ull)) >> 20) & M;
        bool done = !act, direct = false;
        int probes = 0;
        u32 trips = 0;
        // one loop whose only back-edge is taken on a wave-uniform ballot (kmc_device.hip.h, gtable_add: why)
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
                                if (W == 4) __hip_atomic_store(&T.mid[h], mid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
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
                               (W != 4 || __hip_atomic_load(&T.mid[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == mid)) {
                        atomicAdd(&T.cnt[h], 1u);
                        done = true;
                    } else { h = (h + 1) & M; probes++; }
                }
            }
        }
        if (direct) {   // no room in the LDS table: this record's 16 k-mers at once
            for (u32 j = 0; j < 16; ++j) sklog_unfold_one<KW, CANON>(g, k, top, mid, lo, j, 1, mask_hi, mask_lo);
        }
    }
    __syncthreads();
    // every distinct (k+16)-mer of the bin: its 16 k-mers receive its count (item = (slot, j))
    for (u32 it = tid; it < KMC_SKLOG_TCAP * 16u; it += 1024) {
        const u32 s = it >> 4, j = it & 15u;
        if (claim[s] != KMC_EMPTY64) {
            const u64 c = T.cnt[s];
            sklog_unfold_one<KW, CANON>(g, k, W == 4 ? T.top[s] : 0ull, T.mid[s], T.lo[s], j, c, mask_hi, mask_lo);
        }
    }
}
