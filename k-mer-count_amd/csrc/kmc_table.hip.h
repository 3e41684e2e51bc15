// kmc_table.hip.h -- maintenance kernels of the global count table: the GPU side of the
// reference's grouping/ordering step (k-mer-count/src/main.rs:84,87): compaction of occupied
// slots, rehash on growth, merging (key,count) pairs (spill drain, multi-GPU reduce), owner
// partition for the all-to-all, and small utilities.
#pragma once
#include "kmc_device.hip.h"

// occupied slots -> dense (hi, lo, cnt) arrays (+ the identity permutation when asked for) and the sum of
// all counts (counters[KMC_CTR_SUM]) in the same launch.  Every WAVE owns a contiguous span of slots: it
// counts its occupied slots first, reserves their output range with ONE returning atomic, then copies.
// (One atomic per entry -- and later one per 64 slots -- on the same counter cost 70 ms and 12 ms on a
// 64 M-slot table with 21 M entries; the span form needs 8192 atomics whatever the table.)
// The same for a table whose claimed slots are all LISTED (at most KMC_OCC_LIST_CAP keys, nothing spilled -- the host knows
// both from the counters it has just polled): entry i comes from the dense key list and occ_list[i], no scan of the slots and
// no atomics.  (The scan above took 0.20 ms for the 68 k keys of a 16 Mi-slot table: a third of that table's finalize.)
template <int KW>
__global__ __launch_bounds__(256)
void kmc_compact_list_kernel(GTable g, u64 n, u64* __restrict__ out_hi, u64* __restrict__ out_lo, u64* __restrict__ out_cnt) {
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) {
        out_lo[i] = g.occ_key_lo[i];
        if (KW == 2) out_hi[i] = g.occ_key_hi[i];
        out_cnt[i] = g.count[g.occ_list[i]];
    }
}

template <int KW>
__global__ __launch_bounds__(256)
void kmc_compact_kernel(GTable g, u64* out_hi, u64* out_lo, u64* out_cnt, u64* out_idx, int parity) {
    const u64 cap = g.capmask + 1;
    const int c_out = parity ? KMC_CTR_OUT1 : KMC_CTR_OUT, c_sum = parity ? KMC_CTR_SUM1 : KMC_CTR_SUM;
    if (blockIdx.x == 0 && threadIdx.x == 0) {  // clear the pair the NEXT finalize will use
        g.counters[parity ? KMC_CTR_OUT : KMC_CTR_OUT1] = 0;
        g.counters[parity ? KMC_CTR_SUM : KMC_CTR_SUM1] = 0;
    }
    const u32 lane = threadIdx.x & 63;
    const u64 n_waves = (u64)gridDim.x * (blockDim.x >> 6), wave = (u64)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const u64 span = ((cap + n_waves - 1) / n_waves + 63) & ~63ull;
    const u64 s0 = wave * span, s1 = min(s0 + span, cap);
    u64 total = 0;
    for (u64 s = s0 + lane; (s - lane) < s1; s += 64) {
        const bool occ = s < s1 && ((KW == 1) ? (g.key_lo[s] != KMC_EMPTY64) : (g.key_hi[s] != KMC_EMPTY64));
        total += (u64)__popcll(__builtin_amdgcn_ballot_w64(occ));
    }
    if (total == 0) return;  // (wave-uniform)
    unsigned long long base = 0;
    if (lane == 0) base = atomicAdd((unsigned long long*)&g.counters[c_out], (unsigned long long)total);
    base = ((unsigned long long)(u32)__builtin_amdgcn_readfirstlane((int)(u32)(base >> 32)) << 32) | (u32)__builtin_amdgcn_readfirstlane((int)(u32)base);
    u64 sum = 0;
    for (u64 s = s0 + lane; (s - lane) < s1; s += 64) {
        const bool occ = s < s1 && ((KW == 1) ? (g.key_lo[s] != KMC_EMPTY64) : (g.key_hi[s] != KMC_EMPTY64));
        const unsigned long long m = __builtin_amdgcn_ballot_w64(occ);
        if (occ) {
            const u64 idx = base + __builtin_amdgcn_mbcnt_hi((u32)(m >> 32), __builtin_amdgcn_mbcnt_lo((u32)m, 0u));
            if (KW == 2) out_hi[idx] = g.key_hi[s];
            out_lo[idx] = g.key_lo[s];
            const u64 c = g.count[s];
            out_cnt[idx] = c;
            if (out_idx) out_idx[idx] = idx;
            sum += c;
        }
        base += (unsigned long long)__popcll(m);
    }
    sum = wave_sum_u64(sum);
    if (lane == 0 && sum) atomicAdd((unsigned long long*)&g.counters[c_sum], sum);
}

// kmc_reset in one launch: every slot empty, every counter zero.  The device decides how: a table whose claimed
// slots are all listed (at most KMC_OCC_LIST_CAP keys, nothing spilled: every table of generator-style input) is emptied
// through the list -- a few thousand stores instead of a 16 MB memset --, an empty table is left alone (a finalize queued
// without waiting, kmc_finalize_async, drains it when it succeeds and the host cannot know), anything else is cleared
// slot by slot.  Every workgroup must see the SAME counters, so they are cleared by the workgroup that draws the last
// ticket of *done (zero between launches), after all have read them.
// sk: the (k+16)-mer table of the walk path (kmc_walk.hip.h) or an empty GTable.  Counts still pending there belong to the
// batches that are being thrown away: they are cleared with the table (keys stay), so that a caller who queues
// finalize -> reset -> next batch without ever waiting needs no separate unfold launch in between.
template <int KW>
__global__ void kmc_reset_kernel(GTable g, GTable sk, u32* done) {
    const bool sk_pending = sk.key_lo != nullptr && sk.counters[KMC_CTR_KMERS] != 0;
    if (sk_pending) {
        const u64 n_occ = min(sk.counters[KMC_CTR_OCCUPIED], sk.occ_list_cap);
        for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n_occ; i += (u64)gridDim.x * blockDim.x) sk.count[sk.occ_list[i]] = 0;
    }
    const u64 cap = g.capmask + 1;
    const u64 n = g.counters[KMC_CTR_OCCUPIED];
    const bool listed = n <= g.occ_list_cap && g.counters[KMC_CTR_SPILL] == 0 && g.counters[KMC_CTR_ERR] == 0;
    if (listed) {
        for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) {
            const u64 s = g.occ_list[i];
            if (KW == 2) { g.key_hi[s] = KMC_EMPTY64; g.key_lo[s] = 0; } else g.key_lo[s] = KMC_EMPTY64;
            g.count[s] = 0;
        }
    } else {
        for (u64 s = (u64)blockIdx.x * blockDim.x + threadIdx.x; s < cap; s += (u64)gridDim.x * blockDim.x) {
            if (KW == 2) { g.key_hi[s] = KMC_EMPTY64; g.key_lo[s] = 0; } else g.key_lo[s] = KMC_EMPTY64;
            g.count[s] = 0;
        }
    }
    __shared__ u32 s_last;
    __syncthreads();   // (this workgroup has read the counters)
    if (threadIdx.x == 0) s_last = (atomicAdd(done, 1u) == gridDim.x - 1) ? 1u : 0u;
    __syncthreads();
    if (s_last) {
        if (threadIdx.x < KMC_CTR_N) g.counters[threadIdx.x] = 0;
        if (threadIdx.x == 0) *done = 0;
        if (sk_pending && threadIdx.x == 0) { sk.counters[KMC_CTR_SPILL] = 0; sk.counters[KMC_CTR_KMERS] = 0; }
    }
}

// The table as it is before a launch whose size rests on a PREDICTION of how many new keys it brings
// (kmc_api.hip, launch planner): the claimed slots are listed in occ_list while there are at most
// occ_list_cap of them, so a small table is saved by copying those entries -- a few microseconds.
// *s_n = number of entries saved, or ~0 when the table is too large for this form.
template <int KW>
__global__ void kmc_snapshot_kernel(GTable g, u64* __restrict__ s_hi, u64* __restrict__ s_lo, u64* __restrict__ s_cnt, u64* __restrict__ s_n) {
    const u64 n = g.counters[KMC_CTR_OCCUPIED];
    const u64 i0 = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (n > g.occ_list_cap) { if (i0 == 0) *s_n = ~0ull; return; }
    for (u64 i = i0; i < n; i += (u64)gridDim.x * blockDim.x) {
        const u64 slot = g.occ_list[i];
        s_lo[i] = g.key_lo[slot];
        s_cnt[i] = g.count[slot];
        if (KW == 2) s_hi[i] = g.key_hi[slot];
    }
    if (i0 == 0) *s_n = n;
}

// re-insert every entry of `old` into `g` (growth)
template <int KW>
__global__ void kmc_rehash_kernel(GTable old, GTable g) {
    const u64 cap = old.capmask + 1;
    for (u64 s = (u64)blockIdx.x * blockDim.x + threadIdx.x; s < cap; s += (u64)gridDim.x * blockDim.x) {
        bool occ = (KW == 1) ? (old.key_lo[s] != KMC_EMPTY64) : (old.key_hi[s] != KMC_EMPTY64);
        if (occ) gtable_add<KW>(g, KW == 2 ? old.key_hi[s] : 0ull, old.key_lo[s], old.count[s]);
    }
}

// add n (key,count) pairs to the table
template <int KW>
__global__ void kmc_merge_pairs_kernel(GTable g, const u64* hi, const u64* lo, const u64* cnt, u64 n) {
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) {
        u64 c = cnt[i];
        if (c) gtable_add<KW>(g, (KW == 2 && hi) ? hi[i] : 0ull, lo[i], c);
    }
}



// longest read of a batch -> counters[KMC_CTR_MAXLEN] (atomicMax), used when the caller of
// kmc_add_batch_device does not know it
__global__ void kmc_maxlen_kernel(const u64* offsets, u64 n_reads, u64* counters) {
    u64 m = 0;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n_reads; i += (u64)gridDim.x * blockDim.x) {
        u64 l = offsets[i + 1] - offsets[i];
        m = l > m ? l : m;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        u64 t = __shfl_xor(m, o);
        m = t > m ? t : m;
    }
    if ((threadIdx.x & 63) == 0 && m) atomicMax((unsigned long long*)&counters[KMC_CTR_MAXLEN], m);
}

// owner of a key for the multi-GPU all-to-all (same function as kmc_owner_of on the host)
__host__ __device__ inline u32 kmc_owner(u64 hi, u64 lo, u32 n_parts) {
    u64 z = lo ^ (hi * 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (u32)((z >> 32) * (u64)n_parts >> 32);
}


// ---- owner partition for the all-to-all (multi-GPU reduce of large tables) --------------------------
// part_cnt[p] = pairs whose owner is p
__global__ void kmc_owner_count_kernel(const u64* __restrict__ hi, const u64* __restrict__ lo, u64 n, u32 n_parts, unsigned long long* __restrict__ part_cnt) {
    extern __shared__ unsigned int oc_smem[];
    for (u32 p = threadIdx.x; p < n_parts; p += blockDim.x) oc_smem[p] = 0;
    __syncthreads();
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) atomicAdd(&oc_smem[kmc_owner(hi ? hi[i] : 0ull, lo[i], n_parts)], 1u);
    __syncthreads();
    for (u32 p = threadIdx.x; p < n_parts; p += blockDim.x) if (oc_smem[p]) atomicAdd(&part_cnt[p], (unsigned long long)oc_smem[p]);
}
// pairs to their owner's span (cursor[p] starts at the span's begin); the order inside a span is
// arbitrary (the receiver merges pairs into its table).  One global add per (wave, owner present in it).
__global__ void kmc_owner_scatter_kernel(const u64* __restrict__ hi, const u64* __restrict__ lo, const u64* __restrict__ cnt, u64 n, u32 n_parts,
                                         unsigned long long* __restrict__ cursor, u64* __restrict__ o_hi, u64* __restrict__ o_lo, u64* __restrict__ o_cnt) {
    const u64 n_round = (n + 63) & ~63ull;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n_round; i += (u64)gridDim.x * blockDim.x) {
        const bool in = i < n;
        const u32 mine = in ? kmc_owner(hi ? hi[i] : 0ull, lo[i], n_parts) : ~0u;
        bool todo = in;
        u64 pos = 0;
        unsigned long long pending;
        while ((pending = __builtin_amdgcn_ballot_w64(todo)) != 0) {
            const u32 p = (u32)__builtin_amdgcn_readlane((int)mine, (int)__builtin_ctzll(pending));
            const unsigned long long grp = __builtin_amdgcn_ballot_w64(todo && mine == p);
            unsigned long long base = 0;
            if ((threadIdx.x & 63) == (u32)__builtin_ctzll(grp)) base = atomicAdd(&cursor[p], (unsigned long long)__popcll(grp));
            base = ((unsigned long long)(u32)__builtin_amdgcn_readlane((int)(u32)(base >> 32), (int)__builtin_ctzll(grp)) << 32) | (u32)__builtin_amdgcn_readlane((int)(u32)base, (int)__builtin_ctzll(grp));
            if (todo && mine == p) {
                pos = base + __builtin_amdgcn_mbcnt_hi((u32)(grp >> 32), __builtin_amdgcn_mbcnt_lo((u32)grp, 0u));
                todo = false;
            }
        }
        if (in) {
            o_lo[pos] = lo[i];
            o_cnt[pos] = cnt[i];
            if (hi) o_hi[pos] = hi[i];
        }
    }
}

// ---- multi-GPU reduce, small tables: fixed-size slabs ------------------------------------------
// A slab is what one rank contributes to ONE all-gather: an 8-word header followed by fixed-capacity
// arrays, so every rank sends the same number of words and no size exchange (and no host
// synchronisation) is needed.  Layout in u64 words, E = slab_entries:
//   [0] n pairs in this slab, or KMC_SLAB_OVERSIZE when the table has more than E keys (then the
//       payload is absent and the rank's table travels by the partitioned all-to-all instead)
//   [1] sum of the counts   [2..7] reserved (0)
//   [8, 8+E) key_lo   [8+E, 8+2E) count   [8+2E, 8+3E) key_hi (two-word keys only)
#define KMC_SLAB_HEADER 8
#define KMC_SLAB_OVERSIZE (~0ull)
template <int KW>
__global__ void kmc_pack_slab_kernel(const u64* __restrict__ hi, const u64* __restrict__ lo, const u64* __restrict__ cnt,
                                     u64 n, u64 sum, u64 entries, u64* __restrict__ slab) {
    const u64 i0 = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i0 < KMC_SLAB_HEADER) slab[i0] = i0 == 0 ? (n <= entries ? n : KMC_SLAB_OVERSIZE) : (i0 == 1 ? sum : 0ull);
    if (n > entries) return;
    for (u64 i = i0; i < n; i += (u64)gridDim.x * blockDim.x) {
        slab[KMC_SLAB_HEADER + i] = lo[i];
        slab[KMC_SLAB_HEADER + entries + i] = cnt[i];
        if (KW == 2) slab[KMC_SLAB_HEADER + 2 * entries + i] = hi[i];
    }
}

// The same slab straight from the LIVE table (no finalize, no sort -- a slab need not be ordered):
// the first KMC_OCC_LIST_CAP claimed slots are listed in g.occ_list, so a small table is packed
// without scanning it and without the host knowing its size.  Oversize when the table has more
// keys than the slab (or than the list), when anything spilled, when (k+16)-mer counts are pending, or when the host says so.
template <int KW>
__global__ void kmc_pack_slab_live_kernel(GTable g, const u64* __restrict__ sk_counters, u64 entries, int force_oversize, u64* __restrict__ slab) {
    const u64 n = g.counters[KMC_CTR_OCCUPIED];
    // (sk_counters: the (k+16)-mer table's, or null.  Counts pending there are not in the table yet: the host only queues their
    // unfold when a poll has seen that table in use on this source -- if it guessed wrong, the slab says "oversize" and the
    // table travels the other way, through kmc_finalize, which settles them)
    const bool over = force_oversize || n > entries || n > g.occ_list_cap || g.counters[KMC_CTR_SPILL] != 0 || g.counters[KMC_CTR_ERR] != 0 ||
                      (sk_counters && sk_counters[KMC_CTR_KMERS] != 0);
    const u64 i0 = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i0 < KMC_SLAB_HEADER) slab[i0] = i0 == 0 ? (over ? KMC_SLAB_OVERSIZE : n) : (i0 == 1 ? g.counters[KMC_CTR_KMERS] : 0ull);
    if (over) return;
    for (u64 i = i0; i < n; i += (u64)gridDim.x * blockDim.x) {
        const u64 slot = g.occ_list[i];
        slab[KMC_SLAB_HEADER + i] = g.key_lo[slot];
        slab[KMC_SLAB_HEADER + entries + i] = g.count[slot];
        if (KW == 2) slab[KMC_SLAB_HEADER + 2 * entries + i] = g.key_hi[slot];
    }
}

// The owner's side: of the n_slabs gathered slabs, add every pair whose owner(key) is `my_part`.
template <int KW>
__global__ void kmc_merge_slabs_kernel(GTable g, const u64* __restrict__ slabs, u32 n_slabs, u64 slab_words, u64 entries,
                                       u32 my_part, u32 n_parts) {
    const u64 total = (u64)n_slabs * entries;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (u64)gridDim.x * blockDim.x) {
        const u64 sl = i / entries, j = i - sl * entries;
        const u64* s = slabs + sl * slab_words;
        const u64 n = s[0];
        if (n == KMC_SLAB_OVERSIZE) {
            if (j == 0) atomicAdd((unsigned long long*)&g.counters[KMC_CTR_SLABSKIP], 1ull);
            continue;
        }
        if (j >= n) continue;
        const u64 klo = s[KMC_SLAB_HEADER + j], c = s[KMC_SLAB_HEADER + entries + j];
        const u64 khi = KW == 2 ? s[KMC_SLAB_HEADER + 2 * entries + j] : 0ull;
        if (c && kmc_owner(khi, klo, n_parts) == my_part) gtable_add<KW>(g, khi, klo, c);
    }
}


// Fast finalize for small tables (n <= KMC_FIN_KERNEL_MAX claimed slots, listed in g.occ_list): the GPU
// form of the reference's final ordering step (k-mer-count/src/main.rs:87) for the common case of a
// few thousand distinct keys, in ONE launch that uses the whole chip instead of one CU:
//   rank sort -- keys in a table are distinct, so the sorted position of key i is the number of keys
//   smaller than it.  Workgroup b OWNS keys 16 b .. 16 b + 15: it brings every key of the table through
//   LDS in tiles of 4096 (2048 two-word) keys and thread (m, s) counts how many of the tile's keys s, s + 64, ...
//   are smaller than owned key m (the 16 lanes of a quarter wave read the same LDS address: broadcasts, no bank
//   conflict: 52 compares per thread for the benchmark's 3,350 keys, 210 workgroups side by side); the 64 partial
//   ranks of a key are added up by shuffles + LDS and the workgroup writes its 16 (key, count) pairs to their
//   final places.  No global atomic, no second phase.
//   (Round 2's version turned this around -- every workgroup compared ALL keys against its 64 and added
//   partial ranks to a global rank[] array, 177 k device-scope atomics for 3,350 keys, then the workgroup
//   with the last ticket scattered: 20-31 us per launch, a third of the step's time outside the count kernel.)
// It is launched speculatively right behind the count kernels: it reads the occupancy on the device and
// gives up (FASTFIN = 0) unless the table is small, nothing spilled and the (k+16)-mer table holds no
// pending counts, so kmc_finalize needs a single host synchronisation.
//
// DRAIN.  When it succeeds the sorted view holds everything the table held, so the kernel also EMPTIES the
// table: the workgroup that draws the last ticket (all reads of the table are over by then) clears the
// claimed slots, publishes the device counters to the host's pinned mirror (no read-back copy) and zeroes
// them -- the table is as kmc_reset leaves it.  kmc_reset after kmc_finalize then launches nothing, and a
// caller that goes on adding to a finalized ctx gets the view merged back first (kmc_api.hip: undrain).
// (History: a single-workgroup bitonic network took 55-60 us for 3,350 keys whether it ran in LDS
// with workgroup barriers, in LDS with wave-local passes, or in registers with wave shuffles.)
#define KMC_FIN_CHUNK 16
// host_mirror: the ctx's pinned mirror of [count-table counters | (k+16)-mer-table counters].  The host clears
// mirror[KMC_CTR_FASTFIN] before the launch; 1 afterwards means: sorted view written, counters published
// (mirror[KMC_CTR_SUM2] = sum of all counts), table drained; mirror[KMC_CTR_FINSEQ] = seq is written LAST (the host waits on it).
// The device copy of FASTFIN is never written; the device copy of SUM2 carries the sum from workgroup 0 to the publisher.
// ticket[0]: the ticket counter; ticket[1], ticket[2]: launches that produced a view / oversize slabs they saw, since
// kmc_create (a caller that queues several finalizes without waiting -- kmc_finalize_async -- checks afterwards that
// every one of them delivered); seq: this launch's number, published with everything else.
template <int KW>
__global__ __launch_bounds__(1024)
void kmc_small_finalize_kernel(GTable g, const u64* __restrict__ sk_counters, u32* __restrict__ ticket, u64* __restrict__ host_mirror, u64 seq,
                               u64* __restrict__ out_hi, u64* __restrict__ out_lo, u64* __restrict__ out_cnt) {
    constexpr int KMC_FIN_TILE = KW == 1 ? 4096 : 2048;   // keys of the table in LDS at a time (32 KB)
    __shared__ u64 t_lo[KMC_FIN_TILE];
    __shared__ u64 t_hi[KW == 2 ? KMC_FIN_TILE : 1];
    __shared__ u32 s_rank[KMC_FIN_CHUNK];
    __shared__ u32 s_last;
    __shared__ u64 s_sum[16];
    const u32 tid = threadIdx.x;
    const u64 n = g.counters[KMC_CTR_OCCUPIED];
    const bool ok = n > 0 && n <= KMC_FIN_KERNEL_MAX && n <= g.occ_list_cap && n <= (u64)gridDim.x * KMC_FIN_CHUNK &&
                    g.counters[KMC_CTR_SPILL] == 0 && g.counters[KMC_CTR_ERR] == 0 &&
                    sk_counters[KMC_CTR_KMERS] == 0;   // (no counts pending in the (k+16)-mer table; its keys stay across launches)
    // (every workgroup that takes part reads these counters before it draws its ticket, and they change only
    // behind the last ticket: all of them decide alike.  A workgroup past the table that starts that late may
    // read zeros -- it leaves either way.)
    if (!ok) {
        if (blockIdx.x == 0 && tid == 0) {   // "gave up": the table is as it was
            __hip_atomic_store(&host_mirror[KMC_CTR_FASTFIN], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(&host_mirror[KMC_CTR_FINOK], (u64)ticket[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(&host_mirror[KMC_CTR_FINSKIP], (u64)ticket[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __threadfence_system();   // (FINSEQ is the word the host waits on: last)
            __hip_atomic_store(&host_mirror[KMC_CTR_FINSEQ], seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        return;
    }
    const u32 nb = (u32)((n + KMC_FIN_CHUNK - 1) / KMC_FIN_CHUNK);
    if (blockIdx.x >= nb) return;
    // my owned key: thread (m = tid & 15, seg = tid >> 4) -- every quarter wave holds the workgroup's 16 keys, one per lane
    const u32 m = tid & (KMC_FIN_CHUNK - 1u), seg = tid / KMC_FIN_CHUNK;
    constexpr u32 NSEG = 1024 / KMC_FIN_CHUNK;
    const u64 mine_i = (u64)blockIdx.x * KMC_FIN_CHUNK + m;
    // keys come from the dense key list (GTable::occ_key_*: coalesced, no occ_list -> slot -> key chain); counts via the slot
    u64 mlo = ~0ull, mhi = ~0ull, mcnt = 0;
    if (mine_i < n) {
        mlo = g.occ_key_lo[mine_i];
        mhi = KW == 2 ? g.occ_key_hi[mine_i] : 0ull;
        mcnt = g.count[g.occ_list[mine_i]];
    }
    if (tid < KMC_FIN_CHUNK) s_rank[tid] = 0;
    u64 sum = 0;  // (workgroup 0 also adds up all counts)
    u32 r = 0;
    constexpr int KPT = KMC_FIN_TILE / 1024;
    u64 nlo[KPT], nhi[KPT];   // the next tile, on its way while this one is compared
    auto load_tile = [&](u64 t0) {
#pragma unroll
        for (int e = 0; e < KPT; ++e) {
            const u64 i = t0 + tid + 1024u * e;
            nlo[e] = 0; nhi[e] = 0;
            if (i < n) {
                nlo[e] = g.occ_key_lo[i];
                if (KW == 2) nhi[e] = g.occ_key_hi[i];
                if (blockIdx.x == 0) sum += g.count[g.occ_list[i]];
            }
        }
    };
    load_tile(0);
    for (u64 t0 = 0; t0 < n; t0 += KMC_FIN_TILE) {
        const u32 tn = (u32)min((u64)KMC_FIN_TILE, n - t0);
        __syncthreads();  // (the previous tile has been read)
#pragma unroll
        for (int e = 0; e < KPT; ++e) {
            const u32 j = tid + 1024u * e;
            if (j < tn) { t_lo[j] = nlo[e]; if (KW == 2) t_hi[j] = nhi[e]; }
        }
        if (t0 + KMC_FIN_TILE < n) load_tile(t0 + KMC_FIN_TILE);   // (block-uniform)
        __syncthreads();
        // elements seg, seg + 64, ... of the tile against my key (one address per quarter wave: LDS broadcast reads);
        // four independent reads per trip, so that the loop does not pay an LDS round trip per element
        u32 j = seg;
        for (; j + 3 * NSEG < tn; j += 4 * NSEG) {
            u64 cl[4], ch[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { cl[u] = t_lo[j + u * NSEG]; ch[u] = KW == 2 ? t_hi[j + u * NSEG] : 0ull; }
#pragma unroll
            for (int u = 0; u < 4; ++u) r += (KW == 2 ? (ch[u] < mhi || (ch[u] == mhi && cl[u] < mlo)) : cl[u] < mlo) ? 1u : 0u;
        }
        for (; j < tn; j += NSEG) {
            const u64 cl = t_lo[j];
            const u64 ch = KW == 2 ? t_hi[j] : 0ull;
            r += (KW == 2 ? (ch < mhi || (ch == mhi && cl < mlo)) : cl < mlo) ? 1u : 0u;
        }
    }
    r += __shfl_xor(r, 16);   // the four segments of a wave
    r += __shfl_xor(r, 32);
    if ((tid & 63u) < KMC_FIN_CHUNK && r) atomicAdd(&s_rank[m], r);
    if (blockIdx.x == 0) {
        sum = wave_sum_u64(sum);
        if ((tid & 63) == 0) s_sum[tid >> 6] = sum;
    }
    // every read of the table by this workgroup has COMPLETED before its ticket is drawn (the last workgroup
    // empties the table behind the last ticket)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid < 64) {   // wave 0: the workgroup's 16 entries of the view, then its ticket
        if (tid < KMC_FIN_CHUNK && mine_i < n) {
            const u32 pos = s_rank[tid];
            out_lo[pos] = mlo;
            if (KW == 2) out_hi[pos] = mhi;
            out_cnt[pos] = mcnt;
        }
        if (tid == 0 && blockIdx.x == 0) {   // the sum of all counts, for the workgroup that publishes (device word SUM2: otherwise unused)
            u64 tot = 0;
            for (int w = 0; w < 16; ++w) tot += s_sum[w];
            __hip_atomic_store(&g.counters[KMC_CTR_SUM2], tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        // (No fence here.  The host is told "view complete" by the last ticket's workgroup, possibly before this kernel has
        // ended: work queued on the ctx stream sees the entries in stream order; anybody else gets the pointers from
        // kmc_export_device, which waits for the end of this kernel first -- a device-scope release in every one of the
        // ~200 workgroups cost 10 us of a 16 us kernel.)
        if (tid == 0) {
            if (blockIdx.x == 0) __threadfence();   // (the sum above, read by the publishing workgroup, possibly on another XCD)
            s_last = (atomicAdd(ticket, 1u) == nb - 1) ? 1u : 0u;
        }
    }
    __syncthreads();
    if (!s_last) return;
    // ---- the last workgroup: publish the counters, empty the table (every workgroup has finished READING it) ----
    // The host waits on mirror[FINSEQ] (kmc_api.hip: poll_fin), not on the end of the kernel: that word is written
    // last, behind a system-scope fence; everything after it touches device memory only, in stream order before
    // whatever the host launches next.
    if (tid < KMC_CTR_FINSEQ && tid != KMC_CTR_SUM2 && tid != KMC_CTR_FASTFIN)
        __hip_atomic_store(&host_mirror[tid], g.counters[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    else if (tid == KMC_CTR_FINSEQ) {
        const u32 okn = ticket[1] + 1u, skn = ticket[2] + (u32)g.counters[KMC_CTR_SLABSKIP];
        ticket[1] = okn;
        ticket[2] = skn;
        __hip_atomic_store(&host_mirror[KMC_CTR_FINOK], (u64)okn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&host_mirror[KMC_CTR_FINSKIP], (u64)skn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        const u64 tot = __hip_atomic_load(&g.counters[KMC_CTR_SUM2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&host_mirror[KMC_CTR_SUM2], tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&host_mirror[KMC_CTR_FASTFIN], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    else if (tid >= 64 && tid < 64 + KMC_CTR_N)
        __hip_atomic_store(&host_mirror[KMC_CTR_N + tid - 64], sk_counters[tid - 64], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __threadfence_system();
    __syncthreads();
    if (tid == 0) __hip_atomic_store(&host_mirror[KMC_CTR_FINSEQ], seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    for (u64 i = tid; i < n; i += 1024) {
        const u64 slot = g.occ_list[i];
        if (KW == 2) { g.key_hi[slot] = KMC_EMPTY64; g.key_lo[slot] = 0; } else g.key_lo[slot] = KMC_EMPTY64;
        g.count[slot] = 0;
    }
    __syncthreads();   // (the counters above have been read)
    if (tid < KMC_CTR_N) g.counters[tid] = 0;
    if (tid == 0) *ticket = 0;
}
