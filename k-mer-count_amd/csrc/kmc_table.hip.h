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

// kmc_reset in one launch: every slot empty, every counter zero
template <int KW>
__global__ void kmc_reset_kernel(GTable g) {
    const u64 cap = g.capmask + 1;
    for (u64 s = (u64)blockIdx.x * blockDim.x + threadIdx.x; s < cap; s += (u64)gridDim.x * blockDim.x) {
        if (KW == 2) { g.key_hi[s] = KMC_EMPTY64; g.key_lo[s] = 0; } else g.key_lo[s] = KMC_EMPTY64;
        g.count[s] = 0;
    }
    if (blockIdx.x == 0 && threadIdx.x < KMC_CTR_N) g.counters[threadIdx.x] = 0;
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
// keys than the slab (or than the list), when anything spilled, or when the host says so.
template <int KW>
__global__ void kmc_pack_slab_live_kernel(GTable g, u64 entries, int force_oversize, u64* __restrict__ slab) {
    const u64 n = g.counters[KMC_CTR_OCCUPIED];
    const bool over = force_oversize || n > entries || n > g.occ_list_cap || g.counters[KMC_CTR_SPILL] != 0 || g.counters[KMC_CTR_ERR] != 0;
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


// Fast finalize for small tables (n <= KMC_OCC_LIST_CAP claimed slots, listed in g.occ_list): the GPU
// form of the reference's final ordering step (k-mer-count/src/main.rs:87) for the common case of a
// few thousand distinct keys, in ONE launch that uses the whole chip instead of one CU:
//   rank sort -- keys in a table are distinct, so the sorted position of key i is the number of
//   keys smaller than it.  Workgroup b holds 64 of the keys in LDS and adds, for EVERY key i, how
//   many of its 64 are smaller (LDS broadcast reads, no data movement) to rank[i] with an
//   agent-scope atomic; n*n/64 comparisons per workgroup, n/64 workgroups side by side.  The
//   workgroup whose "done" ticket is the last one scatters (key, count) to position rank[i], sums
//   the counts and leaves rank[] and the ticket counter zeroed for the next launch.
// It is launched speculatively right behind the count kernels: it reads the occupancy on the device
// and gives up (FASTFIN = 0) unless the table is small and nothing spilled, so kmc_finalize needs a
// single host synchronisation.
// (History: a single-workgroup bitonic network took 55-60 us for 3,350 keys whether it ran in LDS
// with workgroup barriers, in LDS with wave-local passes, or in registers with wave shuffles --
// stamps: 6 us gather, 39 us network, 9 us output; one CU's LDS pipe carries all the data movement.)
#define KMC_FIN_CHUNK 64
#define KMC_FIN_ROUND 8   // keys per thread and round (a round = 8192 keys of the table)
template <int KW>
__global__ __launch_bounds__(1024)
void kmc_small_finalize_kernel(GTable g, u32* __restrict__ rank, u64* __restrict__ out_hi, u64* __restrict__ out_lo, u64* __restrict__ out_cnt) {
    __shared__ u64 c_lo[KMC_FIN_CHUNK];
    __shared__ u64 c_hi[KW == 2 ? KMC_FIN_CHUNK : 1];
    __shared__ u32 s_last;
    __shared__ u64 s_sum[16];
    const u32 tid = threadIdx.x;
    const u64 n = g.counters[KMC_CTR_OCCUPIED];
    const bool ok = n > 0 && n <= KMC_OCC_LIST_CAP && n <= g.occ_list_cap && n <= (u64)gridDim.x * KMC_FIN_CHUNK &&
                    g.counters[KMC_CTR_SPILL] == 0 && g.counters[KMC_CTR_ERR] == 0;
    if (!ok) {  // (every workgroup reads the same counters; nothing else writes them while this kernel runs)
        if (blockIdx.x == 0 && tid == 0) { g.counters[KMC_CTR_FASTFIN] = 0; g.counters[KMC_CTR_SUM2] = 0; }
        return;
    }
    const u32 nb = (u32)((n + KMC_FIN_CHUNK - 1) / KMC_FIN_CHUNK);
    if (blockIdx.x >= nb) return;
    // this workgroup's 64 keys (padding = all ones: never smaller than a valid key)
    if (tid < KMC_FIN_CHUNK) {
        const u64 j = (u64)blockIdx.x * KMC_FIN_CHUNK + tid;
        u64 lo = ~0ull, hi = ~0ull;
        if (j < n) {
            const u64 slot = g.occ_list[j];
            lo = g.key_lo[slot];
            hi = KW == 2 ? g.key_hi[slot] : 0ull;
        }
        c_lo[tid] = lo;
        if (KW == 2) c_hi[tid] = hi;
    }
    __syncthreads();
    // every key of the table against this workgroup's 64, in rounds of 8 keys per thread (the first
    // version held all of a table of <= 8192 keys in registers; rounds lift that limit to 32768 keys:
    // between 8 k and 32 k keys the general sort costs 0.7 ms in launches and host round trips)
    const u32 n_rounds = (u32)((n + 1024 * KMC_FIN_ROUND - 1) / (1024 * KMC_FIN_ROUND));
    for (u32 rd = 0; rd < n_rounds; ++rd) {
        u64 klo[KMC_FIN_ROUND], khi[KMC_FIN_ROUND];
        u32 r[KMC_FIN_ROUND];
#pragma unroll
        for (int e = 0; e < KMC_FIN_ROUND; ++e) {
            const u64 i = ((u64)rd * KMC_FIN_ROUND + e) * 1024 + tid;
            klo[e] = 0; khi[e] = 0; r[e] = 0;
            if (i < n) {
                const u64 slot = g.occ_list[i];
                klo[e] = g.key_lo[slot];
                if (KW == 2) khi[e] = g.key_hi[slot];
            }
        }
        for (int j = 0; j < KMC_FIN_CHUNK; ++j) {
            const u64 cl = c_lo[j];
            const u64 ch = KW == 2 ? c_hi[j] : 0ull;
#pragma unroll
            for (int e = 0; e < KMC_FIN_ROUND; ++e)
                r[e] += (KW == 2 ? (ch < khi[e] || (ch == khi[e] && cl < klo[e])) : cl < klo[e]) ? 1u : 0u;
        }
#pragma unroll
        for (int e = 0; e < KMC_FIN_ROUND; ++e) {
            const u64 i = ((u64)rd * KMC_FIN_ROUND + e) * 1024 + tid;
            if (i < n && r[e]) atomicAdd(&rank[i], r[e]);
        }
    }
    // every add of this workgroup has been performed before its ticket is drawn
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        __threadfence();
        const u32 t = atomicAdd(&rank[KMC_OCC_LIST_CAP], 1u);
        s_last = (t == nb - 1) ? 1u : 0u;
        if (s_last) __threadfence();
    }
    __syncthreads();
    if (!s_last) return;
    // ---- the last workgroup: scatter to sorted order ----
    u64 sum = 0;
    for (u64 i = tid; i < n; i += 1024) {
        const u64 slot = g.occ_list[i];
        const u32 pos = __hip_atomic_load(&rank[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        rank[i] = 0;
        const u64 c = g.count[slot];
        out_lo[pos] = g.key_lo[slot];
        if (KW == 2) out_hi[pos] = g.key_hi[slot];
        out_cnt[pos] = c;
        sum += c;
    }
    sum = wave_sum_u64(sum);
    if ((tid & 63) == 0) s_sum[tid >> 6] = sum;
    __syncthreads();
    if (tid == 0) {
        u64 tot = 0;
        for (int w = 0; w < 16; ++w) tot += s_sum[w];
        g.counters[KMC_CTR_SUM2] = tot;
        g.counters[KMC_CTR_FASTFIN] = 1;
        rank[KMC_OCC_LIST_CAP] = 0;
    }
}
