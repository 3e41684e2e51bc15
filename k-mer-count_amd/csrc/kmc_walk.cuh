// kmc_walk.cuh -- KMC_ALGO_WALK: memoised successor walk for batches of short reads.
//
// Replaces the reference's window loop + grouping (k-mer-count/src/main.rs:63-87) for
// contiguous k, exactly (same table as KMC_ALGO_STREAM and the CPU oracle), with ONE LDS lookup
// per 8 bases instead of one hash-table update per k-mer.
//
// Idea.  Counting k-mers of a read is a walk in the de Bruijn graph of the input.  Each
// workgroup keeps, in LDS, a memo of that walk at a stride of 8 bases:
//   node  = a context: the last k bases (a k-mer), or -- within the first k bases of a read --
//           the whole prefix read so far (depths 0, 8, 16, ...).  Reads start at ROOT.
//   edge  = (node, next <=8 bases) -> successor node, plus a 32-bit traversal counter.
// A lane owns one read and advances 8 bases per step: key = node|label, one ds_read_b64 of the
// edge entry, one ds_add on its counter, successor comes back with the entry.  No k-mer is
// formed, hashed or compared on this path.  The first time an edge is seen (slow path) the
// successor context is built from the node's stored key and inserted with LDS CAS.  At the end
// of the kernel every edge is unfolded once: its <=8 k-mers (those at depth >= k) each receive
// the edge's counter in the global table (canonical strand chosen there, once per distinct
// k-mer instead of once per occurrence).  Counting is additive, so the result is bit-identical
// to per-occurrence counting.  When a memo table is full the lane counts its k-mers directly
// (global atomics) -- always exact, just slower; KMC_ALGO_AUTO then prefers the stream kernel.
//
// Data movement.  A wave owns 64 consecutive reads = one contiguous byte range of the batch.
// It streams that range with fully coalesced 16-byte loads, packs every 16 ASCII bases into one
// 2-bit word in registers and parks the words in its private LDS staging area (4x smaller than
// the ASCII); each lane then reads its own read back 16 bases at a time, re-aligned with
// v_alignbit.  HBM traffic is the algorithmic minimum: every base byte and offset once.
// Reads that contain a non-ACGT byte are diverted to a scalar kernel (kmc_scalar_reads_kernel).
#pragma once
#include "../../include/kmc.h"
#include "kmc_device.cuh"

#define KMC_WALK_MAX_K 31
#define KMC_WALK_MAX_READ 416
#define KMC_WALK_WAVES 16
#define KMC_WALK_THREADS (KMC_WALK_WAVES * 64)
#define KMC_WALK_STAGE_WORDS (64 * KMC_WALK_MAX_READ / 16 + 4)
#define KMC_WALK_ELOG 11
#define KMC_WALK_ECAP (1 << KMC_WALK_ELOG)
#define KMC_WALK_NLOG 11
#define KMC_WALK_NCAP (1 << KMC_WALK_NLOG)
#define KMC_WALK_DIRECT_ID 8190u  // node-id field of a lane that counts directly (never allocated)
#define KMC_WALK_EMPTY_KEY 0xFFFFFFFFu
#define KMC_WALK_BADWORDS 64

struct WalkEdge {
    u64 kv;   // low 32: key = label(16) | (len-1)<<16 | node<<19 ; high 32: successor, pre-shifted (node<<19 | 7<<16)
    u32 cnt;  // traversals
    u32 pad;
};

struct WalkLds {
    u32 stage[KMC_WALK_WAVES][KMC_WALK_STAGE_WORDS];
    WalkEdge edge[KMC_WALK_ECAP];
    u64 nkeys[KMC_WALK_NCAP];
    u32 badbits[KMC_WALK_WAVES][KMC_WALK_BADWORDS];
    u32 nedges, nnodes;
};

// workspace header (device): [0] number of deferred reads; the u32 read indices follow at +64 B
struct WalkWs {
    unsigned long long n_deferred;
    unsigned long long pad[7];
};

// node keys: k-mer nodes hold the 2k-bit context (top bits clear); prefix nodes (depth < k, read
// start) hold  1<<63 | depth<<56 | 2*depth bits
#define KMC_NODE_PREFIX (1ull << 63)
__device__ __forceinline__ u64 node_encode(u64 ctx, u32 depth, int k, u64 mask) {
    return depth >= (u32)k ? (ctx & mask) : (KMC_NODE_PREFIX | ((u64)depth << 56) | ctx);
}
__device__ __forceinline__ void node_decode(u64 nk, int k, u64& ctx, u32& depth) {
    if (nk >> 63) { depth = (u32)(nk >> 56) & 0x7Fu; ctx = nk & ((1ull << 56) - 1); }
    else { depth = (u32)k; ctx = nk; }
}

// canonical (or forward) key of a k-mer context -> global table
template <bool CANON>
__device__ __forceinline__ void walk_gadd(const GTable& g, u64 kmer, int k, u64 cnt) {
    u64 key = kmer;
    if (CANON) {
        u64 rhi, rlo;
        revcomp_key(0ull, kmer, k, rhi, rlo);
        if (rlo < key) key = rlo;
    }
    gtable_add<1>(g, 0ull, key, cnt);
}

// find-or-insert a node key; returns its id or 0xFFFFFFFF when the node table is full
__device__ __forceinline__ u32 walk_node(WalkLds& L, u64 nk) {
    u32 h = (u32)(kmc_mix64(nk) >> (64 - KMC_WALK_NLOG));
    u32 res = 0xFFFFFFFFu;
    bool done = false;
    int probes = 0;
    while (__builtin_amdgcn_ballot_w64(!done) != 0) {
        if (!done) {
            u64 cur = __hip_atomic_load(&L.nkeys[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (cur == KMC_EMPTY64) {
                if (__hip_atomic_load(&L.nnodes, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >= (u32)(KMC_WALK_NCAP * 3 / 4)) {
                    done = true;  // full
                } else {
                    cur = atomicCAS((unsigned long long*)&L.nkeys[h], KMC_EMPTY64, nk);
                    if (cur == KMC_EMPTY64) { atomicAdd(&L.nnodes, 1u); cur = nk; }
                }
            }
            if (!done) {
                if (cur == nk) { res = h; done = true; }
                else { h = (h + 1) & (KMC_WALK_NCAP - 1); if (++probes > 64) done = true; }
            }
        }
    }
    return res;
}

// Slow path of one step: the edge (key) was not found at its home slot.
// Returns the lane's next state word.  dctx/ddepth: context of a lane in direct mode.
template <bool CANON>
__device__ __forceinline__ u32 walk_slow(WalkLds& L, const GTable& g, u32 key, u32 h, int len, int k, u64 mask,
                                      u64& dctx, u32& ddepth, u64& ndirect) {
    const u32 label = key & 0xFFFFu;
    const u32 s = key >> 19;
    if (s == KMC_WALK_DIRECT_ID) {
        // direct mode: roll the context and count every k-mer with a global atomic
        for (int t = 0; t < len; ++t) {
            u32 c = (label >> (2 * t)) & 3u;
            dctx = ((dctx << 2) | c) & mask;
            if (ddepth < (u32)k) ddepth++;
            if (ddepth >= (u32)k) { walk_gadd<CANON>(g, dctx, k, 1); ndirect++; }
        }
        return (KMC_WALK_DIRECT_ID << 19) | (7u << 16);
    }
    // 1. probe for the edge / an empty slot
    u32 hh = h;
    bool done = false, found = false, have_val = false;
    u32 val = 0;
    u64 sctx = 0;
    u32 sdepth = 0;
    int probes = 0;
    while (__builtin_amdgcn_ballot_w64(!done) != 0) {
        if (!done) {
            u64 kv = __hip_atomic_load(&L.edge[hh].kv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if ((u32)kv == key) {
                atomicAdd(&L.edge[hh].cnt, 1u);
                val = (u32)(kv >> 32);
                found = true;
                done = true;
            } else if ((u32)kv == KMC_WALK_EMPTY_KEY) {
                if (!have_val) {
                    // successor context = node context extended by the label
                    node_decode(__hip_atomic_load(&L.nkeys[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP), k, sctx, sdepth);
                    for (int t = 0; t < len; ++t) {
                        u32 c = (label >> (2 * t)) & 3u;
                        sctx = (sctx << 2) | c;
                        if (sdepth < (u32)k) sdepth++;
                        if (sdepth >= (u32)k) sctx &= mask;
                    }
                    if (len == 8) {
                        u32 id = walk_node(L, node_encode(sctx, sdepth, k, mask));
                        val = id == 0xFFFFFFFFu ? 0xFFFFFFFFu : ((id << 19) | (7u << 16));
                    } else {
                        val = 0;  // a partial step ends the read: no successor needed
                    }
                    have_val = true;
                }
                if (val == 0xFFFFFFFFu || __hip_atomic_load(&L.nedges, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >= (u32)(KMC_WALK_ECAP * 3 / 4)) {
                    done = true;  // memo full
                } else {
                    u64 want = (u64)key | ((u64)val << 32);
                    u64 old = atomicCAS((unsigned long long*)&L.edge[hh].kv, ~0ull, want);
                    if (old == ~0ull) {
                        atomicAdd(&L.nedges, 1u);
                        atomicAdd(&L.edge[hh].cnt, 1u);
                        found = true;
                        done = true;
                    }
                    // else: somebody filled this slot; examine it again next trip
                }
            } else {
                hh = (hh + 1) & (KMC_WALK_ECAP - 1);
                if (++probes > 32) done = true;  // memo (locally) full
            }
        }
    }
    if (found) return val;
    // 2. memo full: count this step's k-mers directly from the node's context
    u64 ctx;
    u32 depth;
    node_decode(__hip_atomic_load(&L.nkeys[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP), k, ctx, depth);
    for (int t = 0; t < len; ++t) {
        u32 c = (label >> (2 * t)) & 3u;
        ctx = (ctx << 2) | c;
        if (depth < (u32)k) depth++;
        if (depth >= (u32)k) { ctx &= mask; walk_gadd<CANON>(g, ctx, k, 1); ndirect++; }
    }
    if (len != 8) return 0;
    if (!have_val) {
        u32 id = walk_node(L, node_encode(ctx, depth, k, mask));
        val = id == 0xFFFFFFFFu ? 0xFFFFFFFFu : ((id << 19) | (7u << 16));
    }
    if (val != 0xFFFFFFFFu) return val;  // successor node exists: stay on the memoised path
    dctx = ctx;
    ddepth = depth;
    return (KMC_WALK_DIRECT_ID << 19) | (7u << 16);
}

template <bool CANON>
__device__ __forceinline__ u32 walk_step(WalkLds& L, const GTable& g, u32 sk, u32 label, int len, int k, u64 mask,
                                         u64& dctx, u32& ddepth, u64& ndirect) {
    const u32 key = (len == 8) ? (sk | label) : ((sk & ~(7u << 16)) | ((u32)(len - 1) << 16) | (label & ((1u << (2 * len)) - 1u)));
    const u32 h = (key * 0x9E3779B1u) >> (32 - KMC_WALK_ELOG);
    const u64 kv = __hip_atomic_load(&L.edge[h].kv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if ((u32)kv == key) {
        atomicAdd(&L.edge[h].cnt, 1u);
        return (u32)(kv >> 32);
    }
    return walk_slow<CANON>(L, g, key, h, len, k, mask, dctx, ddepth, ndirect);
}

template <bool CANON>
__global__ __launch_bounds__(KMC_WALK_THREADS)
void kmc_walk_kernel(const uint8_t* __restrict__ bases, u64 n_bases, const u64* __restrict__ offsets, u64 n_reads,
                     int k, WalkWs* ws, u32* deferred, GTable g) {
    extern __shared__ __align__(16) unsigned char walk_smem[];
    WalkLds& L = *reinterpret_cast<WalkLds*>(walk_smem);
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const u64 mask = (1ull << (2 * k)) - 1;

    for (int i = tid; i < KMC_WALK_ECAP; i += KMC_WALK_THREADS) { L.edge[i].kv = ~0ull; L.edge[i].cnt = 0; }
    for (int i = tid; i < KMC_WALK_NCAP; i += KMC_WALK_THREADS) L.nkeys[i] = KMC_EMPTY64;
    if (tid == 0) { L.nedges = 0; L.nnodes = 1; }
    __syncthreads();
    const u64 root_key = KMC_NODE_PREFIX;  // prefix node of depth 0
    const u32 root_id = (u32)(kmc_mix64(root_key) >> (64 - KMC_WALK_NLOG));
    if (tid == 0) L.nkeys[root_id] = root_key;
    __syncthreads();
    const u32 sk_root = (root_id << 19) | (7u << 16);

    u32* stage = L.stage[wv];
    u64 nk = 0, ndirect = 0;
    const u64 n_tiles = (n_reads + 63) / 64;
    const u64 gw = (u64)blockIdx.x * KMC_WALK_WAVES + wv;
    const u64 total_waves = (u64)gridDim.x * KMC_WALK_WAVES;

    for (u64 tile = gw; tile < n_tiles; tile += total_waves) {
        const u64 r = tile * 64 + lane;
        const bool have = r < n_reads;
        const u64 a = offsets[have ? r : n_reads];
        const u64 e = offsets[have ? r + 1 : n_reads];
        const u64 A = __shfl(a, 0);
        const u64 B = __shfl(e, 63);  // lanes past the last read hold offsets[n_reads] twice
        const u64 A16 = A & ~15ull;
        const u32 n_pieces = (u32)((B - A16 + 15) >> 4);

        // ---- load phase: coalesced 16 B pieces -> 2-bit words in this wave's staging area ----
        if (lane < KMC_WALK_BADWORDS) L.badbits[wv][lane] = 0;
        bool anybad = false;
        for (u32 p0 = 0; p0 < n_pieces; p0 += 256) {
            uint4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                u32 p = p0 + 64 * u + lane;
                u64 pos = A16 + 16ull * p;
                v[u] = make_uint4(0x41414141u, 0x41414141u, 0x41414141u, 0x41414141u);
                if (p < n_pieces && pos < n_bases) v[u] = *reinterpret_cast<const uint4*>(bases + pos);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                u32 p = p0 + 64 * u + lane;
                if (p < n_pieces) {
                    Enc16 en = encode16(v[u]);
                    stage[p] = en.wle;
                    if ((en.x0 | en.x1 | en.x2 | en.x3) != 0) {
                        // exact check, ignoring bytes outside this wave's range [A, B)
                        u32 bad = bad16_from(en);
                        u64 pos = A16 + 16ull * p;
                        if (pos < A) bad &= ~((1u << (u32)(A - pos)) - 1u);
                        if (pos + 16 > B) bad &= (B > pos) ? ((1u << (u32)(B - pos)) - 1u) : 0u;
                        if (bad) { atomicOr(&L.badbits[wv][p >> 5], 1u << (p & 31)); anybad = true; }
                    }
                }
            }
        }
        if (lane == 0) stage[n_pieces] = 0;  // the re-alignment reads one word past the last piece
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

        // ---- per-lane read ----
        u32 len_read = (u32)(e - a);
        bool mine = have && len_read > 0;
        if (__builtin_amdgcn_ballot_w64(anybad) != 0 && mine) {
            // does any piece of my read carry a non-ACGT byte?  (conservative at shared pieces)
            u32 pa = (u32)((a - A16) >> 4), pe = (u32)((e - 1 - A16) >> 4);
            bool hit = false;
            for (u32 w = pa >> 5; w <= (pe >> 5); ++w) {
                u32 bits = __hip_atomic_load(&L.badbits[wv][w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                u32 lo_b = (w == (pa >> 5)) ? (pa & 31) : 0, hi_b = (w == (pe >> 5)) ? (pe & 31) : 31;
                u32 m = (hi_b == 31 ? ~0u : ((1u << (hi_b + 1)) - 1u)) & ~((1u << lo_b) - 1u);
                hit |= (bits & m) != 0;
            }
            if (hit) {
                u64 idx = atomicAdd(&ws->n_deferred, 1ull);
                deferred[idx] = (u32)r;
                mine = false;
            }
        }
        if (mine && len_read >= (u32)k) nk += len_read - (u32)k + 1;

        const u32 rel = (u32)(a - A16);
        const u32 w0 = rel >> 4, sh = 2 * (rel & 15);
        const u32 nfull = mine ? (len_read >> 3) : 0;
        const u32 tail = mine ? (len_read & 7) : 0;
        u32 sk = sk_root;
        u64 dctx = 0;
        u32 ddepth = 0;
        const u32 nsteps = nfull + (tail ? 1u : 0u);
        u32 wc = mine ? stage[w0] : 0, cw = 0;
        for (u32 t = 0; t < KMC_WALK_MAX_READ / 8 + 1; ++t) {
            const bool act = t < nsteps;
            if (__builtin_amdgcn_ballot_w64(act) == 0) break;
            if ((t & 1) == 0) {  // every second step: next 16 bases of my read, re-aligned to its start
                u32 wn = act ? stage[w0 + (t >> 1) + 1] : 0;
                cw = alignbit(wn, wc, sh);
                wc = wn;
            }
            const u32 label = (t & 1) ? (cw >> 16) : (cw & 0xFFFFu);
            if (act) sk = walk_step<CANON>(L, g, sk, label, t < nfull ? 8 : (int)tail, k, mask, dctx, ddepth, ndirect);
        }
        __builtin_amdgcn_wave_barrier();  // staging area is reused by the next tile
    }
    nk = wave_sum_u64(nk);
    ndirect = wave_sum_u64(ndirect);
    if (lane == 0) {
        if (nk) atomicAdd((unsigned long long*)&g.counters[KMC_CTR_KMERS], nk);
        if (ndirect) atomicAdd((unsigned long long*)&g.counters[KMC_CTR_BADBASE], ndirect);
    }

    // ---- flush: unfold every memoised edge into its k-mers ----
    __syncthreads();
    for (int i = tid; i < KMC_WALK_ECAP; i += KMC_WALK_THREADS) {
        const u64 kv = L.edge[i].kv;
        const u32 key = (u32)kv, cnt = L.edge[i].cnt;
        if (key != KMC_WALK_EMPTY_KEY && cnt) {
            const u32 label = key & 0xFFFFu, len = ((key >> 16) & 7u) + 1, s = key >> 19;
            u64 ctx;
            u32 depth;
            node_decode(L.nkeys[s], k, ctx, depth);
            for (u32 t = 0; t < len; ++t) {
                u32 c = (label >> (2 * t)) & 3u;
                ctx = (ctx << 2) | c;
                if (depth < (u32)k) depth++;
                if (depth >= (u32)k) { ctx &= mask; walk_gadd<CANON>(g, ctx, k, cnt); }
            }
        }
    }
}

// One lane per listed read, byte by byte: reads diverted from the walk kernel (non-ACGT bytes).
template <int KW, bool CANON>
__global__ void kmc_scalar_reads_kernel(const uint8_t* __restrict__ bases, const u64* __restrict__ offsets,
                                        const WalkWs* ws, const u32* __restrict__ list, int k, GTable g) {
    const u64 n = ws->n_deferred;
    const int kb = 2 * k;
    const u64 mask_lo = kb >= 64 ? ~0ull : ((1ull << kb) - 1);
    const u64 mask_hi = kb <= 64 ? 0ull : ((1ull << (kb - 64)) - 1);
    u64 nk = 0;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) {
        const u64 r = list[i];
        u64 lo = 0, hi = 0;
        int run = 0;
        for (u64 p = offsets[r]; p < offsets[r + 1]; ++p) {
            const uint8_t b = bases[p];
            int c = b == 'A' ? 0 : b == 'C' ? 1 : b == 'G' ? 2 : b == 'T' ? 3 : -1;
            if (c < 0) { run = 0; lo = hi = 0; continue; }
            hi = ((hi << 2) | (lo >> 62)) & mask_hi;
            lo = ((lo << 2) | (u64)c) & mask_lo;
            if (++run >= k) {
                u64 khi = hi, klo = lo;
                if (CANON) {
                    u64 rhi, rlo;
                    revcomp_key(hi, lo, k, rhi, rlo);
                    if (key_less(rhi, rlo, hi, lo)) { khi = rhi; klo = rlo; }
                }
                gtable_add<KW>(g, khi, klo, 1);
                nk++;
            }
        }
    }
    nk = wave_sum_u64(nk);
    if ((threadIdx.x & 63) == 0 && nk) atomicAdd((unsigned long long*)&g.counters[KMC_CTR_KMERS], nk);
}

// ---- host side ------------------------------------------------------------------------------
static inline bool kmc_walk_supported(int k, int mode, u64 max_read_len) {
    return mode == KMC_MODE_CONTIG && k >= 1 && k <= KMC_WALK_MAX_K && max_read_len >= 1 && max_read_len <= KMC_WALK_MAX_READ;
}
static inline size_t kmc_walk_workspace_bytes(u64 n_reads) { return sizeof(WalkWs) + (size_t)(n_reads + 16) * sizeof(u32); }

static inline int kmc_walk_launch(hipStream_t st, int n_cu, int KW, int k, bool canon, const uint8_t* d_bases,
                                  const u64* d_offsets, u64 n_reads, u64 n_bases, void* ws, GTable g) {
    if (KW != 1 || n_reads >= (1ull << 32)) return KMC_ERR_ARG;
    WalkWs* hdr = (WalkWs*)ws;
    u32* list = (u32*)((char*)ws + sizeof(WalkWs));
    if (hipMemsetAsync(hdr, 0, sizeof(WalkWs), st) != hipSuccess) return KMC_ERR_HIP;
    const u64 n_tiles = (n_reads + 63) / 64;
    u64 want = (n_tiles + KMC_WALK_WAVES - 1) / KMC_WALK_WAVES;
    int grid = (int)(want < (u64)n_cu ? want : (u64)n_cu);  // one 160 KB workgroup per CU is resident
    if (grid < 1) grid = 1;
    const size_t smem = sizeof(WalkLds);
    if (canon) {
        static bool attr1 = false;
        if (!attr1) { (void)hipFuncSetAttribute((const void*)kmc_walk_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); attr1 = true; }
        hipLaunchKernelGGL(kmc_walk_kernel<true>, dim3(grid), dim3(KMC_WALK_THREADS), smem, st, d_bases, n_bases, d_offsets, n_reads, k, hdr, list, g);
        hipLaunchKernelGGL((kmc_scalar_reads_kernel<1, true>), dim3(n_cu), dim3(256), 0, st, d_bases, d_offsets, hdr, list, k, g);
    } else {
        static bool attr0 = false;
        if (!attr0) { (void)hipFuncSetAttribute((const void*)kmc_walk_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); attr0 = true; }
        hipLaunchKernelGGL(kmc_walk_kernel<false>, dim3(grid), dim3(KMC_WALK_THREADS), smem, st, d_bases, n_bases, d_offsets, n_reads, k, hdr, list, g);
        hipLaunchKernelGGL((kmc_scalar_reads_kernel<1, false>), dim3(n_cu), dim3(256), 0, st, d_bases, d_offsets, hdr, list, k, g);
    }
    return hipGetLastError() == hipSuccess ? KMC_OK : KMC_ERR_HIP;
}

static inline int kmc_lr_launch(hipStream_t, int, const uint8_t*, const u64*, u64, u64, GTable) { return KMC_ERR_ARG; }
