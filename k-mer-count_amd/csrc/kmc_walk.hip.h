// kmc_walk.hip.h -- KMC_ALGO_WALK: memoised successor walk for batches of short reads.
//
// Replaces the reference's window loop + grouping (k-mer-count/src/main.rs:63-87) for
// contiguous k, exactly (same table as KMC_ALGO_STREAM and the CPU oracle), with ONE LDS lookup
// per 16 bases instead of one hash-table update per k-mer.
//
// Idea.  Counting k-mers of a read is a walk in the de Bruijn graph of the input.  Each
// workgroup keeps, in LDS, a memo of that walk at a stride of 16 bases:
//   node  = a context: the last k bases (a k-mer), or -- within the first k bases of a read --
//           the whole prefix read so far (depths 0, 16, ...).  Reads start at ROOT.
//   edge  = (node, next <=16 bases) -> successor node, plus a 32-bit traversal counter.  The
//           first continuation seen after a node is its PRIMARY edge and lives inside the node
//           entry (direct-indexed, no hashing); other continuations go to a hashed edge table.
// A lane owns one read and advances 16 bases per step: one ds_read_b64 of the node's
// {label, successor}, one compare with the read's next 16 bases, one ds_add on the counter.  No
// k-mer is formed, hashed or compared on this path.  The first time an edge is seen (slow path) the
// successor context is built from the node's stored key and inserted with LDS CAS.  At the end
// of the kernel every edge is unfolded once: its <=16 k-mers (those at depth >= k) each receive
// the edge's counter in the global table (canonical strand chosen there, once per distinct
// k-mer instead of once per occurrence).  Counting is additive, so the result is bit-identical
// to per-occurrence counting.  When a memo table is full the lane counts its k-mers directly
// (global atomics) -- always exact, just slower; KMC_ALGO_AUTO then prefers the stream kernel.
//
// Data movement.  A tile is 64 consecutive reads = one contiguous byte range of the batch.  A wave
// streams its tile with fully coalesced 16-byte loads (two rounds of 5 KiB in flight, one stream of
// rounds across tiles), packs every 16 ASCII bases into one 2-bit word in registers and parks the
// words in its private LDS staging area (4x smaller than the ASCII); each lane then reads its own
// read back 16 bases at a time, re-aligned with v_alignbit.  HBM traffic is the algorithmic
// minimum: every base byte and offset once.  The waves of a workgroup DRAW their tiles from a
// counter in LDS (the SIMD arbiter favours older waves; with fixed shares half of them idled for
// the last third of the kernel).  Reads longer than KMC_WALK_MAX_READ are walked as pieces that
// overlap by k-1 bases (kmc_vreads_*).  Reads that contain a non-ACGT byte are diverted to a scalar
// kernel (the scalar part of kmc_walk_tail_kernel).  Measured: 1.33 ms for 8.95 G bases = 6.8 TB/s, 85 % of the HBM
// peak.
#pragma once
#include "../../include/kmc.h"
#include "kmc_device.hip.h"
#include <hip/hip_ext.h>

#define KMC_WALK_MAX_K 63
#define KMC_WALK_MAX_READ 416
#define KMC_WALK_WAVES 16
#define KMC_WALK_THREADS (KMC_WALK_WAVES * 64)
#define KMC_WALK_STAGE_WORDS (64 * KMC_WALK_MAX_READ / 16 + 4)
#define KMC_WALK_STRIDE 16
#define KMC_WALK_ELOG 10
#define KMC_WALK_ECAP (1 << KMC_WALK_ELOG)
#define KMC_WALK_NLOG 10
#define KMC_WALK_NCAP (1 << KMC_WALK_NLOG)
#define KMC_WALK_BADWORDS 64
#ifndef KMC_WALK_LPR
// 16-byte loads per lane and round (two rounds = 5-10 KiB in flight per wave).  Same-box A/B on the
// benchmark batch (10 GB FASTA, k=31): 3 -> 1.53-1.55 ms, 4 -> 1.48-1.53, 5 -> 1.44-1.48, 6 -> 1.46-1.48
// (a 64 x 400-base tile is exactly 5 rounds of 5).
#define KMC_WALK_LPR 5
#endif

// A node and its PRIMARY out-edge (the first 16-base continuation seen after this context),
// direct-indexed by node id: the hot step is one ds_read_b64 of {plabel, psucc} + one ds_add.
struct WalkNode {
    u64 prim;  // low 32: label = 16 bases, 2 bits each, first base in the low bits (internal code
               // A0 C1 T2 G3); high 32: byte offset (id*16) of the successor node, 0 = no primary yet
    u32 cnt[2];  // traversals of the primary edge: even lanes count in [0], odd lanes in [1].  All lanes of
                 // a wave are at the same phase of their reads, i.e. on the same few nodes, and same-address
                 // LDS atomics of one instruction serialise (64 % of the LDS cycles were conflict cycles with
                 // one counter); the second counter lives in what was padding
};
// Secondary edges (other continuations, and the <16-base step that ends a read): hashed.
// kv = label(32) | (len-1)<<32 (4 bits) | node<<36 (11 bits) | succ<<47 (11 bits); ~0 = empty
struct WalkEdge {
    u64 kv;
    u32 cnt[2];  // as WalkNode::cnt
};
#define KMC_EDGE_KEYMASK ((1ull << 47) - 1)

template <int KW>
struct WalkLds {
    u32 stage[KMC_WALK_WAVES][KMC_WALK_STAGE_WORDS];
    WalkNode node[KMC_WALK_NCAP];  // node 0 is never allocated: state 0 = "count directly"
    u64 nkeys[KMC_WALK_NCAP];      // node key, low word
    u64 nkeys_hi[KW == 2 ? KMC_WALK_NCAP : 1];  // high word (k >= 32): claimed EMPTY->LOCKED->value
    WalkEdge edge[KMC_WALK_ECAP];
    u32 badbits[KMC_WALK_WAVES][KMC_WALK_BADWORDS];
    u32 nedges, nnodes;
    u32 qnext;  // next unclaimed tile of this workgroup (waves draw their tiles from it)
    u32 logn;   // records this workgroup has put into its span of the (k+16)-mer log (SkLog)
    u32 nsk;    // adds this workgroup has made to the (k+16)-mer table
};

// workspace (device): [WalkWs header | gcnt[NCAP+ECAP] dense snapshot counters | u32 deferred read
// indices]; header and counters are zero before every launch: cleared once by the host when the
// buffer is (re)allocated, afterwards by kmc_walk_tail_kernel, their last reader
struct WalkWs {
    unsigned long long n_deferred;
    unsigned int scalar_done;   // workgroups of the scalar part that have read n_deferred (kmc_walk_tail_kernel)
    unsigned int pad0;
    unsigned long long pad[6];
};
#define KMC_WALK_WS_PREFIX (sizeof(WalkWs) + (size_t)(KMC_WALK_NCAP + KMC_WALK_ECAP) * sizeof(u64))

// The memo (nodes, node keys, edges -- no counters) is input-independent graph structure, so it is
// kept across launches as ONE shared snapshot: workgroup 0 saves its tables at the end of a launch
// and EVERY workgroup of the next launch starts from that snapshot.  That removes the warm-up of the
// slow path, and -- because all workgroups then agree on the slot of every snapshot entry -- lets
// them reduce their per-slot traversal counters with dense, coalesced atomics into one small global
// array (gcnt) that kmc_walk_tail_kernel turns into k-mer counts ONCE, instead of every
// workgroup scattering ~4 k global adds for the same k-mers (the flush was 70 us of a 1.7 ms launch).
// Entries a workgroup discovers during the launch are not in the snapshot and take the scattered
// path.  Two snapshot slots alternate (read A / write B) so that nothing reads a slot being written.
template <int KW>
struct WalkMemoSlot {
    u64 tag;  // KMC_WALK_MEMO_TAG | k when valid
    u32 nedges, nnodes;
    u64 prim[KMC_WALK_NCAP];
    u64 nkeys[KMC_WALK_NCAP];
    u64 nkeys_hi[KW == 2 ? KMC_WALK_NCAP : 1];
    u64 ekv[KMC_WALK_ECAP];
};
#define KMC_WALK_MEMO_TAG 0x4B4D434D454D4F00ull

// A context: up to 63 bases, 2 bits each in the public coding A0 C1 G2 T3, newest base in the low
// bits of lo.  hi stays 0 when KW == 1 (k <= 31).
struct WCtx { u64 hi, lo; };

// node keys: k-mer nodes hold the 2k-bit context (top bits of the top word clear); prefix nodes
// (depth < k, read start) hold  1<<63 | depth<<56  in the top word plus their 2*depth bits.
// Top word == ~0 is EMPTY, ~0-1 is LOCKED (two-word keys while being published).
#define KMC_NODE_PREFIX (1ull << 63)
template <int KW>
__device__ __forceinline__ WCtx node_encode(WCtx c, u32 depth, int k, u64 mask_hi, u64 mask_lo) {
    WCtx n;
    if (depth >= (u32)k) { n.hi = c.hi & mask_hi; n.lo = c.lo & mask_lo; }
    else if (KW == 1) { n.hi = 0; n.lo = KMC_NODE_PREFIX | ((u64)depth << 56) | c.lo; }
    else { n.hi = KMC_NODE_PREFIX | ((u64)depth << 56) | c.hi; n.lo = c.lo; }
    return n;
}
template <int KW>
__device__ __forceinline__ void node_decode(WCtx nk, int k, WCtx& c, u32& depth) {
    const u64 top = KW == 1 ? nk.lo : nk.hi;
    if (top >> 63) {
        depth = (u32)(top >> 56) & 0x7Fu;
        if (KW == 1) { c.hi = 0; c.lo = nk.lo & ((1ull << 56) - 1); }
        else { c.hi = nk.hi & ((1ull << 56) - 1); c.lo = nk.lo; }
    } else {
        depth = (u32)k;
        c = nk;
    }
}
template <int KW>
__device__ __forceinline__ WCtx node_key_load(WalkLds<KW>& L, u32 id) {
    WCtx n;
    n.lo = __hip_atomic_load(&L.nkeys[id], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    n.hi = KW == 2 ? __hip_atomic_load(&L.nkeys_hi[id], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) : 0ull;
    return n;
}

// canonical (or forward) key of a k-mer context -> global table
template <int KW, bool CANON>
__device__ __forceinline__ void walk_gadd(const GTable& g, WCtx kmer, int k, u64 cnt) {
    u64 khi = kmer.hi, klo = kmer.lo;
    if (CANON) {
        u64 rhi, rlo;
        revcomp_key(kmer.hi, kmer.lo, k, rhi, rlo);
        if (key_less(rhi, rlo, khi, klo)) { khi = rhi; klo = rlo; }
    }
    gtable_add<KW>(g, khi, klo, cnt);
}

// Extend a context by `len` bases of a label (internal code -> public code: c ^ (c>>1)).
// COUNT: every k-mer completed on the way receives `cnt` in the global table.
template <int KW, bool CANON, bool COUNT>
__device__ __forceinline__ u64 walk_roll(const GTable& g, WCtx& ctx, u32& depth, u32 label, int len, int k,
                                         u64 mask_hi, u64 mask_lo, u64 cnt) {
    u64 n = 0;
    for (int t = 0; t < len; ++t) {
        u32 c = (label >> (2 * t)) & 3u;
        c ^= c >> 1;
        if (KW == 2) ctx.hi = (ctx.hi << 2) | (ctx.lo >> 62);
        ctx.lo = (ctx.lo << 2) | c;
        if (depth < (u32)k) depth++;
        if (depth >= (u32)k) {
            ctx.lo &= mask_lo;
            if (KW == 2) ctx.hi &= mask_hi;
            if (COUNT) { walk_gadd<KW, CANON>(g, ctx, k, cnt); n++; }
        }
    }
    return n;
}

// The same for a FULL step from a context that is already k bases deep, nothing counted: sixteen shift-and-or rounds are one
// 32-bit shift -- the step's bases, recoded and first base on top, are the new low word.  (Steps that fall off the memo take
// this once or twice each; with the per-base loop the walk kernel spent more time rolling contexts than logging them:
// 0.87 ms per GB at pool 50 against 0.14 on the memo's fast path.)
template <int KW>
__device__ __forceinline__ void walk_roll16(WCtx& ctx, u32 label, u64 mask_hi, u64 mask_lo) {
    const u32 pub = label ^ ((label >> 1) & 0x55555555u);  // A0 C1 T2 G3 -> A0 C1 G2 T3
    const u32 be = le_to_be(pub);
    if (KW == 2) ctx.hi = ((ctx.hi << 32) | (ctx.lo >> 32)) & mask_hi;
    ctx.lo = ((ctx.lo << 32) | be) & mask_lo;
}

// find-or-insert a node key; returns its id (>= 1) or 0 when the node table is full.
// Same wave-uniform loop shape as gtable_add (kmc_device.hip.h).
template <int KW>
__device__ __forceinline__ u32 walk_node(WalkLds<KW>& L, WCtx nk) {
    u32 h = (u32)(kmc_hash_key<KW>(nk.hi, nk.lo) >> (64 - KMC_WALK_NLOG));
    if (h == 0) h = 1;
    u32 res = 0;
    bool done = false;
    int probes = 0;
    u32 trips = 0;
    while (__builtin_amdgcn_ballot_w64(!done) != 0) {
        if (!done) {
            bool advance = false;
            const bool room = __hip_atomic_load(&L.nnodes, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < (u32)(KMC_WALK_NCAP * 3 / 4);
            if (++trips > (1u << 16)) {
                done = true;  // give up (treated as "table full"): every wave must drain
            } else if (KW == 1) {
                u64 cur = __hip_atomic_load(&L.nkeys[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (cur == KMC_EMPTY64) {
                    if (!room) {
                        done = true;  // full
                    } else {
                        cur = atomicCAS((unsigned long long*)&L.nkeys[h], KMC_EMPTY64, nk.lo);
                        if (cur == KMC_EMPTY64) { atomicAdd(&L.nnodes, 1u); cur = nk.lo; }
                    }
                }
                if (!done) {
                    if (cur == nk.lo) { res = h; done = true; }
                    else advance = true;
                }
            } else {
                u64 cur = __hip_atomic_load(&L.nkeys_hi[h], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (cur == KMC_EMPTY64) {
                    if (!room) {
                        done = true;  // full
                    } else {
                        u64 old = atomicCAS((unsigned long long*)&L.nkeys_hi[h], KMC_EMPTY64, KMC_LOCKED64);
                        if (old == KMC_EMPTY64) {
                            __hip_atomic_store(&L.nkeys[h], nk.lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            __hip_atomic_store(&L.nkeys_hi[h], nk.hi, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                            atomicAdd(&L.nnodes, 1u);
                            res = h;
                            done = true;
                        }
                        // lost the race: examine the slot again next trip
                    }
                } else if (cur == KMC_LOCKED64) {
                    // being published; examine it again next trip
                } else if (cur == nk.hi && __hip_atomic_load(&L.nkeys[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == nk.lo) {
                    res = h;
                    done = true;
                } else {
                    advance = true;
                }
            }
            if (advance) {
                h = (h + 1) & (KMC_WALK_NCAP - 1);
                if (h == 0) h = 1;
                // (a full table is only LOOKED UP: a key that is not within a few slots of its home is not
                // worth a long walk -- every step that falls off a full memo used to pay up to 64 probes here)
                if (++probes > (room ? 64 : 6)) done = true;
            }
        }
    }
    return res;
}

// ---- second level of the memo: the stride-16 "super-k-mer" table in global memory -----------------------
// When the LDS memo is full, a full step from a k-mer context used to be counted as 16 separate k-mers
// with 16 global atomics (and KMC_ALGO_AUTO gave the input up to the sort path as soon as 5 % of the
// k-mers went that way: a 100-400x cliff between a pool of 20 and a pool of 26 lines).  Such a step is
// fully described by ONE (k+16)-mer -- the context followed by the 16 bases of the step -- so it is now
// counted as one add into a second global table keyed by that (k+16)-mer (two key words for k <= 47,
// three above), one global atomic per 16 bases; kmc_sk_unfold_kernel later gives the count to each of the (k+16)-mer's
// last 16 k-mers, once per distinct (k+16)-mer instead of once per occurrence.  Additive, so exact.
// The table keeps its keys across launches (like the LDS memo) and is cleared by kmc_forget_source.
#define KMC_SK_MAX_K 47
// the (k+16)-mer of a step: context (2k bits, public code, newest base lowest) followed by the label
// (16 bases, internal code, first base in the low bits)
// count one traversal of the step (ctx, label) in the (k+16)-mer table
// Round 3: the steps that fall off the LDS memo are LOGGED instead -- a fire-and-forget store of the (k+16)-mer into the
// workgroup's own span of a log, position from a counter in LDS -- and counted after the launch by kmc_sklog.hip.h
// (partition the log by a hash of the record, count each part in an LDS table, unfold every distinct (k+16)-mer once).
// A hash-table update per step was what the plateau of pools 32..100 cost: 28 M scattered read-modify-writes per GB of
// reads at ~7 G/s (round 2: 5.6-6.6 ms per GB; the walk itself 1.05 ms with the steps only logged).  A workgroup whose
// span is full goes on with the table update (the table stays: first launch on a new source, overflow of the log).
struct SkLog {
    u64* rec;      // nullptr: no log; else spans of cap_wg records of `words` u64 each, workgroup b's span at b * cap_wg
    u32* count;    // count[b] = records workgroup b wrote
    u32 cap_wg;
    u32 words;     // 2: {lo, mid} (k <= 47); 3: {lo, mid, top} (k >= 48)
};
template <int KW>
__device__ __forceinline__ void sk_add(WalkLds<KW>& L, const GTable& sk, const SkLog& lg, WCtx ctx, u32 label) {
    const u32 pub = label ^ ((label >> 1) & 0x55555555u);  // A0 C1 T2 G3 -> A0 C1 G2 T3
    const u32 be = le_to_be(pub);                           // first base of the step in the top bits
    const u64 lo = (ctx.lo << 32) | be;
    const u64 mid = (ctx.hi << 32) | (ctx.lo >> 32);
    if (lg.rec) {
        const u32 idx = atomicAdd(&L.logn, 1u);
        if (idx < lg.cap_wg) {
            typedef unsigned long long u64x2_t __attribute__((ext_vector_type(2)));
            u64* const rp = lg.rec + ((size_t)blockIdx.x * lg.cap_wg + idx) * lg.words;
            if (lg.words == 2) *reinterpret_cast<u64x2_t*>(rp) = u64x2_t{lo, mid};
            else { rp[0] = lo; rp[1] = mid; rp[2] = ctx.hi >> 32; }   // (24-byte records: 8-byte aligned)
            return;
        }
    }
    atomicAdd(&L.nsk, 1u);   // (counts are pending in the table: kmc_sk_unfold_kernel has work)
    if (sk.key_mid) gtable_add<3>(sk, ctx.hi >> 32, lo, 1, mid);
    else gtable_add<2>(sk, mid, lo, 1);
}

// every (k+16)-mer with a count gives it to its last 16 k-mers; counts are cleared for the next launch
template <int KW, bool CANON>
__global__ __launch_bounds__(256)
void kmc_sk_unfold_kernel(GTable sk, int k, GTable g) {
    // the claimed slots are listed in sk.occ_list (every slot of the table has room in the list), so the
    // work is proportional to the (k+16)-mers the input really has -- none at all for the benchmark input
    const u64 n_occ = min(sk.counters[KMC_CTR_OCCUPIED], sk.occ_list_cap);
    const u64 n_spill = min(sk.counters[KMC_CTR_SPILL], sk.spill_cap);
    if ((n_occ == 0 && n_spill == 0) || sk.counters[KMC_CTR_KMERS] == 0) return;   // (no entries, or keys without pending counts)
    const int kb = 2 * k;
    const u64 mask_lo = kb >= 64 ? ~0ull : ((1ull << kb) - 1);
    const u64 mask_hi = kb <= 64 ? 0ull : ((1ull << (kb - 64)) - 1);
    // item = (entry, j): the k-mer that ends j bases before the end of the (k+16)-mer, j = 0..15
    for (u64 w = (u64)blockIdx.x * blockDim.x + threadIdx.x; w < n_occ * 16; w += (u64)gridDim.x * blockDim.x) {
        const u64 slot = sk.occ_list[w >> 4];
        const u32 j = (u32)w & 15u;
        const u64 c = sk.count[slot];
        if (c) {
            // (two words: {hi, lo}; three words: {top, hi = mid, lo})
            const u64 hi = sk.key_mid ? sk.key_mid[slot] : sk.key_hi[slot], lo = sk.key_lo[slot];
            const u64 top = sk.key_mid ? sk.key_hi[slot] : 0ull;
            const u32 sh = 2 * j;  // drop the last j bases
            WCtx km;
            km.lo = (sh ? ((lo >> sh) | (hi << (64 - sh))) : lo) & mask_lo;
            km.hi = KW == 2 ? ((sh ? ((hi >> sh) | (top << (64 - sh))) : hi) & mask_hi) : 0ull;
            walk_gadd<KW, CANON>(g, km, k, c);
        }
        // the 16 items of an entry sit in 16 consecutive lanes: all have read the count before lane j == 0 clears it
        __builtin_amdgcn_wave_barrier();
        if (c && j == 0) sk.count[slot] = 0;
    }
    // (k+16)-mers that found no slot within the probe budget wait in the table's spill area
    for (u64 w = (u64)blockIdx.x * blockDim.x + threadIdx.x; w < n_spill * 16; w += (u64)gridDim.x * blockDim.x) {
        const u64 e = w >> 4;
        const u32 sh = 2 * ((u32)w & 15u);
        const u64 hi = sk.spill_mid ? sk.spill_mid[e] : sk.spill_hi[e], lo = sk.spill_lo[e], c = sk.spill_cnt[e];
        const u64 top = sk.spill_mid ? sk.spill_hi[e] : 0ull;
        WCtx km;
        km.lo = (sh ? ((lo >> sh) | (hi << (64 - sh))) : lo) & mask_lo;
        km.hi = KW == 2 ? ((sh ? ((hi >> sh) | (top << (64 - sh))) : hi) & mask_hi) : 0ull;
        if (c) walk_gadd<KW, CANON>(g, km, k, c);
    }
}
// (the spill counter is cleared by a second, tiny launch: every workgroup of the unfold reads it)
__global__ void kmc_sk_spill_reset_kernel(GTable sk) {
    if (blockIdx.x == 0 && threadIdx.x == 0) { sk.counters[KMC_CTR_SPILL] = 0; sk.counters[KMC_CTR_KMERS] = 0; }   // (KMERS: adds pending, see kmc_walk_kernel)
}

// Slow path of one step from node offset `s` (s == 0: direct mode).  Returns the next state.
template <int KW, bool CANON>
__device__ __forceinline__ u32 walk_slow(WalkLds<KW>& L, const GTable& g, u32 s, u32 label, int len, int k,
                                         u64 mask_hi, u64 mask_lo, WCtx& dctx, u32& ddepth, u64& ndirect, u32 cpar, const GTable& sk, const SkLog& lg) {
    if (s == 0) {  // direct mode: no node of the LDS memo stands for this lane's context
        if (sk.key_lo && len == KMC_WALK_STRIDE && ddepth >= (u32)k) {
            // a full step from a k-mer context: one add of its (k+16)-mer (second-level memo, above)
            sk_add<KW>(L, sk, lg, dctx, label);
            walk_roll16<KW>(dctx, label, mask_hi, mask_lo);
        } else {
            ndirect += walk_roll<KW, CANON, true>(g, dctx, ddepth, label, len, k, mask_hi, mask_lo, 1);
        }
        return 0;
    }
    const u32 id = s >> 4;
    // successor context (needed to create an edge) -- built once, lazily
    bool have_succ = false;
    u32 sid = 0;  // successor node id, 0 = none / table full
    WCtx sctx = {0, 0};
    u32 sdepth = 0;
    auto make_succ = [&]() {
        node_decode<KW>(node_key_load<KW>(L, id), k, sctx, sdepth);
        if (len == KMC_WALK_STRIDE && sdepth >= (u32)k) walk_roll16<KW>(sctx, label, mask_hi, mask_lo);
        else (void)walk_roll<KW, CANON, false>(g, sctx, sdepth, label, len, k, mask_hi, mask_lo, 0);
        sid = (len == KMC_WALK_STRIDE) ? walk_node<KW>(L, node_encode<KW>(sctx, sdepth, k, mask_hi, mask_lo)) : 0;
        have_succ = true;
    };
    WalkNode* np = reinterpret_cast<WalkNode*>(reinterpret_cast<char*>(L.node) + s);
    u64* pp = &np->prim;
    if (len == KMC_WALK_STRIDE) {
        // 1. the node's primary edge
        u64 pe = __hip_atomic_load(pp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if ((u32)(pe >> 32) == 0) {
            make_succ();
            if (sid) {
                u64 want = (u64)label | ((u64)(sid << 4) << 32);
                u64 old = atomicCAS((unsigned long long*)pp, 0ull, want);
                pe = old == 0 ? want : old;
            }
        }
        if ((u32)pe == label && (u32)(pe >> 32) != 0) {
            atomicAdd(&np->cnt[cpar], 1u);
            return (u32)(pe >> 32);
        }
    }
    // 2. secondary edge: hashed
    const u64 key = (u64)label | ((u64)(len - 1) << 32) | ((u64)id << 36);
    u32 hh = (u32)(kmc_mix64(key) >> (64 - KMC_WALK_ELOG));
    bool done = false, found = false;
    u32 val = 0;
    int probes = 0;
    while (__builtin_amdgcn_ballot_w64(!done) != 0) {
        if (!done) {
            u64 kv = __hip_atomic_load(&L.edge[hh].kv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (((kv ^ key) & KMC_EDGE_KEYMASK) == 0 && kv != ~0ull) {
                atomicAdd(&L.edge[hh].cnt[cpar], 1u);
                val = (u32)(kv >> 47);
                found = true;
                done = true;
            } else if (kv == ~0ull) {
                if (!have_succ) make_succ();
                if ((len == KMC_WALK_STRIDE && sid == 0) ||
                    __hip_atomic_load(&L.nedges, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >= (u32)(KMC_WALK_ECAP * 3 / 4)) {
                    done = true;  // memo full
                } else {
                    u64 want = key | ((u64)sid << 47);
                    u64 old = atomicCAS((unsigned long long*)&L.edge[hh].kv, ~0ull, want);
                    if (old == ~0ull) {
                        atomicAdd(&L.nedges, 1u);
                        atomicAdd(&L.edge[hh].cnt[cpar], 1u);
                        val = sid;
                        found = true;
                        done = true;
                    }
                    // else: somebody filled this slot; examine it again next trip
                }
            } else {
                hh = (hh + 1) & (KMC_WALK_ECAP - 1);
                const bool eroom = __hip_atomic_load(&L.nedges, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < (u32)(KMC_WALK_ECAP * 3 / 4);
                if (++probes > (eroom ? 32 : 6)) done = true;  // memo (locally) full
            }
        }
    }
    if (found) return val << 4;  // (a <16-base step ends the read; its successor is unused)
    // 3. memo full: count this step from the node's context -- as ONE (k+16)-mer in the second-level table
    //    when it is a full step from a k-mer context, k-mer by k-mer otherwise
    WCtx ctx;
    u32 depth;
    node_decode<KW>(node_key_load<KW>(L, id), k, ctx, depth);
    if (sk.key_lo && len == KMC_WALK_STRIDE && depth >= (u32)k) {
        sk_add<KW>(L, sk, lg, ctx, label);
        walk_roll16<KW>(ctx, label, mask_hi, mask_lo);
    } else {
        ndirect += walk_roll<KW, CANON, true>(g, ctx, depth, label, len, k, mask_hi, mask_lo, 1);
    }
    if (len != KMC_WALK_STRIDE) return 0;
    if (!have_succ) make_succ();
    if (sid) return sid << 4;  // successor node exists: stay on the memoised path
    dctx = ctx;
    ddepth = depth;
    return 0;  // direct mode from here on
}

// 16 ASCII bases -> one 32-bit word, 2 bits per base, first base in the low bits, internal code
// A0 C1 T2 G3 = (byte>>1)&3 (no G/T fix-up on the hot path; walk_roll converts).  A byte is one of
// "ACGT" iff it equals LUT[(byte>>1)&3] (v_perm_b32 as a 4-entry byte LUT), so x != 0 marks exactly
// the other bytes.  The selector MUST be masked to 2 bits before the words are combined: bytes past
// the end of the batch are arbitrary, and a third selector bit would spill into a neighbouring --
// valid -- base when the fields are interleaved below (seen as a one-in-10^4 single-base
// corruption in the last piece of a batch on boxes whose fresh memory held garbage;
// tests/test_gpu_parity.py::test_garbage_after_the_batch_is_ignored).
__device__ __forceinline__ u32 walk_encode16(uint4 v, u32& x0, u32& x1, u32& x2, u32& x3) {
    const u32 s0 = (v.x >> 1) & 0x03030303u, s1 = (v.y >> 1) & 0x03030303u;
    const u32 s2 = (v.z >> 1) & 0x03030303u, s3 = (v.w >> 1) & 0x03030303u;
    x0 = v.x ^ __builtin_amdgcn_perm(0u, 0x47544341u, s0);
    x1 = v.y ^ __builtin_amdgcn_perm(0u, 0x47544341u, s1);
    x2 = v.z ^ __builtin_amdgcn_perm(0u, 0x47544341u, s2);
    x3 = v.w ^ __builtin_amdgcn_perm(0u, 0x47544341u, s3);
    // byte j of u = bases j, 4+j, 8+j, 12+j (2 bits each): a 4x4 transpose away from base order
    u32 u = s0 | (s1 << 2) | (s2 << 4) | (s3 << 6);
    u32 t = ((u >> 12) ^ u) & 0x0000F0F0u;
    u ^= t | (t << 12);
    t = ((u >> 6) ^ u) & 0x00CC00CCu;
    u ^= t | (t << 6);
    return u;
}

// Diagnostic build only (tools/build_variant.sh stamps "-DKMC_WALK_STAMPS", tools/walk_stamps.py): the 100 MHz wall clock at
// the kernel's milestones, per workgroup: [0] entry, [1] LDS initialised, [2 + w] wave w left its tile loop, [18] all
// waves there, [19] dense flush done, [20] end, [21] wave 0 has stepped its first tile.
#ifdef KMC_WALK_STAMPS
__device__ unsigned long long kmc_walk_stamps[256 * 24];
#define WALK_STAMP(i) do { kmc_walk_stamps[(blockIdx.x & 255u) * 24 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define WALK_STAMP(i) do { } while (0)
#endif
template <int KW, bool CANON>
__global__ __launch_bounds__(KMC_WALK_THREADS)
void kmc_walk_kernel(const uint8_t* __restrict__ bases, u64 n_bases, const u64* __restrict__ vstart, const u64* __restrict__ vend, u64 n_reads,
                     int k, u64 tile_begin, u64 tile_end, WalkWs* ws, u32* deferred, const WalkMemoSlot<KW>* memo,
                     WalkMemoSlot<KW>* memo_out, u64* gcnt, GTable g, GTable sk, SkLog lg) {
    extern __shared__ __align__(16) unsigned char walk_smem[];
    WalkLds<KW>& L = *reinterpret_cast<WalkLds<KW>*>(walk_smem);
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const u32 cpar = (u32)lane & 1u;  // which of the two traversal counters of a node / edge this lane uses
    const int kb = 2 * k;
    const u64 mask_lo = kb >= 64 ? ~0ull : ((1ull << kb) - 1);
    const u64 mask_hi = kb <= 64 ? 0ull : ((1ull << (kb - 64)) - 1);

    if (tid == 0) WALK_STAMP(0);
    const bool warm = memo && memo->tag == (KMC_WALK_MEMO_TAG | (u64)k);  // workgroup-uniform
    for (int i = tid; i < KMC_WALK_ECAP; i += KMC_WALK_THREADS) { L.edge[i].kv = warm ? memo->ekv[i] : ~0ull; L.edge[i].cnt[0] = 0; L.edge[i].cnt[1] = 0; }
    for (int i = tid; i < KMC_WALK_NCAP; i += KMC_WALK_THREADS) {
        L.nkeys[i] = warm ? memo->nkeys[i] : (KW == 1 ? KMC_EMPTY64 : 0ull);
        if (KW == 2) L.nkeys_hi[i] = warm ? memo->nkeys_hi[i] : KMC_EMPTY64;
        L.node[i].prim = warm ? memo->prim[i] : 0ull;
        L.node[i].cnt[0] = 0;
        L.node[i].cnt[1] = 0;
    }
    if (tid == 0) { L.nedges = warm ? memo->nedges : 0; L.nnodes = warm ? memo->nnodes : 1; L.qnext = KMC_WALK_WAVES; L.logn = 0; L.nsk = 0; }
    __syncthreads();
    const WCtx root_key = node_encode<KW>(WCtx{0, 0}, 0, k, mask_hi, mask_lo);  // prefix node of depth 0
    u32 root_id = (u32)(kmc_hash_key<KW>(root_key.hi, root_key.lo) >> (64 - KMC_WALK_NLOG));
    if (root_id == 0) root_id = 1;
    if (tid == 0 && !warm) { L.nkeys[root_id] = root_key.lo; if (KW == 2) L.nkeys_hi[root_id] = root_key.hi; }
    __syncthreads();
    const u32 s_root = root_id << 4;
    if (tid == 0) WALK_STAMP(1);

    u32* stage = L.stage[wv];
    u64 nk = 0, ndirect = 0;
    const u64 n_tiles = tile_end;  // this launch covers tiles [tile_begin, tile_end) of 64 reads
    // Workgroup b owns the tiles tile_begin + b + gridDim.x * j, j = 0, 1, ... (interleaved over the
    // workgroups, so every CU gets the same share whatever the batch size), and its waves DRAW them
    // from a counter in LDS instead of owning a fixed sixteenth each: the SIMD arbiter favours a
    // workgroup's older waves, which ran 86 tiles in the time the younger ones needed for 68 --
    // with fixed shares they went idle at 70 % of the kernel's run time and the CU finished on half
    // its waves, i.e. half its bytes in flight (in-kernel stamps: tile loops done at 1058 .. 1519 us).
    // (Workgroups still finish up to 50 us apart -- by XCD: the 32 workgroups of an XCD end within 5 us of
    // each other, the XCDs do not, and WHICH ones are slow changes from launch to launch.  Three attempts
    // to even that out did not pay: leaving the last 4 / 8 / 16 % of the tiles to pools shared by 8
    // workgroups (one per XCD), drawn with a global atomic at tile entry, made the kernel 2 / 2.5 / 5 %
    // SLOWER; drawing those tickets two tiles ahead so that nothing waits for the atomic made it 3 / 7 /
    // 12 % slower still -- vector memory returns in issue order, so a returning device-scope atomic
    // (microseconds under load) holds back every load the wave issued after it, however late its
    // result is read; per-XCD tile shares learned from the previous launch's loop times changed
    // nothing (1.319 vs 1.320 ms), the past launch does not predict the next.  Round 3: ONE work queue for the whole launch,
    // tiles dealt in chunks of 16, the ticket drawn two chunks ahead by the wave that takes a chunk's first tile and issued
    // right in front of its step phase, so that nothing ever waits for it -- exact, and 3 % SLOWER at 10 GB (1.393 vs
    // 1.352 ms) and 2 % slower at 1 GB: whatever spreads the XCDs' finish times, it is not a shortage of tiles.)
    // (Round 3, the end of a workgroup's queue: a wave draws its next tile at the entry of the one before, so whoever draws
    // the last ticket has up to two tiles to go while the others leave -- stamps (tools/walk_stamps.py) show the waves of a
    // workgroup leaving 27 us apart (median; up to 47) at 16 us a tile, a seventh of the kernel on the 1 GB batch of
    // BASELINE config 2.  Dealing the last 4 / 8 / 16 tiles of a workgroup in halves, quarters or eighths (16 .. 8 reads per
    // draw) brought that to 13 us -- and made the kernel SLOWER, 0.200-0.223 vs 0.196 ms at 1 GB and 1.35-1.37 vs 1.345 at
    // 10 GB: a tile's time is its chain of 25 dependent steps, which is as long for 16 reads as for 64.
    // profiles/r03_walk_tail_split_ab.txt.)
    auto tile_of = [&](u32 j) { return tile_begin + blockIdx.x + (u64)gridDim.x * j; };
    const u64 gw = tile_of((u32)wv);  // the first tile of every wave is fixed; the counter starts behind them

    // Per-tile geometry: the wave's 64 reads are the byte range [A, B) of the batch.
    // Geometry comes in two steps so that a tile's successor costs no stall: tile_load only ISSUES the
    // two offset loads (at the entry of the tile before), tile_finish turns them into the byte range
    // when that tile's last round is about to be issued, four rounds of loads later -- by then they
    // have long arrived.  (Done in one step at tile entry, the wave-wide shuffles forced
    // s_waitcnt vmcnt(0) there, twice: every tile began by draining the wave's loads in flight and
    // then sat through a second memory round trip with nothing in flight at all.)
    struct TileRaw { u64 a, e; u32 have; };
    struct TileGeo { u64 a, e, A, B, A16; u32 n_pieces; u32 have; };
    auto tile_load = [&](u64 tile) {
        TileRaw w;
        const u64 r = tile * 64 + lane;
        w.have = r < n_reads ? 1u : 0u;
        // read r is bases [vstart[r], vend[r]).  Short-read batches pass offsets and offsets + 1; batches
        // with reads longer than KMC_WALK_MAX_READ pass the pieces made by kmc_vreads_* (consecutive
        // pieces of a read overlap by k-1 bases; starts and ends are non-decreasing either way)
        const u64 idx = w.have ? r : n_reads - 1;
        w.e = vend[idx];
        w.a = vstart[idx];
        return w;
    };
    auto tile_finish = [&](const TileRaw& w) {
        TileGeo t;
        t.have = w.have;
        t.e = w.e;
        t.a = w.have ? w.a : w.e;  // lanes past the last read hold the last end twice
        t.A = __shfl(t.a, 0);
        t.B = __shfl(t.e, 63);
        t.A16 = t.A & ~15ull;
        t.n_pieces = (u32)((t.B - t.A16 + 15) >> 4);
        return t;
    };
    auto tile_geo = [&](u64 tile) { return tile_finish(tile_load(tile)); };
    // One round = KMC_WALK_LPR coalesced 1 KiB wave-loads: pieces base + 64*u + lane of the byte range
    // that starts at A16.  Branch-free on purpose (straight-line code lets the compiler keep counted
    // s_waitcnt vmcnt(N) and two rounds in flight): a piece index past the end is clamped to the
    // last piece -- that re-reads one cache line and is never stored.
    constexpr int LPR = KMC_WALK_LPR;
    constexpr u32 RP = 64u * LPR;  // pieces per round
    const u64 max_off = (n_bases - 1) & ~15ull;  // never touch a 16-byte piece that starts past the last base
    auto issue = [&](uint4 (&v)[LPR], u64 A16, u32 last, u32 base) {
        if (A16 > max_off) A16 = max_off;  // (a tile of empty reads at the very end of the batch)
#pragma unroll
        for (int u = 0; u < LPR; ++u) {
            u32 p = base + 64 * u + lane;
            p = p < last ? p : last;
            // streamed once: non-temporal (nt) loads measured +4 % over the default cache policy
            typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
            u32x4_t q = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(bases + A16 + 16ull * p));
            v[u] = make_uint4(q.x, q.y, q.z, q.w);
        }
    };
    bool anybad = false;
    // Encode one round into the staging area.  Hot path (every piece of the round exists, every
    // byte is ACGT): no per-lane branch at all -- the "is anything not ACGT" test is ONE ballot per
    // round; the exact per-piece check (which also ignores bytes outside this wave's range [A, B))
    // runs only when that ballot fires.
    auto consume = [&](const uint4 (&v)[LPR], const TileGeo& t, u32 base, u32 np_u) {
        u32 xacc = 0;
        if (base + RP <= np_u) {  // wave-uniform
#pragma unroll
            for (int u = 0; u < LPR; ++u) {
                u32 x0, x1, x2, x3;
                stage[base + 64 * u + lane] = walk_encode16(v[u], x0, x1, x2, x3);
                xacc |= x0 | x1 | x2 | x3;
            }
        } else {
#pragma unroll
            for (int u = 0; u < LPR; ++u) {
                const u32 p = base + 64 * u + lane;
                if (p < t.n_pieces) {
                    u32 x0, x1, x2, x3;
                    stage[p] = walk_encode16(v[u], x0, x1, x2, x3);
                    xacc |= x0 | x1 | x2 | x3;
                }
            }
        }
        if (__builtin_amdgcn_ballot_w64(xacc != 0) != 0) {
#pragma unroll
            for (int u = 0; u < LPR; ++u) {
                const u32 p = base + 64 * u + lane;
                if (p < t.n_pieces) {
                    u32 x0, x1, x2, x3;
                    (void)walk_encode16(v[u], x0, x1, x2, x3);
                    if ((x0 | x1 | x2 | x3) != 0) {
                        // exact check, ignoring bytes outside this wave's range [A, B)
                        u32 bad = nonzero_bytes4(x0) | (nonzero_bytes4(x1) << 4) | (nonzero_bytes4(x2) << 8) | (nonzero_bytes4(x3) << 12);
                        const u64 pos = t.A16 + 16ull * p;
                        if (pos < t.A) bad &= ~((1u << (u32)(t.A - pos)) - 1u);
                        if (pos + 16 > t.B) bad &= (t.B > pos) ? ((1u << (u32)(t.B - pos)) - 1u) : 0u;
                        if (bad) { atomicOr(&L.badbits[wv][p >> 5], 1u << (p & 31)); anybad = true; }
                    }
                }
            }
        }
    };

    // ---- step phase of one tile: every lane walks its own read through the memo ----
    auto step_phase = [&](const TileGeo& t, u64 tile) {
        const u64 r = tile * 64 + lane;
        const bool have = t.have != 0;
        const u64 a = t.a, e = t.e, A16 = t.A16;
        if (lane == 0) stage[t.n_pieces] = 0;  // the re-alignment reads one word past the last piece
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

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
        const u32 nfull = mine ? (len_read >> 4) : 0;
        const u32 tail = mine ? (len_read & 15) : 0;
        const u32 nsteps = nfull + (tail ? 1u : 0u);
        u32 s = s_root;
        WCtx dctx = {0, 0};
        u32 ddepth = 0;
        // the staged words of my read are fetched one step ahead (they do not depend on the walk), so
        // the only LDS round trip on the step-to-step critical path is the node lookup
        const u32* sp = stage + (mine ? w0 : 0);
        u32 wc = sp[0], wn = sp[1];
        for (u32 tstep = 0; tstep < KMC_WALK_MAX_READ / KMC_WALK_STRIDE + 1; ++tstep) {
            const bool act = tstep < nsteps;
            if (__builtin_amdgcn_ballot_w64(act) == 0) break;
            if (act) {
                // next 16 bases of my read, re-aligned to its start
                u32 label = alignbit(wn, wc, sh);
                wc = wn;
                wn = sp[tstep + 2];  // (at most 2 words past the read's last word: inside the staging area's slack; unused then)
                const bool full = tstep < nfull;
                WalkNode* np = reinterpret_cast<WalkNode*>(reinterpret_cast<char*>(L.node) + s);
                const u64 pe = __hip_atomic_load(&np->prim, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (full && (u32)pe == label && (u32)(pe >> 32) != 0) {
                    atomicAdd(&np->cnt[cpar], 1u);
                    s = (u32)(pe >> 32);
                } else {
                    if (!full) label &= (1u << (2 * tail)) - 1u;
                    s = walk_slow<KW, CANON>(L, g, s, label, full ? KMC_WALK_STRIDE : (int)tail, k, mask_hi, mask_lo, dctx, ddepth, ndirect, cpar, sk, lg);
                }
            }
        }
        __builtin_amdgcn_wave_barrier();  // staging area is reused by the next tile
    };

    // The wave's loads form ONE stream of rounds across its tiles, and the two register sets
    // alternate strictly along that stream: while round g is encoded out of one set, round g+1 -- the
    // next round of this tile, or round 0 of the wave's next tile -- is already in flight in the
    // other.  So a tile costs ceil(n_pieces / RP) rounds whatever their parity (the first version
    // padded every tile to an even number: 8 instead of 6.25 rounds' worth of loads on 400-base
    // reads), and the first round of the next tile hides behind this tile's step phase.
    // (Issuing the first tile's loads before the LDS initialisation measured 0.5 % slower, not faster.)
    uint4 va[LPR], vb[LPR];
    TileGeo cur, nxt;
    u64 tile = gw;
    bool live = gw < n_tiles;  // wave-uniform
    bool has_next = false;
    u64 nxt_tile = 0;
    TileRaw nxt_raw = {0, 0, 0};
    u32 rbase = 0;
    if (live) { cur = tile_geo(tile); nxt = cur; issue(va, cur.A16, cur.n_pieces ? cur.n_pieces - 1 : 0, 0); }
    auto half = [&](uint4 (&x)[LPR], uint4 (&y)[LPR]) {
        const u32 np_u = (u32)__builtin_amdgcn_readfirstlane((int)cur.n_pieces);
        // The successor's offsets (tile_load, at the entry of this tile) are "used" here, at the top of
        // every half-iteration: in the one after the tile entry that costs a counted wait for loads that
        // are older than the round about to be consumed anyway; in all others nothing.  Without it the
        // compiler has to put an s_waitcnt vmcnt(0) in front of tile_finish, which would drain the
        // round in flight at every tile's last round.
        asm volatile("" ::"v"(nxt_raw.a), "v"(nxt_raw.e));
        const bool last = rbase + RP >= np_u;  // wave-uniform: this is the tile's last round
        if (rbase == 0) {  // tile entry
            if (lane < KMC_WALK_BADWORDS) L.badbits[wv][lane] = 0;
            anybad = false;
            u32 jn = 0;
            if (lane == 0) jn = atomicAdd(&L.qnext, 1u);
            nxt_tile = tile_of((u32)__builtin_amdgcn_readfirstlane((int)jn));
            has_next = nxt_tile < n_tiles;
            if (has_next) nxt_raw = tile_load(nxt_tile);
            // a one-round tile needs its successor's geometry right behind the loads and has to wait for them
            if (last && has_next) { asm volatile("; tile_finish right behind tile_load"); nxt = tile_finish(nxt_raw); }
        } else if (last && has_next) {
            // every other tile: rounds later, and no path from tile_load leads here except through the
            // use above (a second copy on purpose: merged with the one-round case, its s_waitcnt
            // vmcnt(0) would serve both)
            nxt = tile_finish(nxt_raw);
        }
        const u32 cur_last = cur.n_pieces ? cur.n_pieces - 1 : 0, nxt_last = nxt.n_pieces ? nxt.n_pieces - 1 : 0;
        issue(y, last ? nxt.A16 : cur.A16, last ? nxt_last : cur_last, last ? 0u : rbase + RP);
        consume(x, cur, rbase, np_u);
        if (last) {
            step_phase(cur, tile);
#ifdef KMC_WALK_STAMPS
            if (wv == 0 && lane == 0 && tile == gw) WALK_STAMP(21);
#endif
            cur = nxt;
            tile = nxt_tile;
            rbase = 0;
            live = has_next;
        } else {
            rbase += RP;
        }
    };
    while (live) {
        half(va, vb);
        if (!live) break;
        half(vb, va);
    }
    if (lane == 0) WALK_STAMP(2 + wv);
    nk = wave_sum_u64(nk);
    ndirect = wave_sum_u64(ndirect);
    if (lane == 0) {
        if (nk) atomicAdd((unsigned long long*)&g.counters[KMC_CTR_KMERS], nk);
        if (ndirect) atomicAdd((unsigned long long*)&g.counters[KMC_CTR_BADBASE], ndirect);
    }

    // ---- flush: unfold every memoised edge into its k-mers ----
    // Work item = (edge, step): the k-mer completed by the step-th base of an edge's label.  First
    // the used entries are compacted into a list (in wave 0's staging area, free by now), then the
    // items are spread over all 1024 threads: ~4 independent global adds per thread instead of 16-32
    // dependent ones on the few threads that happened to own a used entry (measured: 140 us -> the
    // flush used to be 8 % of the whole kernel).
    __syncthreads();
    if (tid == 0) WALK_STAMP(18);
    if (tid == 0 && lg.count) lg.count[blockIdx.x] = min(L.logn, lg.cap_wg);   // (every wave has finished its tiles)
    // adds to the (k+16)-mer table since its last unfold: what tells "keys only" (they stay across launches) from "counts pending"
    if (tid == 0 && L.nsk) atomicAdd((unsigned long long*)&sk.counters[KMC_CTR_KMERS], (unsigned long long)L.nsk);
    unsigned short* flist = reinterpret_cast<unsigned short*>(L.stage[0]);  // up to NCAP + ECAP entry indices
    u32* fcount = &L.badbits[0][0];
    if (tid == 0) *fcount = 0;
    __syncthreads();
    for (int i = tid; i < KMC_WALK_NCAP + KMC_WALK_ECAP; i += KMC_WALK_THREADS) {
        u32 cnt;
        bool snap;  // the entry came with the shared snapshot (it cannot have changed since: set-once fields)
        if (i < KMC_WALK_NCAP) {
            cnt = ((L.node[i].prim >> 32) != 0) ? L.node[i].cnt[0] + L.node[i].cnt[1] : 0;
            snap = warm && (memo->prim[i] >> 32) != 0;
        } else {
            cnt = (L.edge[i - KMC_WALK_NCAP].kv != ~0ull) ? L.edge[i - KMC_WALK_NCAP].cnt[0] + L.edge[i - KMC_WALK_NCAP].cnt[1] : 0;
            snap = warm && memo->ekv[i - KMC_WALK_NCAP] != ~0ull;
        }
        if (cnt) {
            if (snap) atomicAdd((unsigned long long*)&gcnt[i], (unsigned long long)cnt);  // dense: slot i of every workgroup
            else flist[atomicAdd(fcount, 1u)] = (unsigned short)i;
        }
    }
    __syncthreads();
    if (tid == 0) WALK_STAMP(19);
    const u32 n_items = *fcount * KMC_WALK_STRIDE;
    // every workgroup holds nearly the same entries in nearly the same order (slot = hash of the
    // key): start each one at a different place so that they do not all hit one address at a time
    const u32 rot = n_items ? (u32)(((u64)blockIdx.x * 2654435761u) % n_items) : 0;
    for (u32 w0 = tid; w0 < n_items; w0 += KMC_WALK_THREADS) {
        u32 w = w0 + rot;
        if (w >= n_items) w -= n_items;
        const u32 i = flist[w / KMC_WALK_STRIDE], step = w % KMC_WALK_STRIDE;
        u32 label, len, id, cnt;
        if (i < KMC_WALK_NCAP) {
            id = i; label = (u32)L.node[i].prim; len = KMC_WALK_STRIDE; cnt = L.node[i].cnt[0] + L.node[i].cnt[1];
        } else {
            const u64 kv = L.edge[i - KMC_WALK_NCAP].kv;
            cnt = L.edge[i - KMC_WALK_NCAP].cnt[0] + L.edge[i - KMC_WALK_NCAP].cnt[1];
            label = (u32)kv; len = ((u32)(kv >> 32) & 15u) + 1; id = (u32)(kv >> 36) & (KMC_WALK_NCAP - 1);
        }
        if (step < len) {
            WCtx ctx;
            u32 depth;
            node_decode<KW>(node_key_load<KW>(L, id), k, ctx, depth);
            (void)walk_roll<KW, CANON, false>(g, ctx, depth, label, (int)step + 1, k, mask_hi, mask_lo, 0);  // context after base `step`
            if (depth >= (u32)k) walk_gadd<KW, CANON>(g, ctx, k, cnt);
        }
    }
    // ---- save the memo (structure only) for the next launch ----
    if (memo_out && blockIdx.x == 0) {
        for (int i = tid; i < KMC_WALK_ECAP; i += KMC_WALK_THREADS) memo_out->ekv[i] = L.edge[i].kv;
        for (int i = tid; i < KMC_WALK_NCAP; i += KMC_WALK_THREADS) {
            memo_out->nkeys[i] = L.nkeys[i];
            if (KW == 2) memo_out->nkeys_hi[i] = L.nkeys_hi[i];
            memo_out->prim[i] = L.node[i].prim;
        }
        if (tid == 0) { memo_out->nedges = L.nedges; memo_out->nnodes = L.nnodes; memo_out->tag = KMC_WALK_MEMO_TAG | (u64)k; }
    }
    if (tid == 0) WALK_STAMP(20);
}

// Turns the traversal counters that all workgroups reduced into gcnt (one per snapshot slot) into
// k-mer counts, once per launch (gcnt is cleared by kmc_walk_prepare before the next launch).
template <int KW, bool CANON>
__device__ __forceinline__ void walk_unfold_part(const WalkMemoSlot<KW>* memo, u64* gcnt, int k, const GTable& g, u32 block, u32 n_blocks) {
    if (memo->tag != (KMC_WALK_MEMO_TAG | (u64)k)) return;  // the launch ran without a snapshot: gcnt untouched
    const u32 tid = block * blockDim.x + threadIdx.x, nthreads = n_blocks * blockDim.x;
    const int kb = 2 * k;
    const u64 mask_lo = kb >= 64 ? ~0ull : ((1ull << kb) - 1);
    const u64 mask_hi = kb <= 64 ? 0ull : ((1ull << (kb - 64)) - 1);
    for (u32 w = tid; w < (KMC_WALK_NCAP + KMC_WALK_ECAP) * KMC_WALK_STRIDE; w += nthreads) {
        const u32 i = w / KMC_WALK_STRIDE, step = w % KMC_WALK_STRIDE;
        const u64 cnt = gcnt[i];
        if (!cnt) continue;
        u32 label, len, id;
        if (i < KMC_WALK_NCAP) {
            id = i; label = (u32)memo->prim[i]; len = KMC_WALK_STRIDE;
        } else {
            const u64 kv = memo->ekv[i - KMC_WALK_NCAP];
            label = (u32)kv; len = ((u32)(kv >> 32) & 15u) + 1; id = (u32)(kv >> 36) & (KMC_WALK_NCAP - 1);
        }
        if (step < len) {
            WCtx nk, ctx;
            nk.lo = memo->nkeys[id];
            nk.hi = KW == 2 ? memo->nkeys_hi[id] : 0ull;
            u32 depth;
            node_decode<KW>(nk, k, ctx, depth);
            (void)walk_roll<KW, CANON, false>(g, ctx, depth, label, (int)step + 1, k, mask_hi, mask_lo, 0);
            if (depth >= (u32)k) walk_gadd<KW, CANON>(g, ctx, k, cnt);
        }
    }
    // every entry is read by 16 consecutive threads of ONE block (the grid covers the item space exactly
    // once): after the block's reads, clear its entries for the next launch
    __syncthreads();
    if (threadIdx.x < 256 / KMC_WALK_STRIDE) gcnt[block * (256 / KMC_WALK_STRIDE) + threadIdx.x] = 0;
}

// One lane per listed read, byte by byte: reads diverted from the walk kernel (non-ACGT bytes).
template <int KW, bool CANON>
__device__ __forceinline__ void walk_scalar_part(const uint8_t* __restrict__ bases, const u64* __restrict__ vstart, const u64* __restrict__ vend,
                                                 WalkWs* ws, const u32* __restrict__ list, int k, const GTable& g, u32 block, u32 n_blocks) {
    // every workgroup of this part reads the deferred-read counter; the last one to have read it leaves the
    // workspace clean for the next launch (no memset per launch)
    __shared__ u64 s_n;
    if (threadIdx.x == 0) {
        s_n = __hip_atomic_load(&ws->n_deferred, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // the read above has RETURNED before the ticket below is drawn (the last ticket clears the counter); waiting for
        // the load is all that takes -- a __threadfence() here wrote back and invalidated caches in every one of 256
        // workgroups of a kernel that runs 11 us
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const u32 t = atomicAdd(&ws->scalar_done, 1u);
        if (t == n_blocks - 1) { ws->n_deferred = 0; ws->scalar_done = 0; }
    }
    __syncthreads();
    const u64 n = s_n;
    if (n == 0) return;  // (the common case)
    const int kb = 2 * k;
    const u64 mask_lo = kb >= 64 ? ~0ull : ((1ull << kb) - 1);
    const u64 mask_hi = kb <= 64 ? 0ull : ((1ull << (kb - 64)) - 1);
    u64 nk = 0;
    for (u64 i = (u64)block * blockDim.x + threadIdx.x; i < n; i += (u64)n_blocks * blockDim.x) {
        const u64 r = list[i];
        u64 lo = 0, hi = 0;
        int run = 0;
        for (u64 p = vstart[r]; p < vend[r]; ++p) {
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

// What follows a walk launch, in ONE launch (two launches cost 4 us more per step): the first n_unfold
// workgroups turn the dense traversal counters into k-mer counts, the others count the diverted reads.
// The two parts touch disjoint state (gcnt / the deferred list) and both only add to the count table.
#define KMC_WALK_UNFOLD_BLOCKS ((KMC_WALK_NCAP + KMC_WALK_ECAP) * KMC_WALK_STRIDE / 256)
template <int KW, bool CANON>
__global__ __launch_bounds__(256)
void kmc_walk_tail_kernel(const uint8_t* __restrict__ bases, const u64* __restrict__ vstart, const u64* __restrict__ vend,
                          WalkWs* ws, const u32* __restrict__ list, const WalkMemoSlot<KW>* memo, u64* gcnt, int k, GTable g) {
    if (blockIdx.x < KMC_WALK_UNFOLD_BLOCKS) walk_unfold_part<KW, CANON>(memo, gcnt, k, g, blockIdx.x, KMC_WALK_UNFOLD_BLOCKS);
    else walk_scalar_part<KW, CANON>(bases, vstart, vend, ws, list, k, g, blockIdx.x - KMC_WALK_UNFOLD_BLOCKS, gridDim.x - KMC_WALK_UNFOLD_BLOCKS);
}

// ---- host side ------------------------------------------------------------------------------
static inline bool kmc_walk_supported(int k, int mode, u64 max_read_len) {
    return mode == KMC_MODE_CONTIG && k >= 1 && k <= KMC_WALK_MAX_K && max_read_len >= 1;
}

// ---- reads longer than KMC_WALK_MAX_READ: pieces ("virtual reads") --------------------------------
// A read of L > MAX bases is walked as ceil((L - (k-1)) / S) pieces of at most MAX bases that start
// S = MAX - (k-1) bases apart, i.e. consecutive pieces overlap by k-1 bases.  Piece j then holds
// exactly the windows that END in its last S (or fewer) positions: every window of the read lies in
// exactly one piece, so walking the pieces as independent reads gives the read's counts, bit for bit.
__host__ __device__ inline u64 kmc_vreads_of(u64 len, int k) {
    if (len <= KMC_WALK_MAX_READ) return 1;
    const u64 S = KMC_WALK_MAX_READ - (u64)(k - 1);
    return (len - (u64)(k - 1) + S - 1) / S;
}
// piece j of a read that occupies bases [a, e): [*st, *en)
__host__ __device__ inline void kmc_vread_span(u64 a, u64 e, int k, u64 j, u64* st, u64* en) {
    const u64 S = KMC_WALK_MAX_READ - (u64)(k - 1);
    *st = a + j * S;
    *en = (e - *st > KMC_WALK_MAX_READ) ? *st + KMC_WALK_MAX_READ : e;
}
__global__ void kmc_vreads_count_kernel(const u64* __restrict__ offsets, u64 n_reads, int k, u32* __restrict__ cnt) {
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n_reads; i += (u64)gridDim.x * blockDim.x) {
        const u64 v = kmc_vreads_of(offsets[i + 1] - offsets[i], k);
        cnt[i] = v > 0xFFFFFFFFull ? 0xFFFFFFFFu : (u32)v;  // (a batch with 2^32 pieces is refused by the caller)
    }
}
// pos = exclusive scan of cnt; piece v belongs to the read i with pos[i] <= v < pos[i+1] (every read has >= 1 piece)
__global__ void kmc_vreads_fill_kernel(const u64* __restrict__ offsets, const u32* __restrict__ pos, u64 n_reads, u64 n_v, int k,
                                       u64* __restrict__ vstart, u64* __restrict__ vend) {
    for (u64 v = (u64)blockIdx.x * blockDim.x + threadIdx.x; v < n_v; v += (u64)gridDim.x * blockDim.x) {
        u64 lo = 0, hi = n_reads;  // last i with pos[i] <= v
        while (hi - lo > 1) {
            const u64 mid = (lo + hi) >> 1;
            if (pos[mid] <= v) lo = mid; else hi = mid;
        }
        kmc_vread_span(offsets[lo], offsets[lo + 1], k, v - pos[lo], &vstart[v], &vend[v]);
    }
}
// memo buffer: two snapshot slots + the dense counter array
static inline size_t kmc_walk_slot_bytes(int KW) { return KW == 1 ? sizeof(WalkMemoSlot<1>) : sizeof(WalkMemoSlot<2>); }
static inline size_t kmc_walk_memo_bytes(int, int KW) { return 2 * kmc_walk_slot_bytes(KW); }
static inline size_t kmc_walk_workspace_bytes(u64 n_reads) { return KMC_WALK_WS_PREFIX + (size_t)(n_reads + 16) * sizeof(u32); }

template <int KW, bool CANON>
static inline void kmc_walk_launch_t(hipStream_t st, int grid, int n_cu, const uint8_t* d_bases, const u64* d_vstart, const u64* d_vend,
                                     u64 n_reads, u64 n_bases, int k, u64 tile_begin, u64 tile_end, WalkWs* hdr, u32* list, u64* gcnt, void* memo, int parity, GTable g, GTable sk, SkLog lg, int phase,
                                     hipEvent_t e0, hipEvent_t e1) {
    const size_t smem = sizeof(WalkLds<KW>);
    static std::atomic<unsigned long long> attr{0};  // one flag per instantiation and device
    if (kmc_attr_once(attr)) (void)hipFuncSetAttribute((const void*)kmc_walk_kernel<KW, CANON>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    WalkMemoSlot<KW>* slots = (WalkMemoSlot<KW>*)memo;
    if (phase == 0) {
        // the launch's own start / stop timestamps go to e0 / e1 (hipExtLaunchKernelGGL): what hipEventRecord in front of and
        // behind the launch measured too, without two more packets in the stream per step
        if (e0 && e1)
            hipExtLaunchKernelGGL((kmc_walk_kernel<KW, CANON>), dim3(grid), dim3(KMC_WALK_THREADS), (uint32_t)smem, st, e0, e1, 0u, d_bases, n_bases, d_vstart, d_vend, n_reads, k, tile_begin, tile_end, hdr, list,
                                  (const WalkMemoSlot<KW>*)&slots[parity], &slots[parity ^ 1], gcnt, g, sk, lg);
        else
            hipLaunchKernelGGL((kmc_walk_kernel<KW, CANON>), dim3(grid), dim3(KMC_WALK_THREADS), smem, st, d_bases, n_bases, d_vstart, d_vend, n_reads, k, tile_begin, tile_end, hdr, list,
                               (const WalkMemoSlot<KW>*)&slots[parity], &slots[parity ^ 1], gcnt, g, sk, lg);
    } else {
        static_assert(((KMC_WALK_NCAP + KMC_WALK_ECAP) * KMC_WALK_STRIDE) % 256 == 0 && 256 % KMC_WALK_STRIDE == 0, "unfold grid must cover the items exactly");
        hipLaunchKernelGGL((kmc_walk_tail_kernel<KW, CANON>), dim3(KMC_WALK_UNFOLD_BLOCKS + n_cu), dim3(256), 0, st,
                           d_bases, d_vstart, d_vend, hdr, list, (const WalkMemoSlot<KW>*)&slots[parity], gcnt, k, g);
    }
}

// The (k+16)-mer table is unfolded ONCE per batch, after the batch's last walk launch (its counts add up
// over the launches): with one unfold per launch a 27-launch batch walked its 3 M entries 27 times.
static inline int kmc_sk_unfold_launch(hipStream_t st, int n_cu, int KW, int k, bool canon, GTable sk, GTable g) {
    if (!sk.key_lo) return KMC_OK;
    if (KW == 1) {
        if (canon) hipLaunchKernelGGL((kmc_sk_unfold_kernel<1, true>), dim3((unsigned)n_cu * 8), dim3(256), 0, st, sk, k, g);
        else hipLaunchKernelGGL((kmc_sk_unfold_kernel<1, false>), dim3((unsigned)n_cu * 8), dim3(256), 0, st, sk, k, g);
    } else {
        if (canon) hipLaunchKernelGGL((kmc_sk_unfold_kernel<2, true>), dim3((unsigned)n_cu * 8), dim3(256), 0, st, sk, k, g);
        else hipLaunchKernelGGL((kmc_sk_unfold_kernel<2, false>), dim3((unsigned)n_cu * 8), dim3(256), 0, st, sk, k, g);
    }
    hipLaunchKernelGGL(kmc_sk_spill_reset_kernel, dim3(1), dim3(64), 0, st, sk);
    return hipGetLastError() == hipSuccess ? KMC_OK : KMC_ERR_HIP;
}

// phase 0: the walk kernel over tiles [tile_begin, tile_end); phase 1: the scalar kernel for the reads
// it diverted + the unfold of the dense snapshot counters.  `parity` selects the snapshot slot read
// by this launch (the other one is written); the caller flips it after phase 1.  The caller clears the workspace header (kmc_walk_prepare) before phase 0.
static inline int kmc_walk_prepare(hipStream_t st, void* ws) {
    return hipMemsetAsync(ws, 0, KMC_WALK_WS_PREFIX, st) == hipSuccess ? KMC_OK : KMC_ERR_HIP;
}
// workgroups of a walk launch over n_tiles tiles (one 160 KB workgroup per CU is resident)
static inline int kmc_walk_grid(u64 n_tiles, int n_cu) {
    const u64 want = (n_tiles + KMC_WALK_WAVES - 1) / KMC_WALK_WAVES;
    const int grid = (int)(want < (u64)n_cu ? want : (u64)n_cu);
    return grid < 1 ? 1 : grid;
}
static inline int kmc_walk_launch(hipStream_t st, int n_cu, int KW, int k, bool canon, const uint8_t* d_bases,
                                  const u64* d_vstart, const u64* d_vend, u64 n_reads, u64 n_bases, u64 tile_begin, u64 tile_end, void* ws, void* memo, int parity, GTable g, GTable sk, SkLog lg, int phase,
                                  hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr) {
    if (n_reads >= (1ull << 32) || tile_end <= tile_begin) return KMC_ERR_ARG;
    WalkWs* hdr = (WalkWs*)ws;
    u32* list = (u32*)((char*)ws + KMC_WALK_WS_PREFIX);
    u64* gcnt_ws = (u64*)((char*)ws + sizeof(WalkWs));
    const u64 n_tiles = tile_end - tile_begin;
    const int grid = kmc_walk_grid(n_tiles, n_cu);
    if (KW == 1) {
        if (canon) kmc_walk_launch_t<1, true>(st, grid, n_cu, d_bases, d_vstart, d_vend, n_reads, n_bases, k, tile_begin, tile_end, hdr, list, gcnt_ws, memo, parity, g, sk, lg, phase, e0, e1);
        else kmc_walk_launch_t<1, false>(st, grid, n_cu, d_bases, d_vstart, d_vend, n_reads, n_bases, k, tile_begin, tile_end, hdr, list, gcnt_ws, memo, parity, g, sk, lg, phase, e0, e1);
    } else {
        if (canon) kmc_walk_launch_t<2, true>(st, grid, n_cu, d_bases, d_vstart, d_vend, n_reads, n_bases, k, tile_begin, tile_end, hdr, list, gcnt_ws, memo, parity, g, sk, lg, phase, e0, e1);
        else kmc_walk_launch_t<2, false>(st, grid, n_cu, d_bases, d_vstart, d_vend, n_reads, n_bases, k, tile_begin, tile_end, hdr, list, gcnt_ws, memo, parity, g, sk, lg, phase, e0, e1);
    }
    return hipGetLastError() == hipSuccess ? KMC_OK : KMC_ERR_HIP;
}
