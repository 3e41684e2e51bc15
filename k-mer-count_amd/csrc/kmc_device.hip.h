// kmc_device.hip.h -- device-side building blocks shared by the gfx950 kernels.
//
// Written for CDNA4 (wave64, LDS atomics, v_alignbit/v_perm/v_bfrev) only; there is no other
// backend.  Semantics follow SURVEY.md 8a-def: alphabet A=0 C=1 G=2 T=3 (reference
// k-mer-count/src/main.rs:19-22), MSB-first packing so that unsigned key order equals the
// reference's string order (main.rs:87), canonical = min(fwd, revcomp).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint32_t u32;
typedef uint64_t u64;

#define KMC_EMPTY64 (~0ull)
#define KMC_LOCKED64 (~0ull - 1ull)

// counters[] layout (device, u64 each)
enum { KMC_CTR_OCCUPIED = 0, KMC_CTR_SPILL = 1, KMC_CTR_ERR = 2, KMC_CTR_KMERS = 3, KMC_CTR_MAXLEN = 4,
       KMC_CTR_OUT = 5, KMC_CTR_BADBASE = 6, KMC_CTR_SUM = 7,
       KMC_CTR_OUT1 = 8, KMC_CTR_SUM1 = 9,  // second parity of OUT/SUM: a finalize clears the pair the next one uses
       KMC_CTR_SUM2 = 10,                   // sum of counts of a merged (table + sorted runs) view
       KMC_CTR_FASTFIN = 11,                // 1: the speculative small-table finalize produced the sorted view
       KMC_CTR_SLABSKIP = 12,               // slabs kmc_merge_slabs_kernel skipped (oversize: payload not inline)
       // host mirror only (never device counters): what kmc_small_finalize_kernel publishes about itself
       KMC_CTR_FINSEQ = 13,                 // sequence number of the finalize launch that wrote the mirror last
       KMC_CTR_FINOK = 14,                  // finalize launches of this ctx that produced a view (cumulative)
       KMC_CTR_FINSKIP = 15,                // oversize slabs seen by those launches (cumulative)
       KMC_CTR_N = 16 };

// Global (HBM) open-addressing count table.  One-word keys (KW==1, k<=31) use key_lo only and
// KMC_EMPTY64 as the empty marker (a 62-bit key can never equal it).  Two-word keys (KW==2,
// 32<=k<=63 and the 108-bit LR keys) claim a slot by CAS on key_hi EMPTY->LOCKED, publish
// key_lo, then release key_hi; key_hi < 2^62 so neither marker is a real key.
struct GTable {
    u64* key_hi;
    u64* key_mid;  // third key word (KW == 3: the walk kernel's (k+16)-mer table for k >= 48), else unused
    u64* key_lo;
    u64* count;
    u64  capmask;  // capacity - 1 (capacity is a power of two)
    u64* counters; // KMC_CTR_*
    u64* spill_hi;
    u64* spill_mid;
    u64* spill_lo;
    u64* spill_cnt;
    u64  spill_cap;
    u64* occ_list;     // slot index of the i-th claimed slot, for i < occ_list_cap (small-table fast finalize)
    u64  occ_list_cap;
    u64* occ_key_lo;   // the i-th claimed slot's KEY, dense (nullptr: not kept): the small-table finalize reads the keys of a
    u64* occ_key_hi;   // table with coalesced loads instead of a dependent occ_list -> slot -> key chain per workgroup and tile
};

// claimed slots listed (small-table finalize, reset through the list, slab packing, cheap table snapshots).  Round 3: 32768 ->
// 131072: the rank-sort finalize costs n^2 / 64 compares per workgroup-of-16-keys -- 0.13 ms at 67 k keys, about what the
// general path (compaction + weighted radix sort: 0.8 ms of fixed cost) takes at 131 k -- and the plateau inputs of the
// cardinality sweep (pools of 32..100 lines: 29 k .. 255 k distinct 31-mers) sit right above the old limit
// (Later in round 3 the two meanings were separated: the LIST -- claimed slots + their keys, 24 bytes per entry -- covers a
// million keys, so that finalize compaction, reset and cheap snapshots go through it for every table of the cardinality sweep
// up to pool 100 (572 k distinct 63-mers) instead of scanning millions of slots; the rank-sort KERNEL keeps its 131072.)
#define KMC_OCC_LIST_CAP 1048576
#define KMC_FIN_KERNEL_MAX 131072   // most keys kmc_small_finalize_kernel takes (its grid, its output buffers)

__device__ __forceinline__ u64 kmc_mix64(u64 z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

template <int KW>
__device__ __forceinline__ u64 kmc_hash_key(u64 hi, u64 lo, u64 mid = 0) {
    if (KW >= 2) lo ^= kmc_mix64(hi + 0x9E3779B97F4A7C15ull);
    if (KW == 3) lo ^= kmc_mix64(mid + 0xD6E8FEB86659FD93ull);
    return kmc_mix64(lo);
}

// 32-bit slot hash for the LDS tables: the high half of a 64-bit multiplicative mix (the low bits
// of a plain 32-bit multiply cluster consecutive k-mers, which lengthens linear-probe chains).
template <int KW>
__device__ __forceinline__ u32 kmc_hash32(u64 hi, u64 lo) {
    u64 z = lo;
    if (KW == 2) z ^= hi * 0xC2B2AE3D27D4EB4Full;
    z = (z ^ (z >> 29)) * 0x9E3779B97F4A7C15ull;
    return (u32)(z >> 32) ^ (u32)(z >> 17);
}

__device__ __forceinline__ u64 ld_relaxed(const u64* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ void kmc_spill(const GTable& g, u64 hi, u64 lo, u64 cnt, u64 mid = 0) {
    u64 idx = atomicAdd((unsigned long long*)&g.counters[KMC_CTR_SPILL], 1ull);
    if (idx < g.spill_cap) {
        if (g.spill_hi) g.spill_hi[idx] = hi;
        if (g.spill_mid) g.spill_mid[idx] = mid;
        g.spill_lo[idx] = lo;
        g.spill_cnt[idx] = cnt;
    } else {
        atomicOr((unsigned long long*)&g.counters[KMC_CTR_ERR], 1ull);
    }
}

// Add `cnt` occurrences of key (hi,lo) to the global table.  Lock-free; any number of
// workgroups on any XCD may call it concurrently (device-scope atomics only).
//
// Control flow: ONE loop whose only back-edge is taken on a wave-uniform ballot.  Every active
// lane runs the body once per trip, so a lane that wins a slot (CAS EMPTY->LOCKED) publishes it
// in the same trip, before any sibling lane that saw LOCKED looks again.  A per-lane
// `while (!done)` with a "look again" path must not be used here: the compiler may split that
// path off into an inner spin loop, and a lane spinning on a slot held by a masked-off lane of
// its own wave never terminates.
template <int KW>
__device__ __forceinline__ void gtable_add(const GTable& g, u64 hi, u64 lo, u64 cnt, u64 mid = 0) {
    u64 h = kmc_hash_key<KW>(hi, lo, mid) & g.capmask;
    u64 probes = 0;
    const u64 max_probes = g.capmask < 4095 ? g.capmask + 1 : 4096;
    bool done = false;
    u32 trips = 0;
    while (__builtin_amdgcn_ballot_w64(!done) != 0) {
        if (!done) {
            bool advance = false;
            if (++trips > (1u << 22)) {  // every wave must drain: give up loudly, never spin forever
                atomicOr((unsigned long long*)&g.counters[KMC_CTR_ERR], 2ull);
                done = true;
            } else if (KW == 1) {
                u64 cur = ld_relaxed(&g.key_lo[h]);
                if (cur == KMC_EMPTY64) {
                    cur = atomicCAS((unsigned long long*)&g.key_lo[h], KMC_EMPTY64, lo);
                    if (cur == KMC_EMPTY64) {
                        const u64 i = atomicAdd((unsigned long long*)&g.counters[KMC_CTR_OCCUPIED], 1ull);
                        if (i < g.occ_list_cap) { g.occ_list[i] = h; if (g.occ_key_lo) g.occ_key_lo[i] = lo; }
                        cur = lo;
                    }
                }
                if (cur == lo) {
                    atomicAdd((unsigned long long*)&g.count[h], cnt);
                    done = true;
                } else {
                    advance = true;
                }
            } else {
                // Two-word protocol without agent-scope fences (a release here is a whole-L2
                // write-back, measured 1.2 ms per launch for 6 k keys): every access to the key words
                // is a device-scope atomic (sc1: bypasses L1, write-through), the writer drains its
                // key_lo store (s_waitcnt vmcnt(0)) before it stores key_hi, and a reader loads
                // key_lo only after it has SEEN a matching key_hi (control dependency).
                u64 cur = ld_relaxed(&g.key_hi[h]);
                if (cur == KMC_EMPTY64) {
                    u64 old = atomicCAS((unsigned long long*)&g.key_hi[h], KMC_EMPTY64, KMC_LOCKED64);
                    if (old == KMC_EMPTY64) {
                        __hip_atomic_store(&g.key_lo[h], lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (KW == 3) __hip_atomic_store(&g.key_mid[h], mid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        __hip_atomic_store(&g.key_hi[h], hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        const u64 i = atomicAdd((unsigned long long*)&g.counters[KMC_CTR_OCCUPIED], 1ull);
                        if (i < g.occ_list_cap) { g.occ_list[i] = h; if (g.occ_key_lo) { g.occ_key_lo[i] = lo; g.occ_key_hi[i] = hi; } }
                        atomicAdd((unsigned long long*)&g.count[h], cnt);
                        done = true;
                    }
                    // lost the race: the slot is LOCKED or published now; examine it next trip
                } else if (cur == KMC_LOCKED64) {
                    // being published by another lane/wave; examine it next trip
                } else if (cur == hi) {
                    if (ld_relaxed(&g.key_lo[h]) == lo && (KW != 3 || ld_relaxed(&g.key_mid[h]) == mid)) {
                        atomicAdd((unsigned long long*)&g.count[h], cnt);
                        done = true;
                    } else {
                        advance = true;
                    }
                } else {
                    advance = true;
                }
            }
            if (advance) {
                h = (h + 1) & g.capmask;
                if (++probes >= max_probes) { kmc_spill(g, hi, lo, cnt, mid); done = true; }
            }
        }
    }
}

// ---- bit helpers ---------------------------------------------------------------------------

// (hi:lo) >> s, low 32 bits; s in [0,31]  -> v_alignbit_b32
__device__ __forceinline__ u32 alignbit(u32 hi, u32 lo, u32 s) { return __builtin_amdgcn_alignbit(hi, lo, s); }

// swap the two bits of every 2-bit pair
__device__ __forceinline__ u32 pairswap(u32 x) { return ((x & 0x55555555u) << 1) | ((x >> 1) & 0x55555555u); }

// 16 bases little-endian (base j at bits 2j..2j+1)  <->  big-endian (base j at bits 31-2j..30-2j)
__device__ __forceinline__ u32 le_to_be(u32 wle) { return pairswap(__builtin_bitreverse32(wle)); }

// Reverse-complement word of a big-endian 16-base word, itself big-endian:
// complement every base and reverse their order == ~(little-endian form).
__device__ __forceinline__ u32 rc_word_be(u32 wbe) { return ~le_to_be(wbe); }

// 4 ASCII bytes -> 2-bit codes in place (one code in the low 2 bits of each byte).
// 'A'=0x41 'C'=0x43 'G'=0x47 'T'=0x54: (c>>1)&3 = 0,1,3,2; x ^= x>>1 fixes G/T to 2,3.
__device__ __forceinline__ u32 ascii4_to_codes(u32 v) {
    u32 t = (v >> 1) & 0x03030303u;
    return t ^ ((t >> 1) & 0x01010101u);
}
// the ASCII the codes stand for; differs from the input exactly at non-ACGT bytes (v_perm_b32 as
// a 4-entry byte LUT)
__device__ __forceinline__ u32 codes_to_ascii4(u32 t) { return __builtin_amdgcn_perm(0u, 0x54474341u, t); }
// 4 codes -> 8 bits, first byte in the low pair
__device__ __forceinline__ u32 pack4_le(u32 t) {
    u32 x = t | (t >> 6);
    return (x | (x >> 12)) & 0xFFu;
}
// one bit per non-zero byte of x (bit i <=> byte i)
__device__ __forceinline__ u32 nonzero_bytes4(u32 x) {
    u32 nz = ((x | ((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu)) >> 7) & 0x01010101u;
    return (nz | (nz >> 7) | (nz >> 14) | (nz >> 21)) & 0xFu;
}

// Encode 16 ASCII bases (as loaded: v.x holds bytes 0..3).  wle: little-endian 2-bit word.
// anyx != 0 iff some byte is not one of ACGT; exact per-base flags via bad16_from().
struct Enc16 { u32 wle; u32 x0, x1, x2, x3; };
__device__ __forceinline__ Enc16 encode16(uint4 v) {
    Enc16 e;
    u32 t0 = ascii4_to_codes(v.x), t1 = ascii4_to_codes(v.y), t2 = ascii4_to_codes(v.z), t3 = ascii4_to_codes(v.w);
    e.x0 = v.x ^ codes_to_ascii4(t0);
    e.x1 = v.y ^ codes_to_ascii4(t1);
    e.x2 = v.z ^ codes_to_ascii4(t2);
    e.x3 = v.w ^ codes_to_ascii4(t3);
    e.wle = pack4_le(t0) | (pack4_le(t1) << 8) | (pack4_le(t2) << 16) | (pack4_le(t3) << 24);
    return e;
}
__device__ __forceinline__ u32 bad16_from(const Enc16& e) {
    return nonzero_bytes4(e.x0) | (nonzero_bytes4(e.x1) << 4) | (nonzero_bytes4(e.x2) << 8) | (nonzero_bytes4(e.x3) << 12);
}

// reverse complement of a 2k-bit key held in (hi,lo) (used off the hot loop: flush, LR mode)
__device__ __forceinline__ void revcomp_key(u64 hi, u64 lo, int k, u64& rhi, u64& rlo) {
    // complement + reverse 128 bits pairwise, then shift right by 128-2k
    u64 a = ~lo, b = ~hi;
    u64 ra = __builtin_bitreverse64(a), rb = __builtin_bitreverse64(b);
    ra = ((ra & 0x5555555555555555ull) << 1) | ((ra >> 1) & 0x5555555555555555ull);
    rb = ((rb & 0x5555555555555555ull) << 1) | ((rb >> 1) & 0x5555555555555555ull);
    // 128-bit value (ra:rb) = revcomp of the full 64 bases; keep the top 2k bits
    int sh = 128 - 2 * k;
    if (sh >= 64) { rlo = ra >> (sh - 64); rhi = 0; if (sh == 64) rlo = ra; }
    else if (sh == 0) { rhi = ra; rlo = rb; }
    else { rlo = (rb >> sh) | (ra << (64 - sh)); rhi = ra >> sh; }
}

__device__ __forceinline__ bool key_less(u64 ahi, u64 alo, u64 bhi, u64 blo) {
    return ahi < bhi || (ahi == bhi && alo < blo);
}

__device__ __forceinline__ u64 wave_sum_u64(u64 v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// hipFuncSetAttribute (dynamic LDS beyond 64 KB) is per DEVICE: one flag per (kernel instantiation, device), so
// that a process with contexts on several GPUs (kmc_count_file_multi) sets it on each of them
#include <atomic>
static inline bool kmc_attr_once(std::atomic<unsigned long long>& done) {
    int d = 0;
    (void)hipGetDevice(&d);
    const unsigned long long bit = 1ull << (d & 63);
    if (done.load(std::memory_order_relaxed) & bit) return false;
    done.fetch_or(bit, std::memory_order_relaxed);
    return true;
}

