// kmc_stream.hip.h -- KMC_ALGO_STREAM: the general counting kernel (any read lengths, any bytes).
//
// Replaces the reference's window loop + grouping, k-mer-count/src/main.rs:63-87, for
// contiguous k.
//
// Data layout.  The batch is the concatenation of all reads as ASCII (1 B/base) plus
// offsets[n_reads+1].  The stream is cut into 1024-base chunks; a wave owns a contiguous run of
// chunks and walks them in order.  Per chunk every lane loads ONE 16-byte piece (a fully
// coalesced 1 KiB global_load_dwordx4 per wave), packs it to one 32-bit 2-bit word in
// registers, and obtains the words of the 2 (k<=31) or 4 (k<=63) preceding lanes with wave
// shuffles -- the bases never pass through LDS.  The 16 windows ending in the lane's piece are
// static funnel shifts (v_alignbit_b32) of that register window; the reverse-complement window
// comes from the complemented little-endian words the same way, so there is no per-base
// rolling dependency.  Read starts inside a chunk are scattered from the offsets array into a
// per-wave 64-word LDS bitmap; windows that cross a read start or contain a non-ACGT byte are
// masked with a shift-OR smear of those bits.
//
// Accumulation.  Each workgroup owns an LDS partial histogram of FORWARD-strand keys in buckets of two slots
// (StreamLds below): the hot path reads a key's two home slots and adds on a hit, with no loop and no CAS;
// first sights and keys that do not fit their home bucket go through a per-wave miss buffer and a probing
// insert once per chunk; what fits nowhere goes to the global table with device-scope atomics; at the end
// the LDS table is flushed with one global atomic per distinct key, the canonical strand chosen there.
// (The extraction front end of KMC_ALGO_SORT, which shares this file's window helpers, is kmc_extract.hip.h.)
#pragma once
#include "kmc_device.hip.h"

#define KMC_STREAM_THREADS 1024
#define KMC_STREAM_WAVES (KMC_STREAM_THREADS / 64)
#define KMC_CHUNK 1024

// LDS partial table of one workgroup.  A slot holds a FORWARD-strand key; the canonical strand is
// chosen once per distinct key at the flush (counting is additive, so the table is the same as with
// per-occurrence canonicalisation -- what the walk kernel does too).  Slots are grouped in buckets of
// two, a key's home bucket is 2 * (hash % NB):
//   hot path  (once per k-mer, no loop, no CAS): read the two home slots, compare, ds_add on a hit;
//   miss      (first sight of a key, or a key that did not fit its home bucket): the key is appended
//             to the wave's miss buffer (one ballot + one LDS store) and the wave moves on;
//   drain     (once per chunk, or when the buffer fills): 64 buffered keys at a time go through the
//             probing insert (CAS claim, linear probing over the following slots, global table when
//             the LDS table is full).
// The first version of this kernel ran a probing loop per k-mer whose only exit was a wave-wide ballot:
// with 64 lanes almost every trip had a straggler (190 SALU + 84 VALU wave-instructions per k-mer step,
// 61 ms on the benchmark batch = 150 GB/s).
#define KMC_STREAM_MISSBUF 256
template <int KW> struct StreamSlot;
template <> struct StreamSlot<1> { u64 lo; };
template <> struct StreamSlot<2> { u64 lo, hi; };
template <int KW> struct StreamLds {
    static constexpr int LCAP = (KW == 1) ? 8192 : 4096;  // slots (64 KB of keys either way): one workgroup per CU
    static constexpr int LOGNB = (KW == 1) ? 12 : 11;      // log2(buckets of two)
    StreamSlot<KW> slot[LCAP];
    u32 cnt[LCAP];
    StreamSlot<KW> miss[KMC_STREAM_WAVES][KMC_STREAM_MISSBUF];
    u32 sbits[KMC_STREAM_WAVES][64];
    u32 nfill;
};

// home bucket (index of its first slot).  One 32-bit multiply: the high bits of the product depend on
// every bit of the folded key.  (The first version used a 64-bit multiplicative mix: four quarter-rate
// multiplies per k-mer.)
template <int KW>
__device__ __forceinline__ u32 stream_home(u64 hi, u64 lo) {
    u32 a = (u32)lo ^ __builtin_amdgcn_alignbit((u32)(lo >> 32), (u32)(lo >> 32), 21);
    if (KW == 2) a ^= __builtin_amdgcn_alignbit((u32)hi, (u32)hi, 27) ^ __builtin_amdgcn_alignbit((u32)(hi >> 32), (u32)(hi >> 32), 13);
    const u32 h = a * 0x9E3779B1u;
    return (h >> (32 - StreamLds<KW>::LOGNB)) << 1;
}

template <int KW, bool CANON>
__device__ __forceinline__ void stream_gadd(const GTable& g, u64 hi, u64 lo, int k, u64 cnt) {
    if (CANON) {
        u64 rhi, rlo;
        revcomp_key(hi, lo, k, rhi, rlo);
        if (key_less(rhi, rlo, hi, lo)) { hi = rhi; lo = rlo; }
    }
    gtable_add<KW>(g, hi, lo, cnt);
}

// The probing insert (drain path): find-or-claim a slot for (hi, lo) starting at its home bucket and
// add one; keys that find no slot within the probe budget, or arrive when the table is nearly full,
// are counted in the global table.  Same wave-uniform loop shape as gtable_add (see there for why).
// Returns true when the k-mer went to the global table.
template <int KW, bool CANON>
__device__ __forceinline__ bool lds_insert(StreamLds<KW>& L, const GTable& g, u64 hi, u64 lo, bool active, bool lds_ok, int k) {
    constexpr u32 M = StreamLds<KW>::LCAP - 1;
    u32 h = stream_home<KW>(hi, lo);
    int probes = lds_ok ? 0 : 1000;
    bool done = !active, to_global = false;
    u32 trips = 0;
    while (__builtin_amdgcn_ballot_w64(!done) != 0) {
        if (!done) {
            if (probes >= 16 || ++trips > (1u << 20)) {
                to_global = true;
                done = true;
            } else if (KW == 1) {
                u64 cur = __hip_atomic_load(&L.slot[h].lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (cur == KMC_EMPTY64) {
                    cur = atomicCAS((unsigned long long*)&L.slot[h].lo, KMC_EMPTY64, lo);
                    if (cur == KMC_EMPTY64) { atomicAdd(&L.nfill, 1u); cur = lo; }
                }
                if (cur == lo) { atomicAdd(&L.cnt[h], 1u); done = true; }
                else { h = (h + 1) & M; probes++; }
            } else {
                u64* const phi = &reinterpret_cast<u64*>(&L.slot[h])[1];
                u64* const plo = &reinterpret_cast<u64*>(&L.slot[h])[0];
                u64 cur = __hip_atomic_load(phi, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (cur == KMC_EMPTY64) {
                    u64 old = atomicCAS((unsigned long long*)phi, KMC_EMPTY64, KMC_LOCKED64);
                    if (old == KMC_EMPTY64) {
                        __hip_atomic_store(plo, lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        __hip_atomic_store(phi, hi, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                        atomicAdd(&L.nfill, 1u);
                        atomicAdd(&L.cnt[h], 1u);
                        done = true;
                    }
                } else if (cur == KMC_LOCKED64) {
                    // being published; examine it next trip
                } else if (cur == hi && __hip_atomic_load(plo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == lo) {
                    atomicAdd(&L.cnt[h], 1u);
                    done = true;
                } else { h = (h + 1) & M; probes++; }
            }
        }
    }
    if (to_global) stream_gadd<KW, CANON>(g, hi, lo, k, 1);
    return to_global;
}

// empty the wave's miss buffer through the probing insert (all 64 lanes take part)
template <int KW, bool CANON>
__device__ __attribute__((noinline)) u32 stream_drain(StreamLds<KW>& L, const GTable& g, int wv, int lane, u32 nbuf, int k) {
    u32 nglobal = 0;
    const bool lds_ok = __hip_atomic_load(&L.nfill, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < (u32)(StreamLds<KW>::LCAP * 3 / 4);
    for (u32 b = 0; b < nbuf; b += 64) {
        const bool act = b + lane < nbuf;
        u64 lo = 0, hi = 0;
        if (act) {
            lo = L.miss[wv][b + lane].lo;
            if constexpr (KW == 2) hi = L.miss[wv][b + lane].hi;
        }
        nglobal += lds_insert<KW, CANON>(L, g, hi, lo, act, lds_ok, k) ? 1u : 0u;
    }
    return nglobal;
}

// hot path: one k-mer per lane, straight-line (no branch, so that the sixteen steps of a chunk overlap:
// the next step's hash and LDS reads are issued while this step's compare waits).  Returns true when
// a valid window found its key in neither home slot; the caller collects those per chunk.
template <int KW>
__device__ __forceinline__ bool stream_probe(StreamLds<KW>& L, u64 hi, u64 lo, bool ok) {
    const u32 h = stream_home<KW>(hi, lo);
    bool m0, m1;
    if (KW == 1) {
        const u64 k0 = __hip_atomic_load(&L.slot[h].lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        const u64 k1 = __hip_atomic_load(&L.slot[h + 1].lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        m0 = k0 == lo;
        m1 = k1 == lo;
    } else {
        // (a slot being published shows LOCKED or EMPTY in its high word: a miss, settled by the drain;
        // the low word is read after the high word it was published before)
        const u64* s0 = reinterpret_cast<const u64*>(&L.slot[h]);
        const u64 h0 = __hip_atomic_load(&s0[1], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
        const u64 h1 = __hip_atomic_load(&s0[3], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
        const u64 l0 = __hip_atomic_load(&s0[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        const u64 l1 = __hip_atomic_load(&s0[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        m0 = h0 == hi && l0 == lo;
        m1 = h1 == hi && l1 == lo;
    }
    // branch-free add: every lane adds to one of its home counters, 1 on a hit of a valid window and 0 otherwise
    const bool hit = ok && (m0 || m1);
    atomicAdd(&L.cnt[h + (m0 ? 0u : 1u)], hit ? 1u : 0u);
    return ok && !hit;
}

// The same probe in two halves (one-word keys): issue = hash + the two home slots' loads, finish = compare + add.  The chunk
// loop issues window j + 1 before it finishes window j, so that a wave has two probes' LDS reads in flight instead of
// sitting out every read's latency (the straight-line version above was meant to overlap like that; the compiler put each
// window's s_waitcnt right behind its own reads -- 29 instructions per window, yet 260 SIMD cycles).  Measured: 16.9 -> 16.6 ms
// on the benchmark batch; both home slots in ONE ds_read2_b64 instead of two ds_read_b64: 17.3 (profiles/r03_stream_variants.txt).
struct StreamProbe1 { u32 h; u64 k0, k1; };
__device__ __forceinline__ StreamProbe1 stream_issue1(StreamLds<1>& L, u64 lo) {
    StreamProbe1 p;
    p.h = stream_home<1>(0, lo);
    p.k0 = __hip_atomic_load(&L.slot[p.h].lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    p.k1 = __hip_atomic_load(&L.slot[p.h + 1].lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    return p;
}
__device__ __forceinline__ bool stream_finish1(StreamLds<1>& L, const StreamProbe1& p, u64 lo, bool ok) {
    const bool m0 = p.k0 == lo, m1 = p.k1 == lo;
    const bool hit = ok && (m0 || m1);
    atomicAdd(&L.cnt[p.h + (m0 ? 0u : 1u)], hit ? 1u : 0u);
    return ok && !hit;
}

// wide bit masks for the validity smear: 64 bits cover the 48-base window of KW==1,
// 128 bits the 80-base window of KW==2
template <int KW> struct WMask;
template <> struct WMask<1> {
    u64 v;
    __device__ __forceinline__ static WMask make(u64 lo, u64) { return {lo}; }
    __device__ __forceinline__ WMask shl(int s) const { return {s >= 64 ? 0 : v << s}; }
    __device__ __forceinline__ WMask operator|(WMask o) const { return {v | o.v}; }
    __device__ __forceinline__ u32 bits16_at(int pos) const { return (u32)(v >> pos) & 0xFFFFu; }
};
template <> struct WMask<2> {
    u64 lo, hi;
    __device__ __forceinline__ static WMask make(u64 l, u64 h) { return {l, h}; }
    __device__ __forceinline__ WMask shl(int s) const {
        if (s == 0) return *this;
        if (s >= 128) return {0, 0};
        if (s >= 64) return {0, lo << (s - 64)};
        return {lo << s, (hi << s) | (lo >> (64 - s))};
    }
    __device__ __forceinline__ WMask operator|(WMask o) const { return {lo | o.lo, hi | o.hi}; }
    __device__ __forceinline__ u32 bits16_at(int pos) const {  // pos == 64 here
        return (u32)(hi >> (pos - 64)) & 0xFFFFu;
    }
};

// OR of m << i for i in [0, t)
template <int KW>
__device__ __forceinline__ WMask<KW> smear(WMask<KW> m, int t) {
    if (t <= 0) return WMask<KW>::make(0, 0);
    int cur = 1;
    while (cur * 2 <= t) { m = m | m.shl(cur); cur *= 2; }
    if (cur < t) m = m | m.shl(t - cur);
    return m;
}

// A wave holds 16 consecutive keys per lane (lane l: positions 16 l .. 16 l + 15 of its 1024-position chunk);
// stored as they are, one store instruction touches 64 lines of 128 B with 8 B each.  Transposed through LDS
// (rows of 32 lanes, 32 bits at a time, column index rotated by the row: two-way conflicts writing, none
// reading) every store instruction writes 512 contiguous bytes: out[i * 64 + lane] for i = 0..15.
__device__ __forceinline__ void stream_store_transposed(u32* tr, int lane, const u64 (&v)[16], u64* __restrict__ out) {
    u32 lo32[16], hi32[16];
#pragma unroll
    for (int part = 0; part < 2; ++part) {        // low / high 32 bits
#pragma unroll
        for (int half = 0; half < 2; ++half) {    // rows 0..31 / 32..63 -> outputs 0..7 / 8..15
            const int row = lane - 32 * half;
            if (row >= 0 && row < 32) {
#pragma unroll
                for (int j = 0; j < 16; ++j) tr[row * 16 + ((j + row) & 15)] = part ? (u32)(v[j] >> 32) : (u32)v[j];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int r = 4 * i + (lane >> 4), j = lane & 15;
                const u32 x = tr[r * 16 + ((j + r) & 15)];
                if (part) hi32[8 * half + i] = x; else lo32[8 * half + i] = x;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) out[i * 64 + lane] = ((u64)hi32[i] << 32) | lo32[i];
}
// Count the windows ending in chunks [chunk_begin, chunk_end) into the LDS / global tables (KMC_ALGO_STREAM).
template <int KW, bool CANON>
__global__ __launch_bounds__(KMC_STREAM_THREADS)
void kmc_stream_kernel(const uint8_t* __restrict__ bases, u64 n_bases, const u64* __restrict__ offsets,
                       u64 n_reads, int k, u64 chunk_begin, u64 chunk_end, u64 chunks_per_wave, u64 range_begin, GTable g) {
    constexpr int NW = 2 * KW + 1;  // window words: own + 2*KW preceding lanes
    __shared__ StreamLds<KW> L;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;

    for (int s = tid; s < StreamLds<KW>::LCAP; s += KMC_STREAM_THREADS) {
        if constexpr (KW == 1) L.slot[s].lo = KMC_EMPTY64; else { L.slot[s].hi = KMC_EMPTY64; L.slot[s].lo = 0; }
        L.cnt[s] = 0;
    }
    if (tid == 0) L.nfill = 0;
    __syncthreads();
    u32 nbuf = 0;  // fill of this wave's miss buffer (wave-uniform)

    const u64 gw = (u64)blockIdx.x * KMC_STREAM_WAVES + wv;
    // this launch covers chunks [chunk_begin, chunk_end) of the stream (windows ENDING there)
    u64 c0 = chunk_begin + gw * chunks_per_wave;
    u64 c1 = c0 + chunks_per_wave;
    if (c1 > chunk_end) c1 = chunk_end;

    u64 nk = 0, nglobal = 0;
    if (c0 < c1) {
        // uniform key masks
        const int kb = 2 * k;
        const u64 mask_lo = kb >= 64 ? ~0ull : ((1ull << kb) - 1);
        const u64 mask_hi = kb <= 64 ? 0ull : ((1ull << (kb - 64)) - 1);

        const u64 cfirst = c0 > 0 ? c0 - 1 : 0;  // warm-up chunk supplies the halo of chunk c0
        // first read-start >= first position (binary search, wave-uniform)
        u64 rbase;
        {
            const u64 target = cfirst * KMC_CHUNK;
            u64 lo_i = 0, hi_i = n_reads + 1;  // offsets has n_reads+1 entries
            while (lo_i < hi_i) {
                u64 mid = (lo_i + hi_i) >> 1;
                if (offsets[mid] < target) lo_i = mid + 1; else hi_i = mid;
            }
            rbase = lo_i;
        }
        u64 held = (rbase + lane <= n_reads) ? offsets[rbase + lane] : ~0ull;
        u32 consumed = 0;

        u32 pw = 0, pzb = 0;  // previous chunk's big-endian word and (z | b<<16)

        for (u64 c = cfirst; c < c1; ++c) {
            const u64 cb = c * KMC_CHUNK;
            const u64 pp = cb + 16u * lane;  // this lane's piece
            uint4 v = make_uint4(0, 0, 0, 0);
            if (pp < n_bases) v = *reinterpret_cast<const uint4*>(bases + pp);
            Enc16 e = encode16(v);
            u32 bad = 0;
            if (__builtin_amdgcn_ballot_w64((e.x0 | e.x1 | e.x2 | e.x3) != 0) != 0) bad = bad16_from(e);
            if (pp + 16 > n_bases) {  // bytes past the end of the batch never form windows
                u32 nvalid = pp < n_bases ? (u32)(n_bases - pp) : 0;
                bad |= (0xFFFFu << nvalid) & 0xFFFFu;
            }
            const u32 wbe = le_to_be(e.wle);

            // read starts of this chunk -> per-lane 16-bit mask, through the wave's LDS bitmap
            L.sbits[wv][lane] = 0;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const u64 cend = cb + KMC_CHUNK;
            for (;;) {
                bool in = (lane >= consumed) && (held < cend);
                if (in) {
                    u32 rel = (u32)(held - cb);
                    atomicOr(&L.sbits[wv][rel >> 4], 1u << (rel & 15));
                }
                consumed += (u32)__popcll(__builtin_amdgcn_ballot_w64(in));
                if (consumed < 64) break;
                rbase += 64;
                consumed = 0;
                held = (rbase + lane <= n_reads) ? offsets[rbase + lane] : ~0ull;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const u32 st = __hip_atomic_load(&L.sbits[wv][lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __builtin_amdgcn_wave_barrier();
            const u32 zb = ((st | bad) & 0xFFFFu) | (bad << 16);

            if (c >= c0) {
                // window words X[d] = word of lane-d (previous chunk for lane < d)
                u32 X[NW], ZB[NW];
                X[0] = wbe; ZB[0] = zb;
#pragma unroll
                for (int d = 1; d < NW; ++d) {
                    int src = (lane - d) & 63;
                    u32 a = __shfl(wbe, src), b = __shfl(pw, src);
                    u32 za = __shfl(zb, src), zbb = __shfl(pzb, src);
                    X[d] = lane >= d ? a : b;
                    ZB[d] = lane >= d ? za : zbb;
                }
                // validity: window invalid if a break (bad byte or read start) lies in its last
                // k-1 positions, or a bad byte in its first position
                u32 inv16;
                {
                    u64 zl = 0, zh = 0, bl = 0, bh = 0;
#pragma unroll
                    for (int d = 0; d < NW; ++d) {
                        int pos = 16 * (NW - 1 - d);
                        u64 z = ZB[d] & 0xFFFFu, b = ZB[d] >> 16;
                        if (pos < 64) { zl |= z << pos; bl |= b << pos; } else { zh |= z << (pos - 64); bh |= b << (pos - 64); }
                    }
                    WMask<KW> Z = WMask<KW>::make(zl, zh), B = WMask<KW>::make(bl, bh);
                    WMask<KW> inv = smear<KW>(Z, k - 1) | B.shl(k - 1);
                    inv16 = inv.bits16_at(16 * (NW - 1));
                }
                if (pp < range_begin) {  // windows ending before range_begin belong to an earlier launch
                    u64 nskip = range_begin - pp;
                    inv16 |= nskip >= 16 ? 0xFFFFu : ((1u << (u32)nskip) - 1u);
                }
                {  // (no per-lane skip of lanes without a valid window: the drain inside needs the whole wave)
                    // forward keys are counted; the flush picks the strand
                    u32 missmask = 0;  // bit j: this lane's window j missed both home slots
                    if constexpr (KW == 1) {
                        auto window = [&](int j) { return ((u64)alignbit(X[2], X[1], 30 - 2 * j) << 32 | alignbit(X[1], X[0], 30 - 2 * j)) & mask_lo; };
                        u64 cur = window(0);
                        StreamProbe1 pc = stream_issue1(L, cur);
#pragma unroll
                        for (int j = 0; j < 16; ++j) {
                            u64 nxt = 0;
                            StreamProbe1 pn = pc;
                            if (j + 1 < 16) { nxt = window(j + 1); pn = stream_issue1(L, nxt); }
                            missmask |= stream_finish1(L, pc, cur, !((inv16 >> j) & 1)) ? (1u << j) : 0u;
                            cur = nxt; pc = pn;
                        }
                    } else {
#pragma unroll
                    for (int j = 0; j < 16; ++j) {
                        const int s = 30 - 2 * j;
                        u32 f[2 * KW];
#pragma unroll
                        for (int m = 0; m < 2 * KW; ++m) f[m] = alignbit(X[m + 1], X[m], s);
                        const u64 flo = ((u64)f[1] << 32 | f[0]) & mask_lo;
                        u64 fhi = 0;
                        if constexpr (KW == 2) fhi = ((u64)f[3] << 32 | f[2]) & mask_hi;
                        const bool ok = !((inv16 >> j) & 1);
                        missmask |= stream_probe<KW>(L, fhi, flo, ok) ? (1u << j) : 0u;
                    }
                    }
                    {
                        nk += (u32)__popc(~inv16 & 0xFFFFu);
                        // the chunk's misses (first sight of a key, keys that did not fit their home bucket):
                        // one per lane and round into the wave's miss buffer; the drain inserts them
                        u64 mb;
                        while ((mb = __builtin_amdgcn_ballot_w64(missmask != 0)) != 0) {
                            if (missmask != 0) {
                                const int jm = __ffs((int)missmask) - 1;
                                missmask &= missmask - 1;
                                const u32 sh = 30u - 2u * (u32)jm;
                                u32 f[2 * KW];
#pragma unroll
                                for (int m = 0; m < 2 * KW; ++m) f[m] = alignbit(X[m + 1], X[m], sh);
                                const u32 idx = nbuf + __builtin_amdgcn_mbcnt_hi((u32)(mb >> 32), __builtin_amdgcn_mbcnt_lo((u32)mb, 0u));
                                L.miss[wv][idx].lo = ((u64)f[1] << 32 | f[0]) & mask_lo;
                                if constexpr (KW == 2) L.miss[wv][idx].hi = ((u64)f[3] << 32 | f[2]) & mask_hi;
                            }
                            nbuf += (u32)__popcll(mb);
                            if (nbuf > KMC_STREAM_MISSBUF - 64) {  // no room for another round
                                nglobal += stream_drain<KW, CANON>(L, g, wv, lane, nbuf, k);
                                nbuf = 0;
                            }
                        }
                    }
                }
            }
            pw = wbe;
            pzb = zb;
        }
        if (nbuf) { nglobal += stream_drain<KW, CANON>(L, g, wv, lane, nbuf, k); nbuf = 0; }
    }
    nk = wave_sum_u64(nk);
    nglobal = wave_sum_u64(nglobal);
    if (lane == 0 && nk) atomicAdd((unsigned long long*)&g.counters[KMC_CTR_KMERS], nk);
    if (lane == 0 && nglobal) atomicAdd((unsigned long long*)&g.counters[KMC_CTR_BADBASE], nglobal);  // "direct" k-mers

    __syncthreads();
    for (int s = tid; s < StreamLds<KW>::LCAP; s += KMC_STREAM_THREADS) {
        const u32 c = L.cnt[s];
        if (c) {
            u64 hi = 0;
            if constexpr (KW == 2) hi = L.slot[s].hi;
            stream_gadd<KW, CANON>(g, hi, L.slot[s].lo, k, c);
        }
    }
}
