// kmc_stream.cuh -- KMC_ALGO_STREAM: the general counting kernel (any read lengths, any bytes).
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
// Accumulation.  Each workgroup owns an LDS open-addressing partial histogram (ds_cmpst claim,
// ds_add count); keys that do not fit go straight to the global table with device-scope
// atomics; at the end the LDS table is flushed with one global atomic per distinct key.
#pragma once
#include "kmc_device.cuh"

#define KMC_STREAM_THREADS 1024
#define KMC_STREAM_WAVES (KMC_STREAM_THREADS / 64)
#define KMC_CHUNK 1024

template <int KW> struct StreamLds {
    static constexpr int LCAP = (KW == 1) ? 8192 : 4096;  // 96 KB / 80 KB of keys+counts: one workgroup per CU
    u64 lo[LCAP];
    u64 hi[KW == 2 ? LCAP : 1];
    u32 cnt[LCAP];
    u32 sbits[KMC_STREAM_WAVES][64];
    u32 nfill;
};

// insert-or-increment in the workgroup's LDS table; falls through to the global table when the
// probe budget is exhausted or the table is (nearly) full.  Same wave-uniform loop shape as
// gtable_add (see there for why).
template <int KW>
__device__ __forceinline__ bool lds_add(StreamLds<KW>& L, const GTable& g, u64 hi, u64 lo, bool lds_ok) {
    constexpr u32 M = StreamLds<KW>::LCAP - 1;
    u32 h = kmc_hash32<KW>(hi, lo) & M;
    int probes = lds_ok ? 0 : 1000;
    bool done = false, to_global = false;
    u32 trips = 0;
    while (__builtin_amdgcn_ballot_w64(!done) != 0) {
        if (!done) {
            if (probes >= 24 || ++trips > (1u << 20)) {
                to_global = true;
                done = true;
            } else if (KW == 1) {
                u64 cur = __hip_atomic_load(&L.lo[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (cur == KMC_EMPTY64) {
                    cur = atomicCAS((unsigned long long*)&L.lo[h], KMC_EMPTY64, lo);
                    if (cur == KMC_EMPTY64) { atomicAdd(&L.nfill, 1u); cur = lo; }
                }
                if (cur == lo) { atomicAdd(&L.cnt[h], 1u); done = true; }
                else { h = (h + 1) & M; probes++; }
            } else {
                u64 cur = __hip_atomic_load(&L.hi[h], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (cur == KMC_EMPTY64) {
                    u64 old = atomicCAS((unsigned long long*)&L.hi[h], KMC_EMPTY64, KMC_LOCKED64);
                    if (old == KMC_EMPTY64) {
                        __hip_atomic_store(&L.lo[h], lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        __hip_atomic_store(&L.hi[h], hi, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                        atomicAdd(&L.nfill, 1u);
                        atomicAdd(&L.cnt[h], 1u);
                        done = true;
                    }
                } else if (cur == KMC_LOCKED64) {
                    // being published; examine it next trip
                } else if (cur == hi && __hip_atomic_load(&L.lo[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == lo) {
                    atomicAdd(&L.cnt[h], 1u);
                    done = true;
                } else { h = (h + 1) & M; probes++; }
            }
        }
    }
    if (to_global) gtable_add<KW>(g, hi, lo, 1);
    return to_global;  // counted with a global atomic instead of the LDS partial table
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

// LDS of the extract-only variant (SINK == 1): just the per-wave read-start bitmaps
struct StreamLdsLite {
    u32 sbits[KMC_STREAM_WAVES][64];
};
template <int KW, int SINK> struct StreamLdsSel { typedef StreamLds<KW> type; };
template <int KW> struct StreamLdsSel<KW, 1> { typedef StreamLdsLite type; };

// SINK == 0: count into the LDS / global tables (KMC_ALGO_STREAM).
// SINK == 1: extraction only (front end of KMC_ALGO_SORT): write ONE key per base position of the
//            launch's chunk range to out_lo/out_hi -- the k-mer ending there, or all-ones where no
//            valid window ends -- 16 consecutive keys per lane, fully coalesced, no atomics.
template <int KW, bool CANON, int SINK>
__global__ __launch_bounds__(KMC_STREAM_THREADS)
void kmc_stream_kernel(const uint8_t* __restrict__ bases, u64 n_bases, const u64* __restrict__ offsets,
                       u64 n_reads, int k, u64 chunk_begin, u64 chunk_end, u64 chunks_per_wave, u64 range_begin, GTable g,
                       u64* __restrict__ out_hi, u64* __restrict__ out_lo) {
    constexpr int NW = 2 * KW + 1;  // window words: own + 2*KW preceding lanes
    __shared__ typename StreamLdsSel<KW, SINK>::type L;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;

    if constexpr (SINK == 0) {
        for (int s = tid; s < StreamLds<KW>::LCAP; s += KMC_STREAM_THREADS) {
            if (KW == 1) L.lo[s] = KMC_EMPTY64; else { L.hi[s] = KMC_EMPTY64; L.lo[s] = 0; }
            L.cnt[s] = 0;
        }
        if (tid == 0) L.nfill = 0;
        __syncthreads();
    }

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
        // rc pre-shift: P = 32*NW - 30 - 2k
        const int P = 32 * NW - 30 - kb;
        const int Pq = P >> 5, Pr = P & 31;

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
                if (SINK == 1 || inv16 != 0xFFFFu) {
                    // rc stream words from the LSB end: Yw[m] = rc word of lane-(NW-1-m)
                    u32 Yp[NW + 1];
                    if (CANON) {
                        u32 Yw[2 * NW + 1];
#pragma unroll
                        for (int m = 0; m < NW; ++m) Yw[m] = rc_word_be(X[NW - 1 - m]);
#pragma unroll
                        for (int m = NW; m < 2 * NW + 1; ++m) Yw[m] = 0;
#pragma unroll
                        for (int m = 0; m < NW; ++m) {
                            u32 r = 0;
#pragma unroll
                            for (int q = 0; q < NW; ++q)
                                if (q == Pq) r = alignbit(Yw[m + q + 1], Yw[m + q], Pr);
                            Yp[m] = r;
                        }
                        Yp[NW] = 0;
                    }
                    bool lds_ok = false;
                    if constexpr (SINK == 0)
                        lds_ok = __hip_atomic_load(&L.nfill, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < (u32)(StreamLds<KW>::LCAP * 7 / 8);
                    u64* const o_lo = SINK == 1 ? out_lo + (pp - chunk_begin * KMC_CHUNK) : nullptr;
                    u64* const o_hi = (SINK == 1 && KW == 2) ? out_hi + (pp - chunk_begin * KMC_CHUNK) : nullptr;
#pragma unroll
                    for (int j = 0; j < 16; ++j) {
                        const int s = 30 - 2 * j;
                        u32 f[2 * KW];
#pragma unroll
                        for (int m = 0; m < 2 * KW; ++m) f[m] = alignbit(X[m + 1], X[m], s);
                        u64 flo = ((u64)f[1] << 32 | f[0]) & mask_lo, fhi = 0;
                        if constexpr (KW == 2) fhi = ((u64)f[3] << 32 | f[2]) & mask_hi;
                        u64 klo = flo, khi = fhi;
                        if (CANON) {
                            u32 r[2 * KW];
#pragma unroll
                            for (int m = 0; m < 2 * KW; ++m) r[m] = alignbit(Yp[m + 1], Yp[m], 2 * j);
                            u64 rlo = ((u64)r[1] << 32 | r[0]) & mask_lo, rhi = 0;
                            if constexpr (KW == 2) rhi = ((u64)r[3] << 32 | r[2]) & mask_hi;
                            if (key_less(rhi, rlo, fhi, flo)) { klo = rlo; khi = rhi; }
                        }
                        const bool ok = !((inv16 >> j) & 1);
                        if constexpr (SINK == 0) {
                            if (ok) { nglobal += lds_add<KW>(L, g, khi, klo, lds_ok) ? 1u : 0u; nk++; }
                        } else {
                            o_lo[j] = ok ? klo : ~0ull;
                            if (KW == 2) o_hi[j] = ok ? khi : ~0ull;
                            nk += ok;
                        }
                    }
                }
            }
            pw = wbe;
            pzb = zb;
        }
    }
    nk = wave_sum_u64(nk);
    nglobal = wave_sum_u64(nglobal);
    if (lane == 0 && nk) atomicAdd((unsigned long long*)&g.counters[KMC_CTR_KMERS], nk);
    if (lane == 0 && nglobal) atomicAdd((unsigned long long*)&g.counters[KMC_CTR_BADBASE], nglobal);  // "direct" k-mers

    if constexpr (SINK == 0) {
        __syncthreads();
        for (int s = tid; s < StreamLds<KW>::LCAP; s += KMC_STREAM_THREADS) {
            u32 c = L.cnt[s];
            if (c) gtable_add<KW>(g, KW == 2 ? L.hi[s] : 0ull, L.lo[s], c);
        }
    }
}
