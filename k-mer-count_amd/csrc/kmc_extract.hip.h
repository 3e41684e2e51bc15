// kmc_extract.hip.h -- front end of KMC_ALGO_SORT: one packed key per base position (the k-mer ending there, or
// all-ones where no valid window ends), FUSED with the first level of the MSD radix sort's bookkeeping.
//
// The reference materialises every window (k-mer-count/src/main.rs:63-81: push of a 54-byte String) and then groups
// by sorting (main.rs:9-40,84,87).  Here the window loop writes 8-byte (16-byte for k >= 32) keys, and while a key is
// still in a register the kernel also counts it into the level-0 digit histogram of its 65536-key range
// (kmc_msd.hip.h: 1025 counters per range) and folds it into the range's AND / OR words -- what kmc_msd_hist_kernel would
// otherwise compute by reading every key back from HBM (round 2: 1.46 ms per GB of all-distinct reads for that
// re-read alone).  AND / OR stand in for the range's smallest / largest key: the sort only asks them "are all keys
// equal" and "which is the highest bit in which two keys differ", and min ^ max and AND ^ OR answer both alike.
//
// Work split: a workgroup of 16 waves takes whole ranges (64 chunks of 1024 positions); wave w extracts chunks
// 4 w .. 4 w + 3 of the range (plus one warm-up chunk for the halo, as kmc_stream_kernel does), all waves count into one
// LDS histogram, and the row leaves with one coalesced 4 KB store.  Window extraction, validity (read starts, non-ACGT
// bytes) and canonical strand are those of kmc_stream_kernel (same helper functions).
//
// Stores: a lane owns 16 consecutive positions.  One-word keys leave through the 16-key LDS transpose of
// kmc_stream.hip.h (512 contiguous bytes per store instruction).  Two-word keys go half a lane's keys at a time (8 low
// and 8 high words in registers instead of 16 + 16), each half through an 8-key transpose whose store instructions
// write eight full 64-byte segments -- round 2 stored the high words straight from the lanes, 8 bytes into each of 64
// different lines per instruction (7.5 ms per GB at k = 63 against 1.7 ms for one-word keys).
#pragma once
#include "kmc_stream.hip.h"
#include "kmc_msd.hip.h"

template <int KW> struct ExtractLds {
    u32 sbits[KMC_STREAM_WAVES][64];
    alignas(16) u64 tr[KMC_STREAM_WAVES][KW == 1 ? 256 : 512];   // per wave: 32 x 16 words (one-word keys) / 64 rows x 8 keys
    u32 hist[KMC_MSD_NB + 3];
    u64 s_or[KMC_STREAM_WAVES][2], s_and[KMC_STREAM_WAVES][2];
};

// 8 keys per lane -- lane l holds positions 16 l + 8 h + j, j = 0..7, of the wave's 1024-position chunk -- to eight store
// instructions of eight 64-byte segments each: instruction i, lane m writes position 16 (8 i + (m >> 3)) + 8 h + (m & 7).
// LDS image: row l = 8 keys; the four 16-byte pairs of a row are rotated by l >> 1, which makes the ds_write_b128 of
// eight neighbouring lanes hit 32 different banks; a read instruction covers four whole rows per 32 lanes (every bank once).
__device__ __forceinline__ void extract_store_half(u64* tr, int lane, const u64 (&v)[8], u64* __restrict__ out_chunk, int h) {
    typedef unsigned long long u64x2_t __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        u64x2_t p = {v[2 * q], v[2 * q + 1]};
        *reinterpret_cast<u64x2_t*>(&tr[lane * 8 + 2 * ((q + (lane >> 1)) & 3)]) = p;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    u64 x[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int l = 8 * i + (lane >> 3), j = lane & 7;
        x[i] = tr[l * 8 + 2 * (((j >> 1) + (l >> 1)) & 3) + (j & 1)];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int i = 0; i < 8; ++i) out_chunk[16 * (8 * i + (lane >> 3)) + 8 * h + (lane & 7)] = x[i];
}

// Chunks [chunk_begin + 64 r, chunk_begin + 64 r + 64) are range r of this launch, r < n_ranges; its keys go to
// out_lo / out_hi [65536 r, 65536 r + 65536), its histogram to range hist_r0 + r of hist (msd_hist_idx), its AND / OR words to
// rand[2 r ..] / ror[2 r ..]
// ({high, low} word, as kmc_msd_hist_kernel writes rmin / rmax).  Chunks at or past chunk_end (the padding of the last
// range) and positions at or past n_bases hold filler; windows ending before range_begin belong to an earlier launch.
template <int KW, bool CANON>
__global__ __launch_bounds__(KMC_STREAM_THREADS)
void kmc_extract_hist_kernel(const uint8_t* __restrict__ bases, u64 n_bases, const u64* __restrict__ offsets, u64 n_reads, int k,
                             u64 chunk_begin, u64 chunk_end, u64 range_begin, u32 n_ranges, u64* __restrict__ counters,
                             u64* __restrict__ out_hi, u64* __restrict__ out_lo, u32* __restrict__ hist, u32 hist_r0,
                             u64* __restrict__ rand_, u64* __restrict__ ror_) {
    constexpr int NW = 2 * KW + 1;  // window words: own + 2*KW preceding lanes
    extern __shared__ __align__(16) unsigned char extract_smem[];
    ExtractLds<KW>& L = *reinterpret_cast<ExtractLds<KW>*>(extract_smem);
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int kb = 2 * k;
    const u64 mask_lo = kb >= 64 ? ~0ull : ((1ull << kb) - 1);
    const u64 mask_hi = kb <= 64 ? 0ull : ((1ull << (kb - 64)) - 1);
    const int P = 32 * NW - 30 - kb;   // rc pre-shift
    const int Pq = P >> 5, Pr = P & 31;
    // level-0 digit of the sort: bits [kb - w, kb) with w = min(kb, 10)
    const int dshift = kb > KMC_MSD_BITS ? kb - KMC_MSD_BITS : 0;
    const u32 dmask = kb >= KMC_MSD_BITS ? (u32)KMC_MSD_ND - 1u : (1u << kb) - 1u;
    u64 nk = 0;

    for (u32 r = blockIdx.x; r < n_ranges; r += gridDim.x) {
        __syncthreads();   // (the previous range's histogram row has left)
        for (u32 d = tid; d < KMC_MSD_NB; d += KMC_STREAM_THREADS) L.hist[d] = 0;
        __syncthreads();
        u64 a_lo = ~0ull, a_hi = ~0ull, o_lo = 0, o_hi = 0;   // AND / OR of this lane's valid keys
        const u64 c0 = chunk_begin + (u64)r * (KMC_MSD_RANGE / KMC_CHUNK) + 4u * (u64)wv, c1 = c0 + 4;
        const u64 cfirst = c0 > 0 ? c0 - 1 : 0;  // warm-up chunk supplies the halo of chunk c0
        // first read start >= first position (binary search, wave-uniform)
        u64 rbase;
        {
            const u64 target = cfirst * KMC_CHUNK;
            u64 lo_i = 0, hi_i = n_reads + 1;  // offsets has n_reads + 1 entries
            while (lo_i < hi_i) {
                const u64 mid = (lo_i + hi_i) >> 1;
                if (offsets[mid] < target) lo_i = mid + 1; else hi_i = mid;
            }
            rbase = lo_i;
        }
        u64 held = (rbase + lane <= n_reads) ? offsets[rbase + lane] : ~0ull;
        u32 consumed = 0;
        u32 pw = 0, pzb = 0;  // previous chunk's big-endian word and (z | b << 16)
        for (u64 c = cfirst; c < c1; ++c) {
            const u64 cb = c * KMC_CHUNK;
            const u64 pp = cb + 16u * lane;  // this lane's piece
            const bool live = c < chunk_end;  // (wave-uniform) padding chunks of the last range: filler only
            uint4 v = make_uint4(0, 0, 0, 0);
            if (live && pp < n_bases) v = *reinterpret_cast<const uint4*>(bases + pp);
            Enc16 e = encode16(v);
            u32 bad = 0;
            if (__builtin_amdgcn_ballot_w64((e.x0 | e.x1 | e.x2 | e.x3) != 0) != 0) bad = bad16_from(e);
            if (!live) bad = 0xFFFFu;
            else if (pp + 16 > n_bases) {  // bytes past the end of the batch never form windows
                const u32 nvalid = pp < n_bases ? (u32)(n_bases - pp) : 0;
                bad |= (0xFFFFu << nvalid) & 0xFFFFu;
            }
            const u32 wbe = le_to_be(e.wle);
            // read starts of this chunk -> per-lane 16-bit mask, through the wave's LDS bitmap
            L.sbits[wv][lane] = 0;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const u64 cend = cb + KMC_CHUNK;
            for (;;) {
                const bool in = (lane >= consumed) && (held < cend);
                if (in) {
                    const u32 rel = (u32)(held - cb);
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
                // window words X[d] = word of lane - d (previous chunk for lane < d)
                u32 X[NW], ZB[NW];
                X[0] = wbe; ZB[0] = zb;
#pragma unroll
                for (int d = 1; d < NW; ++d) {
                    const int src = (lane - d) & 63;
                    const u32 a = __shfl(wbe, src), b = __shfl(pw, src);
                    const u32 za = __shfl(zb, src), zbb = __shfl(pzb, src);
                    X[d] = lane >= d ? a : b;
                    ZB[d] = lane >= d ? za : zbb;
                }
                // validity: a window is invalid if a break (bad byte or read start) lies in its last k - 1 positions,
                // or a bad byte in its first position
                u32 inv16;
                {
                    u64 zl = 0, zh = 0, bl = 0, bh = 0;
#pragma unroll
                    for (int d = 0; d < NW; ++d) {
                        const int pos = 16 * (NW - 1 - d);
                        const u64 z = ZB[d] & 0xFFFFu, b = ZB[d] >> 16;
                        if (pos < 64) { zl |= z << pos; bl |= b << pos; } else { zh |= z << (pos - 64); bh |= b << (pos - 64); }
                    }
                    const WMask<KW> Z = WMask<KW>::make(zl, zh), B = WMask<KW>::make(bl, bh);
                    const WMask<KW> inv = smear<KW>(Z, k - 1) | B.shl(k - 1);
                    inv16 = inv.bits16_at(16 * (NW - 1));
                }
                if (pp < range_begin) {  // windows ending before range_begin belong to an earlier launch
                    const u64 nskip = range_begin - pp;
                    inv16 |= nskip >= 16 ? 0xFFFFu : ((1u << (u32)nskip) - 1u);
                }
                // rc stream words from the LSB end
                u32 Yp[NW + 1];
                if (CANON) {
                    u32 Yw[2 * NW + 1];
#pragma unroll
                    for (int m = 0; m < NW; ++m) Yw[m] = rc_word_be(X[NW - 1 - m]);
#pragma unroll
                    for (int m = NW; m < 2 * NW + 1; ++m) Yw[m] = 0;
#pragma unroll
                    for (int m = 0; m < NW; ++m) {
                        u32 rr = 0;
#pragma unroll
                        for (int q = 0; q < NW; ++q)
                            if (q == Pq) rr = alignbit(Yw[m + q + 1], Yw[m + q], Pr);
                        Yp[m] = rr;
                    }
                    Yp[NW] = 0;
                }
                // the wave's 1024 keys of this chunk start at the position of lane 0's first key
                const u64 obase = cb - chunk_begin * KMC_CHUNK;
                auto key_at = [&](int j, u64& khi, u64& klo) -> bool {
                    const int s = 30 - 2 * j;
                    u32 f[2 * KW];
#pragma unroll
                    for (int m = 0; m < 2 * KW; ++m) f[m] = alignbit(X[m + 1], X[m], s);
                    const u64 flo = ((u64)f[1] << 32 | f[0]) & mask_lo;
                    u64 fhi = 0;
                    if constexpr (KW == 2) fhi = ((u64)f[3] << 32 | f[2]) & mask_hi;
                    klo = flo; khi = fhi;
                    if (CANON) {
                        u32 rw[2 * KW];
#pragma unroll
                        for (int m = 0; m < 2 * KW; ++m) rw[m] = alignbit(Yp[m + 1], Yp[m], 2 * j);
                        const u64 rlo = ((u64)rw[1] << 32 | rw[0]) & mask_lo;
                        u64 rhi = 0;
                        if constexpr (KW == 2) rhi = ((u64)rw[3] << 32 | rw[2]) & mask_hi;
                        if (key_less(rhi, rlo, fhi, flo)) { klo = rlo; khi = rhi; }
                    }
                    const bool ok = !((inv16 >> j) & 1);
                    // level-0 bookkeeping of the sort while the key is in registers
                    const u32 d = ok ? (msd_bits<KW>(khi, klo, dshift) & dmask) : (u32)KMC_MSD_ND;
                    atomicAdd(&L.hist[d], 1u);
                    if (ok) {
                        a_lo &= klo; o_lo |= klo;
                        if constexpr (KW == 2) { a_hi &= khi; o_hi |= khi; }
                    } else {
                        klo = ~0ull; khi = ~0ull;
                    }
                    return ok;
                };
                if constexpr (KW == 1) {
                    u64 vlo[16];
#pragma unroll
                    for (int j = 0; j < 16; ++j) { u64 khi; nk += key_at(j, khi, vlo[j]) ? 1u : 0u; }
                    stream_store_transposed(reinterpret_cast<u32*>(L.tr[wv]), lane, vlo, out_lo + obase);
                } else {
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        u64 vlo[8], vhi[8];
#pragma unroll
                        for (int j = 0; j < 8; ++j) nk += key_at(8 * h + j, vhi[j], vlo[j]) ? 1u : 0u;
                        extract_store_half(L.tr[wv], lane, vlo, out_lo + obase, h);
                        extract_store_half(L.tr[wv], lane, vhi, out_hi + obase, h);
                    }
                }
            }
            pw = wbe;
            pzb = zb;
        }
        // the range's AND / OR words
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            a_lo &= __shfl_xor(a_lo, o); o_lo |= __shfl_xor(o_lo, o);
            if (KW == 2) { a_hi &= __shfl_xor(a_hi, o); o_hi |= __shfl_xor(o_hi, o); }
        }
        if (lane == 0) { L.s_and[wv][0] = a_hi; L.s_and[wv][1] = a_lo; L.s_or[wv][0] = o_hi; L.s_or[wv][1] = o_lo; }
        __syncthreads();
        for (u32 d = tid; d < KMC_MSD_NB; d += KMC_STREAM_THREADS) hist[msd_hist_idx((size_t)hist_r0 + r, d)] = L.hist[d];
        if (tid == 0) {
            u64 ah = ~0ull, al = ~0ull, oh = 0, ol = 0;
            for (int w = 0; w < KMC_STREAM_WAVES; ++w) { ah &= L.s_and[w][0]; al &= L.s_and[w][1]; oh |= L.s_or[w][0]; ol |= L.s_or[w][1]; }
            if (KW == 1) { ah = 0; oh = 0; }
            // (a range without a valid key: AND = all ones, OR = 0 -- "smallest" above "largest", which is also what
            //  kmc_msd_hist_kernel leaves for such a range: neutral in the fold)
            if (KW == 1 && al == ~0ull && ol == 0) { ah = ~0ull; }
            rand_[2 * (size_t)r] = ah; rand_[2 * (size_t)r + 1] = al;
            ror_[2 * (size_t)r] = oh; ror_[2 * (size_t)r + 1] = ol;
        }
    }
    nk = wave_sum_u64(nk);
    if (lane == 0 && nk) atomicAdd((unsigned long long*)&counters[KMC_CTR_KMERS], nk);
}
