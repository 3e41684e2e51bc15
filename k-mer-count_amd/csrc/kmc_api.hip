// kmc_api.hip -- the C ABI of libkmc.so (include/kmc.h): context management, batch
// scheduling, table growth, finalisation (compact + device radix sort), export.
//
// One ctx == one GPU.  All work is queued on the ctx stream; kernels are the hand-written
// gfx950 kernels of kmc_stream.hip.h / kmc_walk.hip.h / kmc_table.hip.h.  There is no CPU path: if
// the HIP runtime has no device, kmc_create fails.
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <cstring>

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <climits>
#include <chrono>
#include <deque>
#include <thread>
#include <new>
#include <string>
#include <vector>

#include "../../include/kmc.h"
#include "kmc_device.hip.h"
#include "kmc_stream.hip.h"
#include "kmc_synth.hip.h"
#include "kmc_table.hip.h"
#include "kmc_walk.hip.h"
#include "kmc_sklog.hip.h"
#include "kmc_lr.hip.h"
#include "kmc_msd.hip.h"
#include "kmc_extract.hip.h"
#include "kmc_peak.hip.h"
#include "kmc_ingest.h"

namespace {

thread_local char g_create_err[512] = {0};

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
};

struct Table {
    u64 *hi = nullptr, *lo = nullptr, *cnt = nullptr, *mid = nullptr;  // (mid: three-word keys of the (k+16)-mer table, k >= 48)
    u64 cap = 0;
};

}  // namespace

struct kmc_ctx {
    kmc_config cfg{};
    int KW = 1;     // key words
    int klen = 0;   // characters per key (k, or 54 in LR mode)
    hipStream_t stream = nullptr;
    bool own_stream = false;
    char err[512] = {0};

    Table tab;
    u64* d_counters = nullptr;      // KMC_CTR_N u64
    u64* h_counters = nullptr;      // pinned mirror
    u64* occ_list = nullptr;        // first KMC_OCC_LIST_CAP claimed slots (fast finalize of small tables)
    u64 *occ_key_lo = nullptr, *occ_key_hi = nullptr;   // ... and their keys, dense (GTable::occ_key_*)
    u32* fin_rank = nullptr;        // ticket counter of kmc_small_finalize_kernel (zero between launches)
    u64* d_mirror = nullptr;        // h_counters as the device sees it (the finalize kernel publishes the counters there)
    u64* h_restore = nullptr;       // pinned: the counters to put back when a drained table is filled again (undrain)
    // planner invariant (kmc_stats.n_planner_stale): every kernel queued OUTSIDE the count launches' own accounting that
    // changes the table -- the (k+16)-mer unfold, merges -- bumps table_epoch; a poll records the epoch it has seen; a
    // risky launch must save a table whose counters were polled at the current epoch
    u64 table_epoch = 0, polled_epoch = 0;
    u64 fin_seq = 0;                // number of the last kmc_small_finalize_kernel launch (the kernel publishes it with its result)
    bool async_fin = false;         // kmc_finalize_async: a finalize is queued whose outcome the host has not looked at yet
    bool drained = false;           // the last kmc_finalize emptied the table into the sorted view (kmc_small_finalize_kernel):
                                    // table and device counters are as after kmc_reset, h_counters hold the true totals
    u64 *spill_hi = nullptr, *spill_lo = nullptr, *spill_cnt = nullptr;
    u64 spill_cap = 0;

    // staging for host batches
    DevBuf st_bases, st_offsets;
    // sorted view
    DevBuf o_hi, o_lo, o_cnt, t_hi, t_lo, t_cnt, t_idx0, p_hi, p_lo, p_cnt;
    u64 n_sorted = 0;
    bool sorted_valid = false;
    // walk-kernel workspace
    DevBuf walk_ws;
    DevBuf vr_reads, vr_cnt, vr_pos;  // pieces of long reads for the walk kernel: [starts | ends], per-read counts and their scan
    DevBuf walk_memo;  // two shared memo snapshots + dense counters, kept across launches (kmc_walk.hip.h)
    int memo_parity = 0;  // snapshot slot the next walk launch reads
    bool walk_ws_clean = false;  // workspace header + dense counters are zero (left so by kmc_walk_tail_kernel)
    // KMC_ALGO_SORT: scratch for one sub-batch and the sorted (key,count) runs produced so far
    DevBuf s_lo[2], s_hi[2];
    // the walk kernel's log of steps that fell off its LDS memo (kmc_walk.hip.h SkLog, kmc_sklog.hip.h): per-workgroup spans,
    // their fill counts, the 1024 hash bins the records are partitioned into, the bins' cursors
    DevBuf lg_rec, lg_count, lg_bins, lg_cursor;
    bool sklog_on = false;   // this data source overflows the LDS memo (a poll saw (k+16)-mers in the second-level table): log from now on
    DevBuf a_hist, a_rand, a_ror;   // level-0 histogram rows / AND / OR words of the accumulated key ranges (kmc_extract.hip.h)
    // KMC_ALGO_SORT accumulates: a batch only EXTRACTS its keys behind those of the batches before it (s_lo[0] /
    // s_hi[0]); they are sorted into ONE run when somebody needs the result (kmc_finalize, a reduce) or when 2^31
    // positions have come together.  (Sorting batch by batch left one run per batch -- 16 for a 1 GB file read in
    // 64 MB chunks -- and kmc_finalize then merged them by sorting everything once more: 0.9 s for 760 M 63-mers.)
    u64 acc_n = 0;           // key positions accumulated and not yet sorted
    u64 acc_hint = 0;        // positions the caller expects in all (kmc_count_file: the file size); sizes the first allocation
    DevBuf lr_rank;  // LR mode: rank of every position's 27-mer among the batch's distinct 27-mers
    // hand-written MSD radix sort (kmc_msd.hip.h): per-range histograms, segment lists, terminals
    DevBuf m_hist, m_stot, m_bsum, m_rmin, m_rmax, m_seg[2], m_first, m_cbase, m_skip, m_term, m_ord, m_bitmap, m_rank, m_nd, m_base, m_ctl, m_clist, m_w[2];
    MsdCtl* h_ctl = nullptr;  // pinned mirror of the sort's device counters
    struct Run { u64 *hi = nullptr, *lo = nullptr, *cnt = nullptr; u64 n = 0, cap = 0; u64 total = 0; bool total_known = false; };
    std::vector<Run> runs;       // live runs
    std::vector<Run> run_pool;   // buffers of dropped runs, reused (multi-GB hipMalloc/hipFree per batch is slow)
    Run view_run;                // the merged, sorted view built by the last kmc_finalize (table entries + runs)
    const u64 *v_hi = nullptr, *v_lo = nullptr, *v_cnt = nullptr;  // the sorted view of the last finalize
    bool msd_dup_heavy = false;  // the last large unweighted sort collapsed its keys more than fourfold (leaf size of two-word sorts)
    bool prefer_sort = false;  // AUTO: the data source proved high-cardinality  // per-workgroup memo slots, kept across launches (kmc_walk.hip.h)

    // hipEvent pairs bracketing every count-kernel launch, batch by batch: a batch's events are read once they have
    // completed (harvest_timing), possibly several batches later -- a caller that never synchronises this ctx (the
    // multi-GPU step: count, pack, reset) still gets every batch's kernel time into kernel_ms_lifetime
    struct TimedBatch { std::vector<hipEvent_t> ev; int algo = 0; u64 n_bases = 0; u32 count_launches = 0; };
    std::deque<TimedBatch> tb;                // batches whose events have not been read yet (front = oldest)
    // KMC_ALGO_AUTO chooses by MEASURED cost: kernel milliseconds per base of this ctx's recent walk-path batches (walk
    // kernel + (k+16)-mer unfold + the table merge at finalize) and sort-path batches (< 0: not measured yet)
    double walk_ms_per_base = -1.0, sort_ms_per_base = -1.0;
    bool sort_by_cost = false;   // AUTO: the last comparison of the two rates said "sort" (prefer_sort: structural -- the source overflowed everything)
    std::vector<hipEvent_t> ev_free;          // events to reuse
    kmc_stats st{};
    int fin_parity = 0;    // which OUT/SUM counter pair the next kmc_finalize uses
    u64 fin_hint = 0;      // table entries at the last kmc_finalize (sizes the next speculative small-table finalize)
    // The rank sort of kmc_small_finalize_kernel is quadratic: 16 us for 1 k keys, 0.24 ms for 24 k, 0.8 ms for 68 k (measured).  The
    // weighted radix sort that larger tables take costs 0.25-0.3 ms at that size (a dozen small launches, two polls): tables
    // that were larger than this at the last finalize go there directly.  (KMC_FIN_SMALL_MAX overrides: the parity test of
    // the kernel's size boundaries runs it up to its limit, KMC_FIN_KERNEL_MAX.)
    u64 fin_small_max = 40000;
    bool view_unsynced = false;   // the last small-table finalize was waited for through the mirror, not the stream (poll_fin)
    bool batch_pending = false;  // a COUNT kernel (unknown number of new keys) is queued since the last poll
    u64 unpolled_adds = 0;       // upper bound of keys added by merge kernels since the last poll
    bool walk_overflowed = false;  // the last WALK/STREAM launches counted >5% of their k-mers with global atomics
                                   // (memo / LDS table overflow = high-cardinality input)
    u64 direct_seen = 0, kmers_seen = 0;
    bool pending = false;  // a batch has been queued since the last counter poll
    double rho_last = 0.0; // same, over the most recent sub-batch
    double rho_hist = -1.0; // new keys per k-mer of the previous batch as a whole (< 0: no history); survives kmc_reset
    bool b_open = false; double b_rho_max = 0.0; u64 b_occ0 = 0, b_kmers = 0;  // the batch whose last launch is still unobserved
    double rho_max = 0.0;  // largest observed (new distinct) / (k-mers) over a sub-batch
    int n_cu = 256;
    // A launch sized by a prediction (more k-mers than the table and spill area absorb for certain) is
    // "risky": the table is saved first, and if the spill area overflows the table is put back and the
    // rest of the batch is counted by the sort path, which needs no table (recover_overflow).
    struct Risky {
        bool armed = false;
        int mode = 0;                 // 1: entries listed in occ_list; 2: whole table copied
        bool empty = false;           // mode 1 and the table held nothing: there is nothing to save (no kernel)
        const uint8_t* d_bases = nullptr; const u64* d_offsets = nullptr; u64 n_reads = 0, n_bases = 0;
        u64 base_from = 0;            // first base position the risky launch covers ...
        const u64* d_from = nullptr;  // ... or where to read it on the device (end of the last piece walked before)
        u64 ctr[KMC_CTR_N] = {0};     // the device counters before the launch
    } risky;
    DevBuf snap_hi, snap_lo, snap_cnt, snap_n, snap_occ;
    // second-level memo of the walk kernel: (k+16)-mer table (kmc_walk.hip.h); three key words for k >= 48
    Table sk;
    u64* d_sk_counters = nullptr;
    u64* h_sk_counters = nullptr;   // pinned mirror (valid after a poll)
    u64 *sk_spill_hi = nullptr, *sk_spill_lo = nullptr, *sk_spill_cnt = nullptr, *sk_spill_mid = nullptr;
    u64 sk_spill_cap = 0;
    u64* sk_occ = nullptr;          // list of its claimed slots (what the unfold kernel walks)
    bool recovered = false;  // the last poll found an overflow and recovered: the batch in flight is complete
    bool sk_dirty = false;   // walk launches since the last unfold of the (k+16)-mer table
    bool sk_fixed = false;   // its size was set by KMC_SK_SLOTS (tests): never re-allocated
    bool sk_grow = false;    // a poll found it more than half full: re-allocate larger when it is next empty
    DevBuf rx_hi, rx_lo, rx_cnt;  // receive buffers of the one-process multi-GPU reduce (a peer's sorted table)
};

namespace {

int fail(kmc_ctx* c, int code, const char* fmt, ...) {
    char buf[512] = {0};
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    memcpy(c ? c->err : g_create_err, buf, sizeof(buf));
    return code;
}

#define HIPCHK(c, call)                                                                       \
    do {                                                                                      \
        hipError_t e__ = (call);                                                              \
        if (e__ != hipSuccess)                                                                \
            return fail((c), e__ == hipErrorOutOfMemory ? KMC_ERR_NOMEM : KMC_ERR_HIP, "%s: %s", #call, hipGetErrorString(e__)); \
    } while (0)

int ensure(kmc_ctx* c, DevBuf& b, size_t bytes) {
    if (b.bytes >= bytes && b.p) return KMC_OK;
    if (b.p) { HIPCHK(c, hipFree(b.p)); b.p = nullptr; b.bytes = 0; }
    size_t want = bytes + bytes / 8 + 256;
    HIPCHK(c, hipMalloc(&b.p, want));
    b.bytes = want;
    return KMC_OK;
}

// like ensure, but the first `keep` bytes survive (grows geometrically: the copy is amortised)
int ensure_keep(kmc_ctx* c, DevBuf& b, size_t bytes, size_t keep) {
    if (b.bytes >= bytes && b.p) return KMC_OK;
    if (!b.p || !keep) return ensure(c, b, bytes);
    const size_t want = std::max(bytes + bytes / 8 + 256, b.bytes * 2);
    void* np = nullptr;
    HIPCHK(c, hipMalloc(&np, want));
    if (hipMemcpyAsync(np, b.p, keep, hipMemcpyDeviceToDevice, c->stream) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess) {
        (void)hipFree(np);
        return fail(c, KMC_ERR_HIP, "growing the key accumulator failed");
    }
    (void)hipFree(b.p);
    b.p = np;
    b.bytes = want;
    return KMC_OK;
}

void free_buf(DevBuf& b) {
    if (b.p) (void)hipFree(b.p);
    b.p = nullptr;
    b.bytes = 0;
}

void free_runs(kmc_ctx* c, bool release);

void free_table(Table& t) {
    if (t.hi) (void)hipFree(t.hi);
    if (t.lo) (void)hipFree(t.lo);
    if (t.cnt) (void)hipFree(t.cnt);
    if (t.mid) (void)hipFree(t.mid);
    t = Table{};
}

int alloc_table(kmc_ctx* c, Table& t, u64 cap) {
    t = Table{};
    t.cap = cap;
    HIPCHK(c, hipMalloc((void**)&t.lo, cap * sizeof(u64)));
    HIPCHK(c, hipMalloc((void**)&t.cnt, cap * sizeof(u64)));
    if (c->KW == 2) HIPCHK(c, hipMalloc((void**)&t.hi, cap * sizeof(u64)));
    // EMPTY marker is all ones in the word that is CASed
    if (c->KW == 2) {
        HIPCHK(c, hipMemsetAsync(t.hi, 0xFF, cap * sizeof(u64), c->stream));
        HIPCHK(c, hipMemsetAsync(t.lo, 0, cap * sizeof(u64), c->stream));
    } else {
        HIPCHK(c, hipMemsetAsync(t.lo, 0xFF, cap * sizeof(u64), c->stream));
    }
    HIPCHK(c, hipMemsetAsync(t.cnt, 0, cap * sizeof(u64), c->stream));
    return KMC_OK;
}

// drop the live runs; their buffers go back to the pool (release == true: really free everything)
void free_runs(kmc_ctx* c, bool release = false) {
    c->acc_n = 0;  // (keys extracted and not yet sorted go with the runs)
    for (auto& r : c->runs) { r.n = 0; c->run_pool.push_back(r); }
    c->runs.clear();
    if (c->view_run.lo) { c->view_run.n = 0; c->run_pool.push_back(c->view_run); c->view_run = kmc_ctx::Run{}; }
    if (release) {
        for (auto& r : c->run_pool) {
            if (r.hi) (void)hipFree(r.hi);
            if (r.lo) (void)hipFree(r.lo);
            if (r.cnt) (void)hipFree(r.cnt);
        }
        c->run_pool.clear();
    }
}

// a run with room for `cap` entries: the smallest fitting pooled one, else a new allocation
int take_run(kmc_ctx* c, u64 cap, kmc_ctx::Run* out) {
    int best = -1;
    for (size_t i = 0; i < c->run_pool.size(); ++i)
        if (c->run_pool[i].cap >= cap && (best < 0 || c->run_pool[i].cap < c->run_pool[(size_t)best].cap)) best = (int)i;
    if (best >= 0) {
        *out = c->run_pool[(size_t)best];
        c->run_pool.erase(c->run_pool.begin() + best);
        out->n = 0;
        out->total = 0;
        out->total_known = false;
        return KMC_OK;
    }
    if (!c->run_pool.empty()) {  // nothing fits: recycle the memory of the largest pooled buffer
        size_t big = 0;
        for (size_t i = 1; i < c->run_pool.size(); ++i) if (c->run_pool[i].cap > c->run_pool[big].cap) big = i;
        kmc_ctx::Run r = c->run_pool[big];
        c->run_pool.erase(c->run_pool.begin() + (long)big);
        if (r.hi) (void)hipFree(r.hi);
        (void)hipFree(r.lo);
        (void)hipFree(r.cnt);
    }
    kmc_ctx::Run r;
    r.cap = cap + cap / 16 + 64;
    HIPCHK(c, hipMalloc((void**)&r.lo, r.cap * sizeof(u64)));
    if (hipMalloc((void**)&r.cnt, r.cap * sizeof(u64)) != hipSuccess) { (void)hipFree(r.lo); return fail(c, KMC_ERR_NOMEM, "out of device memory for a sorted run"); }
    if (c->KW == 2 && hipMalloc((void**)&r.hi, r.cap * sizeof(u64)) != hipSuccess) { (void)hipFree(r.lo); (void)hipFree(r.cnt); return fail(c, KMC_ERR_NOMEM, "out of device memory for a sorted run"); }
    *out = r;
    return KMC_OK;
}

GTable gtable_of(const kmc_ctx* c, const Table& t) {
    GTable g{};
    g.key_hi = t.hi;
    g.key_lo = t.lo;
    g.count = t.cnt;
    g.capmask = t.cap - 1;
    g.counters = c->d_counters;
    g.spill_hi = c->spill_hi;
    g.spill_lo = c->spill_lo;
    g.spill_cnt = c->spill_cnt;
    g.spill_cap = c->spill_cap;
    g.occ_list = c->occ_list;
    g.occ_list_cap = c->occ_list ? KMC_OCC_LIST_CAP : 0;
    g.occ_key_lo = c->occ_key_lo;
    g.occ_key_hi = c->occ_key_hi;
    return g;
}

GTable sk_table_of(const kmc_ctx* c) {
    GTable g{};
    if (!c->sk.lo) return g;  // key_lo == nullptr: no second-level memo
    g.key_hi = c->sk.hi;
    g.key_mid = c->sk.mid;
    g.key_lo = c->sk.lo;
    g.count = c->sk.cnt;
    g.capmask = c->sk.cap - 1;
    g.counters = c->d_sk_counters;
    g.spill_hi = c->sk_spill_hi;
    g.spill_mid = c->sk_spill_mid;
    g.spill_lo = c->sk_spill_lo;
    g.spill_cnt = c->sk_spill_cnt;
    g.spill_cap = c->sk_spill_cap;
    g.occ_list = c->sk_occ;
    g.occ_list_cap = c->sk_occ ? c->sk.cap : 0;
    return g;
}

int sk_clear(kmc_ctx* c) {
    if (!c->sk.lo) return KMC_OK;
    HIPCHK(c, hipMemsetAsync(c->sk.hi, 0xFF, c->sk.cap * sizeof(u64), c->stream));
    HIPCHK(c, hipMemsetAsync(c->sk.lo, 0, c->sk.cap * sizeof(u64), c->stream));
    if (c->sk.mid) HIPCHK(c, hipMemsetAsync(c->sk.mid, 0, c->sk.cap * sizeof(u64), c->stream));
    HIPCHK(c, hipMemsetAsync(c->sk.cnt, 0, c->sk.cap * sizeof(u64), c->stream));
    HIPCHK(c, hipMemsetAsync(c->d_sk_counters, 0, KMC_CTR_N * sizeof(u64), c->stream));
    memset(c->h_sk_counters, 0, KMC_CTR_N * sizeof(u64));
    return KMC_OK;
}

// the walk kernel's (k+16)-mer table: allocated at the first walk launch of a ctx.  It starts at 1 Mi slots
// (a fresh ctx's first file -> table run paid 15 ms for allocating and clearing 16 Mi slots it never used)
// and is re-allocated sixteen times larger, once, when a poll finds it more than half full -- at a moment
// when it is empty (right behind its unfold), so nothing has to be re-inserted.
void sk_free(kmc_ctx* c) {
    free_table(c->sk);
    c->sk = Table{};
    u64** p[] = {&c->sk_spill_hi, &c->sk_spill_lo, &c->sk_spill_cnt, &c->sk_spill_mid, &c->sk_occ};
    for (u64** q : p) { if (*q) (void)hipFree(*q); *q = nullptr; }
}
int sk_alloc_parts(kmc_ctx* c, u64 cap) {
    const bool three = c->cfg.k > KMC_SK_MAX_K;  // a (k+16)-mer of more than 63 bases: three key words
    c->sk.cap = cap;
    HIPCHK(c, hipMalloc((void**)&c->sk.hi, cap * sizeof(u64)));
    HIPCHK(c, hipMalloc((void**)&c->sk.lo, cap * sizeof(u64)));
    HIPCHK(c, hipMalloc((void**)&c->sk.cnt, cap * sizeof(u64)));
    if (three) HIPCHK(c, hipMalloc((void**)&c->sk.mid, cap * sizeof(u64)));
    c->d_sk_counters = c->d_counters + KMC_CTR_N;
    c->h_sk_counters = c->h_counters + KMC_CTR_N;
    c->sk_spill_cap = std::max<u64>(cap / 64, 4096);
    HIPCHK(c, hipMalloc((void**)&c->sk_spill_hi, c->sk_spill_cap * sizeof(u64)));
    HIPCHK(c, hipMalloc((void**)&c->sk_spill_lo, c->sk_spill_cap * sizeof(u64)));
    HIPCHK(c, hipMalloc((void**)&c->sk_spill_cnt, c->sk_spill_cap * sizeof(u64)));
    if (three) HIPCHK(c, hipMalloc((void**)&c->sk_spill_mid, c->sk_spill_cap * sizeof(u64)));
    HIPCHK(c, hipMalloc((void**)&c->sk_occ, cap * sizeof(u64)));
    return sk_clear(c);
}
// all or nothing: a table whose spill area or slot list is missing must never reach a kernel (kmc_spill writes
// spill_lo[idx] unconditionally), so a failed allocation leaves NO second-level memo -- the walk then runs
// without it, or the next batch repeats the error cleanly
int sk_alloc(kmc_ctx* c, u64 cap) {
    const int rc = sk_alloc_parts(c, cap);
    if (rc) { sk_free(c); c->sk_spill_cap = 0; }
    return rc;
}
int sk_ensure(kmc_ctx* c) {
    if (c->sk.lo || c->cfg.mode != KMC_MODE_CONTIG) return KMC_OK;
    u64 cap = 1ull << 20;
    if (const char* e = getenv("KMC_SK_SLOTS")) { u64 v = strtoull(e, nullptr, 10); if (v >= 1024) { cap = 1; while (cap < v) cap <<= 1; c->sk_fixed = true; } }
    return sk_alloc(c, cap);
}
// the (k+16)-mer table is empty (stream order: right behind its unfold) and was found too small: once, x16
int sk_regrow(kmc_ctx* c) {
    if (!c->sk_grow || !c->sk.lo) return KMC_OK;
    c->sk_grow = false;
    const u64 cap = std::min<u64>(c->sk.cap * 16, 1ull << 24);
    if (cap <= c->sk.cap) return KMC_OK;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    sk_free(c);
    return sk_alloc(c, cap);
}

u64 next_pow2(u64 v) {
    u64 p = 1;
    while (p < v) p <<= 1;
    return p;
}

int grid_for(const kmc_ctx* c, u64 n, int threads) {
    u64 blocks = (n + threads - 1) / threads;
    u64 cap = (u64)c->n_cu * 8;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    return (int)blocks;
}

// workgroups of kmc_reset_kernel: every one of them draws a ticket from ONE word at the end (the counters are cleared behind
// the last), so few of them while the table is small enough to be cleared by few
int reset_grid(const kmc_ctx* c) {
    return c->tab.cap <= (16ull << 20) ? 256 : (int)std::min<u64>(c->tab.cap >> 14, 2048);
}

template <typename F1, typename F2>
auto kw_dispatch(int KW, F1 f1, F2 f2) { return KW == 1 ? f1() : f2(); }

// what a poll learns from fresh h_counters
int poll_book(kmc_ctx* c);
// read the device counters (synchronises the stream)
int poll(kmc_ctx* c) {
    HIPCHK(c, hipMemcpyAsync(c->h_counters, c->d_counters, (c->sk.lo ? 2 : 1) * KMC_CTR_N * sizeof(u64), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return poll_book(c);
}
// The poll right behind a speculative kmc_small_finalize_kernel: when the kernel succeeded it has PUBLISHED the
// counters to h_counters itself (and emptied the table): no read-back copy, one synchronisation.
// The kernel writes mirror[FINSEQ] = its number LAST, whatever it decided: the host waits for that word in the pinned
// mirror rather than for the end of the kernel (the end-of-kernel release + the completion signal cost microseconds of a
// 1.4 ms step, and what the kernel still does after the word -- clearing slots -- is device work in stream order).
// A kernel that never publishes (a fault) is found by the synchronisation this falls back to.
int poll_fin(kmc_ctx* c) {
    {
        const volatile u64* seqw = (const volatile u64*)&c->h_counters[KMC_CTR_FINSEQ];
        static const bool no_spin = getenv("KMC_NO_MIRROR_SPIN") != nullptr;
        bool seen = false;
        if (!no_spin) {
            const auto t0 = std::chrono::steady_clock::now();
            for (u32 it = 1; !(seen = (*seqw == c->fin_seq)); ++it) {
                __builtin_ia32_pause();
                if ((it & 0xfffu) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(200)) break;
            }
            std::atomic_thread_fence(std::memory_order_acquire);
        }
        if (!seen) HIPCHK(c, hipStreamSynchronize(c->stream));
        c->view_unsynced = seen;   // (the kernel may still be running: kmc_export_device waits for its end before it hands out pointers)
    }
    c->st.n_async_ok = c->h_counters[KMC_CTR_FINOK];
    c->st.n_async_slabs_skipped = c->h_counters[KMC_CTR_FINSKIP];
    if (c->h_counters[KMC_CTR_FASTFIN] != 1 || c->h_counters[KMC_CTR_FINSEQ] != c->fin_seq) {   // it gave up: read the counters the usual way
        c->h_counters[KMC_CTR_FASTFIN] = 0;
        int rc = poll(c);
        c->h_counters[KMC_CTR_FASTFIN] = 0;   // (the device word is never written; whatever the copy brought is not a verdict)
        return rc;
    }
    c->drained = true;
    c->fin_parity = 0;
    return poll_book(c);
}
int poll_book(kmc_ctx* c) {
    c->polled_epoch = c->table_epoch;
    c->pending = false;
    c->batch_pending = false;
    c->unpolled_adds = 0;
    c->st.n_slabs_skipped = c->h_counters[KMC_CTR_SLABSKIP];
    c->st.n_direct = c->h_counters[KMC_CTR_BADBASE];
    {
        // share of k-mers the walk kernel had to count directly since the previous poll
        u64 d = c->h_counters[KMC_CTR_BADBASE], n = c->h_counters[KMC_CTR_KMERS];
        u64 dd = d - c->direct_seen, dn = n - c->kmers_seen;
        if (d >= c->direct_seen && n > c->kmers_seen && (c->st.algo_last == KMC_ALGO_WALK || c->st.algo_last == KMC_ALGO_STREAM)) c->walk_overflowed = dd * 20 > dn;
        if (c->sk.lo && c->h_sk_counters[KMC_CTR_OCCUPIED] + c->h_sk_counters[KMC_CTR_SPILL] != 0) c->sklog_on = true;   // (the memo overflows on this source)
        // the second-level memo more than half full: this input has too many distinct (k+16)-mers for it
        if (c->sk.lo && (c->h_sk_counters[KMC_CTR_OCCUPIED] + c->h_sk_counters[KMC_CTR_SPILL]) * 2 > c->sk.cap) {
            if (!c->sk_fixed && c->sk.cap < (1ull << 24)) c->sk_grow = true;  // (first: a larger one)
            else c->walk_overflowed = true;
        }
        c->direct_seen = d;
        c->kmers_seen = n;
    }
    if (c->b_open) {
        // history for the next batch's launch plan: new keys per k-mer of the batch just finished
        u64 occ = c->h_counters[KMC_CTR_OCCUPIED] + c->h_counters[KMC_CTR_SPILL];
        // What limits a launch is the NUMBER of new keys it can bring, so the predictor is the whole
        // batch's ratio (not the burstiest sub-batch's).
        double whole = (double)(occ > c->b_occ0 ? occ - c->b_occ0 : 0) / (double)c->b_kmers;
        c->rho_hist = whole;
        c->b_open = false;
    }
    return KMC_OK;
}

int recover_overflow(kmc_ctx* c);

int grow_to(kmc_ctx* c, u64 newcap) {
    Table nt;
    int rc = alloc_table(c, nt, newcap);
    if (rc) { free_table(nt); return rc; }
    // fresh occupancy count: rehash re-counts every claimed slot
    HIPCHK(c, hipMemsetAsync(&c->d_counters[KMC_CTR_OCCUPIED], 0, sizeof(u64), c->stream));
    GTable go = gtable_of(c, c->tab), gn = gtable_of(c, nt);
    int grid = grid_for(c, c->tab.cap, 256);
    if (c->KW == 1) hipLaunchKernelGGL(kmc_rehash_kernel<1>, dim3(grid), dim3(256), 0, c->stream, go, gn);
    else hipLaunchKernelGGL(kmc_rehash_kernel<2>, dim3(grid), dim3(256), 0, c->stream, go, gn);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream));
    free_table(c->tab);
    c->tab = nt;
    c->st.table_capacity = newcap;
    return KMC_OK;
}

// Called with fresh h_counters: drain the spill area and grow when the table is over half full.
int settle(kmc_ctx* c) {
    for (int iter = 0; iter < 40; ++iter) {
        u64 occ = c->h_counters[KMC_CTR_OCCUPIED], spill = c->h_counters[KMC_CTR_SPILL], err = c->h_counters[KMC_CTR_ERR];
        if (c->sk.lo && c->h_sk_counters[KMC_CTR_ERR]) err |= (c->h_sk_counters[KMC_CTR_ERR] & 2) ? 2 : 1;  // the (k+16)-mer table dropped pairs
        if (err & 4) return fail(c, KMC_ERR_ALPHABET, "Unexpected charactor appears in a sequence (reference mode accepts ACGT only)");
        if (err & 2) return fail(c, KMC_ERR_HIP, "table insert gave up after too many retries (internal error)");
        if (err == 1 && c->risky.armed) return recover_overflow(c);  // a prediction was wrong: put the table back, count that part by sorting
        if (err) return fail(c, KMC_ERR_CAPACITY, "count table and spill area exhausted (capacity %llu slots, %llu spilled); raise capacity_hint",
                             (unsigned long long)c->tab.cap, (unsigned long long)spill);
        c->risky.armed = false;  // (whatever was predicted has been absorbed)
        if (!spill && occ * 2 <= c->tab.cap) return KMC_OK;
        u64 need = (occ + spill) * 2;
        u64 newcap = c->tab.cap;
        while (newcap < need) newcap <<= 1;
        if (newcap == c->tab.cap && spill) newcap <<= 1;
        if (newcap != c->tab.cap) {
            int rc = grow_to(c, newcap);
            if (rc) return rc;
        }
        if (spill) {
            // move the spill entries aside conceptually: merge them, then clear the counter
            u64 n = spill;
            c->st.n_spilled += n;
            GTable g = gtable_of(c, c->tab);
            // the merge may itself spill (appends after position n); handled by the next iteration
            u64 *sh = nullptr, *sl = nullptr, *sc = nullptr;
            HIPCHK(c, hipMalloc((void**)&sl, n * sizeof(u64)));
            HIPCHK(c, hipMalloc((void**)&sc, n * sizeof(u64)));
            if (c->KW == 2) HIPCHK(c, hipMalloc((void**)&sh, n * sizeof(u64)));
            HIPCHK(c, hipMemcpyAsync(sl, c->spill_lo, n * sizeof(u64), hipMemcpyDeviceToDevice, c->stream));
            HIPCHK(c, hipMemcpyAsync(sc, c->spill_cnt, n * sizeof(u64), hipMemcpyDeviceToDevice, c->stream));
            if (sh) HIPCHK(c, hipMemcpyAsync(sh, c->spill_hi, n * sizeof(u64), hipMemcpyDeviceToDevice, c->stream));
            HIPCHK(c, hipMemsetAsync(&c->d_counters[KMC_CTR_SPILL], 0, sizeof(u64), c->stream));
            int grid = grid_for(c, n, 256);
            if (c->KW == 1) hipLaunchKernelGGL(kmc_merge_pairs_kernel<1>, dim3(grid), dim3(256), 0, c->stream, g, (const u64*)sh, (const u64*)sl, (const u64*)sc, n);
            else hipLaunchKernelGGL(kmc_merge_pairs_kernel<2>, dim3(grid), dim3(256), 0, c->stream, g, (const u64*)sh, (const u64*)sl, (const u64*)sc, n);
            HIPCHK(c, hipGetLastError());
            HIPCHK(c, hipStreamSynchronize(c->stream));
            (void)hipFree(sl); (void)hipFree(sc); if (sh) (void)hipFree(sh);
        }
        int rc = poll(c);
        if (rc) return rc;
    }
    return fail(c, KMC_ERR_CAPACITY, "table growth did not converge");
}

int launch_begin(kmc_ctx* c);
int launch_end(kmc_ctx* c);
int poll_and_settle(kmc_ctx* c) {
    int rc = poll(c);
    if (rc) return rc;
    return settle(c);
}
int poll_fin_and_settle(kmc_ctx* c) {
    int rc = poll_fin(c);
    if (rc) return rc;
    return settle(c);
}

// one workgroup per KMC_FIN_CHUNK keys; the grid follows the size of the last table seen (+ 25 %), at least 64
// workgroups.  (Workgroups past the table leave at once, but hundreds of them still cost microseconds.)  A table that
// outgrew the grid is noticed by kmc_finalize and finalized again with the full grid.
int small_finalize_grid(const kmc_ctx* c) {
    return (int)std::min<u64>(KMC_FIN_KERNEL_MAX / KMC_FIN_CHUNK, std::max<u64>(64, (c->fin_hint + c->fin_hint / 4) / KMC_FIN_CHUNK + 8));
}
// The speculative small-table finalize (kmc_table.hip.h): queued behind whatever is still running.
int launch_small_finalize(kmc_ctx* c, int grid) {
    GTable g = gtable_of(c, c->tab);
    // the kernel publishes {FASTFIN, FINSEQ = seq, the counters} through d_mirror; the host trusts a verdict only
    // when it carries this launch's number (earlier launches may still be in flight: kmc_finalize_async)
    const u64 seq = ++c->fin_seq;
    const u64* skc = c->d_counters + KMC_CTR_N;
    if (c->KW == 1) hipLaunchKernelGGL(kmc_small_finalize_kernel<1>, dim3(grid), dim3(1024), 0, c->stream, g, skc, c->fin_rank, c->d_mirror, seq, (u64*)nullptr, (u64*)c->o_lo.p, (u64*)c->o_cnt.p);
    else hipLaunchKernelGGL(kmc_small_finalize_kernel<2>, dim3(grid), dim3(1024), 0, c->stream, g, skc, c->fin_rank, c->d_mirror, seq, (u64*)c->o_hi.p, (u64*)c->o_lo.p, (u64*)c->o_cnt.p);
    HIPCHK(c, hipGetLastError());
    return KMC_OK;
}

// A finalize that drained the table (c->drained) left the counts in the sorted view only.  Before anything adds
// to the table again -- a further batch, a merge -- the view's pairs go back in and the counters are restored.
// Nothing is synchronised: h_counters already hold what the device counters will read once the merge has run.
int undrain(kmc_ctx* c) {
    if (!c->drained) return KMC_OK;
    c->drained = false;
    memcpy(c->h_restore, c->h_counters, KMC_CTR_N * sizeof(u64));
    const int zero[] = {KMC_CTR_OCCUPIED, KMC_CTR_OUT, KMC_CTR_SUM, KMC_CTR_OUT1, KMC_CTR_SUM1, KMC_CTR_SUM2, KMC_CTR_FASTFIN};
    for (int i : zero) c->h_restore[i] = 0;   // (the merge claims the slots again and counts them)
    HIPCHK(c, hipMemcpyAsync(c->d_counters, c->h_restore, KMC_CTR_N * sizeof(u64), hipMemcpyHostToDevice, c->stream));
    c->fin_parity = 0;
    const u64 n = c->n_sorted;
    if (n) {
        GTable g = gtable_of(c, c->tab);
        if (c->KW == 1) hipLaunchKernelGGL(kmc_merge_pairs_kernel<1>, dim3(grid_for(c, n, 256)), dim3(256), 0, c->stream, g, (const u64*)nullptr, (const u64*)c->o_lo.p, (const u64*)c->o_cnt.p, n);
        else hipLaunchKernelGGL(kmc_merge_pairs_kernel<2>, dim3(grid_for(c, n, 256)), dim3(256), 0, c->stream, g, (const u64*)c->o_hi.p, (const u64*)c->o_lo.p, (const u64*)c->o_cnt.p, n);
        HIPCHK(c, hipGetLastError());
    }
    return KMC_OK;
}

// kmc_finalize_async queued a finalize and returned; before the host does anything else with the ctx it has to learn
// how that went (one synchronisation): a view + a drained table, or "gave up" and a table that is as it was.
int resolve_async(kmc_ctx* c) {
    if (!c->async_fin) return KMC_OK;
    c->async_fin = false;
    int rc = poll_fin(c);
    if (rc) return rc;
    if (c->drained) {
        const u64 n = c->h_counters[KMC_CTR_OCCUPIED];
        c->v_hi = c->KW == 2 ? (const u64*)c->o_hi.p : nullptr;
        c->v_lo = (const u64*)c->o_lo.p;
        c->v_cnt = (const u64*)c->o_cnt.p;
        c->n_sorted = n;
        c->sorted_valid = true;
        c->fin_hint = n;
        c->st.n_distinct = n;
        c->st.n_kmers = n ? c->h_counters[KMC_CTR_SUM2] : 0;
        return KMC_OK;
    }
    return settle(c);
}

// Give the counts of the (k+16)-mer table to their k-mers (kmc_sk_unfold_kernel).  What the last poll
// saw of that table is certain to come (16 k-mers per entry): room is made for it first.  Entries
// added since then are covered like any prediction: by the table saved in front of the launch
// that added them.  (Growing re-inserts the table, which a saved COPY of the table would not
// survive; a saved entry list does.)
// The unfold is DEFERRED: a batch's walk launches leave sk_dirty set, and whoever looks at the count
// table next settles it -- the next batch (two tiny launches, no synchronisation), kmc_finalize (which
// polls anyway and launches nothing when the poll shows the (k+16)-mer table empty: the benchmark's
// steady state, where the unconditional unfold + spill reset cost 9 us of a 280 us step), pack / reset.
int flush_sk(kmc_ctx* c) {
    const u64 sk_known = c->h_sk_counters ? c->h_sk_counters[KMC_CTR_OCCUPIED] + c->h_sk_counters[KMC_CTR_SPILL] : 0;
    const u64 occ0 = c->h_counters[KMC_CTR_OCCUPIED];
    if (sk_known && (occ0 + 16 * sk_known) * 10 > c->tab.cap * 7 && !(c->risky.armed && c->risky.mode == 2)) {
        int r = grow_to(c, next_pow2((occ0 + 16 * sk_known) * 2));
        if (r) return r;
    }
    int r = launch_begin(c);   // (the unfold is part of what the walk path costs: kernel_ms_*, KMC_ALGO_AUTO's comparison)
    if (r) return r;
    r = kmc_sk_unfold_launch(c->stream, c->n_cu, c->KW, c->cfg.k, c->cfg.canonical != 0, sk_table_of(c), gtable_of(c, c->tab));
    if (r) return fail(c, r, "(k+16)-mer unfold launch failed");
    r = launch_end(c);
    if (r) return r;
    c->sk_dirty = false;
    c->pending = true;
    c->table_epoch++;
    return sk_regrow(c);
}

// right after a poll: launch nothing when the poll shows the (k+16)-mer table empty
int settle_sk_polled(kmc_ctx* c) {
    if (!c->sk_dirty) return KMC_OK;
    if (!c->h_sk_counters || c->h_sk_counters[KMC_CTR_KMERS] == 0) { c->sk_dirty = false; return sk_regrow(c); }   // (keys may be there -- they stay across launches -- but no counts are pending)
    return flush_sk(c);
}

// ---- launching the counting kernels ---------------------------------------------------------

// hipEvent pair around one count-kernel launch; the sum over a batch is kmc_stats.kernel_ms_last
int take_event(kmc_ctx* c, hipEvent_t* e) {
    if (!c->ev_free.empty()) { *e = c->ev_free.back(); c->ev_free.pop_back(); return KMC_OK; }
    HIPCHK(c, hipEventCreate(e));
    return KMC_OK;
}
int launch_begin(kmc_ctx* c) {
    if (c->tb.empty()) c->tb.emplace_back();
    if (c->tb.back().ev.size() & 1) { c->ev_free.push_back(c->tb.back().ev.back()); c->tb.back().ev.pop_back(); }   // (a bracket left open by a failed launch)
    hipEvent_t e;
    { int rc = take_event(c, &e); if (rc) return rc; }
    c->tb.back().ev.push_back(e);
    HIPCHK(c, hipEventRecord(e, c->stream));
    return KMC_OK;
}
// an event pair for a launch that records its own timestamps (hipExtLaunchKernelGGL): joins the batch's list unrecorded
int launch_events(kmc_ctx* c, hipEvent_t* e0, hipEvent_t* e1) {
    if (c->tb.empty()) c->tb.emplace_back();
    if (c->tb.back().ev.size() & 1) { c->ev_free.push_back(c->tb.back().ev.back()); c->tb.back().ev.pop_back(); }
    { int rc = take_event(c, e0); if (rc) return rc; }
    { int rc = take_event(c, e1); if (rc) { c->ev_free.push_back(*e0); return rc; } }
    c->tb.back().ev.push_back(*e0);
    c->tb.back().ev.push_back(*e1);
    return KMC_OK;
}
int launch_end(kmc_ctx* c) {
    hipEvent_t e;
    { int rc = take_event(c, &e); if (rc) return rc; }
    c->tb.back().ev.push_back(e);
    HIPCHK(c, hipEventRecord(e, c->stream));
    c->batch_pending = true;
    return KMC_OK;
}

int launch_stream(kmc_ctx* c, const uint8_t* d_bases, const u64* d_offsets, u64 n_reads, u64 n_bases,
                  u64 chunk_begin, u64 chunk_end, u64 range_begin) {
    // the kernel indexes chunks from 0; a sub-range is expressed by offsetting the chunk ids
    u64 n_chunks = chunk_end - chunk_begin;
    if (!n_chunks) return KMC_OK;
    u64 max_waves = (u64)c->n_cu * 2 * KMC_STREAM_WAVES;   // one workgroup per CU is resident (LDS); 2 rounds
    u64 cpw = (n_chunks + max_waves - 1) / max_waves;
    if (cpw < 4) cpw = std::min<u64>(4, n_chunks);
    u64 waves = (n_chunks + cpw - 1) / cpw;
    int grid = (int)((waves + KMC_STREAM_WAVES - 1) / KMC_STREAM_WAVES);
    GTable g = gtable_of(c, c->tab);
    const bool canon = c->cfg.canonical != 0;
    { int rc = launch_begin(c); if (rc) return rc; }
#define LAUNCH_STREAM(KWV, CAN)                                                                             \
    hipLaunchKernelGGL((kmc_stream_kernel<KWV, CAN>), dim3(grid), dim3(KMC_STREAM_THREADS), 0, c->stream, \
                       d_bases, n_bases, d_offsets, n_reads, c->cfg.k, chunk_begin, chunk_end, cpw, range_begin, g)
    if (c->KW == 1) { if (canon) LAUNCH_STREAM(1, true); else LAUNCH_STREAM(1, false); }
    else { if (canon) LAUNCH_STREAM(2, true); else LAUNCH_STREAM(2, false); }
#undef LAUNCH_STREAM
    HIPCHK(c, hipGetLastError());
    return launch_end(c);
}

// ---- KMC_ALGO_SORT -------------------------------------------------------------------------------

// extraction front end (kmc_extract.hip.h): one key per base position of chunks [chunk_begin, chunk_end), padded with
// filler to whole ranges of KMC_MSD_RANGE positions, plus each range's level-0 histogram row and AND / OR words
int launch_extract(kmc_ctx* c, const uint8_t* d_bases, const u64* d_offsets, u64 n_reads, u64 n_bases,
                   u64 chunk_begin, u64 chunk_end, u64 range_begin, u32 n_ranges, u64* out_hi, u64* out_lo, u32* hist, u32 hist_r0, u64* rand_, u64* ror_) {
    if (!n_ranges) return KMC_OK;
    const int grid = (int)std::min<u64>(n_ranges, (u64)c->n_cu);   // one 1024-thread workgroup per CU is resident (registers)
    const bool canon = c->cfg.canonical != 0;
#define LAUNCH_EXTRACT(KWV, CAN)                                                                                                   \
    do {                                                                                                                           \
        static std::atomic<unsigned long long> attr{0};                                                                            \
        if (kmc_attr_once(attr)) (void)hipFuncSetAttribute((const void*)kmc_extract_hist_kernel<KWV, CAN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(ExtractLds<KWV>)); \
        hipLaunchKernelGGL((kmc_extract_hist_kernel<KWV, CAN>), dim3(grid), dim3(KMC_STREAM_THREADS), sizeof(ExtractLds<KWV>), c->stream,    \
                           d_bases, n_bases, d_offsets, n_reads, c->cfg.k, chunk_begin, chunk_end, range_begin, n_ranges, c->d_counters, out_hi, out_lo, hist, hist_r0, rand_, ror_); \
    } while (0)
    if (c->KW == 1) { if (canon) LAUNCH_EXTRACT(1, true); else LAUNCH_EXTRACT(1, false); }
    else { if (canon) LAUNCH_EXTRACT(2, true); else LAUNCH_EXTRACT(2, false); }
#undef LAUNCH_EXTRACT
    HIPCHK(c, hipGetLastError());
    return KMC_OK;
}

// ---- hand-written MSD radix sort + run-length (kmc_msd.hip.h) ------------------------------------------
// Sorts the n keys in lo[0] (hi[0] for two-word keys; w[0] = weights to sum, or null: every key counts
// once), dropping all-ones filler keys, and appends the resulting sorted (key, count) run to c->runs.
// lo[1] / hi[1] / w[1] are scratch of the same size.  One host synchronisation per level (the number
// of segments that go on) and one for the size of the run.
// KW: words per key (1: lo only).  out == nullptr: the run joins c->runs; otherwise it is handed to the
// caller (n == 0 when nothing but filler came in).
// pre0: level 0's per-range histograms and AND / OR words are already in c->a_hist / a_rand / a_ror (the extraction kernel
// computed them while it wrote the keys: kmc_extract.hip.h); n is then a multiple of KMC_MSD_RANGE.
int msd_sort_to_run(kmc_ctx* c, u64* const hi[2], u64* const lo[2], u64* const w[2], u64 n, unsigned kb, int KW, kmc_ctx::Run* out = nullptr, bool repeated_keys = false, bool pre0 = false) {
    if (out) *out = kmc_ctx::Run{};
    if (!n) return KMC_OK;
    if (n >= (1ull << 32) - KMC_MSD_RANGE) return fail(c, KMC_ERR_ARG, "msd sort: more than 2^32 keys in one pass");
    const bool weights = w[0] != nullptr;
    // two-word keys: leaves of 2048 keys save a level on random keys (1270 keys per child after two levels:
    // 84 -> 57 ms for 760 M 63-mers) but cost on heavily repeated, clustered keys (pool = 1000: 87 -> 102 ms);
    // a ctx whose last sort collapsed its keys more than fourfold keeps the smaller leaves
    u32 leaf_cap = KW == 1 ? KMC_MSD_LEAF1 : (w[0] ? KMC_MSD_LEAF2W : KMC_MSD_LEAF2);
    if (KW == 2 && !w[0] && c->msd_dup_heavy) leaf_cap = 1024;
    const bool clustered = repeated_keys || c->msd_dup_heavy;   // (one-word leaves: the larger wave scratch)
    const u64 max_seg = n / leaf_cap + 257;
    const u32 cnt_bits = weights ? 0u : (u32)KMC_MSD_CNT_BITS;   // long spans with few bits left: LDS histograms (plain counts only)
    const u32 rsz = pre0 ? (u32)KMC_MSD_RANGE : msd_range_for(n);   // keys per range (a workgroup of the histogram / scatter passes)
    const u64 max_ranges = n / rsz + max_seg + 1;
    const u64 term_cap = 16 * (n / leaf_cap) + 65536;
    const u64 n_words = (n + 63) / 64;
    int rc;
#define MSD_ENSURE(buf, bytes) do { rc = ensure(c, (buf), (size_t)(bytes)); if (rc) return rc; } while (0)
    MSD_ENSURE(c->m_hist, msd_hist_words(max_ranges) * sizeof(u32));
    MSD_ENSURE(c->m_rmin, max_ranges * 2 * sizeof(u64));
    MSD_ENSURE(c->m_rmax, max_ranges * 2 * sizeof(u64));
    MSD_ENSURE(c->m_seg[0], max_seg * sizeof(MsdSeg));
    MSD_ENSURE(c->m_seg[1], max_seg * sizeof(MsdSeg));
    MSD_ENSURE(c->m_first, (max_seg + 1) * sizeof(u32));
    MSD_ENSURE(c->m_cbase, max_seg * KMC_MSD_NB * sizeof(u32));
    MSD_ENSURE(c->m_stot, max_seg * KMC_MSD_NB * sizeof(u32));
    MSD_ENSURE(c->m_skip, max_seg * sizeof(u32));
    MSD_ENSURE(c->m_term, term_cap * sizeof(MsdTerm));
    MSD_ENSURE(c->m_ord, term_cap * sizeof(MsdTerm));
    MSD_ENSURE(c->m_bitmap, n_words * sizeof(u64));
    MSD_ENSURE(c->m_rank, n_words * sizeof(u32));
    MSD_ENSURE(c->m_nd, term_cap * sizeof(u32));
    MSD_ENSURE(c->m_clist, max_seg * sizeof(u32));   // (kind-2 terminals hold more than leaf_cap keys each)
    MSD_ENSURE(c->m_base, term_cap * sizeof(u32));
    MSD_ENSURE(c->m_ctl, sizeof(MsdCtl));
    MSD_ENSURE(c->m_bsum, (std::max<u64>(n_words, term_cap) / KMC_SCAN_PER_BLOCK + 2) * sizeof(u32));
#undef MSD_ENSURE
    if (!c->h_ctl) HIPCHK(c, hipHostMalloc((void**)&c->h_ctl, sizeof(MsdCtl)));
    MsdCtl* ctl = (MsdCtl*)c->m_ctl.p;
    HIPCHK(c, hipMemsetAsync(ctl, 0, sizeof(MsdCtl), c->stream));
    HIPCHK(c, hipMemsetAsync(c->m_bitmap.p, 0, n_words * sizeof(u64), c->stream));
    const MsdSeg root{0u, (u32)n, kb, 0u};  // (in buffer 0)
    HIPCHK(c, hipMemcpyAsync(c->m_seg[0].p, &root, sizeof(root), hipMemcpyHostToDevice, c->stream));
    u32 n_seg = 1;
    int cur = 0;
    // (every level takes at least one bit off every active segment, normally ten: kb levels at most)
    for (int l = 0; l < (int)kb && n_seg; ++l) {
        MsdSeg* seg = (MsdSeg*)c->m_seg[cur].p;
        MsdSeg* next = (MsdSeg*)c->m_seg[cur ^ 1].p;
        u32* first = (u32*)c->m_first.p;
        hipLaunchKernelGGL(kmc_msd_ranges_kernel, dim3(1), dim3(1024), 0, c->stream, (const MsdSeg*)seg, n_seg, rsz, first, ctl);
        const u32 grid = (u32)std::min<u64>(n / rsz + n_seg + 1, max_ranges);
        const bool have = pre0 && l == 0;   // this level's histogram rows came with the keys
        u32* const hist_l = have ? (u32*)c->a_hist.p : (u32*)c->m_hist.p;
        u64* const rmin_l = have ? (u64*)c->a_rand.p : (u64*)c->m_rmin.p;
        u64* const rmax_l = have ? (u64*)c->a_ror.p : (u64*)c->m_rmax.p;
        if (have) { /* nothing to read back */ }
        else if (KW == 1) hipLaunchKernelGGL(kmc_msd_hist_kernel<1>, dim3(grid), dim3(KMC_MSD_THREADS), 0, c->stream, (const u64*)hi[0], (const u64*)lo[0], (const u64*)hi[1], (const u64*)lo[1],
                                        (const MsdSeg*)seg, n_seg, (const u32*)first, rsz, (int)kb, l == 0 ? 1 : 0, hist_l, rmin_l, rmax_l, (const MsdCtl*)ctl);
        else hipLaunchKernelGGL(kmc_msd_hist_kernel<2>, dim3(grid), dim3(KMC_MSD_THREADS), 0, c->stream, (const u64*)hi[0], (const u64*)lo[0], (const u64*)hi[1], (const u64*)lo[1],
                                (const MsdSeg*)seg, n_seg, (const u32*)first, rsz, (int)kb, l == 0 ? 1 : 0, hist_l, rmin_l, rmax_l, (const MsdCtl*)ctl);
        HIPCHK(c, hipMemsetAsync(&ctl->n_next, 0, sizeof(u32), c->stream));
        // few segments = long ones: a wave per digit column (257 workgroups of four waves for a single segment: level 0),
        // 64 workgroups per segment while there are few, one when there are many (short ones)
        const u32 S = n_seg <= 8 ? 257u : (n_seg <= 4096 ? 64u : 1u);
        hipLaunchKernelGGL(kmc_msd_scan_a_kernel, dim3(n_seg * S), dim3(KMC_MSD_THREADS), 0, c->stream, n_seg, S, (const u32*)first, hist_l, (u32*)c->m_stot.p);
        hipLaunchKernelGGL(kmc_msd_scan_kernel, dim3(n_seg), dim3(KMC_MSD_ND), 0, c->stream, (const MsdSeg*)seg, n_seg, (const u32*)first, hist_l, (const u32*)c->m_stot.p,
                           (const u64*)rmin_l, (const u64*)rmax_l, (u32*)c->m_cbase.p, (u32*)c->m_skip.p, l == 0 ? 1 : 0, n_seg <= 8 ? 1 : 0, leaf_cap, cnt_bits,
                           next, (u32)max_seg, (MsdTerm*)c->m_term.p, (u32)term_cap, (unsigned long long*)c->m_bitmap.p, ctl);
#define MSD_SCATTER(KWV, WV)                                                                                                              \
        do {                                                                                                                              \
            static std::atomic<unsigned long long> attr{0};                                                                               \
            if (kmc_attr_once(attr)) (void)hipFuncSetAttribute((const void*)kmc_msd_scatter_kernel<KWV, WV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(MsdScatterLds<KWV, WV>)); \
            hipLaunchKernelGGL((kmc_msd_scatter_kernel<KWV, WV>), dim3(grid), dim3(1024), sizeof(MsdScatterLds<KWV, WV>), c->stream, hi[0], lo[0], w[0], hi[1], lo[1], w[1], \
                               (const MsdSeg*)seg, n_seg, (const u32*)first, rsz, (const u32*)hist_l, (const u32*)c->m_cbase.p, (const u32*)c->m_skip.p, \
                               (int)kb, l == 0 ? 1 : 0, (const MsdCtl*)ctl);                                                 \
        } while (0)
        if (KW == 1) { if (weights) MSD_SCATTER(1, true); else MSD_SCATTER(1, false); }
        else { if (weights) MSD_SCATTER(2, true); else MSD_SCATTER(2, false); }
#undef MSD_SCATTER
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipMemcpyAsync(c->h_ctl, ctl, sizeof(MsdCtl), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (c->h_ctl->overflow) return fail(c, KMC_ERR_CAPACITY, "msd sort: segment / terminal list overflow (%u)", c->h_ctl->overflow);
        n_seg = c->h_ctl->n_next;
        cur ^= 1;
    }
    const u32 n_term = c->h_ctl->n_term;
    if (!n_term) return KMC_OK;  // nothing but filler
    // terminals in position order, their pairs, the dense run
    {   // rank of every bitmap word = exclusive prefix of the popcounts
        const u32 nb = (u32)((n_words + KMC_SCAN_PER_BLOCK - 1) / KMC_SCAN_PER_BLOCK);
        hipLaunchKernelGGL(kmc_scan_sums_kernel<1>, dim3(nb), dim3(256), 0, c->stream, (const void*)c->m_bitmap.p, (u32)n_words, (u32*)c->m_bsum.p);
        hipLaunchKernelGGL(kmc_scan_top_kernel, dim3(1), dim3(1024), 0, c->stream, (u32*)c->m_bsum.p, nb, &ctl->scan_total);
        hipLaunchKernelGGL(kmc_scan_final_kernel<1>, dim3(nb), dim3(256), 0, c->stream, (const void*)c->m_bitmap.p, (u32)n_words, (const u32*)c->m_bsum.p, (u32*)c->m_rank.p);
    }
    hipLaunchKernelGGL(kmc_msd_order_kernel, dim3(grid_for(c, n_term, 256)), dim3(256), 0, c->stream, (const MsdTerm*)c->m_term.p, n_term,
                       (const unsigned long long*)c->m_bitmap.p, (const u32*)c->m_rank.p, (MsdTerm*)c->m_ord.p, (u32*)c->m_clist.p, ctl);
    const u32 n_cnt = c->h_ctl->n_cnt;   // (as of the last level's poll: only the levels' scans add to it)
    // the run the leaves write into: at most one pair per valid key
    const u64 run_cap = std::max<u64>(weights ? n : std::min<u64>(n, (u64)c->h_ctl->n_valid), 1);
    kmc_ctx::Run run;
    rc = take_run(c, run_cap, &run);
    if (rc) return rc;
#define MSD_LEAF(KWV, WV, CAPV, SCRV)                                                                                                       \
    do {                                                                                                                                    \
        static std::atomic<unsigned long long> attr{0};                                                                                     \
        if (kmc_attr_once(attr)) (void)hipFuncSetAttribute((const void*)kmc_msd_leaf_kernel<KWV, WV, CAPV, SCRV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(MsdLeafLds<KWV, WV, CAPV, SCRV>)); \
        hipLaunchKernelGGL((kmc_msd_leaf_kernel<KWV, WV, CAPV, SCRV>), dim3(n_term), dim3(KMC_MSD_THREADS), sizeof(MsdLeafLds<KWV, WV, CAPV, SCRV>), c->stream, \
                           (const u64*)hi[0], (const u64*)lo[0], (const u64*)(weights ? w[0] : nullptr), (const u64*)hi[1], (const u64*)lo[1], (const u64*)(weights ? w[1] : nullptr), \
                           (const MsdTerm*)c->m_ord.p, n_term, (int)kb, hi[0], lo[0], (u64*)(weights ? w[0] : nullptr), hi[1], lo[1], (u64*)(weights ? w[1] : nullptr), \
                           run.hi, run.lo, run.cnt, (u32*)c->m_nd.p, ctl);                                                                  \
    } while (0)
    // (leaf size and wave scratch by what the keys look like: kmc_msd.hip.h, MsdLeafLds)
    if (KW == 1) {
        if (weights) MSD_LEAF(1, true, KMC_MSD_LEAF1, 128);
        else if (clustered) MSD_LEAF(1, false, KMC_MSD_LEAF1, 256);
        else MSD_LEAF(1, false, KMC_MSD_LEAF1, 2);   // (seven leaves per CU at 512-byte granules)
    } else if (weights) MSD_LEAF(2, true, KMC_MSD_LEAF2W, 64);
    else if (leaf_cap > 1024) MSD_LEAF(2, false, KMC_MSD_LEAF2, 16);
    else MSD_LEAF(2, false, 1024, 128);
#undef MSD_LEAF
    if (n_cnt) {
        if (n_cnt > max_seg) { c->run_pool.push_back(run); return fail(c, KMC_ERR_HIP, "msd sort: more counted spans than segments (internal error)"); }
        if (KW == 1) hipLaunchKernelGGL(kmc_msd_count_kernel<1>, dim3(n_cnt), dim3(1024), 0, c->stream, (const u64*)hi[0], (const u64*)lo[0], (const u64*)hi[1], (const u64*)lo[1],
                                        (const MsdTerm*)c->m_ord.p, (const u32*)c->m_clist.p, n_cnt, run.hi, run.lo, run.cnt, (u32*)c->m_nd.p, ctl);
        else hipLaunchKernelGGL(kmc_msd_count_kernel<2>, dim3(n_cnt), dim3(1024), 0, c->stream, (const u64*)hi[0], (const u64*)lo[0], (const u64*)hi[1], (const u64*)lo[1],
                                (const MsdTerm*)c->m_ord.p, (const u32*)c->m_clist.p, n_cnt, run.hi, run.lo, run.cnt, (u32*)c->m_nd.p, ctl);
    }
    auto give_back = [&](int code, const char* what) { c->run_pool.push_back(run); return fail(c, code, "msd sort: %s", what); };
    if (hipGetLastError() != hipSuccess) return give_back(KMC_ERR_HIP, "leaf launch failed");
    if (hipMemcpyAsync(c->h_ctl, ctl, sizeof(MsdCtl), hipMemcpyDeviceToHost, c->stream) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess)
        return give_back(KMC_ERR_HIP, "leaf kernel failed");
    if (c->h_ctl->overflow) return give_back(KMC_ERR_CAPACITY, "segment / terminal list overflow");
    const u64 n_keys_in = (u64)c->h_ctl->n_valid;
    u64 n_dups = 0;
    for (u32 d : c->h_ctl->n_dups) n_dups += d;
    const u64 n_pairs = n_keys_in - std::min<u64>(n_keys_in, n_dups);
    if (n_dups) {
        // some keys repeat: the run has holes behind the terminals that hold them -- scan the pair counts, move the pairs
        // together into a run of the right size (work proportional to the distinct keys), give the sparse one back
        kmc_ctx::Run dense;
        rc = take_run(c, std::max<u64>(n_pairs, 1), &dense);
        if (rc) { c->run_pool.push_back(run); return rc; }
        const u32 nb = (n_term + KMC_SCAN_PER_BLOCK - 1) / KMC_SCAN_PER_BLOCK;
        hipLaunchKernelGGL(kmc_scan_sums_kernel<0>, dim3(nb), dim3(256), 0, c->stream, (const void*)c->m_nd.p, n_term, (u32*)c->m_bsum.p);
        hipLaunchKernelGGL(kmc_scan_top_kernel, dim3(1), dim3(1024), 0, c->stream, (u32*)c->m_bsum.p, nb, &ctl->n_pairs);
        hipLaunchKernelGGL(kmc_scan_final_kernel<0>, dim3(nb), dim3(256), 0, c->stream, (const void*)c->m_nd.p, n_term, (const u32*)c->m_bsum.p, (u32*)c->m_base.p);
        if (KW == 1) hipLaunchKernelGGL(kmc_msd_gather_kernel<1>, dim3(grid_for(c, (u64)n_term * 64, 256)), dim3(256), 0, c->stream, (const MsdTerm*)c->m_ord.p, n_term,
                                        (const u32*)c->m_nd.p, (const u32*)c->m_base.p, (const u64*)run.hi, (const u64*)run.lo, (const u64*)run.cnt, dense.hi, dense.lo, dense.cnt);
        else hipLaunchKernelGGL(kmc_msd_gather_kernel<2>, dim3(grid_for(c, (u64)n_term * 64, 256)), dim3(256), 0, c->stream, (const MsdTerm*)c->m_ord.p, n_term,
                                (const u32*)c->m_nd.p, (const u32*)c->m_base.p, (const u64*)run.hi, (const u64*)run.lo, (const u64*)run.cnt, dense.hi, dense.lo, dense.cnt);
        c->run_pool.push_back(run);   // (stream order: the gather is queued before anything can reuse it)
        run = dense;
        if (hipGetLastError() != hipSuccess) return give_back(KMC_ERR_HIP, "gather launch failed");
    }
    if (n_pairs > run_cap) return give_back(KMC_ERR_HIP, "more pairs than keys (internal error)");
    run.n = n_pairs;
    run.total = weights ? (u64)c->h_ctl->w_total : (u64)c->h_ctl->n_valid;  // what the run's counts sum to
    run.total_known = true;
    if (!weights && c->h_ctl->n_valid >= (1u << 20)) c->msd_dup_heavy = (u64)c->h_ctl->n_valid >= 4 * std::max<u64>(n_pairs, 1);
    if (out) *out = run;
    else if (n_pairs) c->runs.push_back(run);
    else c->run_pool.push_back(run);
    return KMC_OK;
}

// Count the windows ending in [range_begin, n_bases) by extract -> sort -> run-length, in sub-batches
// of at most 2^31 base positions (2 x 16-32 GiB of keys in flight).  Each sub-batch leaves one run.
// Sort what the batches since the last flush have extracted: one run.
int flush_acc(kmc_ctx* c) {
    if (!c->acc_n) return KMC_OK;
    const u64 n = c->acc_n;
    int rc = ensure(c, c->s_lo[1], (size_t)n * sizeof(u64));
    if (rc) return rc;
    if (c->KW == 2) { rc = ensure(c, c->s_hi[1], (size_t)n * sizeof(u64)); if (rc) return rc; }
    rc = launch_begin(c);  // (the event pair brackets the whole sort: levels, leaves, gather)
    if (rc) return rc;
    u64* const khi[2] = {(u64*)c->s_hi[0].p, (u64*)c->s_hi[1].p};
    u64* const klo[2] = {(u64*)c->s_lo[0].p, (u64*)c->s_lo[1].p};
    u64* const kwt[2] = {nullptr, nullptr};
    c->acc_n = 0;  // (whatever happens below, these keys are not sorted twice)
    rc = msd_sort_to_run(c, khi, klo, kwt, n, 2u * (unsigned)c->klen, c->KW, nullptr, false, true);
    if (rc) return rc;
    rc = launch_end(c);
    if (rc) return rc;
    c->pending = true;
    return KMC_OK;
}

int run_sort_path(kmc_ctx* c, const uint8_t* d_bases, const u64* d_offsets, u64 n_reads, u64 n_bases, u64 range_begin) {
    const u64 n_chunks = (n_bases + KMC_CHUNK - 1) / KMC_CHUNK;
    const u64 SB = 1ull << 21;  // chunks per sort (2^31 positions: the run kernels index with u32)
    const u64 CPR = KMC_MSD_RANGE / KMC_CHUNK;   // chunks per range of the sort
    for (u64 cb = range_begin / KMC_CHUNK; cb < n_chunks; cb += SB) {
        const u64 ce = std::min(n_chunks, cb + SB);
        // (every batch's keys start on a range boundary of the accumulated array: its last range is padded with filler)
        const u64 n_ranges = (ce - cb + CPR - 1) / CPR;
        const u64 n = n_ranges * KMC_MSD_RANGE;
        if (c->acc_n + n > SB * KMC_CHUNK) { int rc = flush_acc(c); if (rc) return rc; }
        // room behind the keys already there (sized by the caller's hint the first time)
        const u64 want = std::max<u64>(c->acc_n + n, std::min<u64>(c->acc_hint + KMC_MSD_RANGE, SB * KMC_CHUNK));
        int rc = ensure_keep(c, c->s_lo[0], (size_t)(c->s_lo[0].bytes >= (c->acc_n + n) * sizeof(u64) ? (c->acc_n + n) : want) * sizeof(u64), (size_t)c->acc_n * sizeof(u64));
        if (rc) return rc;
        if (c->KW == 2) {
            rc = ensure_keep(c, c->s_hi[0], (size_t)(c->s_hi[0].bytes >= (c->acc_n + n) * sizeof(u64) ? (c->acc_n + n) : want) * sizeof(u64), (size_t)c->acc_n * sizeof(u64));
            if (rc) return rc;
        }
        // the level-0 histogram rows and AND / OR words of the accumulated ranges
        const u64 r0 = c->acc_n / KMC_MSD_RANGE, r_have = r0 + n_ranges, r_want = std::max<u64>(r_have, want / KMC_MSD_RANGE + 1);
        const size_t hb_have = msd_hist_words(r_have) * sizeof(u32), hb_want = msd_hist_words(r_want) * sizeof(u32);
        rc = ensure_keep(c, c->a_hist, c->a_hist.bytes >= hb_have ? hb_have : hb_want, msd_hist_words(r0) * sizeof(u32));
        if (rc) return rc;
        rc = ensure_keep(c, c->a_rand, (c->a_rand.bytes >= r_have * 16 ? r_have : r_want) * 16, (size_t)r0 * 16);
        if (rc) return rc;
        rc = ensure_keep(c, c->a_ror, (c->a_ror.bytes >= r_have * 16 ? r_have : r_want) * 16, (size_t)r0 * 16);
        if (rc) return rc;
        rc = launch_begin(c);  // (this event pair brackets the extraction; the sort has its own at the flush)
        if (rc) return rc;
        rc = launch_extract(c, d_bases, d_offsets, n_reads, n_bases, cb, ce, range_begin, (u32)n_ranges,
                            c->KW == 2 ? (u64*)c->s_hi[0].p + c->acc_n : nullptr, (u64*)c->s_lo[0].p + c->acc_n,
                            (u32*)c->a_hist.p, (u32)r0, (u64*)c->a_rand.p + 2 * r0, (u64*)c->a_ror.p + 2 * r0);
        if (rc) return rc;
        rc = launch_end(c);
        if (rc) return rc;
        if (!c->tb.empty()) c->tb.back().count_launches++;
        c->acc_n += n;
    }
    c->pending = true;
    return KMC_OK;
}

// Save the table in front of a risky launch (see kmc_ctx::Risky).  Returns false when the table is too
// large to be saved cheaply: the caller then keeps the launch within what is certain to fit.
bool arm_risky(kmc_ctx* c, const uint8_t* d_bases, const u64* d_offsets, u64 n_reads, u64 n_bases, u64 base_from, const u64* d_from) {
    const u64 occ = c->h_counters[KMC_CTR_OCCUPIED];
    if (c->polled_epoch != c->table_epoch) c->st.n_planner_stale++;   // (the counters this decision rests on are older than a queued unfold / merge)
    kmc_ctx::Risky& r = c->risky;
    GTable g = gtable_of(c, c->tab);
    r.empty = false;
    if (occ == 0 && c->h_counters[KMC_CTR_SPILL] == 0 && !c->pending) {
        // a table that is known to be empty (the usual case: reset, then one batch) needs no snapshot at all
        r.mode = 1;
        r.empty = true;
    } else if (occ <= KMC_OCC_LIST_CAP && c->occ_list && c->h_counters[KMC_CTR_SPILL] == 0) {
        const size_t nb = (size_t)KMC_OCC_LIST_CAP * sizeof(u64);
        if (ensure(c, c->snap_lo, nb) || ensure(c, c->snap_cnt, nb) || ensure(c, c->snap_n, 64) || (c->KW == 2 && ensure(c, c->snap_hi, nb))) return false;
        if (c->KW == 1) hipLaunchKernelGGL(kmc_snapshot_kernel<1>, dim3(occ > 32768 ? 256 : 32), dim3(256), 0, c->stream, g, (u64*)nullptr, (u64*)c->snap_lo.p, (u64*)c->snap_cnt.p, (u64*)c->snap_n.p);
        else hipLaunchKernelGGL(kmc_snapshot_kernel<2>, dim3(occ > 32768 ? 256 : 32), dim3(256), 0, c->stream, g, (u64*)c->snap_hi.p, (u64*)c->snap_lo.p, (u64*)c->snap_cnt.p, (u64*)c->snap_n.p);
        if (hipGetLastError() != hipSuccess) return false;
        r.mode = 1;
    } else if (c->tab.cap <= (16ull << 20)) {
        const size_t nb = (size_t)c->tab.cap * sizeof(u64);
        if (ensure(c, c->snap_lo, nb) || ensure(c, c->snap_cnt, nb) || (c->KW == 2 && ensure(c, c->snap_hi, nb)) || ensure(c, c->snap_occ, 3 * (size_t)KMC_OCC_LIST_CAP * sizeof(u64))) return false;
        bool ok = hipMemcpyAsync(c->snap_lo.p, c->tab.lo, nb, hipMemcpyDeviceToDevice, c->stream) == hipSuccess &&
                  hipMemcpyAsync(c->snap_cnt.p, c->tab.cnt, nb, hipMemcpyDeviceToDevice, c->stream) == hipSuccess &&
                  hipMemcpyAsync(c->snap_occ.p, c->occ_list, (size_t)KMC_OCC_LIST_CAP * sizeof(u64), hipMemcpyDeviceToDevice, c->stream) == hipSuccess &&
                  hipMemcpyAsync((u64*)c->snap_occ.p + KMC_OCC_LIST_CAP, c->occ_key_lo, (size_t)KMC_OCC_LIST_CAP * sizeof(u64), hipMemcpyDeviceToDevice, c->stream) == hipSuccess &&
                  hipMemcpyAsync((u64*)c->snap_occ.p + 2 * (size_t)KMC_OCC_LIST_CAP, c->occ_key_hi, (size_t)KMC_OCC_LIST_CAP * sizeof(u64), hipMemcpyDeviceToDevice, c->stream) == hipSuccess;
        if (ok && c->KW == 2) ok = hipMemcpyAsync(c->snap_hi.p, c->tab.hi, nb, hipMemcpyDeviceToDevice, c->stream) == hipSuccess;
        if (!ok) { (void)hipGetLastError(); return false; }
        r.mode = 2;
    } else {
        return false;
    }
    r.armed = true;
    r.d_bases = d_bases; r.d_offsets = d_offsets; r.n_reads = n_reads; r.n_bases = n_bases;
    r.base_from = base_from; r.d_from = d_from;
    memcpy(r.ctr, c->h_counters, sizeof(r.ctr));
    return true;
}

// The spill area overflowed in a risky launch: some (key, count) pairs were dropped, so the table is put
// back as it was in front of that launch and everything from the launch's first base position to the
// end of the batch is counted by the sort path (extract, sort, run-length: no table, any cardinality).
int recover_overflow(kmc_ctx* c) {
    kmc_ctx::Risky r = c->risky;
    c->risky.armed = false;
    GTable g = gtable_of(c, c->tab);
    u64 ctr[KMC_CTR_N];
    memcpy(ctr, r.ctr, sizeof(ctr));
    ctr[KMC_CTR_ERR] = 0;
    ctr[KMC_CTR_SPILL] = 0;
    ctr[KMC_CTR_FASTFIN] = 0;  // (a speculative finalize queued in front of this poll looked at the overflowed table)
    if (r.mode == 1) {
        u64 n_snap = 0;
        if (!r.empty) {
            HIPCHK(c, hipMemcpyAsync(&n_snap, c->snap_n.p, sizeof(u64), hipMemcpyDeviceToHost, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
        }
        if (n_snap == ~0ull) return fail(c, KMC_ERR_CAPACITY, "count table and spill area exhausted and the table could not be restored; raise capacity_hint");
        const int grid = reset_grid(c);
        if (c->KW == 1) hipLaunchKernelGGL(kmc_reset_kernel<1>, dim3(grid), dim3(256), 0, c->stream, g, sk_table_of(c), c->fin_rank + 3);
        else hipLaunchKernelGGL(kmc_reset_kernel<2>, dim3(grid), dim3(256), 0, c->stream, g, sk_table_of(c), c->fin_rank + 3);
        ctr[KMC_CTR_OCCUPIED] = 0;  // (the merge below claims the slots again and counts them)
        HIPCHK(c, hipMemcpyAsync(c->d_counters, ctr, sizeof(ctr), hipMemcpyHostToDevice, c->stream));
        if (n_snap) {
            if (c->KW == 1) hipLaunchKernelGGL(kmc_merge_pairs_kernel<1>, dim3(grid_for(c, n_snap, 256)), dim3(256), 0, c->stream, g, (const u64*)nullptr, (const u64*)c->snap_lo.p, (const u64*)c->snap_cnt.p, n_snap);
            else hipLaunchKernelGGL(kmc_merge_pairs_kernel<2>, dim3(grid_for(c, n_snap, 256)), dim3(256), 0, c->stream, g, (const u64*)c->snap_hi.p, (const u64*)c->snap_lo.p, (const u64*)c->snap_cnt.p, n_snap);
        }
        HIPCHK(c, hipGetLastError());
    } else {
        const size_t nb = (size_t)c->tab.cap * sizeof(u64);
        HIPCHK(c, hipMemcpyAsync(c->tab.lo, c->snap_lo.p, nb, hipMemcpyDeviceToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(c->tab.cnt, c->snap_cnt.p, nb, hipMemcpyDeviceToDevice, c->stream));
        if (c->KW == 2) HIPCHK(c, hipMemcpyAsync(c->tab.hi, c->snap_hi.p, nb, hipMemcpyDeviceToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(c->occ_list, c->snap_occ.p, (size_t)KMC_OCC_LIST_CAP * sizeof(u64), hipMemcpyDeviceToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(c->occ_key_lo, (u64*)c->snap_occ.p + KMC_OCC_LIST_CAP, (size_t)KMC_OCC_LIST_CAP * sizeof(u64), hipMemcpyDeviceToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(c->occ_key_hi, (u64*)c->snap_occ.p + 2 * (size_t)KMC_OCC_LIST_CAP, (size_t)KMC_OCC_LIST_CAP * sizeof(u64), hipMemcpyDeviceToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(c->d_counters, ctr, sizeof(ctr), hipMemcpyHostToDevice, c->stream));
    }
    { int rs = sk_clear(c); if (rs) return rs; }  // (its counts belong to the launch that is being undone)
    c->sk_dirty = false;
    u64 from = r.base_from;
    if (r.d_from) HIPCHK(c, hipMemcpyAsync(&from, r.d_from, sizeof(u64), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));  // (ctr[] and `from` are stack memory)
    memcpy(c->h_counters, ctr, sizeof(ctr));
    c->polled_epoch = c->table_epoch;   // (the table is what the snapshot says)
    c->h_counters[KMC_CTR_OCCUPIED] = r.ctr[KMC_CTR_OCCUPIED];
    c->direct_seen = ctr[KMC_CTR_BADBASE];
    c->kmers_seen = ctr[KMC_CTR_KMERS];
    // this data source is not what the history said: forget it, and leave the per-occurrence kernels alone
    c->rho_hist = -1.0;
    c->rho_last = 0.0;
    c->b_open = false;
    if (c->cfg.algo == KMC_ALGO_AUTO) c->prefer_sort = true;
    c->st.algo_last = KMC_ALGO_SORT;
    c->recovered = true;
    return run_sort_path(c, r.d_bases, r.d_offsets, r.n_reads, r.n_bases, from);
}

// AUTO hands a batch to the sort path after a launch or two of the walk kernel.  When the table held nothing at
// the start of the batch, what those launches put into it is dropped and the WHOLE batch is sorted: the table
// stays empty, so kmc_finalize has a single run to show (zero copy) instead of merging 14 M table entries with
// a 745 M-entry run by one more sort of everything (first step on 1 GB of random 63-mers: 100 ms of kernels
// and 60 GB of buffers less).  ctr0 = the device counters at the start of the batch.
int drop_batch_from_table(kmc_ctx* c, const u64* ctr0) {
    GTable g = gtable_of(c, c->tab);
    const int grid = reset_grid(c);
    if (c->KW == 1) hipLaunchKernelGGL(kmc_reset_kernel<1>, dim3(grid), dim3(256), 0, c->stream, g, sk_table_of(c), c->fin_rank + 3);
    else hipLaunchKernelGGL(kmc_reset_kernel<2>, dim3(grid), dim3(256), 0, c->stream, g, sk_table_of(c), c->fin_rank + 3);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(c->d_counters, ctr0, KMC_CTR_N * sizeof(u64), hipMemcpyHostToDevice, c->stream));
    { int rs = sk_clear(c); if (rs) return rs; }
    c->sk_dirty = false;
    HIPCHK(c, hipStreamSynchronize(c->stream));  // (ctr0 is the caller's stack memory)
    memcpy(c->h_counters, ctr0, KMC_CTR_N * sizeof(u64));
    c->polled_epoch = c->table_epoch;
    c->direct_seen = ctr0[KMC_CTR_BADBASE];
    c->kmers_seen = ctr0[KMC_CTR_KMERS];
    c->risky.armed = false;
    c->fin_parity = 0;
    return KMC_OK;
}

// pieces of at most KMC_WALK_MAX_READ bases for a batch with longer reads (kmc_walk.hip.h): vr_reads = [starts | ends]
int build_vreads(kmc_ctx* c, const u64* d_offsets, u64 n_reads, u64* n_v_out) {
    // pieces per read (u32), their exclusive prefix (the three-kernel scan of kmc_msd.hip.h), then the pieces
    if (n_reads >= (1ull << 32)) return fail(c, KMC_ERR_ARG, "batch too large for one walk pass");
    const u32 nb = (u32)((n_reads + KMC_SCAN_PER_BLOCK - 1) / KMC_SCAN_PER_BLOCK);
    int rc = ensure(c, c->vr_cnt, (size_t)n_reads * sizeof(u32));
    if (rc) return rc;
    rc = ensure(c, c->vr_pos, (size_t)n_reads * sizeof(u32));
    if (rc) return rc;
    rc = ensure(c, c->m_bsum, (size_t)(nb + 2) * sizeof(u32));
    if (rc) return rc;
    rc = ensure(c, c->m_ctl, sizeof(MsdCtl));
    if (rc) return rc;
    if (!c->h_ctl) HIPCHK(c, hipHostMalloc((void**)&c->h_ctl, sizeof(MsdCtl)));
    MsdCtl* ctl = (MsdCtl*)c->m_ctl.p;
    u32 *cnt = (u32*)c->vr_cnt.p, *pos = (u32*)c->vr_pos.p;
    hipLaunchKernelGGL(kmc_vreads_count_kernel, dim3(grid_for(c, n_reads, 256)), dim3(256), 0, c->stream, d_offsets, n_reads, c->cfg.k, cnt);
    hipLaunchKernelGGL(kmc_scan_sums_kernel<0>, dim3(nb), dim3(256), 0, c->stream, (const void*)cnt, (u32)n_reads, (u32*)c->m_bsum.p);
    hipLaunchKernelGGL(kmc_scan_top_kernel, dim3(1), dim3(1024), 0, c->stream, (u32*)c->m_bsum.p, nb, &ctl->scan_total);
    hipLaunchKernelGGL(kmc_scan_final_kernel<0>, dim3(nb), dim3(256), 0, c->stream, (const void*)cnt, (u32)n_reads, (const u32*)c->m_bsum.p, pos);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(c->h_ctl, ctl, sizeof(MsdCtl), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    const u64 n_v = c->h_ctl->scan_total;
    rc = ensure(c, c->vr_reads, (size_t)n_v * 2 * sizeof(u64));
    if (rc) return rc;
    u64* vs = (u64*)c->vr_reads.p;
    hipLaunchKernelGGL(kmc_vreads_fill_kernel, dim3(grid_for(c, n_v, 256)), dim3(256), 0, c->stream, d_offsets, (const u32*)pos, n_reads, n_v, c->cfg.k, vs, vs + n_v);
    HIPCHK(c, hipGetLastError());
    *n_v_out = n_v;
    return KMC_OK;
}

void harvest_timing(kmc_ctx* c);
// the three kernels behind a walk launch that logged (kmc_sklog.hip.h): bin totals, partition, count + unfold
int launch_sklog(kmc_ctx* c, const SkLog& lg, u32 wgrid) {
    GTable g = gtable_of(c, c->tab);
    const bool canon = c->cfg.canonical != 0;
    const int k = c->cfg.k;
    u64* binned = (u64*)c->lg_bins.p;
    u32* total = (u32*)c->lg_cursor.p;              // [1024] records per bin
    u32* cur = total + KMC_SKLOG_BINS;              // [1024] reservations so far
    u32* span_hist = (u32*)c->lg_count.p + c->n_cu; // [wgrid][1024]
#define SKLOG_LAUNCH(KWV, CAN, WV)                                                                                                           \
    do {                                                                                                                                     \
        static std::atomic<unsigned long long> attr{0}, attr2{0};                                                                            \
        if (kmc_attr_once(attr)) (void)hipFuncSetAttribute((const void*)kmc_sklog_consume_kernel<KWV, CAN, WV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(SklogTable<WV>)); \
        if (kmc_attr_once(attr2)) (void)hipFuncSetAttribute((const void*)kmc_sklog_partition_kernel<WV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(SklogPartLds<WV>)); \
        hipLaunchKernelGGL((kmc_sklog_hist_kernel<WV>), dim3(wgrid), dim3(1024), 0, c->stream, (const u64*)lg.rec, (const u32*)lg.count, lg.cap_wg, span_hist, total); \
        hipLaunchKernelGGL((kmc_sklog_partition_kernel<WV>), dim3(wgrid), dim3(1024), sizeof(SklogPartLds<WV>), c->stream, (const u64*)lg.rec, (const u32*)lg.count, lg.cap_wg, \
                           (const u32*)span_hist, (const u32*)total, cur, binned);                                                           \
        hipLaunchKernelGGL((kmc_sklog_consume_kernel<KWV, CAN, WV>), dim3(KMC_SKLOG_BINS), dim3(1024), sizeof(SklogTable<WV>), c->stream, (const u64*)binned, (const u32*)total, k, g); \
    } while (0)
    if (lg.words == 2) {
        if (c->KW == 1) { if (canon) SKLOG_LAUNCH(1, true, 2); else SKLOG_LAUNCH(1, false, 2); }
        else { if (canon) SKLOG_LAUNCH(2, true, 2); else SKLOG_LAUNCH(2, false, 2); }
    } else {
        if (canon) SKLOG_LAUNCH(2, true, 3); else SKLOG_LAUNCH(2, false, 3);
    }
#undef SKLOG_LAUNCH
    HIPCHK(c, hipGetLastError());
    return KMC_OK;
}

// kernel milliseconds per base of the sort path on all-distinct reads (1 GB FASTA = 0.914 G bases; profiles/r03_sort_*):
// the prior of KMC_ALGO_AUTO's cost comparison until the ctx has sorted something itself
#define KMC_SORT_MS_PER_BASE_1 2.9e-8
#define KMC_SORT_MS_PER_BASE_2 5.0e-8
int count_batch_device_body(kmc_ctx* c, const uint8_t* d_bases, const u64* d_offsets, u64 n_reads, u64 n_bases, u64 max_read_len);
int count_batch_device(kmc_ctx* c, const uint8_t* d_bases, const u64* d_offsets, u64 n_reads, u64 n_bases, u64 max_read_len) {
    const int rc = count_batch_device_body(c, d_bases, d_offsets, n_reads, n_bases, max_read_len);
    // the events of finished batches are read here, BEHIND this batch's launches (or in kmc_get_stats / kmc_poll): in front of
    // them the microseconds would lie between a step's synchronisation and its successor's first launch
    if (c->tb.size() >= 4) harvest_timing(c);
    return rc;
}
int count_batch_device_body(kmc_ctx* c, const uint8_t* d_bases, const u64* d_offsets, u64 n_reads, u64 n_bases, u64 max_read_len) {
    { int rc = resolve_async(c); if (rc) return rc; }
    { int rc = undrain(c); if (rc) return rc; }
    if (c->pending) { int rc = poll_and_settle(c); if (rc) return rc; }
    // the previous batch's (k+16)-mer counts, if it left any: the counters are as of a poll that came after
    // its last launch (every launch sets `pending`), so an empty table is known to be empty
    {
        const bool unfolds = c->sk_dirty && c->h_sk_counters && c->h_sk_counters[KMC_CTR_KMERS] != 0;
        int rc = settle_sk_polled(c);
        if (rc) return rc;
        // the unfold just queued changes the table: the launch planner (and the table it saves in front of a risky
        // launch) must see the table as it will be, not as the poll above found it.  (Without this second poll a
        // second high-cardinality batch under KMC_ALGO_WALK saved "a small table" that held millions of entries by
        // the time the snapshot ran: counts of the first batch were lost or KMC_ERR_CAPACITY raised --
        // tools/stress_sort_lr.py found it, test_walk_two_high_cardinality_batches pins it.)
        if (unfolds) { rc = poll_and_settle(c); if (rc) return rc; }
    }
    c->recovered = false;
    c->sorted_valid = false;
    // (for drop_batch_from_table: did this batch start on an empty table?  The counters are as of a poll.)
    const bool batch_on_empty_table = !c->pending && c->h_counters[KMC_CTR_OCCUPIED] == 0 && c->h_counters[KMC_CTR_SPILL] == 0;
    u64 ctr0[KMC_CTR_N];
    memcpy(ctr0, c->h_counters, sizeof(ctr0));
    c->st.n_reads += n_reads;
    c->st.n_bases += n_bases;
    c->st.n_batches += 1;
    if (!n_reads || !n_bases) return KMC_OK;

    int algo = c->cfg.algo;
    if (c->cfg.mode == KMC_MODE_LR) algo = KMC_ALGO_STREAM;  // LR runs its own kernel (kmc_lr.hip.h)
    if (algo == KMC_ALGO_AUTO && !c->prefer_sort && c->walk_ms_per_base > 0) {
        // by measured cost: the walk path (memo overflow: logged steps, (k+16)-mer table updates, unfold, table merge) against what
        // sorting costs -- this ctx's own measurement on this source when it has one (heavily repeated keys sort at half the
        // rate of distinct ones), else the rate of all-distinct reads on this GPU.  Both ways: a source that
        // turned out to sort slower than it walks goes back to walking.
        const double sort_est = c->sort_ms_per_base > 0 ? c->sort_ms_per_base : (c->KW == 1 ? KMC_SORT_MS_PER_BASE_1 : KMC_SORT_MS_PER_BASE_2);
        c->sort_by_cost = c->walk_ms_per_base > 1.1 * sort_est;
    }
    if (algo == KMC_ALGO_AUTO && (c->prefer_sort || c->sort_by_cost)) algo = KMC_ALGO_SORT;
    if (algo == KMC_ALGO_AUTO || algo == KMC_ALGO_WALK) {
        if (!max_read_len) {
            HIPCHK(c, hipMemsetAsync(&c->d_counters[KMC_CTR_MAXLEN], 0, sizeof(u64), c->stream));
            hipLaunchKernelGGL(kmc_maxlen_kernel, dim3(grid_for(c, n_reads, 256)), dim3(256), 0, c->stream, d_offsets, n_reads, c->d_counters);
            HIPCHK(c, hipGetLastError());
            int rc = poll(c);
            if (rc) return rc;
            max_read_len = c->h_counters[KMC_CTR_MAXLEN];
        }
        bool walk_ok = kmc_walk_supported(c->cfg.k, c->cfg.mode, max_read_len) && n_reads < (1ull << 32);
        if (algo == KMC_ALGO_AUTO && c->walk_overflowed) walk_ok = false;  // high-cardinality input: memo tables do not help
        if (algo == KMC_ALGO_WALK && !walk_ok)
            return fail(c, KMC_ERR_ARG, "KMC_ALGO_WALK needs contiguous-k mode, k <= %d and fewer than 2^32 reads", KMC_WALK_MAX_K);
        algo = walk_ok ? KMC_ALGO_WALK : KMC_ALGO_STREAM;
    }
    c->st.algo_last = algo;

    c->tb.emplace_back();    // this batch's launch events
    c->tb.back().n_bases = n_bases;
    int rc = KMC_OK;
    bool batch_mixed = false;   // the batch changed path in the middle: not a rate sample of either
    {
        // Sub-batches.  A launch over n k-mers can add at most n new keys, so without history the
        // first launch is sized to what the table and spill area absorb for certain and later ones
        // ramp up (x16 at most) using the observed ratio rho = new keys per k-mer.  With history
        // (the previous batch on this ctx, kept across kmc_reset) the batch goes out in as few
        // launches as 8 x the predicted number of new keys allows -- one for the benchmark input.
        // A launch larger than what is certain to fit is "risky": the table is saved in front of it
        // (arm_risky) and, should the prediction be wrong enough to exhaust table AND spill area, the
        // next poll puts the table back and counts the rest of the batch by sorting (recover_overflow):
        // nothing is dropped and nothing fails.  The caller's device buffers must therefore stay valid
        // until the next call on the ctx that synchronises (kmc.h, kmc_add_batch_device).
        double batch_rho_max = 0.0;
        u64 plan_safe = 0;  // units of the last plan() that fit for certain
        c->b_occ0 = c->h_counters[KMC_CTR_OCCUPIED];
        c->b_kmers = std::max<u64>(n_bases, 1);
        auto plan = [&](u64 units_left, u64 kmers_per_unit, u64 prev) -> u64 {
            u64 occ = c->h_counters[KMC_CTR_OCCUPIED];
            u64 freeslots = (c->tab.cap * 7 / 10 > occ ? c->tab.cap * 7 / 10 - occ : 0) + c->spill_cap / 2;
            u64 safe = std::max<u64>(freeslots / kmers_per_unit, 1);
            plan_safe = safe;
            u64 take = safe;
            if (prev) {
                double opt = (double)freeslots / (4.0 * std::max(c->rho_last, 1e-9)) / (double)kmers_per_unit;
                take = std::max<u64>(safe, (u64)std::min<double>(opt, (double)prev * 16.0));
            } else if (c->rho_hist >= 0.0) {
                // history says rho_hist new keys per k-mer.  A launch may bring 8 x that -- 3 x when the table is known to be
                // empty: then "saving" it costs nothing, a whole batch of the same source brings what the last one brought
                // (rho_hist is a whole-batch ratio), and a table with room for 3 x of it takes the batch in ONE launch
                // (pool 100 of the cardinality sweep went out in four launches with a host poll behind each: 2.9 ms per GB)
                const bool empty = occ == 0 && c->h_counters[KMC_CTR_SPILL] == 0 && !c->pending;
                double opt = (double)freeslots / ((empty ? 3.0 : 8.0) * std::max(c->rho_hist, 1e-9)) / (double)kmers_per_unit;
                take = std::max<u64>(safe, (u64)std::min<double>(opt, 1e18));
            }
            return std::min<u64>(take, units_left);
        };
        double observed_kmers = 0.0;
        u64 sk_seen = c->sk.lo ? c->h_sk_counters[KMC_CTR_OCCUPIED] + c->h_sk_counters[KMC_CTR_SPILL] : 0;
        auto observe = [&](u64 occ_before, u64 units, u64 kmers_per_unit) -> int {
            int r = poll_and_settle(c);
            if (r) return r;
            if (c->recovered) return KMC_OK;
            u64 occ_after = c->h_counters[KMC_CTR_OCCUPIED];
            // keys still waiting in the (k+16)-mer table count as well: each entry becomes up to 16 k-mers at the unfold
            const u64 sk_after = c->sk.lo ? c->h_sk_counters[KMC_CTR_OCCUPIED] + c->h_sk_counters[KMC_CTR_SPILL] : 0;
            const u64 sk_new = sk_after > sk_seen ? sk_after - sk_seen : 0;
            sk_seen = sk_after;
            double rho = ((double)(occ_after > occ_before ? occ_after - occ_before : 0) + 16.0 * (double)sk_new) / ((double)units * (double)kmers_per_unit);
            c->rho_last = rho;
            c->rho_max = std::max(c->rho_max, rho);
            batch_rho_max = std::max(batch_rho_max, rho);
            // history for later batches from what has been observed of this one so far; the poll after
            // the batch's last launch (kmc_finalize) replaces it with the whole batch's ratio -- but a
            // caller that never finalizes this ctx (multi-GPU reduce: the live table is packed and the
            // ctx reset) must not start every batch without history
            observed_kmers += (double)units * (double)kmers_per_unit;
            u64 occ_all = occ_after + c->h_counters[KMC_CTR_SPILL];
            c->rho_hist = (double)(occ_all > c->b_occ0 ? occ_all - c->b_occ0 : 0) / std::max(observed_kmers, 1.0);
            return KMC_OK;
        };
        const bool lr = c->cfg.mode == KMC_MODE_LR;
        if (lr) {
            // Reference mode (main.rs:63-81): every window start contributes up to 61 keys (27 + gap + 27 for
            // sizes 80..=140), and almost every key is new (1.08 M distinct of 3.55 M on the fixture): the keys
            // are FORMED and grouped by the radix sort + run-length, like the reference's own push + sort()
            // (main.rs:79,87) -- no hash table, no per-occurrence atomics -- as pairs of 27-mer ranks
            // (kmc_lr.hip.h): sort the batch's 27-mers once, then sort one-word rank pairs.
            // Sub-batches of 2^25 window starts (61 x 8 B x 2 buffers = 32 GiB of keys in flight at most).
            const u64 per = KMC_LRX_NS;
            const u64 SB = 1ull << 25;
            for (u64 p0 = 0; p0 < n_bases; p0 += SB) {
                const u64 p1 = std::min(n_bases, p0 + SB);
                const u64 n = (p1 - p0) * per;
                const u64 q0 = p0, q1 = std::min(n_bases, p1 + KMC_LRX_DMAX);
                const u64 nq = q1 - q0;
                for (int i = 0; i < 2; ++i) {
                    rc = ensure(c, c->s_lo[i], (size_t)std::max(n, nq) * sizeof(u64)); if (rc) return rc;
                }
                rc = ensure(c, c->lr_rank, (size_t)nq * sizeof(u32)); if (rc) return rc;
                rc = launch_begin(c);
                if (rc) return rc;
                u64* const khi[2] = {(u64*)c->s_hi[0].p, (u64*)c->s_hi[1].p};
                u64* const klo[2] = {(u64*)c->s_lo[0].p, (u64*)c->s_lo[1].p};
                u64* const kwt[2] = {nullptr, nullptr};
                // 1. the batch's distinct 27-mers, ordered: the dictionary
                const unsigned mer_grid = (unsigned)((nq + KMC_LRX_POS - 1) / KMC_LRX_POS);
                hipLaunchKernelGGL(kmc_lr_mer_kernel<0>, dim3(mer_grid), dim3(KMC_LRX_THREADS), 0, c->stream,
                                   d_bases, n_bases, q0, q1, klo[0], (const u64*)nullptr, 0u, (u32*)nullptr);
                HIPCHK(c, hipGetLastError());
                kmc_ctx::Run mers;
                rc = msd_sort_to_run(c, khi, klo, kwt, nq, 2u * KMC_LR_L, 1, &mers, true);
                if (rc) return rc;
                const u64 n_distinct = mers.n;
                if (n_distinct) {  // 2. every position's rank in it
                    hipLaunchKernelGGL(kmc_lr_mer_kernel<1>, dim3(mer_grid), dim3(KMC_LRX_THREADS), 0, c->stream,
                                       d_bases, n_bases, q0, q1, (u64*)nullptr, (const u64*)mers.lo, (u32)n_distinct, (u32*)c->lr_rank.p);
                    HIPCHK(c, hipGetLastError());
                }
                if (n_distinct) {
                    // 3. every key as a pair of ranks, sorted and run-length counted; 4. back to 108-bit keys
                    int B = 1;
                    while ((1ull << B) < n_distinct) ++B;
                    hipLaunchKernelGGL(kmc_lr_pair_kernel, dim3((unsigned)((p1 - p0 + KMC_LRX_POS - 1) / KMC_LRX_POS)), dim3(KMC_LRX_THREADS), 0, c->stream,
                                       d_offsets, n_reads, p0, p1, q0, nq, (const u32*)c->lr_rank.p, B, klo[0], c->d_counters);
                    HIPCHK(c, hipGetLastError());
                    kmc_ctx::Run run;
                    rc = msd_sort_to_run(c, khi, klo, kwt, n, 2u * (unsigned)B, 1, &run, true);
                    if (rc) { c->run_pool.push_back(mers); return rc; }
                    hipLaunchKernelGGL(kmc_lr_addcount_kernel, dim3(1), dim3(64), 0, c->stream, (const u32*)&((const MsdCtl*)c->m_ctl.p)->n_valid, c->d_counters);
                    if (run.n) {
                        hipLaunchKernelGGL(kmc_lr_compose_kernel, dim3(grid_for(c, run.n, 256)), dim3(256), 0, c->stream, run.hi, run.lo, run.n, (const u64*)mers.lo, B);
                        HIPCHK(c, hipGetLastError());
                        c->runs.push_back(run);
                    } else if (run.lo) c->run_pool.push_back(run);
                }
                if (mers.lo) c->run_pool.push_back(mers);  // (stream order: the compose kernel is queued before any later use)
                rc = launch_end(c);
                if (rc) return rc;
                c->pending = true;
            }
        }
        u64 stream_from = 0;  // base position from which the stream / sort path takes over
        bool run_stream = (algo == KMC_ALGO_STREAM) && !lr;
        bool run_sort = (algo == KMC_ALGO_SORT);
        const bool is_auto = c->cfg.algo == KMC_ALGO_AUTO;
        if (algo == KMC_ALGO_WALK) {
            if (!c->walk_memo.p || c->walk_overflowed) {
                // (re)start from an empty memo: first use, or the last batch overflowed it (its entries
                // were not representative; keeping them would only hold the tables full)
                rc = ensure(c, c->walk_memo, kmc_walk_memo_bytes(c->n_cu, c->KW));
                if (rc) return rc;
                HIPCHK(c, hipMemsetAsync(c->walk_memo.p, 0, kmc_walk_memo_bytes(c->n_cu, c->KW), c->stream));
                c->memo_parity = 0;
                c->walk_overflowed = false;
            }
            // the reads the kernel walks: the batch's own, or pieces of <= KMC_WALK_MAX_READ bases
            const u64 *d_vs = d_offsets, *d_ve = d_offsets + 1;
            u64 n_v = n_reads;
            if (max_read_len > KMC_WALK_MAX_READ) {
                rc = build_vreads(c, d_offsets, n_reads, &n_v);
                if (rc) return rc;
                d_vs = (const u64*)c->vr_reads.p;
                d_ve = d_vs + n_v;
            }
            rc = sk_ensure(c);
            if (rc) return rc;
            if (n_v >= (1ull << 32)) return fail(c, KMC_ERR_ARG, "batch too large for one walk pass: %llu read pieces; feed smaller batches", (unsigned long long)n_v);
            {
                void* before = c->walk_ws.p;
                rc = ensure(c, c->walk_ws, kmc_walk_workspace_bytes(n_v));
                if (rc) return rc;
                if (c->walk_ws.p != before) c->walk_ws_clean = false;  // fresh memory: the host clears header + counters once
            }
            const u64 n_tiles = (n_v + 63) / 64;
            const u64 kpt = 64 * std::min<u64>(std::max<u64>(max_read_len, 1), KMC_WALK_MAX_READ);  // k-mers per tile, upper bound
            u64 done = 0, prev = 0;
            while (done < n_tiles) {
                u64 occ = c->h_counters[KMC_CTR_OCCUPIED];
                u64 take = plan(n_tiles - done, kpt, prev);
                if (c->sk_dirty && take > plan_safe) {
                    // A launch that rests on a prediction is about to save the table: the (k+16)-mer counts of
                    // the launches before it must be IN that table first (a recovery drops the (k+16)-mer
                    // table's counts together with the failed launch's).  This is a poll point: the unfold is
                    // sized exactly, and one more poll tells the planner what the table looks like now.
                    rc = flush_sk(c);
                    if (rc) return rc;
                    rc = poll_and_settle(c);
                    if (rc) return rc;
                    if (c->recovered) break;
                    take = plan(n_tiles - done, kpt, prev);
                }
                if (take > plan_safe && !arm_risky(c, d_bases, d_offsets, n_reads, n_bases, 0, done ? d_ve + (done * 64 - 1) : nullptr))
                    take = plan_safe;  // (the table cannot be saved cheaply: stay within what fits for certain)
                // the (k+16)-mer table takes one entry per step at most: a launch that could fill it is risky too;
                // if the count table cannot be saved, this launch runs without the second-level memo
                GTable skt = sk_table_of(c);
                if (skt.key_lo && !c->risky.armed) {
                    const u64 max_adds = take * 64ull * (KMC_WALK_MAX_READ / KMC_WALK_STRIDE + 1);
                    if ((c->h_sk_counters[KMC_CTR_OCCUPIED] + max_adds) * 4 > c->sk.cap * 3 && c->sk_dirty) {
                        rc = flush_sk(c);   // (as above: nothing of earlier launches may be lost with this one)
                        if (rc) return rc;
                        rc = poll_and_settle(c);
                        if (rc) return rc;
                        if (c->recovered) break;
                        skt = sk_table_of(c);  // (the flush may have re-allocated it larger)
                    }
                    if ((c->h_sk_counters[KMC_CTR_OCCUPIED] + max_adds) * 4 > c->sk.cap * 3 &&
                        !arm_risky(c, d_bases, d_offsets, n_reads, n_bases, 0, done ? d_ve + (done * 64 - 1) : nullptr))
                        skt = GTable{};
                }
                // the log of the steps that fall off the LDS memo (kmc_sklog.hip.h), sized for the worst case -- every step of the
                // launch -- up to 12 GiB; a workgroup whose span is full goes on with (k+16)-mer table updates
                SkLog lg{};
                const int wgrid = kmc_walk_grid(take, c->n_cu);
                if (c->sklog_on && skt.key_lo && !getenv("KMC_NO_SKLOG")) {
                    const u32 words = c->cfg.k > KMC_SK_MAX_K ? 3u : 2u;
                    const u64 steps_per_read = std::min<u64>(std::max<u64>(max_read_len, 1), KMC_WALK_MAX_READ) / KMC_WALK_STRIDE + 1;
                    u64 cap = ((take + wgrid - 1) / wgrid + 1) * 64ull * steps_per_read;
                    cap = std::min<u64>(cap, (12ull << 30) / ((u64)wgrid * words * sizeof(u64)));
                    cap = std::max<u64>(cap, 1024);
                    // (the binned copy holds exactly the logged records: the same room as the log; fewer than 2^32 records)
                    if ((u64)wgrid * cap < (1ull << 32) &&
                        !ensure(c, c->lg_rec, (size_t)wgrid * cap * words * sizeof(u64)) && !ensure(c, c->lg_bins, (size_t)wgrid * cap * words * sizeof(u64)) &&
                        !ensure(c, c->lg_count, ((size_t)c->n_cu + (size_t)c->n_cu * KMC_SKLOG_BINS) * sizeof(u32)) && !ensure(c, c->lg_cursor, 2 * KMC_SKLOG_BINS * sizeof(u32))) {
                        HIPCHK(c, hipMemsetAsync(c->lg_count.p, 0, (size_t)c->n_cu * sizeof(u32), c->stream));
                        HIPCHK(c, hipMemsetAsync(c->lg_cursor.p, 0, 2 * KMC_SKLOG_BINS * sizeof(u32), c->stream));
                        lg = SkLog{(u64*)c->lg_rec.p, (u32*)c->lg_count.p, (u32)cap, words};
                    } else {
                        c->err[0] = 0;   // (no memory for a log: the launch runs with table updates, as before)
                    }
                }
                if (!c->walk_ws_clean) {  // (normally the unfold kernel of the previous launch left it clean)
                    rc = kmc_walk_prepare(c->stream, c->walk_ws.p);
                    if (rc) return fail(c, rc, "walk workspace reset failed");
                }
                c->walk_ws_clean = false;
                hipEvent_t we0 = nullptr, we1 = nullptr;   // the walk kernel's own start / stop timestamps (no event packets in the stream)
                static const bool ev_records = getenv("KMC_WALK_EVENT_RECORDS") != nullptr;   // (A/B switch: hipEventRecord around the launch instead)
                if (ev_records) rc = launch_begin(c); else rc = launch_events(c, &we0, &we1);
                if (rc) return rc;
                rc = kmc_walk_launch(c->stream, c->n_cu, c->KW, c->cfg.k, c->cfg.canonical != 0, d_bases, d_vs, d_ve, n_v, n_bases,
                                     done, done + take, c->walk_ws.p, c->walk_memo.p, c->memo_parity, gtable_of(c, c->tab), skt, lg, 0, we0, we1);
                if (rc) return fail(c, rc, "walk kernel launch failed: %s", hipGetErrorString(hipGetLastError()));
                if (ev_records) { rc = launch_end(c); if (rc) return rc; }
                c->batch_pending = true;
                if (skt.key_lo) c->sk_dirty = true;
                rc = kmc_walk_launch(c->stream, c->n_cu, c->KW, c->cfg.k, c->cfg.canonical != 0, d_bases, d_vs, d_ve, n_v, n_bases,
                                     done, done + take, c->walk_ws.p, c->walk_memo.p, c->memo_parity, gtable_of(c, c->tab), skt, lg, 1);
                if (rc) return fail(c, rc, "scalar/unfold kernel launch failed: %s", hipGetErrorString(hipGetLastError()));
                if (lg.rec) {   // count what the launch logged: partition by hash, LDS tables, one unfold per distinct (k+16)-mer
                    rc = launch_begin(c);
                    if (rc) return rc;
                    rc = launch_sklog(c, lg, (u32)wgrid);
                    if (rc) return rc;
                    rc = launch_end(c);
                    if (rc) return rc;
                }
                c->walk_ws_clean = true;
                if (!c->tb.empty()) c->tb.back().count_launches++;
                c->memo_parity ^= 1;
                c->pending = true;
                done += take;
                prev = take;
                if (done < n_tiles) {
                    rc = observe(occ, take, kpt);
                    if (rc) return rc;
                    if (c->recovered) break;  // (the sort path has counted the rest of the batch)
                    if (c->cfg.algo == KMC_ALGO_AUTO && (c->walk_overflowed || (c->rho_last > 0.5 && (take >= 512 || done * 2 >= n_tiles)))) {
                        // (the new-key rate only counts once a launch was large -- 512 tiles, 13 M k-mers: the third launch of the
                        // ramp used to put another 200 M k-mers into the table before the hand-over -- or half the batch is through: the
                        // first tiles of ANY input are all new)
                        // the memos do not help on this input (both levels overflow, or more than one k-mer in
                        // five is new: per-occurrence table updates are the wrong tool): hand the rest
                        // of the batch to the sort path
                        // ... from the end of the last piece walked: the windows ENDING before it are counted
                        u64 pos = 0;
                        if (batch_on_empty_table) {
                            rc = drop_batch_from_table(c, ctr0);   // (the whole batch goes to the sort path)
                            if (rc) return rc;
                        } else {
                            HIPCHK(c, hipMemcpyAsync(&pos, d_ve + (done * 64 - 1), sizeof(u64), hipMemcpyDeviceToHost, c->stream));
                            HIPCHK(c, hipStreamSynchronize(c->stream));
                        }
                        stream_from = pos;
                        run_sort = true;
                        batch_mixed = true;
                        c->prefer_sort = true;
                        c->st.algo_last = KMC_ALGO_SORT;
                        break;
                    }
                }
            }
        }
        if (c->recovered) { run_stream = false; run_sort = false; c->sk_dirty = false; }
        if (run_stream) {
            const u64 n_chunks = (n_bases + KMC_CHUNK - 1) / KMC_CHUNK;
            u64 done = stream_from / KMC_CHUNK, prev = 0;
            while (done < n_chunks) {
                u64 occ = c->h_counters[KMC_CTR_OCCUPIED];
                u64 take = plan(n_chunks - done, KMC_CHUNK, prev);
                if (take > plan_safe && !arm_risky(c, d_bases, d_offsets, n_reads, n_bases, std::max<u64>(done * KMC_CHUNK, stream_from), nullptr))
                    take = plan_safe;
                rc = launch_stream(c, d_bases, d_offsets, n_reads, n_bases, done, done + take, stream_from);
                if (rc) return rc;
                c->pending = true;
                done += take;
                prev = take;
                if (done < n_chunks) {
                    rc = observe(occ, take, KMC_CHUNK);
                    if (rc) return rc;
                    if (c->recovered) break;
                    if (is_auto && (c->rho_last > 0.2 || c->walk_overflowed)) {
                        // many new keys per k-mer, or the LDS partial tables overflow and most k-mers go to
                        // global atomics anyway: per-occurrence hashing is the wrong tool
                        stream_from = done * KMC_CHUNK;
                        run_sort = true;
                        batch_mixed = true;
                        c->prefer_sort = true;
                        c->st.algo_last = KMC_ALGO_SORT;
                        break;
                    }
                }
            }
        }
        if (c->recovered) run_sort = false;
        if (run_sort) {
            rc = run_sort_path(c, d_bases, d_offsets, n_reads, n_bases, stream_from);
            if (rc) return rc;
        }
        c->b_rho_max = batch_rho_max;  // the last launch's share is folded in by the next poll()
        c->b_open = true;
    }
    if (!c->tb.empty()) c->tb.back().algo = ((batch_mixed || c->recovered) ? 0 : c->st.algo_last);
    return KMC_OK;
}

// kernel time of the last batch = sum over its count-kernel launches (host polls between sub-batches
// are not kernel time).  Only once the batch has finished on the GPU.
void harvest_timing(kmc_ctx* c) {
    while (!c->tb.empty()) {
        std::vector<hipEvent_t>& b = c->tb.front().ev;
        const bool current = c->tb.size() == 1;   // (the newest batch may still be queueing launches: leave it until it is complete AND idle)
        if (b.size() >= 2 && (b.size() & 1) == 0) {
            hipError_t q = hipEventQuery(b.back());
            if (q != hipSuccess && c->tb.size() > 64) q = hipEventSynchronize(b.back());   // (bounded backlog)
            if (q != hipSuccess) { (void)hipGetLastError(); return; }
            float ms = 0.f;
            double sum = 0.0;
            for (size_t i = 0; i + 1 < b.size(); i += 2) {
                if (hipEventElapsedTime(&ms, b[i], b[i + 1]) == hipSuccess) sum += ms; else (void)hipGetLastError();
            }
            c->st.kernel_ms_last = sum;
            c->st.kernel_ms_total += sum;
            c->st.kernel_ms_lifetime += sum;
            c->st.launches_last = (int32_t)(b.size() / 2);  // launches in that batch
            c->st.launches_lifetime += b.size() / 2;
            // what this data source costs on the path it took (KMC_ALGO_AUTO's choice between the walk and the sort path)
            const kmc_ctx::TimedBatch& tbk = c->tb.front();
            // (a rate sample = a large batch that went out in ONE count launch: small batches are launch overhead, and the
            //  first batch on a new source -- no history: a ramp of launches with polls in between, a memo still learning -- is not
            //  what later batches cost)
            if (tbk.n_bases >= (1u << 26) && tbk.count_launches == 1) {
                const double per = sum / (double)tbk.n_bases;
                double* dst = tbk.algo == KMC_ALGO_WALK ? &c->walk_ms_per_base : (tbk.algo == KMC_ALGO_SORT ? &c->sort_ms_per_base : nullptr);
                if (dst) *dst = *dst < 0 ? per : 0.5 * (*dst + per);
            }
        } else if (!b.empty() && current) {
            return;   // (an odd number: a launch is being bracketed right now)
        }
        for (hipEvent_t e : b) c->ev_free.push_back(e);
        c->tb.pop_front();
        (void)current;
    }
}

}  // namespace

// ---- ABI -------------------------------------------------------------------------------------

extern "C" const char* kmc_version(void) { return "libkmc 0.1 (gfx950)"; }

extern "C" const char* kmc_status_string(int s) {
    switch (s) {
        case KMC_OK: return "ok";
        case KMC_ERR_ARG: return "bad argument";
        case KMC_ERR_NO_DEVICE: return "no usable HIP device (libkmc has no CPU fallback)";
        case KMC_ERR_HIP: return "HIP runtime error";
        case KMC_ERR_NOMEM: return "out of memory";
        case KMC_ERR_IO: return "Error during opening the file";
        case KMC_ERR_FORMAT: return "Expected > at record start.";
        case KMC_ERR_ALPHABET: return "Unexpected charactor in sequence";
        case KMC_ERR_CAPACITY: return "count table capacity exhausted";
        case KMC_ERR_STATE: return "call out of order";
        default: return "unknown status";
    }
}

extern "C" const char* kmc_last_error(const kmc_ctx* ctx) { return ctx ? ctx->err : g_create_err; }

extern "C" void kmc_destroy(kmc_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->cfg.device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    free_table(c->tab);
    if (c->d_counters) (void)hipFree(c->d_counters);
    if (c->h_counters) (void)hipHostFree(c->h_counters);
    if (c->h_restore) (void)hipHostFree(c->h_restore);
    if (c->occ_list) (void)hipFree(c->occ_list);
    if (c->occ_key_lo) (void)hipFree(c->occ_key_lo);
    if (c->occ_key_hi) (void)hipFree(c->occ_key_hi);
    if (c->fin_rank) (void)hipFree(c->fin_rank);
    if (c->spill_hi) (void)hipFree(c->spill_hi);
    if (c->spill_lo) (void)hipFree(c->spill_lo);
    if (c->spill_cnt) (void)hipFree(c->spill_cnt);
    DevBuf* bufs[] = {&c->st_bases, &c->st_offsets, &c->o_hi, &c->o_lo, &c->o_cnt, &c->t_hi, &c->t_lo, &c->t_cnt,
                      &c->t_idx0, &c->p_hi, &c->p_lo, &c->p_cnt, &c->walk_ws, &c->walk_memo, &c->vr_reads, &c->vr_cnt, &c->vr_pos,
                      &c->s_lo[0], &c->s_lo[1], &c->s_hi[0], &c->s_hi[1], &c->lr_rank, &c->a_hist, &c->a_rand, &c->a_ror,
                      &c->lg_rec, &c->lg_count, &c->lg_bins, &c->lg_cursor,
                      &c->m_hist, &c->m_stot, &c->m_bsum, &c->m_rmin, &c->m_rmax, &c->m_seg[0], &c->m_seg[1], &c->m_first, &c->m_cbase, &c->m_skip, &c->m_term, &c->m_ord,
                      &c->m_bitmap, &c->m_rank, &c->m_nd, &c->m_base, &c->m_ctl, &c->m_clist, &c->m_w[0], &c->m_w[1],
                      &c->snap_hi, &c->snap_lo, &c->snap_cnt, &c->snap_n, &c->snap_occ, &c->rx_hi, &c->rx_lo, &c->rx_cnt};
    if (c->h_ctl) (void)hipHostFree(c->h_ctl);
    sk_free(c);
    try { free_runs(c, true); } catch (...) { /* (only the pool bookkeeping can throw; the buffers it could not list leak with the process) */ }
    for (DevBuf* b : bufs) free_buf(*b);
    for (auto& b : c->tb) for (hipEvent_t e : b.ev) (void)hipEventDestroy(e);
    for (hipEvent_t e : c->ev_free) (void)hipEventDestroy(e);
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

static int kmc_create_impl(kmc_ctx** out, const kmc_config* cfg) {
    if (!out || !cfg) return fail(nullptr, KMC_ERR_ARG, "null argument");
    *out = nullptr;
    if (cfg->struct_size != sizeof(kmc_config)) return fail(nullptr, KMC_ERR_ARG, "kmc_config.struct_size mismatch (%u != %zu)", cfg->struct_size, sizeof(kmc_config));
    if (cfg->mode != KMC_MODE_CONTIG && cfg->mode != KMC_MODE_LR) return fail(nullptr, KMC_ERR_ARG, "bad mode %d", cfg->mode);
    if (cfg->mode == KMC_MODE_CONTIG && (cfg->k < 1 || cfg->k > 63)) return fail(nullptr, KMC_ERR_ARG, "k must be in 1..63 (got %d)", cfg->k);
    if (cfg->algo < KMC_ALGO_AUTO || cfg->algo > KMC_ALGO_SORT) return fail(nullptr, KMC_ERR_ARG, "bad algo %d", cfg->algo);
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) return fail(nullptr, KMC_ERR_NO_DEVICE, "no usable HIP device (%s); libkmc has no CPU fallback", e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    if (cfg->device < 0 || cfg->device >= ndev) return fail(nullptr, KMC_ERR_ARG, "device %d out of range (0..%d)", cfg->device, ndev - 1);
    kmc_ctx* c = new (std::nothrow) kmc_ctx();
    if (!c) return fail(nullptr, KMC_ERR_NOMEM, "out of memory");
    c->cfg = *cfg;
    if (cfg->mode == KMC_MODE_LR) { c->KW = 2; c->klen = 54; c->cfg.k = 54; c->cfg.canonical = 0; }
    else { c->KW = cfg->k <= 31 ? 1 : 2; c->klen = cfg->k; }
    if (const char* e = getenv("KMC_FIN_SMALL_MAX")) c->fin_small_max = std::min<u64>(strtoull(e, nullptr, 10), KMC_FIN_KERNEL_MAX);
    int rc = KMC_OK;
    auto body = [&]() -> int {
        HIPCHK(c, hipSetDevice(cfg->device));
        hipDeviceProp_t prop;
        HIPCHK(c, hipGetDeviceProperties(&prop, cfg->device));
        c->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        if (cfg->stream) c->stream = (hipStream_t)cfg->stream;
        else { HIPCHK(c, hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)); c->own_stream = true; }
        HIPCHK(c, hipMalloc((void**)&c->d_counters, 2 * KMC_CTR_N * sizeof(u64)));  // [count table | (k+16)-mer table]: one read-back
        HIPCHK(c, hipMemsetAsync(c->d_counters, 0, 2 * KMC_CTR_N * sizeof(u64), c->stream));
        HIPCHK(c, hipHostMalloc((void**)&c->h_counters, 2 * KMC_CTR_N * sizeof(u64)));
        memset(c->h_counters, 0, 2 * KMC_CTR_N * sizeof(u64));
        HIPCHK(c, hipHostGetDevicePointer((void**)&c->d_mirror, c->h_counters, 0));
        HIPCHK(c, hipHostMalloc((void**)&c->h_restore, KMC_CTR_N * sizeof(u64)));
        u64 cap = next_pow2(std::max<u64>(cfg->capacity_hint * 2, 1ull << 20));
        c->spill_cap = std::max<u64>(cap / 4, 1ull << 18);
        HIPCHK(c, hipMalloc((void**)&c->occ_list, KMC_OCC_LIST_CAP * sizeof(u64)));
        HIPCHK(c, hipMalloc((void**)&c->occ_key_lo, KMC_OCC_LIST_CAP * sizeof(u64)));
        HIPCHK(c, hipMalloc((void**)&c->occ_key_hi, KMC_OCC_LIST_CAP * sizeof(u64)));
        HIPCHK(c, hipMalloc((void**)&c->fin_rank, 16 * sizeof(u32)));
        HIPCHK(c, hipMemsetAsync(c->fin_rank, 0, 16 * sizeof(u32), c->stream));
        HIPCHK(c, hipMalloc((void**)&c->spill_lo, c->spill_cap * sizeof(u64)));
        HIPCHK(c, hipMalloc((void**)&c->spill_cnt, c->spill_cap * sizeof(u64)));
        if (c->KW == 2) HIPCHK(c, hipMalloc((void**)&c->spill_hi, c->spill_cap * sizeof(u64)));
        int r = alloc_table(c, c->tab, cap);
        if (r) return r;
        c->st.table_capacity = cap;
        HIPCHK(c, hipStreamSynchronize(c->stream));
        return KMC_OK;
    };
    rc = body();
    if (rc) {
        memcpy(g_create_err, c->err, sizeof(g_create_err));
        kmc_destroy(c);
        return rc;
    }
    *out = c;
    return KMC_OK;
}

static int kmc_reset_impl(kmc_ctx* c) {
    if (!c) return KMC_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    if (c->async_fin) {
        // a finalize is queued whose outcome nobody has looked at (kmc_finalize_async): it has drained the table or left it
        // as it was -- the reset kernel decides on the device, nothing waits
        c->async_fin = false;
        c->drained = false;
        c->sk_dirty = false;   // (kmc_finalize_async queued the unfold in front of its kernel, or the reset kernel drops what is pending)
        GTable g = gtable_of(c, c->tab);
        const int grid = reset_grid(c);
        if (c->KW == 1) hipLaunchKernelGGL(kmc_reset_kernel<1>, dim3(grid), dim3(256), 0, c->stream, g, sk_table_of(c), c->fin_rank + 3);
        else hipLaunchKernelGGL(kmc_reset_kernel<2>, dim3(grid), dim3(256), 0, c->stream, g, sk_table_of(c), c->fin_rank + 3);
        HIPCHK(c, hipGetLastError());
        c->fin_parity = 0;
    } else if (c->drained) {
        // the last kmc_finalize emptied the table into its sorted view (kmc_small_finalize_kernel): table and device
        // counters are already what the reset kernel would leave -- nothing to launch
        c->drained = false;
        c->fin_parity = 0;
    } else {
        // (counts pending in the (k+16)-mer table go with the table: the reset kernel clears them, nothing is unfolded first)
        c->sk_dirty = false;
        if (c->sk_grow) { int r = sk_regrow(c); if (r) return r; }
        GTable g = gtable_of(c, c->tab);
        const int grid = reset_grid(c);
        if (c->KW == 1) hipLaunchKernelGGL(kmc_reset_kernel<1>, dim3(grid), dim3(256), 0, c->stream, g, sk_table_of(c), c->fin_rank + 3);
        else hipLaunchKernelGGL(kmc_reset_kernel<2>, dim3(grid), dim3(256), 0, c->stream, g, sk_table_of(c), c->fin_rank + 3);
        HIPCHK(c, hipGetLastError());
        c->fin_parity = 0;
    }
    memset(c->h_counters, 0, KMC_CTR_N * sizeof(u64));
    c->pending = false;
    c->polled_epoch = c->table_epoch;   // (an empty table is a known state)
    c->sorted_valid = false;
    c->n_sorted = 0;
    free_runs(c, false);
    c->direct_seen = c->kmers_seen = 0;
    c->batch_pending = false;
    c->unpolled_adds = 0;
    c->b_open = false;  // (rho_hist itself is kept: it describes the data source)
    c->risky.armed = false;
    c->recovered = false;
    const kmc_stats keep = c->st;
    c->st = kmc_stats{};
    c->st.table_capacity = keep.table_capacity;
    c->st.kernel_ms_lifetime = keep.kernel_ms_lifetime;
    c->st.launches_lifetime = keep.launches_lifetime;
    c->st.n_planner_stale = keep.n_planner_stale;
    c->st.n_async_ok = keep.n_async_ok;
    c->st.n_async_slabs_skipped = keep.n_async_slabs_skipped;
    return KMC_OK;
}

static int kmc_add_batch_device_impl(kmc_ctx* c, const void* d_bases, const void* d_offsets, uint64_t n_reads,
                                    uint64_t n_bases, uint64_t max_read_len) {
    if (!c) return KMC_ERR_ARG;
    if (n_reads && (!d_bases || !d_offsets)) return fail(c, KMC_ERR_ARG, "null device pointer");
    if (((uintptr_t)d_bases & 15) != 0) return fail(c, KMC_ERR_ARG, "d_bases must be 16-byte aligned");
    if (((uintptr_t)d_offsets & 7) != 0) return fail(c, KMC_ERR_ARG, "d_offsets must be 8-byte aligned");
    HIPCHK(c, hipSetDevice(c->cfg.device));
    return count_batch_device(c, (const uint8_t*)d_bases, (const u64*)d_offsets, n_reads, n_bases, max_read_len);
}

static int kmc_add_batch_impl(kmc_ctx* c, const uint8_t* bases, const uint64_t* offsets, uint64_t n_reads) {
    if (!c) return KMC_ERR_ARG;
    if (!n_reads) { c->st.n_batches += 1; return KMC_OK; }
    if (!bases || !offsets) return fail(c, KMC_ERR_ARG, "null buffer");
    if (offsets[0] != 0) return fail(c, KMC_ERR_ARG, "offsets[0] must be 0");
    u64 maxlen = 0;
    for (u64 i = 0; i < n_reads; ++i) {
        if (offsets[i + 1] < offsets[i]) return fail(c, KMC_ERR_ARG, "offsets must be non-decreasing (read %llu)", (unsigned long long)i);
        maxlen = std::max<u64>(maxlen, offsets[i + 1] - offsets[i]);
    }
    const u64 n_bases = offsets[n_reads];
    HIPCHK(c, hipSetDevice(c->cfg.device));
    if (c->pending) { int rc = poll_and_settle(c); if (rc) return rc; }  // previous batch may still read the staging area
    int rc = ensure(c, c->st_bases, n_bases + 64);
    if (rc) return rc;
    rc = ensure(c, c->st_offsets, (n_reads + 1) * sizeof(u64));
    if (rc) return rc;
    if (n_bases) HIPCHK(c, hipMemcpyAsync(c->st_bases.p, bases, n_bases, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->st_offsets.p, offsets, (n_reads + 1) * sizeof(u64), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));  // caller may reuse its buffers on return
    return count_batch_device(c, (const uint8_t*)c->st_bases.p, (const u64*)c->st_offsets.p, n_reads, n_bases, maxlen);
}

static int kmc_merge_pairs_device_impl(kmc_ctx* c, const void* d_key_hi, const void* d_key_lo, const void* d_count, uint64_t n) {
    if (!c) return KMC_ERR_ARG;
    if (!n) return KMC_OK;
    if (!d_key_lo || !d_count) return fail(c, KMC_ERR_ARG, "null device pointer");
    HIPCHK(c, hipSetDevice(c->cfg.device));
    { int rc = resolve_async(c); if (rc) return rc; }
    { int rc = undrain(c); if (rc) return rc; }
    // make room for the worst case (every pair new).  Merges queued since the last poll are
    // accounted with their upper bound, so a series of merges (one per peer in the multi-GPU reduce)
    // needs no host synchronisation in between.
    if (c->batch_pending || (c->h_counters[KMC_CTR_OCCUPIED] + c->unpolled_adds + n) * 2 > c->tab.cap) {
        if (c->pending) { int rc = poll_and_settle(c); if (rc) return rc; }
        u64 occ = c->h_counters[KMC_CTR_OCCUPIED];
        if ((occ + n) * 2 > c->tab.cap) {
            int rc = grow_to(c, next_pow2((occ + n) * 2));
            if (rc) return rc;
        }
    }
    c->unpolled_adds += n;
    c->sorted_valid = false;
    GTable g = gtable_of(c, c->tab);
    int grid = grid_for(c, n, 256);
    if (c->KW == 1) hipLaunchKernelGGL(kmc_merge_pairs_kernel<1>, dim3(grid), dim3(256), 0, c->stream, g, (const u64*)nullptr, (const u64*)d_key_lo, (const u64*)d_count, n);
    else hipLaunchKernelGGL(kmc_merge_pairs_kernel<2>, dim3(grid), dim3(256), 0, c->stream, g, (const u64*)d_key_hi, (const u64*)d_key_lo, (const u64*)d_count, n);
    HIPCHK(c, hipGetLastError());
    c->pending = true;
    c->table_epoch++;
    return KMC_OK;
}

static int kmc_finalize_impl(kmc_ctx* c, uint64_t* n_distinct, uint64_t* n_total) {
    if (!c) return KMC_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    int rc;
    bool tried_fast = false;
    int fgrid_used = 0;
    rc = resolve_async(c);   // (a finalize queued by kmc_finalize_async: its view, when it produced one, is the result)
    if (rc) return rc;
    if (c->drained && c->sorted_valid) {  // nothing was added since the last finalize (which emptied the table into the view)
        if (n_distinct) *n_distinct = c->n_sorted;
        if (n_total) *n_total = c->st.n_kmers;
        return KMC_OK;
    }
    rc = flush_acc(c);  // keys the sort path has extracted since the last flush: one run
    if (rc) return rc;
    if (c->runs.empty() && c->fin_hint <= c->fin_small_max) {
        // speculative small-table finalize, queued behind whatever is still running
        const size_t fb = (size_t)KMC_FIN_KERNEL_MAX * sizeof(u64);
        rc = ensure(c, c->o_lo, fb); if (rc) return rc;
        rc = ensure(c, c->o_cnt, fb); if (rc) return rc;
        if (c->KW == 2) { rc = ensure(c, c->o_hi, fb); if (rc) return rc; }
        fgrid_used = small_finalize_grid(c);
        rc = launch_small_finalize(c, fgrid_used);
        if (rc) return rc;
        tried_fast = true;
        if (c->tb.size() >= 4) harvest_timing(c);   // (older batches' events, while the GPU is busy with this one)
    }
    rc = tried_fast ? poll_fin_and_settle(c) : poll_and_settle(c);
    if (!rc && c->acc_n) rc = flush_acc(c);  // (the poll may have recovered an overflow by extracting the rest of a batch)
    if (rc) return rc;
    if (c->sk_dirty) {
        // the last batch's walk launches may have left counts in the (k+16)-mer table: the poll tells
        const bool had = c->h_sk_counters && c->h_sk_counters[KMC_CTR_KMERS] != 0;
        rc = settle_sk_polled(c);
        if (rc) return rc;
        if (had) {  // the table changed under the speculative finalize: once more
            // (the unfold just queued fills the table: the poll behind it may find it grown or spilled -- settled there;
            // the second finalize kernel sees the (k+16)-mer table empty again and may drain)
            bool again = false;
            if (tried_fast && c->runs.empty()) {
                fgrid_used = (int)(std::min<u64>(c->fin_small_max + c->fin_small_max / 4, KMC_FIN_KERNEL_MAX) / KMC_FIN_CHUNK + 8);
                rc = launch_small_finalize(c, fgrid_used);
                if (rc) return rc;
                again = true;
            }
            rc = again ? poll_fin_and_settle(c) : poll_and_settle(c);
    if (!rc && c->acc_n) rc = flush_acc(c);  // (the poll may have recovered an overflow by extracting the rest of a batch)
            if (rc) return rc;
        }
    }
    c->fin_hint = c->h_counters[KMC_CTR_OCCUPIED];
    if (tried_fast && c->runs.empty() && c->h_counters[KMC_CTR_FASTFIN] != 1 && c->h_counters[KMC_CTR_SPILL] == 0 &&
        c->h_counters[KMC_CTR_OCCUPIED] > (u64)fgrid_used * KMC_FIN_CHUNK && c->h_counters[KMC_CTR_OCCUPIED] <= c->fin_small_max) {
        // the table outgrew the speculative grid (first finalize of a larger source): once more, full grid
        fgrid_used = (int)(std::min<u64>(c->fin_small_max + c->fin_small_max / 4, KMC_FIN_KERNEL_MAX) / KMC_FIN_CHUNK + 8);
        rc = launch_small_finalize(c, fgrid_used);
        if (rc) return rc;
        rc = poll_fin_and_settle(c);
    if (!rc && c->acc_n) rc = flush_acc(c);  // (the poll may have recovered an overflow by extracting the rest of a batch)
        if (rc) return rc;
    }
    const bool fast_done = tried_fast && c->h_counters[KMC_CTR_FASTFIN] == 1 && c->runs.empty();  // (the poll may have recovered an overflow: runs exist now)
    const u64 n_tab = c->h_counters[KMC_CTR_OCCUPIED];
    u64 n_runs_total = 0;
    for (auto& r : c->runs) n_runs_total += r.n;
    u64 n = n_tab + n_runs_total;  // entries before merging duplicates across sources
    u64 n_kmers = 0;
    const bool single_run = n_tab == 0 && c->runs.size() == 1;
    if (c->view_run.lo) { c->view_run.n = 0; c->run_pool.push_back(c->view_run); c->view_run = kmc_ctx::Run{}; }  // the previous finalize's view
    c->v_hi = c->KW == 2 ? (const u64*)c->o_hi.p : nullptr;
    c->v_lo = (const u64*)c->o_lo.p;
    c->v_cnt = (const u64*)c->o_cnt.p;
    if (fast_done) {
        // small table: the rank-sort kernel already gathered, sorted and wrote the view
        n_kmers = c->h_counters[KMC_CTR_SUM2];
    } else if (single_run) {
        // one sorted run and an empty table: it IS the sorted view (no copy); the sort knows its total
        auto& r = c->runs[0];
        c->v_hi = r.hi; c->v_lo = r.lo; c->v_cnt = r.cnt;
        n_kmers = r.total;
    } else if (n) {
        // table entries and / or several runs: ONE weighted sort merges and orders them -- the entries are
        // concatenated as (key, count) pairs, the hand-written radix sort (kmc_msd.hip.h) orders them and
        // its run-length step sums the counts of equal keys (the grouping of main.rs:84,87 once more)
        if (n >= (1ull << 32) - KMC_MSD_RANGE) return fail(c, KMC_ERR_ARG, "kmc_finalize: more than 2^32 table entries + run entries to merge in one pass");
        if (!c->runs.empty()) {
            // a merge of big runs needs room: give back the pooled buffers
            for (auto& r : c->run_pool) { if (r.hi) (void)hipFree(r.hi); (void)hipFree(r.lo); (void)hipFree(r.cnt); }
            c->run_pool.clear();
        }
        const size_t nb = (size_t)n * sizeof(u64);
        DevBuf* need[] = {&c->o_lo, &c->o_cnt, &c->t_lo, &c->t_cnt};
        for (DevBuf* b : need) { rc = ensure(c, *b, nb); if (rc) return rc; }
        if (c->KW == 2) { rc = ensure(c, c->o_hi, nb); if (rc) return rc; rc = ensure(c, c->t_hi, nb); if (rc) return rc; }
        if (n_tab && n_tab <= KMC_OCC_LIST_CAP && c->occ_list && c->h_counters[KMC_CTR_SPILL] == 0 && c->h_counters[KMC_CTR_ERR] == 0) {
            // every claimed slot is listed (and its key kept in the dense key list): no scan of the table
            GTable g = gtable_of(c, c->tab);
            const int grid = grid_for(c, n_tab, 256);
            if (c->KW == 1) hipLaunchKernelGGL(kmc_compact_list_kernel<1>, dim3(grid), dim3(256), 0, c->stream, g, n_tab, (u64*)nullptr, (u64*)c->t_lo.p, (u64*)c->t_cnt.p);
            else hipLaunchKernelGGL(kmc_compact_list_kernel<2>, dim3(grid), dim3(256), 0, c->stream, g, n_tab, (u64*)c->t_hi.p, (u64*)c->t_lo.p, (u64*)c->t_cnt.p);
            HIPCHK(c, hipGetLastError());
        } else if (n_tab) {
            const int parity = c->fin_parity;
            c->fin_parity ^= 1;
            GTable g = gtable_of(c, c->tab);
            int grid = grid_for(c, c->tab.cap, 256);
            if (c->KW == 1) hipLaunchKernelGGL(kmc_compact_kernel<1>, dim3(grid), dim3(256), 0, c->stream, g, (u64*)nullptr, (u64*)c->t_lo.p, (u64*)c->t_cnt.p, (u64*)nullptr, parity);
            else hipLaunchKernelGGL(kmc_compact_kernel<2>, dim3(grid), dim3(256), 0, c->stream, g, (u64*)c->t_hi.p, (u64*)c->t_lo.p, (u64*)c->t_cnt.p, (u64*)nullptr, parity);
            HIPCHK(c, hipGetLastError());
        }
        u64 off = n_tab;
        for (auto& r : c->runs) {
            HIPCHK(c, hipMemcpyAsync((u64*)c->t_lo.p + off, r.lo, r.n * sizeof(u64), hipMemcpyDeviceToDevice, c->stream));
            HIPCHK(c, hipMemcpyAsync((u64*)c->t_cnt.p + off, r.cnt, r.n * sizeof(u64), hipMemcpyDeviceToDevice, c->stream));
            if (c->KW == 2) HIPCHK(c, hipMemcpyAsync((u64*)c->t_hi.p + off, r.hi, r.n * sizeof(u64), hipMemcpyDeviceToDevice, c->stream));
            off += r.n;
        }
        u64* const khi[2] = {(u64*)c->t_hi.p, (u64*)c->o_hi.p};
        u64* const klo[2] = {(u64*)c->t_lo.p, (u64*)c->o_lo.p};
        u64* const kwt[2] = {(u64*)c->t_cnt.p, (u64*)c->o_cnt.p};
        const size_t before = c->runs.size();
        rc = launch_begin(c);   // (the merge is part of what the path that left a large table costs)
        if (rc) return rc;
        rc = msd_sort_to_run(c, khi, klo, kwt, n, 2u * (unsigned)c->klen, c->KW);
        if (rc) return rc;
        rc = launch_end(c);
        if (rc) return rc;
        // The merged view is read by other streams right after this call (kmc_export_device consumers, the
        // peer copies of kmc_count_file_multi on the destination ctx's stream): unlike the fast and single-run
        // branches, whose view was complete before the poll above synchronised, this one's last kernels are
        // still queued -- wait for them (it is already a multi-synchronisation path).
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (c->runs.size() > before) {
            c->view_run = c->runs.back();   // the merged view is not one of the ctx's runs (they stay as they are)
            c->runs.pop_back();
            n = c->view_run.n;
            n_kmers = c->view_run.total;
            c->v_hi = c->view_run.hi; c->v_lo = c->view_run.lo; c->v_cnt = c->view_run.cnt;
        } else {
            n = 0;
        }
    }
    c->n_sorted = n;
    c->sorted_valid = true;
    c->st.n_distinct = n;
    c->st.n_kmers = n ? n_kmers : 0;
    if (!fast_done && c->tb.size() >= 4) harvest_timing(c);
    if (n_distinct) *n_distinct = n;
    if (n_total) *n_total = c->st.n_kmers;
    return KMC_OK;
}

// kmc_finalize without the wait, for small tables: the unfold of pending (k+16)-mer counts and the small-table finalize
// are queued behind the work in flight and the call returns.  Anything but a small table (sorted runs, extracted keys)
// is finalized the ordinary, synchronous way.
static int kmc_finalize_async_impl(kmc_ctx* c) {
    if (!c) return KMC_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    if (c->async_fin || (c->drained && c->sorted_valid)) return KMC_OK;   // (queued already / final already)
    if (!c->runs.empty() || c->acc_n) return kmc_finalize(c, nullptr, nullptr);
    const size_t fb = (size_t)KMC_FIN_KERNEL_MAX * sizeof(u64);
    int rc = ensure(c, c->o_lo, fb); if (rc) return rc;
    rc = ensure(c, c->o_cnt, fb); if (rc) return rc;
    if (c->KW == 2) { rc = ensure(c, c->o_hi, fb); if (rc) return rc; }
    // Pending (k+16)-mer counts have to be in the table first.  Only the device knows whether there are any; while no poll has
    // ever seen that table in use on this source (sklog_on), none are expected and nothing is launched for them: the kernel
    // below checks, and gives up if the guess was wrong -- the next synchronising call then finalizes the ordinary way, and a
    // kmc_reset in between throws those counts away with the table (kmc_reset_kernel).
    if (c->sk_dirty && c->sklog_on) { rc = flush_sk(c); if (rc) return rc; }
    // (no second try here as in kmc_finalize: room for 8192 keys at least, whatever the last table looked like)
    rc = launch_small_finalize(c, std::max(small_finalize_grid(c), 8192 / KMC_FIN_CHUNK));
    if (rc) return rc;
    c->async_fin = true;
    c->sorted_valid = false;   // (until somebody has looked)
    return KMC_OK;
}

static int kmc_export_impl(kmc_ctx* c, uint64_t* key_hi, uint64_t* key_lo, uint64_t* count, uint64_t cap) {
    if (!c) return KMC_ERR_ARG;
    { int rc = resolve_async(c); if (rc) return rc; }
    if (!c->sorted_valid) return fail(c, KMC_ERR_STATE, "kmc_export before kmc_finalize");
    const u64 n = c->n_sorted;
    if (cap < n) return fail(c, KMC_ERR_ARG, "export capacity %llu < %llu distinct keys", (unsigned long long)cap, (unsigned long long)n);
    if (!n) return KMC_OK;
    if (!key_lo || !count) return fail(c, KMC_ERR_ARG, "null buffer");
    HIPCHK(c, hipSetDevice(c->cfg.device));
    HIPCHK(c, hipMemcpyAsync(key_lo, c->v_lo, n * sizeof(u64), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(count, c->v_cnt, n * sizeof(u64), hipMemcpyDeviceToHost, c->stream));
    if (key_hi) {
        if (c->KW == 2) HIPCHK(c, hipMemcpyAsync(key_hi, c->v_hi, n * sizeof(u64), hipMemcpyDeviceToHost, c->stream));
        else memset(key_hi, 0, n * sizeof(u64));
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return KMC_OK;
}

static int kmc_export_device_impl(kmc_ctx* c, const void** d_key_hi, const void** d_key_lo, const void** d_count, uint64_t* n_distinct) {
    if (!c) return KMC_ERR_ARG;
    { int rc = resolve_async(c); if (rc) return rc; }
    if (!c->sorted_valid) return fail(c, KMC_ERR_STATE, "kmc_export_device before kmc_finalize");
    if (c->view_unsynced) {   // (see poll_fin: the finalize kernel told the host it was done before it ended)
        HIPCHK(c, hipSetDevice(c->cfg.device));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        c->view_unsynced = false;
    }
    if (d_key_hi) *d_key_hi = c->KW == 2 ? c->v_hi : nullptr;
    if (d_key_lo) *d_key_lo = c->v_lo;
    if (d_count) *d_count = c->v_cnt;
    if (n_distinct) *n_distinct = c->n_sorted;
    return KMC_OK;
}

extern "C" uint32_t kmc_owner_of(uint64_t key_hi, uint64_t key_lo, uint32_t n_parts) { return kmc_owner(key_hi, key_lo, n_parts); }

static int kmc_partition_device_impl(kmc_ctx* c, uint32_t n_parts, uint64_t* part_begin, const void** d_key_hi,
                                    const void** d_key_lo, const void** d_count) {
    if (!c || !n_parts || !part_begin) return KMC_ERR_ARG;
    { int rc = resolve_async(c); if (rc) return rc; }
    if (!c->sorted_valid) return fail(c, KMC_ERR_STATE, "kmc_partition_device before kmc_finalize");
    HIPCHK(c, hipSetDevice(c->cfg.device));
    const u64 n = c->n_sorted;
    const size_t nb = (size_t)std::max<u64>(n, 1) * sizeof(u64);
    int rc;
    DevBuf* need[] = {&c->p_lo, &c->p_cnt};
    for (DevBuf* b : need) { rc = ensure(c, *b, nb); if (rc) return rc; }
    rc = ensure(c, c->t_idx0, (size_t)std::max<u64>(n_parts, 1) * sizeof(u64)); if (rc) return rc;
    if (c->KW == 2) { rc = ensure(c, c->p_hi, nb); if (rc) return rc; }
    if (n_parts > 4096) return fail(c, KMC_ERR_ARG, "kmc_partition_device: more than 4096 parts");
    std::vector<unsigned long long> cnt((size_t)n_parts, 0ull);
    unsigned long long* d_cnt = (unsigned long long*)c->t_idx0.p;  // (scratch: n_parts counters, then cursors)
    if (n) {
        const int g2 = grid_for(c, n, 256);
        const u64* vhi = c->KW == 2 ? c->v_hi : (const u64*)nullptr;
        HIPCHK(c, hipMemsetAsync(d_cnt, 0, (size_t)n_parts * sizeof(unsigned long long), c->stream));
        hipLaunchKernelGGL(kmc_owner_count_kernel, dim3(g2), dim3(256), (size_t)n_parts * sizeof(unsigned int), c->stream, vhi, c->v_lo, n, n_parts, d_cnt);
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipMemcpyAsync(cnt.data(), d_cnt, (size_t)n_parts * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    u64 pos = 0;
    std::vector<unsigned long long> cursor((size_t)n_parts);
    for (u32 p = 0; p < n_parts; ++p) { part_begin[p] = pos; cursor[p] = pos; pos += cnt[p]; }
    part_begin[n_parts] = pos;
    if (n) {
        const u64* vhi = c->KW == 2 ? c->v_hi : (const u64*)nullptr;
        HIPCHK(c, hipMemcpyAsync(d_cnt, cursor.data(), (size_t)n_parts * sizeof(unsigned long long), hipMemcpyHostToDevice, c->stream));
        hipLaunchKernelGGL(kmc_owner_scatter_kernel, dim3(grid_for(c, n, 256)), dim3(256), 0, c->stream, vhi, c->v_lo, c->v_cnt, n, n_parts, d_cnt,
                           (u64*)c->p_hi.p, (u64*)c->p_lo.p, (u64*)c->p_cnt.p);
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipStreamSynchronize(c->stream));  // (cursor[] is host memory of this call)
    }
    if (d_key_hi) *d_key_hi = c->KW == 2 ? c->p_hi.p : nullptr;
    if (d_key_lo) *d_key_lo = c->p_lo.p;
    if (d_count) *d_count = c->p_cnt.p;
    return KMC_OK;
}

extern "C" uint64_t kmc_slab_words(const kmc_ctx* c, uint64_t slab_entries) {
    return c ? KMC_SLAB_HEADER + slab_entries * (u64)(c->KW + 1) : 0;
}

static int kmc_pack_slab_device_impl(kmc_ctx* c, void* d_slab, uint64_t slab_entries) {
    if (!c || !d_slab || !slab_entries) return c ? fail(c, KMC_ERR_ARG, "kmc_pack_slab_device: null slab or zero capacity") : KMC_ERR_ARG;
    if (((uintptr_t)d_slab & 7) != 0) return fail(c, KMC_ERR_ARG, "d_slab must be 8-byte aligned");
    HIPCHK(c, hipSetDevice(c->cfg.device));
    { int rc = resolve_async(c); if (rc) return rc; }
    if (c->sk_dirty && (c->sklog_on || c->sorted_valid)) { int r = flush_sk(c); if (r) return r; }   // (else: none expected -- the pack kernel checks)
    if (!c->sorted_valid) {
        // not finalized: pack the live table (unsorted) -- the device decides whether it fits
        GTable g = gtable_of(c, c->tab);
        const u64* skc = c->sk.lo ? c->d_counters + KMC_CTR_N : nullptr;
        const int force = (c->runs.empty() && !c->acc_n) ? 0 : 1;  // sorted runs / extracted keys exist only for high-cardinality input: far too large
        const int grid = grid_for(c, std::min<u64>(slab_entries, KMC_OCC_LIST_CAP), 256);
        if (c->KW == 1) hipLaunchKernelGGL(kmc_pack_slab_live_kernel<1>, dim3(grid), dim3(256), 0, c->stream, g, skc, (u64)slab_entries, force, (u64*)d_slab);
        else hipLaunchKernelGGL(kmc_pack_slab_live_kernel<2>, dim3(grid), dim3(256), 0, c->stream, g, skc, (u64)slab_entries, force, (u64*)d_slab);
        HIPCHK(c, hipGetLastError());
        return KMC_OK;
    }
    const u64 n = c->n_sorted;
    const int grid = grid_for(c, std::max<u64>(std::min(n, slab_entries), KMC_SLAB_HEADER), 256);
    if (c->KW == 1) hipLaunchKernelGGL(kmc_pack_slab_kernel<1>, dim3(grid), dim3(256), 0, c->stream, (const u64*)nullptr, c->v_lo, c->v_cnt, n, c->st.n_kmers, slab_entries, (u64*)d_slab);
    else hipLaunchKernelGGL(kmc_pack_slab_kernel<2>, dim3(grid), dim3(256), 0, c->stream, c->v_hi, c->v_lo, c->v_cnt, n, c->st.n_kmers, slab_entries, (u64*)d_slab);
    HIPCHK(c, hipGetLastError());
    return KMC_OK;
}

static int kmc_merge_slabs_device_impl(kmc_ctx* c, const void* d_slabs, uint32_t n_slabs, uint64_t slab_entries,
                                      uint32_t my_part, uint32_t n_parts) {
    if (!c) return KMC_ERR_ARG;
    if (!d_slabs || !n_slabs || !slab_entries || !n_parts || my_part >= n_parts) return fail(c, KMC_ERR_ARG, "kmc_merge_slabs_device: bad argument");
    if (((uintptr_t)d_slabs & 7) != 0) return fail(c, KMC_ERR_ARG, "d_slabs must be 8-byte aligned");
    HIPCHK(c, hipSetDevice(c->cfg.device));
    { int rc = resolve_async(c); if (rc) return rc; }
    { int rc = undrain(c); if (rc) return rc; }
    // room for the worst case (every pair of every slab new and owned here), accounted like
    // kmc_merge_pairs_device so that a reset table needs no host synchronisation
    const u64 n = (u64)n_slabs * slab_entries;
    if (c->batch_pending || (c->h_counters[KMC_CTR_OCCUPIED] + c->unpolled_adds + n) * 2 > c->tab.cap) {
        if (c->pending) { int rc = poll_and_settle(c); if (rc) return rc; }
        u64 occ = c->h_counters[KMC_CTR_OCCUPIED];
        if ((occ + n) * 2 > c->tab.cap) {
            int rc = grow_to(c, next_pow2((occ + n) * 2));
            if (rc) return rc;
        }
    }
    c->unpolled_adds += n;
    c->sorted_valid = false;
    GTable g = gtable_of(c, c->tab);
    const u64 words = KMC_SLAB_HEADER + slab_entries * (u64)(c->KW + 1);
    const int grid = grid_for(c, n, 256);
    if (c->KW == 1) hipLaunchKernelGGL(kmc_merge_slabs_kernel<1>, dim3(grid), dim3(256), 0, c->stream, g, (const u64*)d_slabs, n_slabs, words, slab_entries, my_part, n_parts);
    else hipLaunchKernelGGL(kmc_merge_slabs_kernel<2>, dim3(grid), dim3(256), 0, c->stream, g, (const u64*)d_slabs, n_slabs, words, slab_entries, my_part, n_parts);
    HIPCHK(c, hipGetLastError());
    c->pending = true;
    c->table_epoch++;
    return KMC_OK;
}

// How the walk kernel cuts a read of read_len bases into pieces (host copy of the device arithmetic,
// kmc_vreads_of / kmc_vread_span): used by the CPU tests to check that every window lies in exactly one piece.
extern "C" uint64_t kmc_read_pieces(uint64_t read_len, int k, uint64_t* starts, uint64_t* ends, uint64_t cap) {
    if (k < 1 || k > KMC_WALK_MAX_K) return 0;
    const u64 n = kmc_vreads_of(read_len, k);
    for (u64 j = 0; j < n && j < cap && starts && ends; ++j) kmc_vread_span(0, read_len, k, j, &starts[j], &ends[j]);
    return n;
}

static int kmc_poll_impl(kmc_ctx* c) {
    if (!c) return KMC_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    { int rc = resolve_async(c); if (rc) return rc; }
    if (c->drained) {  // (nothing can be pending behind a finalize; the device counters are zero, h_counters hold the totals)
        HIPCHK(c, hipStreamSynchronize(c->stream));
        harvest_timing(c);
        return KMC_OK;
    }
    int rc = poll_and_settle(c);
    if (rc) return rc;
    if (c->runs.empty() && !c->acc_n) c->st.n_kmers = c->h_counters[KMC_CTR_KMERS];  // (the sort path counts at finalize)
    harvest_timing(c);
    return KMC_OK;
}

static int kmc_forget_source_impl(kmc_ctx* c, int what) {
    if (!c) return KMC_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    { int rc = resolve_async(c); if (rc) return rc; }
    if ((what & KMC_FORGET_MEMO) && c->walk_memo.p) {
        HIPCHK(c, hipMemsetAsync(c->walk_memo.p, 0, kmc_walk_memo_bytes(c->n_cu, c->KW), c->stream));  // tag 0 = no snapshot
        c->memo_parity = 0;
    }
    if (what & KMC_FORGET_MEMO) c->sklog_on = false;
    if (what & KMC_FORGET_MEMO) {
        // the (k+16)-mer table holds COUNTS as well as structure (walk launches add to it, the unfold into the
        // count table is deferred): give them to their k-mers first -- stream order puts the unfold ahead of
        // the memsets -- so that forgetting the source never touches counts
        if (c->sk_dirty) {
            if (c->pending) { int rp = poll_and_settle(c); if (rp) return rp; }   // (sizes the unfold exactly; recovers a wrong prediction first)
            int rf = settle_sk_polled(c);
            if (rf) return rf;
        }
        int rs = sk_clear(c);
        if (rs) return rs;
    }
    if (what & KMC_FORGET_HISTORY) {
        c->rho_hist = -1.0;
        c->rho_last = c->rho_max = 0.0;
        c->prefer_sort = false;
        c->walk_ms_per_base = c->sort_ms_per_base = -1.0;
        c->sort_by_cost = false;
        c->msd_dup_heavy = false;
        c->walk_overflowed = false;
    }
    return KMC_OK;
}

static int kmc_get_stats_impl(const kmc_ctx* c, kmc_stats* out) {
    if (!c || !out) return KMC_ERR_ARG;
    harvest_timing(const_cast<kmc_ctx*>(c));  // (kernel_ms_* of a batch that has finished since the last call)
    *out = c->st;
    return KMC_OK;
}

// Whole file through kmc_parse_fasta, then batches of <= 1 GiB (fallback when the file cannot be mapped)
static int count_file_whole(kmc_ctx* c, const char* path, uint64_t* n_distinct, uint64_t* n_total) {
    kmc_reads rd;
    char eb[256] = {0};
    int rc = kmc_parse_fasta(path, &rd, eb, sizeof(eb));
    if (rc) return fail(c, rc, "%s: %s", path, eb);
    // (LR mode: a byte outside ACGT is found by the kernel, which looks at exactly the bytes the windows
    // read -- like the reference's bucket_sort, main.rs:17-23, which never sees reads shorter than 80
    // bases or the bytes no window covers)
    const u64 BATCH = 1ull << 30;
    u64 r0 = 0;
    std::vector<u64> offs;
    while (r0 < rd.n_reads && !rc) {
        u64 r1 = r0 + 1;
        while (r1 < rd.n_reads && rd.offsets[r1 + 1] - rd.offsets[r0] <= BATCH) r1++;
        offs.resize((size_t)(r1 - r0 + 1));
        for (u64 i = r0; i <= r1; ++i) offs[(size_t)(i - r0)] = rd.offsets[i] - rd.offsets[r0];
        rc = kmc_add_batch(c, rd.bases + rd.offsets[r0], offs.data(), r1 - r0);
        r0 = r1;
    }
    kmc_free_reads(&rd);
    if (rc) return rc;
    return kmc_finalize(c, n_distinct, n_total);
}

static int count_file_pipeline(kmc_ctx** ctxs, uint32_t n_ctx, const char* path, uint64_t* n_distinct, uint64_t* n_total) {
    kmc_ctx* c0 = ctxs[0];
    // chunk size: a sixteenth of the file between 32 and 128 MiB (measured on 1 and 4 GB files, 16 host
    // threads: 64-128 MiB is best -- 0.072 s / 0.24 s; 256 MiB chunks cost 0.10 / 0.26 s: the two pinned
    // buffers take longer to allocate and the first upload starts later); KMC_INGEST_CHUNK_BYTES overrides
    u64 chunk_bytes = 0;
    if (const char* e = getenv("KMC_INGEST_CHUNK_BYTES")) { u64 v = strtoull(e, nullptr, 10); if (v) chunk_bytes = v; }
    u64 fsize = 0;
    {
        FILE* f = fopen(path, "rb");
        if (f) { if (fseeko(f, 0, SEEK_END) == 0) fsize = (u64)ftello(f); fclose(f); }
    }
    if (!chunk_bytes) chunk_bytes = std::min<u64>(std::max<u64>(fsize / 16, 32ull << 20), 128ull << 20);
    // should the input turn out high-cardinality, the sort path's key accumulator is allocated once for the
    // file's share of this ctx (a base position per byte of text at most) instead of growing chunk by chunk
    for (uint32_t i = 0; i < n_ctx; ++i) ctxs[i]->acc_hint = fsize / n_ctx + chunk_bytes;
    KmcFastaIngest ing;
    std::string err;
    int rc = ing.open(path, chunk_bytes, &err);
    if (rc == KMC_ERR_IO && err != "Error during opening the file" && n_ctx == 1) return count_file_whole(c0, path, n_distinct, n_total);  // (not mappable)
    if (rc) return fail(c0, rc, "%s: %s", path, err.c_str());
    const u64 cap = ing.chunk_capacity();
    struct Pinned { uint8_t* bases = nullptr; u64* offs = nullptr; u64 offs_cap = 0; hipEvent_t ev = nullptr; bool busy = false; };
    std::vector<Pinned> pin((size_t)n_ctx * 2);  // two per ctx
    auto cleanup = [&]() {
        for (uint32_t i = 0; i < n_ctx; ++i) { (void)hipSetDevice(ctxs[i]->cfg.device); (void)hipStreamSynchronize(ctxs[i]->stream); }
        for (auto& p : pin) {
            if (p.bases) (void)hipHostFree(p.bases);
            if (p.offs) (void)hipHostFree(p.offs);
            if (p.ev) (void)hipEventDestroy(p.ev);
        }
    };
    auto body = [&]() -> int {
        KmcIngestChunk ck;
        for (u64 j = 0;; ++j) {
            kmc_ctx* c = ctxs[j % n_ctx];  // chunks go round-robin over the GPUs
            Pinned& p = pin[(size_t)(j % n_ctx) * 2 + ((j / n_ctx) & 1)];
            HIPCHK(c, hipSetDevice(c->cfg.device));
            if (!p.bases) {
                HIPCHK(c, hipHostMalloc((void**)&p.bases, (size_t)cap));
                HIPCHK(c, hipEventCreateWithFlags(&p.ev, hipEventDisableTiming));
            }
            if (p.busy) { HIPCHK(c, hipEventSynchronize(p.ev)); p.busy = false; }  // its upload two rounds ago has finished
            int r;
            try { r = ing.next(p.bases, false, &ck, &err); } catch (const std::bad_alloc&) { return fail(c0, KMC_ERR_NOMEM, "out of memory while parsing %s", path); }
            if (r) return fail(c0, r, "%s: %s", path, err.c_str());
            if (ck.n_reads) {
                if (p.offs_cap < ck.n_reads + 1) {
                    if (p.offs) { HIPCHK(c, hipHostFree(p.offs)); p.offs = nullptr; }
                    p.offs_cap = (ck.n_reads + 1) * 5 / 4 + 1024;
                    HIPCHK(c, hipHostMalloc((void**)&p.offs, (size_t)p.offs_cap * sizeof(u64)));
                }
                memcpy(p.offs, ck.offsets.data(), (size_t)(ck.n_reads + 1) * sizeof(u64));
                // settle the ctx's previous batch first (it may still read the staging buffers; its kernels
                // finished long ago -- this chunk took longer to parse), then queue upload + count without waiting
                auto fwd = [&](int code) { if (c != c0) memcpy(c0->err, c->err, sizeof(c0->err)); return code; };
                if (c->pending) { r = poll_and_settle(c); if (r) return fwd(r); }
                r = ensure(c, c->st_bases, ck.n_bases + 64);
                if (r) return fwd(r);
                r = ensure(c, c->st_offsets, (ck.n_reads + 1) * sizeof(u64));
                if (r) return fwd(r);
                for (const auto& pc : ck.pieces)
                    HIPCHK(c, hipMemcpyAsync((uint8_t*)c->st_bases.p + pc.dst_off, p.bases + pc.src_off, (size_t)pc.n_bytes, hipMemcpyHostToDevice, c->stream));
                HIPCHK(c, hipMemcpyAsync(c->st_offsets.p, p.offs, (size_t)(ck.n_reads + 1) * sizeof(u64), hipMemcpyHostToDevice, c->stream));
                HIPCHK(c, hipEventRecord(p.ev, c->stream));
                p.busy = true;
                r = count_batch_device(c, (const uint8_t*)c->st_bases.p, (const u64*)c->st_offsets.p, ck.n_reads, ck.n_bases, ck.max_read_len);
                if (r) return fwd(r);
            }
            if (ck.eof) break;
        }
        return KMC_OK;
    };
    // One GPU: parse chunk j+1 while the GPU uploads and counts chunk j (body()).  Several GPUs: ONE FEEDER
    // THREAD PER GPU, each with its own pinned pair, parsing the chunks i, i+N, i+2N, ... of the mapped file
    // itself (the first version parsed every chunk in one host loop and dealt them out round-robin: N GPUs
    // waited on one parser).  The parser threads of the host are split between the feeders.
    if (n_ctx == 1) {
        rc = body();
        cleanup();
        if (rc) return rc;
    } else {
        cleanup();  // (nothing allocated yet; the feeders own their buffers)
        const size_t n_chunks = ing.n_chunks();
        const unsigned per = std::max(1u, ing.threads() / n_ctx);
        std::atomic<long long> stop_at{LLONG_MAX};   // first chunk in which an empty record ended the input
        std::vector<int> frc(n_ctx, KMC_OK);
        std::vector<long long> queued_max(n_ctx, -1);
        auto feed = [&](uint32_t i) {
            kmc_ctx* c = ctxs[i];
            Pinned pp[2];
            auto fin = [&]() {
                (void)hipStreamSynchronize(c->stream);
                for (auto& p : pp) { if (p.bases) (void)hipHostFree(p.bases); if (p.offs) (void)hipHostFree(p.offs); if (p.ev) (void)hipEventDestroy(p.ev); }
            };
            auto run = [&]() -> int {
                HIPCHK(c, hipSetDevice(c->cfg.device));
                KmcIngestChunk ck;
                std::string e2;
                size_t it = 0;
                for (size_t j = i; j < n_chunks; j += n_ctx, ++it) {
                    if ((long long)j > stop_at.load()) break;
                    Pinned& p = pp[it & 1];
                    if (!p.bases) {
                        HIPCHK(c, hipHostMalloc((void**)&p.bases, (size_t)cap));
                        HIPCHK(c, hipEventCreateWithFlags(&p.ev, hipEventDisableTiming));
                    }
                    if (p.busy) { HIPCHK(c, hipEventSynchronize(p.ev)); p.busy = false; }
                    int r;
                    try { r = ing.parse_chunk(j, per, p.bases, false, &ck, &e2); } catch (const std::bad_alloc&) { return fail(c, KMC_ERR_NOMEM, "out of memory while parsing %s", path); }
                    if (r) return fail(c, r, "%s: %s", path, e2.c_str());
                    if (ck.terminated) { long long cur = stop_at.load(); while ((long long)j < cur && !stop_at.compare_exchange_weak(cur, (long long)j)) {} }
                    if ((long long)j > stop_at.load()) break;
                    if (!ck.n_reads) continue;
                    if (p.offs_cap < ck.n_reads + 1) {
                        if (p.offs) { HIPCHK(c, hipHostFree(p.offs)); p.offs = nullptr; }
                        p.offs_cap = (ck.n_reads + 1) * 5 / 4 + 1024;
                        HIPCHK(c, hipHostMalloc((void**)&p.offs, (size_t)p.offs_cap * sizeof(u64)));
                    }
                    memcpy(p.offs, ck.offsets.data(), (size_t)(ck.n_reads + 1) * sizeof(u64));
                    if (c->pending) { r = poll_and_settle(c); if (r) return r; }
                    r = ensure(c, c->st_bases, ck.n_bases + 64);
                    if (r) return r;
                    r = ensure(c, c->st_offsets, (ck.n_reads + 1) * sizeof(u64));
                    if (r) return r;
                    for (const auto& pc : ck.pieces)
                        HIPCHK(c, hipMemcpyAsync((uint8_t*)c->st_bases.p + pc.dst_off, p.bases + pc.src_off, (size_t)pc.n_bytes, hipMemcpyHostToDevice, c->stream));
                    HIPCHK(c, hipMemcpyAsync(c->st_offsets.p, p.offs, (size_t)(ck.n_reads + 1) * sizeof(u64), hipMemcpyHostToDevice, c->stream));
                    HIPCHK(c, hipEventRecord(p.ev, c->stream));
                    p.busy = true;
                    r = count_batch_device(c, (const uint8_t*)c->st_bases.p, (const u64*)c->st_offsets.p, ck.n_reads, ck.n_bases, ck.max_read_len);
                    if (r) return r;
                    queued_max[i] = (long long)j;
                }
                return KMC_OK;
            };
            int r;
            try { r = run(); } catch (const std::bad_alloc&) { r = fail(c, KMC_ERR_NOMEM, "out of host memory"); } catch (...) { r = fail(c, KMC_ERR_HIP, "internal error in a feeder thread"); }
            frc[i] = r;
            fin();
        };
        {
            std::vector<std::thread> th;
            th.reserve(n_ctx);
            struct Join { std::vector<std::thread>& t; ~Join() { for (auto& x : t) if (x.joinable()) x.join(); } } join{th};
            for (uint32_t i = 1; i < n_ctx; ++i) th.emplace_back(feed, i);
            feed(0);
        }
        for (uint32_t i = 0; i < n_ctx; ++i)
            if (frc[i]) { if (i) memcpy(c0->err, ctxs[i]->err, sizeof(c0->err)); return frc[i]; }
        bool raced = false;  // a chunk behind the terminating record was counted before that record was found
        for (uint32_t i = 0; i < n_ctx; ++i) raced |= queued_max[i] > stop_at.load();
        if (raced) {
            for (uint32_t i = 0; i < n_ctx; ++i) { int r = kmc_reset(ctxs[i]); if (r) return r; }
            KmcFastaIngest again;
            std::string e3;
            rc = again.open(path, chunk_bytes, &e3);
            if (rc) return fail(c0, rc, "%s: %s", path, e3.c_str());
            // (the sequential loop stops exactly at the terminating record)
            KmcIngestChunk ck;
            std::vector<uint8_t> hostbuf((size_t)again.chunk_capacity());
            for (u64 j = 0;; ++j) {
                kmc_ctx* c = ctxs[j % n_ctx];
                int r = again.next(hostbuf.data(), false, &ck, &e3);
                if (r) return fail(c0, r, "%s: %s", path, e3.c_str());
                if (ck.n_reads) {
                    std::vector<uint8_t> dense((size_t)ck.n_bases);
                    for (const auto& pc : ck.pieces) memcpy(dense.data() + pc.dst_off, hostbuf.data() + pc.src_off, (size_t)pc.n_bytes);
                    r = kmc_add_batch(c, dense.data(), ck.offsets.data(), ck.n_reads);
                    if (r) { if (c != c0) memcpy(c0->err, c->err, sizeof(c0->err)); return r; }
                }
                if (ck.eof) break;
            }
        }
    }
    // reduce: pairwise (a tree of depth log2 N): in round r the contexts i with i % 2^(r+1) == 0 take the sorted
    // table of context i + 2^r over a peer copy (xGMI) into receive buffers they keep, and merge it; the pairs
    // of a round run side by side.  ctxs[0] ends up with everything.  (The first version finalized, allocated
    // and copied every other GPU's table into GPU 0 one after the other.)
    for (uint32_t stride = 1; stride < n_ctx; stride *= 2) {
        for (uint32_t i = 0; i + stride < n_ctx; i += 2 * stride) {
            kmc_ctx *dst = ctxs[i], *src = ctxs[i + stride];
            u64 nd = 0, nt = 0;
            rc = kmc_finalize(src, &nd, &nt);
            if (rc) { memcpy(c0->err, src->err, sizeof(c0->err)); return rc; }
            if (!nd) continue;
            if (src->view_unsynced) {   // (its view is read on ANOTHER ctx's stream below: poll_fin, kmc_export_device)
                HIPCHK(c0, hipSetDevice(src->cfg.device));
                HIPCHK(c0, hipStreamSynchronize(src->stream));
                src->view_unsynced = false;
            }
            HIPCHK(c0, hipSetDevice(dst->cfg.device));
            const size_t nb = (size_t)nd * sizeof(u64);
            rc = ensure(dst, dst->rx_lo, nb); if (!rc) rc = ensure(dst, dst->rx_cnt, nb); if (!rc && dst->KW == 2) rc = ensure(dst, dst->rx_hi, nb);
            if (rc) { if (dst != c0) memcpy(c0->err, dst->err, sizeof(c0->err)); return rc; }
            hipError_t e = hipMemcpyPeerAsync(dst->rx_lo.p, dst->cfg.device, src->v_lo, src->cfg.device, nb, dst->stream);
            if (e == hipSuccess) e = hipMemcpyPeerAsync(dst->rx_cnt.p, dst->cfg.device, src->v_cnt, src->cfg.device, nb, dst->stream);
            if (e == hipSuccess && dst->KW == 2) e = hipMemcpyPeerAsync(dst->rx_hi.p, dst->cfg.device, src->v_hi, src->cfg.device, nb, dst->stream);
            if (e != hipSuccess) return fail(c0, KMC_ERR_HIP, "peer copy from device %d failed: %s", src->cfg.device, hipGetErrorString(e));
            rc = kmc_merge_pairs_device(dst, dst->KW == 2 ? dst->rx_hi.p : nullptr, dst->rx_lo.p, dst->rx_cnt.p, nd);
            if (rc) { if (dst != c0) memcpy(c0->err, dst->err, sizeof(c0->err)); return rc; }
        }
        for (uint32_t i = 0; i + stride < n_ctx; i += 2 * stride) {  // the round's copies and merges have finished
            HIPCHK(c0, hipSetDevice(ctxs[i]->cfg.device));
            HIPCHK(c0, hipStreamSynchronize(ctxs[i]->stream));
        }
    }
    HIPCHK(c0, hipSetDevice(c0->cfg.device));
    return kmc_finalize(c0, n_distinct, n_total);
}

// FASTA path in, table out (the reference's File::open + Reader + record loop, main.rs:44-46,58-62,
// feeding the window loop).  Pipeline: the streaming reader (kmc_ingest.h) parses chunk j+1 on the
// host cores into one of two pinned buffers while the GPU uploads and counts chunk j; the reader's
// per-thread pieces go straight to their dense place in the device buffer, so the host never
// stitches or copies the sequence a second time.
static int kmc_count_file_impl(kmc_ctx* c, const char* path, uint64_t* n_distinct, uint64_t* n_total) {
    if (!c || !path) return KMC_ERR_ARG;
    return count_file_pipeline(&c, 1, path, n_distinct, n_total);
}

// The same on several GPUs of ONE process (the CLI's --gpus N): chunks go round-robin to the ctxs,
// each with its own pinned double buffer, and the tables are reduced into ctxs[0] by peer copies.
// (Scaling runs use one process per GPU and RCCL instead: k-mer-count_amd/distributed.py.)
static int kmc_count_file_multi_impl(kmc_ctx** ctxs, uint32_t n_ctx, const char* path, uint64_t* n_distinct, uint64_t* n_total) {
    if (!ctxs || !n_ctx || !path) return KMC_ERR_ARG;
    for (uint32_t i = 0; i < n_ctx; ++i) {
        if (!ctxs[i]) return KMC_ERR_ARG;
        if (ctxs[i]->cfg.k != ctxs[0]->cfg.k || ctxs[i]->cfg.mode != ctxs[0]->cfg.mode || ctxs[i]->cfg.canonical != ctxs[0]->cfg.canonical)
            return fail(ctxs[0], KMC_ERR_ARG, "kmc_count_file_multi: ctx %u differs from ctx 0 in k, mode or canonical", i);
        for (uint32_t j = 0; j < i; ++j) if (ctxs[j] == ctxs[i]) return fail(ctxs[0], KMC_ERR_ARG, "kmc_count_file_multi: ctx %u listed twice", i);
    }
    return count_file_pipeline(ctxs, n_ctx, path, n_distinct, n_total);
}

static int kmc_synth_reads_device_impl(const kmc_synth* s, uint64_t first_record, uint64_t n_records, void* d_bases,
                                      void* d_offsets, int device, void* stream) {
    if (!s || !d_bases || !d_offsets || !s->line_len || !s->lines_per_record) return KMC_ERR_ARG;
    if (hipSetDevice(device) != hipSuccess) return KMC_ERR_NO_DEVICE;
    hipStream_t st = (hipStream_t)stream;
    const u64 read_len = (u64)s->lines_per_record * s->line_len;
    const u64 n_bases = n_records * read_len;
    uint8_t* d_pool = nullptr;
    if (s->pool) {
        std::vector<uint8_t> pool((size_t)s->pool * s->line_len);
        for (u32 p = 0; p < s->pool; ++p)
            for (u32 x = 0; x < s->line_len; ++x) pool[(size_t)p * s->line_len + x] = kmc_synth_pool_base(s->seed, s->line_len, p, x);
        if (hipMalloc((void**)&d_pool, pool.size()) != hipSuccess) return KMC_ERR_NOMEM;
        if (hipMemcpy(d_pool, pool.data(), pool.size(), hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(d_pool); return KMC_ERR_HIP; }
    }
    u64 n16 = (n_bases + 15) / 16;
    int grid = (int)std::min<u64>((n16 + 255) / 256, 256 * 16);
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(kmc_synth_kernel, dim3(grid), dim3(256), 0, st, s->seed, s->pool, s->line_len, s->lines_per_record,
                       first_record, n_bases, (const uint8_t*)d_pool, (uint8_t*)d_bases);
    int g2 = (int)std::min<u64>((n_records + 256) / 256, 256 * 16);
    hipLaunchKernelGGL(kmc_synth_offsets_kernel, dim3(g2), dim3(256), 0, st, n_records, read_len, (u64*)d_offsets);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (d_pool) (void)hipFree(d_pool);
    return e == hipSuccess ? KMC_OK : KMC_ERR_HIP;
}

// Measured streaming-read peak (kmc_peak.hip.h): `iters` launches of the plain read kernel over
// [d_buf, d_buf + n_bytes) after one warm-up launch, bracketed by one hipEvent pair on `stream`.
static int kmc_read_peak_device_impl(const void* d_buf, uint64_t n_bytes, int device, void* stream, int shape, int iters,
                                     double* ms_avg, uint64_t* xor_out) {
    if (!d_buf || n_bytes < 16 || iters < 1 || !ms_avg || ((uintptr_t)d_buf & 15) != 0) return KMC_ERR_ARG;
    if (hipSetDevice(device) != hipSuccess) return KMC_ERR_NO_DEVICE;
    hipStream_t st = (hipStream_t)stream;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return KMC_ERR_HIP;
    const int n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    unsigned long long* d_out = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int rc = KMC_OK;
    float ms = 0.f;
    unsigned long long h = 0;
    const u64 n16 = n_bytes / 16;
    auto launch = [&]() {
        switch (shape) {
            case 0: hipLaunchKernelGGL(kmc_read_peak_kernel<5>, dim3(n_cu), dim3(1024), 0, st, (const uint8_t*)d_buf, n16, d_out); break;      // the walk kernel's shape
            case 1: hipLaunchKernelGGL(kmc_read_peak_kernel<8>, dim3(n_cu * 8), dim3(256), 0, st, (const uint8_t*)d_buf, n16, d_out); break;
            case 2: hipLaunchKernelGGL(kmc_read_peak_kernel<4>, dim3(n_cu * 2), dim3(1024), 0, st, (const uint8_t*)d_buf, n16, d_out); break;
            default: hipLaunchKernelGGL(kmc_read_peak_kernel<8>, dim3(n_cu * 4), dim3(512), 0, st, (const uint8_t*)d_buf, n16, d_out); break;
        }
    };
    if (hipMalloc((void**)&d_out, sizeof(unsigned long long)) != hipSuccess) return KMC_ERR_NOMEM;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess || hipMemsetAsync(d_out, 0, sizeof(unsigned long long), st) != hipSuccess) rc = KMC_ERR_HIP;
    if (!rc) {
        launch();   // warm-up (clocks, code object)
        (void)hipEventRecord(e0, st);
        for (int i = 0; i < iters; ++i) launch();
        (void)hipEventRecord(e1, st);
        if (hipGetLastError() != hipSuccess || hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess ||
            hipMemcpy(&h, d_out, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) rc = KMC_ERR_HIP;
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    (void)hipFree(d_out);
    if (rc) return rc;
    *ms_avg = (double)ms / iters;
    if (xor_out) *xor_out = h;   // (iters + 1 launches: the xor of an even number of passes is 0, of an odd number the buffer's checksum)
    return KMC_OK;
}

#ifdef KMC_LEAF_STAMPS
extern "C" int kmc_debug_leaf_stamps(uint64_t* out16, int reset) {
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(kmc_leaf_stamps), 16 * sizeof(uint64_t)) != hipSuccess) return KMC_ERR_HIP;
    if (reset) { uint64_t z[16] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(kmc_leaf_stamps), z, sizeof(z)) != hipSuccess) return KMC_ERR_HIP; }
    return KMC_OK;
}
#endif
#ifdef KMC_WALK_STAMPS
extern "C" int kmc_debug_walk_stamps(uint64_t* out, uint32_t n_words) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(kmc_walk_stamps), std::min<size_t>(n_words, 256 * 24) * sizeof(uint64_t)) == hipSuccess ? KMC_OK : KMC_ERR_HIP;
}
#endif
// ---- the ABI proper: no C++ exception leaves the library (kmc.h: "no exception or abort crosses the ABI") ----
namespace {
template <typename F>
int guarded(kmc_ctx* c, F&& f) noexcept {
    try {
        return f();
    } catch (const std::bad_alloc&) {
        return fail(c, KMC_ERR_NOMEM, "out of host memory");
    } catch (const std::exception& e) {
        return fail(c, KMC_ERR_HIP, "internal error: %s", e.what());
    } catch (...) {
        return fail(c, KMC_ERR_HIP, "internal error (unknown C++ exception)");
    }
}
}  // namespace
extern "C" int kmc_create(kmc_ctx** out, const kmc_config* cfg) {
    return guarded(nullptr, [&]() -> int { return kmc_create_impl(out, cfg); });
}
extern "C" int kmc_reset(kmc_ctx* c) {
    return guarded(c, [&]() -> int { return kmc_reset_impl(c); });
}
extern "C" int kmc_add_batch_device(kmc_ctx* c, const void* d_bases, const void* d_offsets, uint64_t n_reads,
                                    uint64_t n_bases, uint64_t max_read_len) {
    return guarded(c, [&]() -> int { return kmc_add_batch_device_impl(c, d_bases, d_offsets, n_reads, n_bases, max_read_len); });
}
extern "C" int kmc_add_batch(kmc_ctx* c, const uint8_t* bases, const uint64_t* offsets, uint64_t n_reads) {
    return guarded(c, [&]() -> int { return kmc_add_batch_impl(c, bases, offsets, n_reads); });
}
extern "C" int kmc_merge_pairs_device(kmc_ctx* c, const void* d_key_hi, const void* d_key_lo, const void* d_count, uint64_t n) {
    return guarded(c, [&]() -> int { return kmc_merge_pairs_device_impl(c, d_key_hi, d_key_lo, d_count, n); });
}
extern "C" int kmc_finalize(kmc_ctx* c, uint64_t* n_distinct, uint64_t* n_total) {
    return guarded(c, [&]() -> int { return kmc_finalize_impl(c, n_distinct, n_total); });
}
extern "C" int kmc_finalize_async(kmc_ctx* c) {
    return guarded(c, [&]() -> int { return kmc_finalize_async_impl(c); });
}
extern "C" int kmc_export(kmc_ctx* c, uint64_t* key_hi, uint64_t* key_lo, uint64_t* count, uint64_t cap) {
    return guarded(c, [&]() -> int { return kmc_export_impl(c, key_hi, key_lo, count, cap); });
}
extern "C" int kmc_export_device(kmc_ctx* c, const void** d_key_hi, const void** d_key_lo, const void** d_count, uint64_t* n_distinct) {
    return guarded(c, [&]() -> int { return kmc_export_device_impl(c, d_key_hi, d_key_lo, d_count, n_distinct); });
}
extern "C" int kmc_partition_device(kmc_ctx* c, uint32_t n_parts, uint64_t* part_begin, const void** d_key_hi,
                                    const void** d_key_lo, const void** d_count) {
    return guarded(c, [&]() -> int { return kmc_partition_device_impl(c, n_parts, part_begin, d_key_hi, d_key_lo, d_count); });
}
extern "C" int kmc_pack_slab_device(kmc_ctx* c, void* d_slab, uint64_t slab_entries) {
    return guarded(c, [&]() -> int { return kmc_pack_slab_device_impl(c, d_slab, slab_entries); });
}
extern "C" int kmc_merge_slabs_device(kmc_ctx* c, const void* d_slabs, uint32_t n_slabs, uint64_t slab_entries,
                                      uint32_t my_part, uint32_t n_parts) {
    return guarded(c, [&]() -> int { return kmc_merge_slabs_device_impl(c, d_slabs, n_slabs, slab_entries, my_part, n_parts); });
}
extern "C" int kmc_poll(kmc_ctx* c) {
    return guarded(c, [&]() -> int { return kmc_poll_impl(c); });
}
extern "C" int kmc_sync(kmc_ctx* c) {
    if (!c) return KMC_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return KMC_OK;
}

extern "C" int kmc_forget_source(kmc_ctx* c, int what) {
    return guarded(c, [&]() -> int { return kmc_forget_source_impl(c, what); });
}
extern "C" int kmc_get_stats(const kmc_ctx* c, kmc_stats* out) {
    return guarded(const_cast<kmc_ctx*>(c), [&]() -> int { return kmc_get_stats_impl(c, out); });
}
extern "C" int kmc_count_file(kmc_ctx* c, const char* path, uint64_t* n_distinct, uint64_t* n_total) {
    return guarded(c, [&]() -> int { return kmc_count_file_impl(c, path, n_distinct, n_total); });
}
extern "C" int kmc_count_file_multi(kmc_ctx** ctxs, uint32_t n_ctx, const char* path, uint64_t* n_distinct, uint64_t* n_total) {
    return guarded((ctxs && n_ctx ? ctxs[0] : nullptr), [&]() -> int { return kmc_count_file_multi_impl(ctxs, n_ctx, path, n_distinct, n_total); });
}
extern "C" int kmc_read_peak_device(const void* d_buf, uint64_t n_bytes, int device, void* stream, int shape, int iters, double* ms_avg, uint64_t* xor_out) {
    return guarded(nullptr, [&]() -> int { return kmc_read_peak_device_impl(d_buf, n_bytes, device, stream, shape, iters, ms_avg, xor_out); });
}
extern "C" int kmc_synth_reads_device(const kmc_synth* s, uint64_t first_record, uint64_t n_records, void* d_bases,
                                      void* d_offsets, int device, void* stream) {
    return guarded(nullptr, [&]() -> int { return kmc_synth_reads_device_impl(s, first_record, n_records, d_bases, d_offsets, device, stream); });
}
