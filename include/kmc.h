/*
 * kmc.h -- C ABI of the MI355X k-mer counter (libkmc.so).
 *
 * This is the drop-in boundary for the reference's hot path.  The reference
 * (jaxonwang/k-mer-count @ v1, paths below relative to /root/reference) has no FFI of its own:
 * its whole pipeline is one function, k-mer-count/src/main.rs:43-91.  The entry points below
 * are what a Rust `extern "C"` block in that crate would bind so that main() keeps FASTA
 * reading (main.rs:44-46,58-62) and printing (main.rs:88-90) and hands the loop nest in
 * between (main.rs:63-87) to the GPU.  INTEGRATION.md shows the binding.
 *
 *   reference code being replaced                         entry point
 *   ----------------------------------------------------  ---------------------------------
 *   constants l_len/r_len/80..141, main.rs:48-49,63       kmc_create(kmc_config)
 *   per-record window loop + push, main.rs:58-81          kmc_add_batch / kmc_add_batch_device
 *   radix_sort + sort (grouping), main.rs:84,87           kmc_finalize
 *   iteration over the sorted result, main.rs:88-90       kmc_export (sorted ascending)
 *   File::open + Reader + loop, main.rs:44-46,58-62       kmc_count_file (convenience; host parser)
 *   random_fasta_generator.py:5-15                        kmc_synth_* (seeded, sized re-creation)
 *
 * Conventions
 *   - Every function returns 0 (KMC_OK) or a negative kmc_status; no exception or abort crosses
 *     the ABI (the reference panics instead: main.rs:23,35,44,59).
 *   - Alphabet A=0 C=1 G=2 T=3 (main.rs:19-22); keys are packed MSB-first so unsigned order of
 *     (key_hi,key_lo) equals the reference's string order (main.rs:87).  key_hi is 0 for k<=32.
 *   - The caller owns every host buffer passed in or out; buffers passed to kmc_add_batch may be
 *     reused as soon as it returns.  The library owns the ctx and all device memory.
 *   - A ctx is bound to ONE GPU and is not thread-safe; distinct ctxs are independent.
 *   - There is NO CPU fallback: if no HIP device is usable kmc_create fails with
 *     KMC_ERR_NO_DEVICE.
 */
#ifndef KMC_H
#define KMC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KMC_VERSION_MAJOR 0
#define KMC_VERSION_MINOR 1

typedef enum kmc_status {
    KMC_OK = 0,
    KMC_ERR_ARG = -1,        /* bad argument / unsupported k */
    KMC_ERR_NO_DEVICE = -2,  /* no usable HIP device (no CPU fallback exists) */
    KMC_ERR_HIP = -3,        /* a HIP runtime call failed; see kmc_last_error */
    KMC_ERR_NOMEM = -4,      /* host or device allocation failed */
    KMC_ERR_IO = -5,         /* cannot open/read file            (main.rs:44) */
    KMC_ERR_FORMAT = -6,     /* "Expected > at record start."    (main.rs:59) */
    KMC_ERR_ALPHABET = -7,   /* non-ACGT byte in LR mode         (main.rs:23) */
    KMC_ERR_CAPACITY = -8,   /* count table and spill area exhausted */
    KMC_ERR_STATE = -9       /* call out of order (e.g. export before finalize) */
} kmc_status;

typedef enum kmc_mode {
    KMC_MODE_CONTIG = 0, /* contiguous k-mers, SURVEY.md 8a-def */
    KMC_MODE_LR = 1      /* reference mode: 27 + gap + 27, chunk sizes 80..=140 (main.rs:48-49,63) */
} kmc_mode;

typedef enum kmc_algo {
    KMC_ALGO_AUTO = 0,   /* pick per batch: WALK for short reads, SORT once the input proves high-cardinality, else STREAM */
    KMC_ALGO_STREAM = 1, /* per-k-mer LDS partial histogram + global atomics (any input) */
    KMC_ALGO_WALK = 2,   /* memoised successor walk: one LDS lookup per 16 bases (longer reads as overlapping pieces of 416) */
    KMC_ALGO_SORT = 3    /* extract every window; batches accumulate; hand-written MSD radix sort + run-length when the
                            result is needed (high-cardinality input) */
} kmc_algo;

/* Deviations from the config sketched in SURVEY.md 8b (`n_devices`, `device_ids*`, `backend`), on purpose:
 *   - ONE `device` per ctx instead of n_devices/device_ids: a ctx is one GPU's table and stream.  Several
 *     GPUs of one process = several ctxs handed to kmc_count_file_multi (the CLI's --gpus N); scaling
 *     runs use one process per GPU and RCCL (kmc_pack_slab_device / kmc_merge_slabs_device /
 *     kmc_partition_device are the device-side halves of that reduce).
 *   - no `backend` field and no `--backend cpu`: the library has no CPU path to select (kmc_create
 *     fails with KMC_ERR_NO_DEVICE instead); the CPU restatement lives in oracle/ as test infrastructure.
 *   - `algo` and `stream` are additions (kernel choice for measurements; running on the caller's stream so
 *     that kernels and collectives are ordered on the device).
 * Alphabet rule of KMC_MODE_LR: the reference panics on a character outside ACGT that its bucket_sort
 * inspects (main.rs:17-23: chunk indices 1..53 of every emitted chunk).  Here KMC_ERR_ALPHABET is raised
 * for such a byte at ANY index of an emitted chunk (index 0 too: a 2-bit key cannot hold it); bytes no
 * window reads -- reads shorter than 80 bases, the uncovered middle of reads of 80..105 bases -- are
 * accepted, as in the reference.  KMC_MODE_CONTIG skips windows that contain such a byte (8a-def). */
typedef struct kmc_config {
    uint32_t struct_size;   /* = sizeof(kmc_config) */
    int32_t  k;             /* 1..63 in KMC_MODE_CONTIG; ignored in KMC_MODE_LR */
    int32_t  mode;          /* kmc_mode */
    int32_t  canonical;     /* 1: key = min(fwd, revcomp); 0: forward strand (the reference is forward-only) */
    int32_t  device;        /* HIP device ordinal */
    int32_t  algo;          /* kmc_algo */
    uint64_t capacity_hint; /* expected distinct keys; 0 = default.  The table grows between batches. */
    void*    stream;        /* hipStream_t to run on, or NULL for a ctx-owned stream */
} kmc_config;

typedef struct kmc_ctx kmc_ctx;

/* Per-ctx counters, all cumulative since kmc_create / kmc_reset. */
typedef struct kmc_stats {
    uint64_t n_reads;
    uint64_t n_bases;
    uint64_t n_kmers;         /* valid windows counted (== sum of counts) */
    uint64_t n_distinct;      /* valid after kmc_finalize */
    uint64_t table_capacity;  /* slots */
    uint64_t n_spilled;       /* pairs that went through the spill area */
    uint64_t n_batches;
    double   kernel_ms_last;  /* sum of hipEvent-bracketed count-kernel launches of the last batch */
    double   kernel_ms_total;
    int32_t  algo_last;       /* kmc_algo actually used for the last batch */
    int32_t  launches_last;   /* count-kernel launches in the last batch */
    uint64_t n_slabs_skipped; /* oversize slabs seen by kmc_merge_slabs_device (valid after kmc_finalize) */
    uint64_t n_direct;        /* k-mers the WALK / STREAM kernels counted with one global atomic each because their
                                 LDS memo / partial table was full (valid after kmc_finalize / kmc_poll) */
    double   kernel_ms_lifetime;  /* like kernel_ms_total, but since kmc_create: kmc_reset does not clear it (a caller that */
    uint64_t launches_lifetime;   /* resets the ctx per step reads the kernel time of all steps once, after the last) */
    uint64_t n_async_ok;          /* finalizes of this ctx that produced their view through the small-table kernel, and the oversize */
    uint64_t n_async_slabs_skipped; /* slabs they saw, both since kmc_create (as of the last call that synchronised): see kmc_finalize_async */
    uint64_t n_planner_stale;     /* debug invariant of the launch planner: risky launches whose table snapshot was armed on
                                     counters older than the last queued unfold / merge (must stay 0) */
} kmc_stats;

const char* kmc_version(void);
const char* kmc_status_string(int status);

int  kmc_create(kmc_ctx** out, const kmc_config* cfg);
void kmc_destroy(kmc_ctx* ctx);
/* Message of the last failing call on this ctx (owned by the ctx; "" if none).  ctx may be NULL:
 * then the message of the last failing kmc_create on this thread. */
const char* kmc_last_error(const kmc_ctx* ctx);

/* Forget all counts (table kept allocated). */
int kmc_reset(kmc_ctx* ctx);

/* Count one batch of reads.  `bases`: ASCII bases of all reads concatenated (no newlines);
 * `offsets[n_reads+1]`: start of each read, offsets[0] == 0, offsets[n_reads] == total bases.
 * Host buffers; copied to the device before return. */
int kmc_add_batch(kmc_ctx* ctx, const uint8_t* bases, const uint64_t* offsets, uint64_t n_reads);

/* Same, for buffers already resident in this ctx's GPU memory (HBM-resident timing, pipelines
 * that parse into pinned/device memory).  d_bases must be 16-byte aligned and readable up to the
 * next 16-byte boundary past the last base.  max_read_len: longest read in the batch, or 0 if
 * unknown (then it is computed on the device).  Asynchronous on the ctx stream.  The buffers must stay
 * valid and unchanged until the next call on this ctx that synchronises (kmc_add_batch*, kmc_finalize,
 * kmc_poll, kmc_export): the batch may go out in one launch sized by what earlier batches looked like, and
 * if that prediction proves wrong (far more distinct k-mers than the table and its spill area hold) the
 * library puts the table back and counts the affected part of the batch again by sorting -- nothing is
 * dropped and no KMC_ERR_CAPACITY is raised. */
int kmc_add_batch_device(kmc_ctx* ctx, const void* d_bases, const void* d_offsets,
                         uint64_t n_reads, uint64_t n_bases, uint64_t max_read_len);

/* Merge (key,count) pairs that already live on this GPU into the table (multi-GPU reduce:
 * pairs received from a peer over RCCL).  d_key_hi may be NULL when k <= 32 / not LR. */
int kmc_merge_pairs_device(kmc_ctx* ctx, const void* d_key_hi, const void* d_key_lo,
                           const void* d_count, uint64_t n_pairs);

/* Compact the table, sort by key on the device, report sizes.  May be called repeatedly;
 * more batches may be added afterwards (the sorted view is then stale until the next finalize). */
int kmc_finalize(kmc_ctx* ctx, uint64_t* n_distinct, uint64_t* n_total);

/* kmc_finalize for a pipeline that does not want to wait: the device work of a SMALL table's finalize (at most 131072
 * keys, nothing spilled: every table of generator-style input) is queued on the ctx stream and the call returns.  Behind
 * it in stream order the sorted view is in place (the pointers kmc_export_device last returned stay valid for small
 * tables) and the table is empty; kmc_reset after it launches one small kernel and does not wait either.  The next call
 * that needs the outcome on the host (kmc_finalize, kmc_export*, kmc_add_batch*, ...) synchronises once and takes it from
 * there -- kmc_finalize then returns the sizes of the view this call produced.  A caller that queues many steps
 * (count -> kmc_finalize_async -> kmc_reset -> ...) checks afterwards that every step delivered:
 * kmc_stats.n_async_ok grows by one per finalize that produced its view, n_async_slabs_skipped by the oversize slabs
 * those finalizes saw (both valid after a synchronising call).  Larger tables are finalized synchronously, as by kmc_finalize. */
int kmc_finalize_async(kmc_ctx* ctx);

/* Copy the sorted table to caller-allocated host arrays of `cap` entries (cap >= n_distinct).
 * key_hi may be NULL if the caller knows k <= 32. */
int kmc_export(kmc_ctx* ctx, uint64_t* key_hi, uint64_t* key_lo, uint64_t* count, uint64_t cap);

/* Device pointers to the sorted table of the last kmc_finalize (owned by the ctx, valid until the
 * next finalize/reset/destroy).  d_key_hi is NULL when keys fit one word.
 * Ordering contract: when kmc_export_device returns, every kernel that writes the view has FINISHED, so the arrays may
 * be read from any stream, by a peer copy or by a collective without further synchronisation.  (kmc_finalize itself may
 * return earlier than that for a small table: its kernel tells the host through pinned memory that the view and the
 * counters are complete while it is still clearing table slots -- work queued on the ctx stream is ordered behind it
 * anyway, and this call waits for the kernel's end, once, before it hands out pointers.) */
int kmc_export_device(kmc_ctx* ctx, const void** d_key_hi, const void** d_key_lo,
                      const void** d_count, uint64_t* n_distinct);

/* Owner-partitioned view for an all-to-all exchange: after kmc_finalize, reorders the sorted
 * table so that pairs with owner(key) == p are contiguous, p = 0..n_parts-1, where
 * owner = mix(key) % n_parts (kmc_owner_of gives the same function on the host).
 * part_begin[n_parts+1] (host) receives the boundaries.  Pointers as in kmc_export_device. */
int kmc_partition_device(kmc_ctx* ctx, uint32_t n_parts, uint64_t* part_begin,
                         const void** d_key_hi, const void** d_key_lo, const void** d_count);
uint32_t kmc_owner_of(uint64_t key_hi, uint64_t key_lo, uint32_t n_parts);

/* Multi-GPU reduce for small tables: ONE fixed-size all-gather instead of size exchange +
 * all-to-all (the reduce of main.rs:87's grouping across GPUs; for the generator's input a table is
 * a few thousand keys, so the exchange is latency-bound and every host synchronisation counts).
 * A slab holds up to slab_entries (key,count) pairs behind an 8-word header; kmc_slab_words gives
 * its size in 64-bit words for this ctx's key width.  kmc_pack_slab_device writes this ctx's table
 * into d_slab -- the sorted view after kmc_finalize, otherwise straight from the live table
 * (unsorted; no finalize and no host synchronisation needed first) -- or marks the slab "oversize"
 * when the table has more than slab_entries keys (live table: also more than 1048576, the length of its list of claimed slots).  kmc_merge_slabs_device adds, from n_slabs consecutive slabs (the all-gather
 * result), every pair with kmc_owner_of(key, n_parts) == my_part; oversize slabs are skipped and
 * counted in kmc_stats.n_slabs_skipped at the next kmc_finalize (the caller then moves those
 * tables with kmc_partition_device + all-to-all + kmc_merge_pairs_device).  Both calls are
 * asynchronous on the ctx stream and never synchronise with the host. */
uint64_t kmc_slab_words(const kmc_ctx* ctx, uint64_t slab_entries);
int kmc_pack_slab_device(kmc_ctx* ctx, void* d_slab, uint64_t slab_entries);
int kmc_merge_slabs_device(kmc_ctx* ctx, const void* d_slabs, uint32_t n_slabs, uint64_t slab_entries,
                           uint32_t my_part, uint32_t n_parts);

/* Drop what the ctx has learned about its data source (the walk kernel's memo of the input's
 * de Bruijn graph structure and the launch planner's new-keys-per-k-mer history); kmc_reset keeps
 * both because later batches of the same source profit from them.  Counts are not affected. */
/* Wait for everything queued on the ctx and read its device counters: brings kmc_stats up to date
 * (n_kmers, kernel_ms_*) and lets the launch planner learn from the batch just counted -- what
 * kmc_finalize does on the way, for callers that reset a ctx without finalizing it (multi-GPU
 * reduce: the live table is packed and shipped, only the owner's table is finalized). */
int kmc_poll(kmc_ctx* ctx);

/* Wait until everything queued on the ctx's stream has finished (nothing else: no counters are read).
 * For callers that hand buffers written by ctx kernels to another stream or library (the RCCL
 * all-gather of a slab packed on a ctx-owned stream). */
int kmc_sync(kmc_ctx* ctx);

/* How KMC_ALGO_WALK cuts a read of read_len bases into pieces of at most 416 bases that overlap by
 * k-1 (every window of the read lies in exactly one piece): the number of pieces, and, for the
 * first `cap`, their [start, end) within the read.  Pure host arithmetic (no GPU needed). */
uint64_t kmc_read_pieces(uint64_t read_len, int k, uint64_t* starts, uint64_t* ends, uint64_t cap);

#define KMC_FORGET_MEMO 1     /* the walk kernel's memo snapshot */
#define KMC_FORGET_HISTORY 2  /* the launch planner's history (and the AUTO algorithm choice) */
int kmc_forget_source(kmc_ctx* ctx, int what);

int kmc_get_stats(const kmc_ctx* ctx, kmc_stats* out);

/* Convenience used by the CLI: parse `path` on the host (restating the reader the reference
 * uses, main.rs:45-46,59-62), feed batches, finalize.  Results via kmc_export. */
int kmc_count_file(kmc_ctx* ctx, const char* path, uint64_t* n_distinct, uint64_t* n_total);

/* The same on several GPUs of this process (the CLI's --gpus N): ctxs[0..n_ctx) are distinct
 * contexts with the same k / mode / canonical, normally one per GPU; chunks of the file go
 * round-robin to them and the tables are reduced into ctxs[0] (peer copies + kmc_merge_pairs_device),
 * which holds the result (kmc_export).  Scaling runs use one process per GPU and RCCL instead
 * (k-mer-count_amd/distributed.py). */
int kmc_count_file_multi(kmc_ctx** ctxs, uint32_t n_ctx, const char* path, uint64_t* n_distinct, uint64_t* n_total);

/* Host FASTA reader on its own (library-owned buffers; free with kmc_free_reads). */
typedef struct kmc_reads {
    uint8_t*  bases;
    uint64_t* offsets;
    uint64_t  n_reads;
    uint64_t  n_bases;
    uint64_t  max_read_len;
} kmc_reads;
int  kmc_parse_fasta(const char* path, kmc_reads* out, char* errbuf, size_t errbuf_len);
void kmc_free_reads(kmc_reads* r);

/* Streaming form of the reader: the file is handed out in chunks of about chunk_bytes of FASTA text
 * (0 = 256 MiB), each ending at a record boundary, parsed by worker threads.  out->bases/offsets
 * point into the stream's own buffers and stay valid until the next call on the stream (do NOT
 * pass them to kmc_free_reads).  *eof is set to 1 with the last chunk.  This is what a host
 * program (the Rust main() of INTEGRATION.md) feeds to kmc_add_batch chunk by chunk;
 * kmc_count_file uses the same reader internally and overlaps parsing with upload and counting.
 * Extension (not in the reference): a file whose first byte is '@' is read as
 * four-line FASTQ and only its sequence lines are handed on. */
typedef struct kmc_fasta_stream kmc_fasta_stream;
int  kmc_fasta_stream_open(const char* path, uint64_t chunk_bytes, kmc_fasta_stream** out, char* errbuf, size_t errbuf_len);
int  kmc_fasta_stream_next(kmc_fasta_stream* s, kmc_reads* out, int* eof, char* errbuf, size_t errbuf_len);
void kmc_fasta_stream_close(kmc_fasta_stream* s);

/* Decode a key into klen ASCII characters (no terminator). */
void kmc_decode_key(uint64_t key_hi, uint64_t key_lo, int klen, char* out);

/* ---- synthetic input: seeded, size-parameterised re-creation of the distribution of
 * random_fasta_generator.py:5-15 (pool of `pool` random lines of `line_len` bases; each record
 * = `lines_per_record` lines drawn uniformly from the pool; pool == 0: every line fresh random).
 * Counter-based PRNG, so any record range can be generated independently and identically on
 * host and device. ---- */
typedef struct kmc_synth {
    uint64_t seed;
    uint32_t pool;             /* 10 in the reference (:5) */
    uint32_t line_len;         /* 80 (:6) */
    uint32_t lines_per_record; /* 5 (:13) */
    uint32_t reserved;
} kmc_synth;

/* Number of records whose FASTA text (header ">dummy_sequence_NNN Nth record\n", :11-12, plus
 * lines) first reaches `file_bytes` bytes; also the exact byte size of that text. */
uint64_t kmc_synth_records_for_bytes(const kmc_synth* s, uint64_t file_bytes, uint64_t* exact_bytes);
/* Parsed form of records [first, first+n): bases (n*lines*line_len bytes) and offsets[n+1]. */
int kmc_synth_reads_host(const kmc_synth* s, uint64_t first_record, uint64_t n_records,
                         uint8_t* bases, uint64_t* offsets);
int kmc_synth_reads_device(const kmc_synth* s, uint64_t first_record, uint64_t n_records,
                           void* d_bases, void* d_offsets, int device, void* stream);
/* FASTA text of records [first, first+n) appended to `FILE_ptr` (a FILE*). */
int kmc_synth_write_fasta(const kmc_synth* s, uint64_t first_record, uint64_t n_records, void* FILE_ptr);

/* ---- measurement aid: the streaming-read rate this GPU actually reaches, so that a kernel's
 * fraction of the HBM roofline can be quoted against the measured peak beside the nominal 8 TB/s.  A plain
 * read-only kernel (non-temporal 16-byte loads of [d_buf, d_buf + n_bytes), xor-reduced; no product code) is
 * launched iters times after one warm-up, bracketed by one hipEvent pair on `stream`; *ms_avg = time per launch.
 * shape: 0 = the walk kernel's grid (one 1024-thread workgroup per CU), 1 = 8 x 256 threads per CU,
 * 2 = 2 x 1024 per CU, 3 = 4 x 512 per CU.  *xor_out (may be NULL) receives the checksum that keeps the loads alive. */
int kmc_read_peak_device(const void* d_buf, uint64_t n_bytes, int device, void* stream, int shape, int iters,
                         double* ms_avg, uint64_t* xor_out);

#ifdef __cplusplus
}
#endif
#endif /* KMC_H */
