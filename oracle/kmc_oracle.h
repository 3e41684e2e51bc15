/*
 * kmc_oracle.h -- CPU oracle for the k-mer counting hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and only
 * as the checker or as the reported CPU baseline -- never as the thing measured or shipped.
 *
 * It restates, in plain C, the algorithm of the reference (paths relative to /root/reference):
 *   - FASTA record reading        : bio 0.41.0 io::fasta::Reader::read, call sites
 *                                   k-mer-count/src/main.rs:45-46,59-62 (crate pinned in
 *                                   k-mer-count/Cargo.lock:36-38; the crate is NOT vendored,
 *                                   its published algorithm is restated in kmc_oracle.c)
 *   - LR-gapped window extraction : k-mer-count/src/main.rs:63-81 (== test.py:22-38)
 *   - grouping / ordering         : k-mer-count/src/main.rs:87-90 (== test.py:39-40),
 *                                   alphabet order A<C<G<T from main.rs:18-22,27-30
 *   - contiguous canonical k-mers : SURVEY.md section 8a-def (the reference has no k; the
 *                                   definition reuses the reference's record scope, alphabet
 *                                   and ordering conventions)
 *
 * Parity pin: the LR mode reproduces the sha256 digests of the reference's own test.py
 * output on k-mer-count/sample.fasta (goldens G-full, G-1, G-3, G-empty recorded in
 * SURVEY.md section 8c and committed as tests/golden/lr_goldens.json).  The contiguous-k
 * mode reuses the same reader, window arithmetic and ordering and is additionally pinned by
 * the known-answer table of SURVEY.md section 8c-KAT.
 */
#ifndef KMC_ORACLE_H
#define KMC_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* error codes (negative) */
#define KMO_OK 0
#define KMO_ERR_IO (-1)       /* cannot open/read file (main.rs:44 panics here) */
#define KMO_ERR_FORMAT (-2)   /* "Expected > at record start." (main.rs:59 unwrap panics) */
#define KMO_ERR_ALPHABET (-3) /* non-ACGT in LR mode (main.rs:23 panics) */
#define KMO_ERR_ARG (-4)
#define KMO_ERR_NOMEM (-5)

/* A parsed FASTA file: all record sequences concatenated (no newlines), offsets[n_reads+1]. */
typedef struct {
    uint8_t*  bases;
    uint64_t* offsets;
    uint64_t  n_reads;
    uint64_t  n_bases;
} kmo_reads;

/* A sorted count table.  Keys are MSB-first 2-bit packed (A=0,C=1,G=2,T=3); key_hi is all
 * zero when 2*klen <= 64.  Sorted ascending by (hi,lo) == byte-lexicographic string order. */
typedef struct {
    uint64_t* key_hi;
    uint64_t* key_lo;
    uint64_t* count;
    uint64_t  n_distinct;
    uint64_t  n_total;
    int       klen; /* characters per key: k, or 54 in LR mode */
} kmo_table;

int  kmo_parse_fasta(const char* path, kmo_reads* out);
void kmo_free_reads(kmo_reads* r);
void kmo_free_table(kmo_table* t);

/* method: 0 = collect every occurrence, sort, run-length (the reference's algorithm shape on
 * packed keys); 1 = open-addressing hash map then sort the distinct keys (B2 baseline). */
int kmo_count_kmers(const uint8_t* bases, const uint64_t* offsets, uint64_t n_reads, int k,
                    int canonical, int method, kmo_table* out);

/* Reference mode: 27 + gap + 27 for chunk sizes 80..=140 (main.rs:48-49,63). */
int kmo_count_lr(const uint8_t* bases, const uint64_t* offsets, uint64_t n_reads, kmo_table* out);

/* B1 baseline: the reference's literal algorithm shape applied to contiguous k -- materialise
 * every window as a heap string (main.rs:78-79), comparison-sort (main.rs:87), run-length.
 * Single-threaded like the reference.  Produces the same table as kmo_count_kmers. */
int kmo_count_kmers_strings(const uint8_t* bases, const uint64_t* offsets, uint64_t n_reads, int k,
                            int canonical, kmo_table* out);

/* Decode key -> ASCII (klen chars, no terminator). */
void kmo_decode_key(uint64_t hi, uint64_t lo, int klen, char* out);

/* Write the table: expand=0 -> "KMER\tCOUNT\n"; expand=1 -> KMER repeated COUNT times, one per
 * line (byte-identical to main.rs:88-90).  Returns bytes written or <0. */
int64_t kmo_write_table(const kmo_table* t, int expand, void* FILE_ptr);

const char* kmo_strerror(int code);

#ifdef __cplusplus
}
#endif
#endif
