/*
 * kmc_oracle.c -- CPU oracle (TEST INFRASTRUCTURE ONLY; see kmc_oracle.h).
 *
 * Every function cites the reference lines it follows (paths relative to /root/reference).
 * Parity: LR mode is pinned against the reference's own test.py output digests
 * (tests/golden/lr_goldens.json); see tests/test_oracle_goldens.py.
 */
#define _GNU_SOURCE
#include "kmc_oracle.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 u128;

const char* kmo_strerror(int code) {
    switch (code) {
        case KMO_OK: return "ok";
        case KMO_ERR_IO: return "Error during opening the file";          /* main.rs:44 */
        case KMO_ERR_FORMAT: return "Expected > at record start.";         /* bio 0.41 fasta.rs */
        case KMO_ERR_ALPHABET: return "Unexpected charactor appears";      /* main.rs:23 */
        case KMO_ERR_ARG: return "bad argument";
        case KMO_ERR_NOMEM: return "out of memory";
        default: return "unknown error";
    }
}

/* Alphabet order A<C<G<T: the bucket order of main.rs:19-22 / concatenation order :27-30,
 * which is also byte order ('A'<'C'<'G'<'T'), so packed-key order == main.rs:87 sort order. */
static inline int base_code(uint8_t c) {
    switch (c) {
        case 'A': return 0;
        case 'C': return 1;
        case 'G': return 2;
        case 'T': return 3;
        default: return -1; /* anything else, lower case included */
    }
}

/* ------------------------------------------------------------------------------------------
 * FASTA reading.  Restates bio 0.41.0 io::fasta::Reader::read as used at main.rs:45-46,59-62:
 *   - if no line is buffered, read one; EOF -> the record stays empty (main.rs:60 breaks);
 *   - the buffered line must start with '>' else Err("Expected > at record start.");
 *   - header = line[1..] trim_end, split once on whitespace into id / desc;
 *   - following lines, until EOF or a line starting with '>', are trim_end()ed and appended;
 *   - Record::is_empty() <=> id empty && desc none && seq empty, which main.rs:60 treats as EOF.
 * trim_end strips Unicode White_Space; for ASCII input that is " \t\n\v\f\r" (bytes 0x09-0x0d,
 * 0x20).  test.py:9-10 (Biopython) joins stripped lines the same way for well-formed input.
 * ---------------------------------------------------------------------------------------- */
static inline int is_ws(uint8_t c) { return c == ' ' || (c >= 0x09 && c <= 0x0d); }

int kmo_parse_fasta(const char* path, kmo_reads* out) {
    memset(out, 0, sizeof(*out));
    FILE* f = fopen(path, "rb");
    if (!f) return KMO_ERR_IO;
    fseek(f, 0, SEEK_END);
    long fsz = ftell(f);
    fseek(f, 0, SEEK_SET);
    if (fsz < 0) { fclose(f); return KMO_ERR_IO; }
    uint8_t* buf = (uint8_t*)malloc((size_t)fsz + 1);
    if (!buf) { fclose(f); return KMO_ERR_NOMEM; }
    if (fsz > 0 && fread(buf, 1, (size_t)fsz, f) != (size_t)fsz) { free(buf); fclose(f); return KMO_ERR_IO; }
    fclose(f);

    size_t cap_reads = 1024;
    uint8_t* bases = (uint8_t*)malloc((size_t)fsz + 1);
    uint64_t* offsets = (uint64_t*)malloc((cap_reads + 1) * sizeof(uint64_t));
    if (!bases || !offsets) { free(buf); free(bases); free(offsets); return KMO_ERR_NOMEM; }
    uint64_t nb = 0, nr = 0;
    offsets[0] = 0;

    size_t pos = 0, n = (size_t)fsz;
    while (pos < n) {
        /* one line = up to and including '\n' (read_line semantics) */
        size_t ls = pos;
        size_t le = ls;
        while (le < n && buf[le] != '\n') le++;
        size_t next = (le < n) ? le + 1 : le;
        if (buf[ls] != '>') { free(buf); free(bases); free(offsets); return KMO_ERR_FORMAT; }
        /* header: id/desc only matter for is_empty() */
        size_t he = le;
        while (he > ls + 1 && is_ws(buf[he - 1])) he--;
        int header_empty = (he <= ls + 1); /* id "" and no desc */
        pos = next;
        uint64_t seq_start = nb;
        while (pos < n && buf[pos] != '>') {
            size_t s = pos, e = pos;
            while (e < n && buf[e] != '\n') e++;
            size_t nx = (e < n) ? e + 1 : e;
            size_t te = e;
            while (te > s && is_ws(buf[te - 1])) te--;
            memcpy(bases + nb, buf + s, te - s);
            nb += te - s;
            pos = nx;
        }
        if (header_empty && nb == seq_start) break; /* Record::is_empty() -> main.rs:60-62 */
        if (nr + 1 > cap_reads) {
            cap_reads *= 2;
            uint64_t* no = (uint64_t*)realloc(offsets, (cap_reads + 1) * sizeof(uint64_t));
            if (!no) { free(buf); free(bases); free(offsets); return KMO_ERR_NOMEM; }
            offsets = no;
        }
        nr++;
        offsets[nr] = nb;
    }
    free(buf);
    out->bases = bases;
    out->offsets = offsets;
    out->n_reads = nr;
    out->n_bases = nb;
    return KMO_OK;
}

void kmo_free_reads(kmo_reads* r) {
    if (!r) return;
    free(r->bases);
    free(r->offsets);
    memset(r, 0, sizeof(*r));
}

void kmo_free_table(kmo_table* t) {
    if (!t) return;
    free(t->key_hi);
    free(t->key_lo);
    free(t->count);
    memset(t, 0, sizeof(*t));
}

/* ------------------------------------------------------------------------------------------
 * key helpers
 * ---------------------------------------------------------------------------------------- */
static inline u128 mask_bits(int nbits) { return nbits >= 128 ? ~(u128)0 : (((u128)1 << nbits) - 1); }

/* reverse complement of a klen-base key (SURVEY 8a-def "Canonical"): complement = 3-code */
static inline u128 revcomp(u128 x, int klen) {
    u128 r = 0;
    for (int i = 0; i < klen; i++) {
        r = (r << 2) | (3 - (x & 3));
        x >>= 2;
    }
    return r;
}

void kmo_decode_key(uint64_t hi, uint64_t lo, int klen, char* out) {
    u128 x = ((u128)hi << 64) | lo;
    for (int i = klen - 1; i >= 0; i--) {
        out[i] = "ACGT"[(int)(x & 3)];
        x >>= 2;
    }
}

/* growable array of u128 */
typedef struct { u128* v; size_t n, cap; } vec128;
static int vpush(vec128* a, u128 x) {
    if (a->n == a->cap) {
        size_t nc = a->cap ? a->cap * 2 : (1u << 16);
        u128* nv = (u128*)realloc(a->v, nc * sizeof(u128));
        if (!nv) return -1;
        a->v = nv; a->cap = nc;
    }
    a->v[a->n++] = x;
    return 0;
}

static int cmp128(const void* a, const void* b) {
    u128 x = *(const u128*)a, y = *(const u128*)b;
    return (x > y) - (x < y);
}

/* LSD radix sort of u128 restricted to `nbits` significant bits, 8 bits per pass: the packed
 * analogue of main.rs:9-40 (one stable counting pass per digit, least significant first). */
static int radix_sort128(u128* v, size_t n, int nbits) {
    if (n < 2) return 0;
    u128* tmp = (u128*)malloc(n * sizeof(u128));
    if (!tmp) { qsort(v, n, sizeof(u128), cmp128); return 0; }
    u128 *src = v, *dst = tmp;
    for (int shift = 0; shift < nbits; shift += 8) {
        size_t cnt[257] = {0};
        for (size_t i = 0; i < n; i++) cnt[((unsigned)(src[i] >> shift) & 0xff) + 1]++;
        for (int d = 0; d < 256; d++) cnt[d + 1] += cnt[d];
        for (size_t i = 0; i < n; i++) dst[cnt[(unsigned)(src[i] >> shift) & 0xff]++] = src[i];
        u128* t = src; src = dst; dst = t;
    }
    if (src != v) memcpy(v, src, n * sizeof(u128));
    free(tmp);
    return 0;
}

/* sorted occurrences -> run-length table (what `uniq -c` does to main.rs:88-90 output) */
static int rle_to_table(const u128* v, size_t n, int klen, kmo_table* out) {
    size_t nd = 0;
    for (size_t i = 0; i < n; i++) if (i == 0 || v[i] != v[i - 1]) nd++;
    out->key_hi = (uint64_t*)malloc((nd + 1) * sizeof(uint64_t));
    out->key_lo = (uint64_t*)malloc((nd + 1) * sizeof(uint64_t));
    out->count = (uint64_t*)malloc((nd + 1) * sizeof(uint64_t));
    if (!out->key_hi || !out->key_lo || !out->count) return KMO_ERR_NOMEM;
    size_t j = 0;
    for (size_t i = 0; i < n;) {
        size_t e = i + 1;
        while (e < n && v[e] == v[i]) e++;
        out->key_hi[j] = (uint64_t)(v[i] >> 64);
        out->key_lo[j] = (uint64_t)v[i];
        out->count[j] = e - i;
        j++;
        i = e;
    }
    out->n_distinct = nd;
    out->n_total = n;
    out->klen = klen;
    return KMO_OK;
}

/* ------------------------------------------------------------------------------------------
 * open-addressing hash map u128 -> u64 (method 1 / B2 baseline)
 * ---------------------------------------------------------------------------------------- */
typedef struct { u128* keys; uint64_t* vals; size_t cap, n; } hmap;
#define HM_EMPTY (~(u128)0)
static inline uint64_t mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static int hm_init(hmap* h, size_t cap) {
    h->cap = cap; h->n = 0;
    h->keys = (u128*)malloc(cap * sizeof(u128));
    h->vals = (uint64_t*)calloc(cap, sizeof(uint64_t));
    if (!h->keys || !h->vals) return -1;
    for (size_t i = 0; i < cap; i++) h->keys[i] = HM_EMPTY;
    return 0;
}
static int hm_add(hmap* h, u128 k, uint64_t c);
static int hm_grow(hmap* h) {
    hmap nh;
    if (hm_init(&nh, h->cap * 2)) return -1;
    for (size_t i = 0; i < h->cap; i++) if (h->keys[i] != HM_EMPTY) hm_add(&nh, h->keys[i], h->vals[i]);
    free(h->keys); free(h->vals);
    *h = nh;
    return 0;
}
static int hm_add(hmap* h, u128 k, uint64_t c) {
    if ((h->n + 1) * 10 > h->cap * 7) if (hm_grow(h)) return -1;
    size_t m = h->cap - 1;
    size_t i = (size_t)mix64((uint64_t)k ^ mix64((uint64_t)(k >> 64))) & m;
    for (;;) {
        if (h->keys[i] == k) { h->vals[i] += c; return 0; }
        if (h->keys[i] == HM_EMPTY) { h->keys[i] = k; h->vals[i] = c; h->n++; return 0; }
        i = (i + 1) & m;
    }
}

typedef struct { u128 k; uint64_t c; } kc_pair;
static int cmp_pair(const void* a, const void* b) {
    u128 x = ((const kc_pair*)a)->k, y = ((const kc_pair*)b)->k;
    return (x > y) - (x < y);
}

static int hm_to_table(hmap* h, int klen, uint64_t total, kmo_table* out) {
    kc_pair* p = (kc_pair*)malloc((h->n + 1) * sizeof(kc_pair));
    if (!p) return KMO_ERR_NOMEM;
    size_t j = 0;
    for (size_t i = 0; i < h->cap; i++) if (h->keys[i] != HM_EMPTY) { p[j].k = h->keys[i]; p[j].c = h->vals[i]; j++; }
    qsort(p, j, sizeof(kc_pair), cmp_pair);
    out->key_hi = (uint64_t*)malloc((j + 1) * sizeof(uint64_t));
    out->key_lo = (uint64_t*)malloc((j + 1) * sizeof(uint64_t));
    out->count = (uint64_t*)malloc((j + 1) * sizeof(uint64_t));
    if (!out->key_hi || !out->key_lo || !out->count) { free(p); return KMO_ERR_NOMEM; }
    for (size_t i = 0; i < j; i++) {
        out->key_hi[i] = (uint64_t)(p[i].k >> 64);
        out->key_lo[i] = (uint64_t)p[i].k;
        out->count[i] = p[i].c;
    }
    free(p);
    out->n_distinct = j;
    out->n_total = total;
    out->klen = klen;
    return KMO_OK;
}

/* ------------------------------------------------------------------------------------------
 * Contiguous k-mers.  Record scope and stride follow main.rs:58-75: one
 * record at a time, window start advances by one (main.rs:71), a window never leaves the
 * record (main.rs:73-75), a record shorter than the window contributes nothing.  A byte
 * outside ACGT restarts the window after it.  canonical = min(fwd, revcomp) numerically.
 * ---------------------------------------------------------------------------------------- */
int kmo_count_kmers(const uint8_t* bases, const uint64_t* offsets, uint64_t n_reads, int k,
                    int canonical, int method, kmo_table* out) {
    memset(out, 0, sizeof(*out));
    if (k < 1 || k > 64 || (method != 0 && method != 1)) return KMO_ERR_ARG;
    const u128 mask = mask_bits(2 * k);
    vec128 occ = {0};
    hmap hm = {0};
    if (method == 1 && hm_init(&hm, 1u << 12)) return KMO_ERR_NOMEM;
    uint64_t total = 0;
    for (uint64_t r = 0; r < n_reads; r++) {
        u128 fwd = 0, rc = 0;
        int run = 0; /* valid bases ending here, within this record */
        for (uint64_t p = offsets[r]; p < offsets[r + 1]; p++) {
            int c = base_code(bases[p]);
            if (c < 0) { run = 0; fwd = 0; rc = 0; continue; }
            fwd = ((fwd << 2) | (u128)c) & mask;
            rc = (rc >> 2) | ((u128)(3 - c) << (2 * (k - 1)));
            if (++run >= k) {
                u128 key = (canonical && rc < fwd) ? rc : fwd;
                total++;
                if (method == 0) { if (vpush(&occ, key)) { free(occ.v); return KMO_ERR_NOMEM; } }
                else if (hm_add(&hm, key, 1)) return KMO_ERR_NOMEM;
            }
        }
    }
    int rcode;
    if (method == 0) {
        radix_sort128(occ.v, occ.n, 2 * k);
        rcode = rle_to_table(occ.v, occ.n, k, out);
        free(occ.v);
    } else {
        rcode = hm_to_table(&hm, k, total, out);
        free(hm.keys); free(hm.vals);
    }
    return rcode;
}

/* ------------------------------------------------------------------------------------------
 * Reference mode, main.rs:58-90.  For every record, every dna_chunk_size in 80..141
 * (main.rs:63), every window_start from 0 (main.rs:64) while r_end <= seq.len()
 * (main.rs:73-75): l = seq[l_start..l_end], r = seq[r_start..r_end] with l_len = r_len = 27
 * (main.rs:48-49,66-70), emit l+r (main.rs:78-79).  Then sort (main.rs:87).  A character
 * outside ACGT in an emitted chunk aborts (main.rs:23, reached through radix_sort at :84).
 * An empty chunk list panics in the reference (main.rs:35: source[0]); test.py:40 prints a
 * lone "\n".  Here: empty table, KMO_OK.
 * ---------------------------------------------------------------------------------------- */
int kmo_count_lr(const uint8_t* bases, const uint64_t* offsets, uint64_t n_reads, kmo_table* out) {
    memset(out, 0, sizeof(*out));
    const int l_len = 27, r_len = 27;
    vec128 occ = {0};
    for (uint64_t rd = 0; rd < n_reads; rd++) {
        const uint8_t* seq = bases + offsets[rd];
        uint64_t len = offsets[rd + 1] - offsets[rd];
        for (int dna_chunk_size = 80; dna_chunk_size < 141; dna_chunk_size++) {
            uint64_t window_start = 0;
            for (;;) {
                uint64_t m_len = (uint64_t)(dna_chunk_size - l_len - r_len);
                uint64_t l_start = window_start;
                uint64_t l_end = l_start + l_len;
                uint64_t r_start = l_end + m_len;
                uint64_t r_end = r_start + r_len;
                window_start += 1;
                if (r_end > len) break;
                u128 key = 0;
                for (uint64_t i = l_start; i < l_end; i++) {
                    int c = base_code(seq[i]);
                    if (c < 0) { free(occ.v); return KMO_ERR_ALPHABET; }
                    key = (key << 2) | (u128)c;
                }
                for (uint64_t i = r_start; i < r_end; i++) {
                    int c = base_code(seq[i]);
                    if (c < 0) { free(occ.v); return KMO_ERR_ALPHABET; }
                    key = (key << 2) | (u128)c;
                }
                if (vpush(&occ, key)) { free(occ.v); return KMO_ERR_NOMEM; }
            }
        }
    }
    radix_sort128(occ.v, occ.n, 2 * (l_len + r_len));
    int rcode = rle_to_table(occ.v, occ.n, l_len + r_len, out);
    free(occ.v);
    return rcode;
}

/* ------------------------------------------------------------------------------------------
 * B1 baseline: literal algorithm shape of the reference on contiguous k -- one heap string per
 * window (main.rs:78-79 `to_owned() + ...; push`), Vec<String>::sort (main.rs:87, a stable
 * comparison sort of byte strings; here qsort+memcmp, order-equivalent), then run-length.
 * ---------------------------------------------------------------------------------------- */
static int g_cmp_len;
static int cmp_strptr(const void* a, const void* b) {
    return memcmp(*(char* const*)a, *(char* const*)b, (size_t)g_cmp_len);
}

int kmo_count_kmers_strings(const uint8_t* bases, const uint64_t* offsets, uint64_t n_reads, int k,
                            int canonical, kmo_table* out) {
    memset(out, 0, sizeof(*out));
    if (k < 1 || k > 64) return KMO_ERR_ARG;
    size_t cap = 1u << 16, n = 0;
    char** v = (char**)malloc(cap * sizeof(char*));
    if (!v) return KMO_ERR_NOMEM;
    char rcbuf[64];
    for (uint64_t r = 0; r < n_reads; r++) {
        const uint8_t* seq = bases + offsets[r];
        uint64_t len = offsets[r + 1] - offsets[r];
        for (uint64_t ws = 0; ws + (uint64_t)k <= len; ws++) {
            int ok = 1;
            for (int i = 0; i < k; i++) if (base_code(seq[ws + i]) < 0) { ok = 0; break; }
            if (!ok) continue;
            char* s = (char*)malloc((size_t)k);
            if (!s) return KMO_ERR_NOMEM;
            memcpy(s, seq + ws, (size_t)k);
            if (canonical) {
                for (int i = 0; i < k; i++) rcbuf[i] = "TGCA"[base_code((uint8_t)s[k - 1 - i])];
                if (memcmp(rcbuf, s, (size_t)k) < 0) memcpy(s, rcbuf, (size_t)k);
            }
            if (n == cap) {
                cap *= 2;
                char** nv = (char**)realloc(v, cap * sizeof(char*));
                if (!nv) return KMO_ERR_NOMEM;
                v = nv;
            }
            v[n++] = s;
        }
    }
    g_cmp_len = k;
    qsort(v, n, sizeof(char*), cmp_strptr);
    size_t nd = 0;
    for (size_t i = 0; i < n; i++) if (i == 0 || memcmp(v[i], v[i - 1], (size_t)k)) nd++;
    out->key_hi = (uint64_t*)malloc((nd + 1) * sizeof(uint64_t));
    out->key_lo = (uint64_t*)malloc((nd + 1) * sizeof(uint64_t));
    out->count = (uint64_t*)malloc((nd + 1) * sizeof(uint64_t));
    if (!out->key_hi || !out->key_lo || !out->count) return KMO_ERR_NOMEM;
    size_t j = 0;
    for (size_t i = 0; i < n;) {
        size_t e = i + 1;
        while (e < n && !memcmp(v[e], v[i], (size_t)k)) e++;
        u128 key = 0;
        for (int c = 0; c < k; c++) key = (key << 2) | (u128)base_code((uint8_t)v[i][c]);
        out->key_hi[j] = (uint64_t)(key >> 64);
        out->key_lo[j] = (uint64_t)key;
        out->count[j] = e - i;
        j++;
        i = e;
    }
    for (size_t i = 0; i < n; i++) free(v[i]);
    free(v);
    out->n_distinct = nd;
    out->n_total = n;
    out->klen = k;
    return KMO_OK;
}

/* Output, main.rs:88-90: one line per occurrence (expand) or "KMER\tCOUNT\n". */
int64_t kmo_write_table(const kmo_table* t, int expand, void* FILE_ptr) {
    FILE* f = (FILE*)FILE_ptr;
    char line[80];
    int64_t written = 0;
    for (uint64_t i = 0; i < t->n_distinct; i++) {
        kmo_decode_key(t->key_hi[i], t->key_lo[i], t->klen, line);
        if (expand) {
            line[t->klen] = '\n';
            for (uint64_t c = 0; c < t->count[i]; c++) {
                if (fwrite(line, 1, (size_t)t->klen + 1, f) != (size_t)t->klen + 1) return -1;
                written += t->klen + 1;
            }
        } else {
            int m = snprintf(line + t->klen, sizeof(line) - (size_t)t->klen, "\t%llu\n", (unsigned long long)t->count[i]);
            if (fwrite(line, 1, (size_t)(t->klen + m), f) != (size_t)(t->klen + m)) return -1;
            written += t->klen + m;
        }
    }
    return written;
}
