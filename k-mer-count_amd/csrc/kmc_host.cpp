// kmc_host.cpp -- host-only parts of libkmc: the FASTA reader that stays on the host
// (north star: "FASTA parsing stays on the host"), key decoding, and the host half of the
// synthetic-input generator.  No GPU code here.
//
// Reader semantics mirror what the reference relies on, k-mer-count/src/main.rs:45-46,59-62
// (bio 0.41.0 io::fasta::Reader, pinned in k-mer-count/Cargo.lock:36-38) and test.py:7-11:
// a record starts at a line beginning with '>', its sequence is every following line up to the
// next '>' line with trailing whitespace removed and joined; a non-empty line that is not a
// header where a header is required is "Expected > at record start."; a header-less,
// sequence-less record means end of input (Record::is_empty(), main.rs:60).
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <new>
#include <string>
#include <vector>

#include "../../include/kmc.h"
#include "kmc_synth.cuh"

namespace {

inline bool is_space(unsigned char c) { return c == ' ' || (c >= 9 && c <= 13); }

void set_err(char* errbuf, size_t n, const char* msg) {
    if (errbuf && n) { snprintf(errbuf, n, "%s", msg); }
}

}  // namespace

extern "C" int kmc_parse_fasta(const char* path, kmc_reads* out, char* errbuf, size_t errbuf_len) {
    if (!out || !path) return KMC_ERR_ARG;
    memset(out, 0, sizeof(*out));
    FILE* f = fopen(path, "rb");
    if (!f) { set_err(errbuf, errbuf_len, "Error during opening the file"); return KMC_ERR_IO; }
    std::vector<uint8_t> seq;
    std::vector<uint64_t> offs;
    offs.push_back(0);
    try {
        // streaming line reader: the file is never held whole
        const size_t BUFSZ = 1u << 22;
        std::vector<char> buf(BUFSZ);
        std::string line;       // current (possibly partial) line
        bool in_record = false; // a header has been seen
        bool header_blank = false;
        uint64_t rec_start = 0;
        uint64_t maxlen = 0;
        bool stop = false;
        auto finish_record = [&]() {
            // Record::is_empty(): id "", no desc, no sequence -> treated as EOF by main.rs:60-62
            if (header_blank && seq.size() == rec_start) { stop = true; return; }
            offs.push_back(seq.size());
            uint64_t l = seq.size() - rec_start;
            if (l > maxlen) maxlen = l;
        };
        auto handle_line = [&](const char* s, size_t n) -> int {
            if (n && s[0] == '>') {
                if (in_record) finish_record();
                if (stop) return 0;
                size_t e = n;
                while (e > 1 && is_space((unsigned char)s[e - 1])) e--;
                header_blank = (e <= 1);
                in_record = true;
                rec_start = seq.size();
                return 0;
            }
            if (!in_record) return KMC_ERR_FORMAT;  // includes a blank first line, like bio
            size_t e = n;
            while (e > 0 && is_space((unsigned char)s[e - 1])) e--;
            seq.insert(seq.end(), (const uint8_t*)s, (const uint8_t*)s + e);
            return 0;
        };
        int rc = 0;
        size_t got;
        while (!stop && (got = fread(buf.data(), 1, BUFSZ, f)) > 0) {
            size_t pos = 0;
            while (pos < got && !stop) {
                const char* nl = (const char*)memchr(buf.data() + pos, '\n', got - pos);
                size_t end = nl ? (size_t)(nl - buf.data()) : got;
                if (!line.empty() || !nl) {
                    line.append(buf.data() + pos, end - pos);
                    if (nl) { rc = handle_line(line.data(), line.size()); line.clear(); }
                } else {
                    rc = handle_line(buf.data() + pos, end - pos);
                }
                if (rc) break;
                pos = nl ? end + 1 : got;
            }
            if (rc) break;
        }
        if (!rc && !stop && !line.empty()) rc = handle_line(line.data(), line.size());
        fclose(f);
        f = nullptr;
        if (rc) { set_err(errbuf, errbuf_len, "Expected > at record start."); return rc; }
        if (in_record && !stop) finish_record();
        out->n_reads = offs.size() - 1;
        out->n_bases = seq.size();
        out->max_read_len = maxlen;
        out->bases = (uint8_t*)malloc(seq.size() + 64);
        out->offsets = (uint64_t*)malloc(offs.size() * sizeof(uint64_t));
        if (!out->bases || !out->offsets) {
            free(out->bases); free(out->offsets);
            memset(out, 0, sizeof(*out));
            set_err(errbuf, errbuf_len, "out of memory");
            return KMC_ERR_NOMEM;
        }
        if (!seq.empty()) memcpy(out->bases, seq.data(), seq.size());
        memcpy(out->offsets, offs.data(), offs.size() * sizeof(uint64_t));
        return KMC_OK;
    } catch (const std::bad_alloc&) {
        if (f) fclose(f);
        memset(out, 0, sizeof(*out));
        set_err(errbuf, errbuf_len, "out of memory");
        return KMC_ERR_NOMEM;
    }
}

extern "C" void kmc_free_reads(kmc_reads* r) {
    if (!r) return;
    free(r->bases);
    free(r->offsets);
    memset(r, 0, sizeof(*r));
}

extern "C" void kmc_decode_key(uint64_t hi, uint64_t lo, int klen, char* out) {
    for (int i = klen - 1; i >= 0; --i) {
        out[i] = "ACGT"[lo & 3];
        lo = (lo >> 2) | (hi << 62);
        hi >>= 2;
    }
}

// ---- synthetic input (host half) -------------------------------------------------------------

static inline int ndigits(uint64_t v) { int d = 1; while (v >= 10) { v /= 10; d++; } return d; }

// ">dummy_sequence_" (16) + zfill(3) + " " + digits + "th record" (9) + "\n"
static inline uint64_t header_len(uint64_t i) { int d = ndigits(i); return 16 + (d < 3 ? 3 : d) + 1 + d + 9 + 1; }

extern "C" uint64_t kmc_synth_records_for_bytes(const kmc_synth* s, uint64_t file_bytes, uint64_t* exact_bytes) {
    if (!s) return 0;
    const uint64_t body = (uint64_t)s->lines_per_record * (s->line_len + 1);
    uint64_t n = 0, bytes = 0, lo = 1;
    while (bytes < file_bytes) {
        // records lo .. hi share one digit count
        int d = ndigits(lo);
        uint64_t hi = 1;
        for (int i = 0; i < d; ++i) hi *= 10;
        hi -= 1;  // last record with d digits
        uint64_t per = header_len(lo) + body;
        uint64_t avail = hi - lo + 1;
        uint64_t need = (file_bytes - bytes + per - 1) / per;
        uint64_t take = need < avail ? need : avail;
        n += take;
        bytes += take * per;
        lo += take;
    }
    if (exact_bytes) *exact_bytes = bytes;
    return n;
}

extern "C" int kmc_synth_reads_host(const kmc_synth* s, uint64_t first_record, uint64_t n_records,
                                    uint8_t* bases, uint64_t* offsets) {
    if (!s || !bases || !offsets || !s->line_len || !s->lines_per_record) return KMC_ERR_ARG;
    const uint64_t read_len = (uint64_t)s->lines_per_record * s->line_len;
    std::vector<uint8_t> pool((size_t)s->pool * s->line_len);
    for (uint32_t p = 0; p < s->pool; ++p)
        for (uint32_t x = 0; x < s->line_len; ++x) pool[(size_t)p * s->line_len + x] = kmc_synth_pool_base(s->seed, s->line_len, p, x);
    for (uint64_t r = 0; r < n_records; ++r) {
        offsets[r] = r * read_len;
        for (uint32_t j = 0; j < s->lines_per_record; ++j) {
            uint64_t gl = (first_record + r) * s->lines_per_record + j;
            uint8_t* dst = bases + r * read_len + (uint64_t)j * s->line_len;
            if (s->pool) {
                memcpy(dst, &pool[(size_t)kmc_synth_choice(s->seed, s->pool, gl) * s->line_len], s->line_len);
            } else {
                for (uint32_t x = 0; x < s->line_len; ++x) dst[x] = kmc_synth_fresh_base(s->seed, s->line_len, gl, x);
            }
        }
    }
    offsets[n_records] = n_records * read_len;
    return KMC_OK;
}

extern "C" int kmc_synth_write_fasta(const kmc_synth* s, uint64_t first_record, uint64_t n_records, void* FILE_ptr) {
    if (!s || !FILE_ptr || !s->line_len || !s->lines_per_record) return KMC_ERR_ARG;
    FILE* f = (FILE*)FILE_ptr;
    const uint64_t read_len = (uint64_t)s->lines_per_record * s->line_len;
    std::vector<uint8_t> bases(read_len);
    uint64_t offs[2];
    std::string rec;
    for (uint64_t r = 0; r < n_records; ++r) {
        uint64_t i = first_record + r + 1;  // 1-based like range(1, 201), generator :10
        kmc_synth_reads_host(s, first_record + r, 1, bases.data(), offs);
        char hdr[96];
        int hl = snprintf(hdr, sizeof(hdr), ">dummy_sequence_%03llu %lluth record\n", (unsigned long long)i, (unsigned long long)i);
        rec.assign(hdr, (size_t)hl);
        for (uint32_t j = 0; j < s->lines_per_record; ++j) {
            rec.append((const char*)&bases[(size_t)j * s->line_len], s->line_len);
            rec.push_back('\n');
        }
        if (fwrite(rec.data(), 1, rec.size(), f) != rec.size()) return KMC_ERR_IO;
    }
    return KMC_OK;
}
