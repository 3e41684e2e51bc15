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

#include <algorithm>
#include <new>
#include <string>
#include <system_error>
#include <thread>
#include <vector>

#include "../../include/kmc.h"
#include "kmc_synth.cuh"

namespace {

inline bool is_space(unsigned char c) { return c == ' ' || (c >= 9 && c <= 13); }

void set_err(char* errbuf, size_t n, const char* msg) {
    if (errbuf && n) { snprintf(errbuf, n, "%s", msg); }
}

}  // namespace

// Multi-threaded reader.  The file is cut into byte segments; every worker snaps its segment to
// line starts, then makes one memchr pass over it: a line starting with '>' opens a record, any
// other line is trimmed on the right and appended to the worker's local sequence buffer.  The
// workers' results are stitched in order (sequence bytes before a worker's first header belong to
// the last record of an earlier worker), which gives exactly the sequential reader's result.
namespace {

struct SegOut {
    std::vector<uint8_t> seq;        // sequence bytes of this segment, in order
    std::vector<uint64_t> rec_start; // local offset in `seq` where each record of this segment starts
    std::vector<uint8_t> rec_blank;  // header was empty (id "", no desc)
    bool bad_first_line = false;     // segment 0 only: a non-header line before any header
    int err = 0;
};

void parse_segment(const char* path, uint64_t begin, uint64_t end, uint64_t fsize, bool first, SegOut* out) {
    FILE* f = fopen(path, "rb");
    if (!f) { out->err = KMC_ERR_IO; return; }
    // snap `begin` to the first line start >= begin (unless it is the file start)
    const size_t BUF = 1u << 22;
    std::vector<char> buf(BUF);
    uint64_t pos = begin;
    if (begin > 0) {
        // the line containing byte begin-1 belongs to the previous segment: skip to after its '\n'
        fseeko(f, (off_t)(begin - 1), SEEK_SET);
        pos = begin - 1;
        bool found = false;
        while (!found && pos < fsize) {
            size_t got = fread(buf.data(), 1, BUF, f);
            if (!got) break;
            const char* nl = (const char*)memchr(buf.data(), '\n', got);
            if (nl) { pos += (uint64_t)(nl - buf.data()) + 1; found = true; } else pos += got;
        }
        if (!found) { fclose(f); return; }  // no line starts in this segment
    }
    if (pos >= end && !(first && fsize == 0)) { if (pos >= end) { fclose(f); return; } }
    fseeko(f, (off_t)pos, SEEK_SET);
    out->seq.reserve((size_t)(end > pos ? end - pos : 0) + 64);
    std::string carry;  // a line cut by the read buffer
    bool in_record = false;
    bool stop = false;
    auto handle_line = [&](const char* s, size_t n, uint64_t line_start) {
        if (line_start >= end) { stop = true; return; }  // first line of the next segment
        if (n && s[0] == '>') {
            size_t e = n;
            while (e > 1 && is_space((unsigned char)s[e - 1])) e--;
            out->rec_start.push_back(out->seq.size());
            out->rec_blank.push_back(e <= 1);
            in_record = true;
            return;
        }
        if (first && !in_record) { out->bad_first_line = true; stop = true; return; }
        size_t e = n;
        while (e > 0 && is_space((unsigned char)s[e - 1])) e--;
        out->seq.insert(out->seq.end(), (const uint8_t*)s, (const uint8_t*)s + e);
    };
    uint64_t line_start = pos;
    while (!stop) {
        size_t got = fread(buf.data(), 1, BUF, f);
        if (!got) break;
        size_t p = 0;
        while (p < got && !stop) {
            const char* nl = (const char*)memchr(buf.data() + p, '\n', got - p);
            size_t e = nl ? (size_t)(nl - buf.data()) : got;
            if (!carry.empty() || !nl) {
                carry.append(buf.data() + p, e - p);
                if (nl) { handle_line(carry.data(), carry.size(), line_start); line_start += carry.size() + 1; carry.clear(); }
            } else {
                handle_line(buf.data() + p, e - p, line_start);
                line_start += (e - p) + 1;
            }
            p = nl ? e + 1 : got;
        }
    }
    if (!stop && !carry.empty()) handle_line(carry.data(), carry.size(), line_start);  // last line without '\n'
    fclose(f);
}

}  // namespace

extern "C" int kmc_parse_fasta(const char* path, kmc_reads* out, char* errbuf, size_t errbuf_len) {
    if (!out || !path) return KMC_ERR_ARG;
    memset(out, 0, sizeof(*out));
    FILE* f = fopen(path, "rb");
    if (!f) { set_err(errbuf, errbuf_len, "Error during opening the file"); return KMC_ERR_IO; }
    fseeko(f, 0, SEEK_END);
    const uint64_t fsize = (uint64_t)ftello(f);
    fclose(f);
    try {
        unsigned hw = std::thread::hardware_concurrency();
        uint64_t nseg = std::min<uint64_t>(std::max(1u, std::min(hw, 32u)), std::max<uint64_t>(1, fsize >> 24));  // >= 16 MiB each
        std::vector<SegOut> segs((size_t)nseg);
        std::vector<std::thread> th;
        for (uint64_t i = 0; i < nseg; ++i) {
            uint64_t b = fsize * i / nseg, e = fsize * (i + 1) / nseg;
            th.emplace_back(parse_segment, path, b, e, fsize, i == 0, &segs[(size_t)i]);
        }
        for (auto& t : th) t.join();
        for (auto& sg : segs) if (sg.err) { set_err(errbuf, errbuf_len, "Error during opening the file"); return sg.err; }
        if (segs[0].bad_first_line) { set_err(errbuf, errbuf_len, "Expected > at record start."); return KMC_ERR_FORMAT; }
        // stitch
        uint64_t total = 0, nrec = 0;
        for (auto& sg : segs) { total += sg.seq.size(); nrec += sg.rec_start.size(); }
        out->bases = (uint8_t*)malloc((size_t)total + 64);
        out->offsets = (uint64_t*)malloc((size_t)(nrec + 1) * sizeof(uint64_t));
        if (!out->bases || !out->offsets) {
            free(out->bases); free(out->offsets);
            memset(out, 0, sizeof(*out));
            set_err(errbuf, errbuf_len, "out of memory");
            return KMC_ERR_NOMEM;
        }
        std::vector<uint8_t> blank((size_t)nrec);
        uint64_t base = 0, r = 0;
        std::vector<std::thread> cp;
        for (auto& sg : segs) {
            for (size_t j = 0; j < sg.rec_start.size(); ++j) { out->offsets[r] = base + sg.rec_start[j]; blank[(size_t)r] = sg.rec_blank[j]; r++; }
            if (!sg.seq.empty()) cp.emplace_back([dst = out->bases + base, &sg]() { memcpy(dst, sg.seq.data(), sg.seq.size()); });
            base += sg.seq.size();
        }
        for (auto& t : cp) t.join();
        out->offsets[nrec] = total;
        // Record::is_empty(): a record with empty header and no sequence ends the input (main.rs:60-62)
        uint64_t keep = nrec;
        for (uint64_t i = 0; i < nrec; ++i)
            if (blank[(size_t)i] && out->offsets[i + 1] == out->offsets[i]) { keep = i; break; }
        if (keep < nrec) { total = out->offsets[keep]; nrec = keep; }
        uint64_t maxlen = 0;
        for (uint64_t i = 0; i < nrec; ++i) maxlen = std::max<uint64_t>(maxlen, out->offsets[i + 1] - out->offsets[i]);
        out->n_reads = nrec;
        out->n_bases = total;
        out->max_read_len = maxlen;
        return KMC_OK;
    } catch (const std::bad_alloc&) {
        free(out->bases); free(out->offsets);
        memset(out, 0, sizeof(*out));
        set_err(errbuf, errbuf_len, "out of memory");
        return KMC_ERR_NOMEM;
    } catch (const std::system_error&) {
        free(out->bases); free(out->offsets);
        memset(out, 0, sizeof(*out));
        set_err(errbuf, errbuf_len, "cannot start parser threads");
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
