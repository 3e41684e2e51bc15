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
#include "kmc_synth.hip.h"

namespace {

inline bool is_space(unsigned char c) { return c == ' ' || (c >= 9 && c <= 13); }

void set_err(char* errbuf, size_t n, const char* msg) {
    if (errbuf && n) { snprintf(errbuf, n, "%s", msg); }
}

// Worker threads that are joined on EVERY way out of a scope: if starting the i-th thread throws
// (std::system_error: no resources), unwinding must not destroy joinable threads -- that would be
// std::terminate, i.e. an abort across the C ABI (kmc.h promises a status code instead).
struct ThreadGroup {
    std::vector<std::thread> th;
    explicit ThreadGroup(size_t n) { th.reserve(n); }  // (so that emplace_back never reallocates)
    template <typename... A> void start(A&&... a) { th.emplace_back(std::forward<A>(a)...); }
    void join() { for (auto& t : th) if (t.joinable()) t.join(); }
    ~ThreadGroup() { join(); }
};

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

void parse_segment_body(const char* path, uint64_t begin, uint64_t end, uint64_t fsize, bool first, SegOut* out);
// thread entry: an exception leaving a thread function is std::terminate
void parse_segment(const char* path, uint64_t begin, uint64_t end, uint64_t fsize, bool first, SegOut* out) {
    try { parse_segment_body(path, begin, end, fsize, first, out); } catch (...) { out->err = KMC_ERR_NOMEM; }
}
void parse_segment_body(const char* path, uint64_t begin, uint64_t end, uint64_t fsize, bool first, SegOut* out) {
    struct FileCloser { FILE* f; ~FileCloser() { if (f) fclose(f); } };
    FILE* f = fopen(path, "rb");
    if (!f) { out->err = KMC_ERR_IO; return; }
    FileCloser closer{f};
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
        if (!found) return;  // no line starts in this segment
    }
    if (pos >= end) return;
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
        {
            ThreadGroup th((size_t)nseg);
            for (uint64_t i = 0; i < nseg; ++i) {
                uint64_t b = fsize * i / nseg, e = fsize * (i + 1) / nseg;
                th.start(parse_segment, path, b, e, fsize, i == 0, &segs[(size_t)i]);
            }
        }
        for (auto& sg : segs) if (sg.err) { set_err(errbuf, errbuf_len, sg.err == KMC_ERR_NOMEM ? "out of memory" : "Error during opening the file"); return sg.err; }
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
        {
            ThreadGroup cp(segs.size());
            for (auto& sg : segs) {
                for (size_t j = 0; j < sg.rec_start.size(); ++j) { out->offsets[r] = base + sg.rec_start[j]; blank[(size_t)r] = sg.rec_blank[j]; r++; }
                if (!sg.seq.empty()) cp.start([dst = out->bases + base, &sg]() { memcpy(dst, sg.seq.data(), sg.seq.size()); });
                base += sg.seq.size();
            }
        }
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
    } catch (...) {
        free(out->bases); free(out->offsets);
        memset(out, 0, sizeof(*out));
        set_err(errbuf, errbuf_len, "internal error in the FASTA reader");
        return KMC_ERR_NOMEM;
    }
}

// ---- streaming reader (kmc_ingest.h) -----------------------------------------------------------
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include "kmc_ingest.h"

namespace {

struct PieceOut {
    bool oom = false;  // the worker ran out of memory (an exception must not leave a thread function)
    uint64_t src_off = 0, n_bytes = 0;
    std::vector<uint64_t> rec_start;  // local: sequence bytes of this piece that precede the record
    std::vector<uint8_t> rec_blank;
    bool bad_first_line = false;
    int bad_byte = -1;
};

// lines of text[b, e) (b is a line start; the last line may end at e without '\n')
void parse_piece_body(const char* text, uint64_t b, uint64_t e, bool file_start, bool check_alphabet, uint8_t* out, PieceOut* po);
void parse_piece(const char* text, uint64_t b, uint64_t e, bool file_start, bool check_alphabet, uint8_t* out, PieceOut* po) {
    try { parse_piece_body(text, b, e, file_start, check_alphabet, out, po); } catch (...) { po->oom = true; }
}
void parse_piece_body(const char* text, uint64_t b, uint64_t e, bool file_start, bool check_alphabet, uint8_t* out, PieceOut* po) {
    const char* p = text + b;
    const char* const end = text + e;
    uint8_t* o = out;
    bool in_record = false;
    while (p < end) {
        const char* nl = (const char*)memchr(p, '\n', (size_t)(end - p));
        const char* le = nl ? nl : end;
        if (le > p && *p == '>') {
            const char* t = le;
            while (t > p + 1 && is_space((unsigned char)t[-1])) t--;
            po->rec_start.push_back((uint64_t)(o - out));
            po->rec_blank.push_back(t - p <= 1);
            in_record = true;
        } else {
            if (file_start && !in_record) { po->bad_first_line = true; break; }
            const char* t = le;
            while (t > p && is_space((unsigned char)t[-1])) t--;
            const size_t n = (size_t)(t - p);
            memcpy(o, p, n);
            if (check_alphabet && po->bad_byte < 0) {
                for (size_t i = 0; i < n; ++i) {
                    const uint8_t c = (uint8_t)p[i];
                    if (c != 'A' && c != 'C' && c != 'G' && c != 'T') { po->bad_byte = c; break; }
                }
            }
            o += n;
        }
        if (!nl) break;
        p = nl + 1;
    }
    po->n_bytes = (uint64_t)(o - out);
}

// four-line FASTQ records of text[b, e) (b is a record start)
void parse_piece_fastq_body(const char* text, uint64_t b, uint64_t e, bool check_alphabet, uint8_t* out, PieceOut* po);
void parse_piece_fastq(const char* text, uint64_t b, uint64_t e, bool check_alphabet, uint8_t* out, PieceOut* po) {
    try { parse_piece_fastq_body(text, b, e, check_alphabet, out, po); } catch (...) { po->oom = true; }
}
void parse_piece_fastq_body(const char* text, uint64_t b, uint64_t e, bool check_alphabet, uint8_t* out, PieceOut* po) {
    const char* p = text + b;
    const char* const end = text + e;
    uint8_t* o = out;
    unsigned phase = 0;
    while (p < end) {
        const char* nl = (const char*)memchr(p, '\n', (size_t)(end - p));
        const char* le = nl ? nl : end;
        if (phase == 0) {
            if (le == p && !nl) break;            // (nothing after the last newline)
            if (*p != '@') { po->bad_first_line = true; break; }
            po->rec_start.push_back((uint64_t)(o - out));
            po->rec_blank.push_back(0);
        } else if (phase == 1) {
            const char* t = le;
            while (t > p && is_space((unsigned char)t[-1])) t--;
            const size_t n = (size_t)(t - p);
            memcpy(o, p, n);
            if (check_alphabet && po->bad_byte < 0) {
                for (size_t i = 0; i < n; ++i) {
                    const uint8_t c = (uint8_t)p[i];
                    if (c != 'A' && c != 'C' && c != 'G' && c != 'T') { po->bad_byte = c; break; }
                }
            }
            o += n;
        } else if (phase == 2) {
            if (le == p || *p != '+') { po->bad_first_line = true; break; }
        }
        phase = (phase + 1) & 3;
        if (!nl) break;
        p = nl + 1;
    }
    po->n_bytes = (uint64_t)(o - out);
}

}  // namespace

// first FASTQ record start at or after q (< limit), else limit: a line that starts with '@' whose
// second successor starts with '+' (a quality line may start with '@' too, but then the line two
// further down is a sequence, which never starts with '+')
uint64_t KmcFastaIngest::fastq_record_start(uint64_t q, uint64_t limit) const {
    if (q == 0) return 0;
    const char* nl = (const char*)memchr(map_ + q - 1, '\n', (size_t)(limit - q + 1));
    if (!nl) return limit;
    uint64_t ls = (uint64_t)(nl - map_) + 1;
    for (int tries = 0; tries < 8 && ls < limit; ++tries) {
        const char* n1 = (const char*)memchr(map_ + ls, '\n', (size_t)(size_ - ls));
        if (!n1) return limit;
        const uint64_t l1 = (uint64_t)(n1 - map_) + 1;
        const char* n2 = l1 < size_ ? (const char*)memchr(map_ + l1, '\n', (size_t)(size_ - l1)) : nullptr;
        if (!n2) return limit;
        const uint64_t l2 = (uint64_t)(n2 - map_) + 1;
        if (map_[ls] == '@' && l2 < size_ && map_[l2] == '+') return ls;
        ls = l1;
    }
    return limit;
}

KmcFastaIngest::~KmcFastaIngest() {
    if (map_ && size_) munmap((void*)map_, (size_t)size_);
}

int KmcFastaIngest::open(const char* path, uint64_t chunk_bytes, std::string* err) {
    int fd = ::open(path, O_RDONLY);
    if (fd < 0) { if (err) *err = "Error during opening the file"; return KMC_ERR_IO; }
    struct stat st;
    if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode)) { ::close(fd); if (err) *err = "not a regular file"; return KMC_ERR_IO; }
    size_ = (uint64_t)st.st_size;
    if (size_) {
        void* m = mmap(nullptr, (size_t)size_, PROT_READ, MAP_PRIVATE, fd, 0);
        if (m == MAP_FAILED) { ::close(fd); size_ = 0; if (err) *err = "cannot map the file"; return KMC_ERR_IO; }
        map_ = (const char*)m;
        (void)madvise(m, (size_t)size_, MADV_SEQUENTIAL);
    }
    ::close(fd);
    chunk_bytes_ = std::max<uint64_t>(chunk_bytes, 1);
    unsigned hw = std::thread::hardware_concurrency();
    threads_ = std::max(1u, std::min(hw ? hw : 1u, 32u));
    fastq_ = size_ > 0 && map_[0] == '@';
    // chunk boundaries: the first record start ("\n>") at or after every multiple of chunk_bytes
    cuts_.assign(1, 0);
    uint64_t at = 0;
    while (size_ - at > chunk_bytes_) {
        uint64_t q = at + chunk_bytes_;
        uint64_t cut = size_;
        if (fastq_) { cut = fastq_record_start(q, size_); q = size_; }
        while (q < size_) {
            const char* nl = (const char*)memchr(map_ + q, '\n', (size_t)(size_ - q));
            if (!nl) break;
            const uint64_t after = (uint64_t)(nl - map_) + 1;
            if (after < size_ && map_[after] == '>') { cut = after; break; }
            q = after;
        }
        if (cut >= size_) break;
        cuts_.push_back(cut);
        at = cut;
    }
    cuts_.push_back(size_);
    cap_ = 64;
    for (size_t i = 0; i + 1 < cuts_.size(); ++i) cap_ = std::max<uint64_t>(cap_, cuts_[i + 1] - cuts_[i] + 64);
    next_cut_ = 0;
    done_ = false;
    return KMC_OK;
}

uint64_t KmcFastaIngest::chunk_capacity() const { return cap_; }

size_t KmcFastaIngest::n_chunks() const { return cuts_.size() > 1 ? cuts_.size() - 1 : 0; }

// Chunk `idx` on its own (no reader state is touched: several feeder threads may parse different chunks of
// one file at the same time).  ck->terminated tells that a record with an empty header and no sequence
// ended the input inside this chunk (Record::is_empty(), main.rs:60-62): nothing after it counts.
int KmcFastaIngest::parse_chunk(size_t idx, unsigned threads, uint8_t* out_buf, bool check_alphabet, KmcIngestChunk* ck, std::string* err) const {
    ck->pieces.clear();
    ck->offsets.assign(1, 0);
    ck->n_reads = ck->n_bases = ck->max_read_len = 0;
    ck->bad_byte = -1;
    ck->eof = true;
    ck->terminated = false;
    if (idx + 1 >= cuts_.size()) return KMC_OK;
    const uint64_t cb = cuts_[idx], ce = cuts_[idx + 1];
    // segments of >= 4 MiB, snapped forward to line starts
    const uint64_t len = ce - cb;
    uint64_t nseg = std::min<uint64_t>(std::max(1u, threads), std::max<uint64_t>(1, len >> 22));
    std::vector<uint64_t> sb((size_t)nseg + 1);
    sb[0] = cb;
    for (uint64_t i = 1; i < nseg; ++i) {
        uint64_t q = cb + len * i / nseg;
        if (q < sb[(size_t)i - 1]) q = sb[(size_t)i - 1];
        if (fastq_) { sb[(size_t)i] = std::max(sb[(size_t)i - 1], fastq_record_start(q, ce)); continue; }  // FASTQ pieces start at records
        const char* nl = q < ce ? (const char*)memchr(map_ + q - 1, '\n', (size_t)(ce - q + 1)) : nullptr;  // line start: byte after a '\n' at >= q-1
        sb[(size_t)i] = nl ? (uint64_t)(nl - map_) + 1 : ce;
    }
    sb[(size_t)nseg] = ce;
    std::vector<PieceOut> po((size_t)nseg);
    try {
        ThreadGroup th((size_t)nseg);  // joined on every way out, also when starting a thread throws
        for (uint64_t i = 1; i < nseg; ++i) {
            po[(size_t)i].src_off = sb[(size_t)i] - cb;
            if (fastq_) th.start(parse_piece_fastq, map_, sb[(size_t)i], sb[(size_t)i + 1], check_alphabet, out_buf + po[(size_t)i].src_off, &po[(size_t)i]);
            else th.start(parse_piece, map_, sb[(size_t)i], sb[(size_t)i + 1], false, check_alphabet, out_buf + po[(size_t)i].src_off, &po[(size_t)i]);
        }
        if (fastq_) parse_piece_fastq(map_, sb[0], sb[1], check_alphabet, out_buf, &po[0]);
        else parse_piece(map_, sb[0], sb[1], cb == 0, check_alphabet, out_buf, &po[0]);
    } catch (const std::system_error&) {
        if (err) *err = "cannot start parser threads";
        return KMC_ERR_NOMEM;
    }
    for (auto& p : po) if (p.oom) throw std::bad_alloc();  // (callers turn it into KMC_ERR_NOMEM)
    if (fastq_) {
        for (auto& p : po)
            if (p.bad_first_line) { if (err) *err = "malformed FASTQ record (expected '@' header and '+' separator lines)"; return KMC_ERR_FORMAT; }
    }
    if (po[0].bad_first_line) { if (err) *err = "Expected > at record start."; return KMC_ERR_FORMAT; }
    uint64_t base = 0;
    std::vector<uint8_t> blank;
    ck->offsets.clear();
    for (auto& p : po) {
        for (size_t j = 0; j < p.rec_start.size(); ++j) { ck->offsets.push_back(base + p.rec_start[j]); blank.push_back(p.rec_blank[j]); }
        if (p.n_bytes) ck->pieces.push_back(KmcIngestPiece{p.src_off, p.n_bytes, base});
        if (ck->bad_byte < 0 && p.bad_byte >= 0) ck->bad_byte = p.bad_byte;
        base += p.n_bytes;
    }
    uint64_t nrec = ck->offsets.size();
    ck->offsets.push_back(base);
    ck->eof = idx + 2 >= cuts_.size();
    // Record::is_empty(): a record with empty header and no sequence ends the input (main.rs:60-62)
    for (uint64_t i = 0; i < nrec && !fastq_; ++i) {
        if (blank[(size_t)i] && ck->offsets[(size_t)i + 1] == ck->offsets[(size_t)i]) {
            nrec = i;
            base = ck->offsets[(size_t)i];
            ck->offsets.resize((size_t)nrec + 1);
            // drop / shorten the pieces past the cut
            std::vector<KmcIngestPiece> keep;
            for (auto& pc : ck->pieces) {
                if (pc.dst_off >= base) continue;
                if (pc.dst_off + pc.n_bytes > base) pc.n_bytes = base - pc.dst_off;
                keep.push_back(pc);
            }
            ck->pieces.swap(keep);
            ck->eof = true;
            ck->terminated = true;
            break;
        }
    }
    ck->n_reads = nrec;
    ck->n_bases = base;
    uint64_t m = 0;
    for (uint64_t i = 0; i < nrec; ++i) m = std::max<uint64_t>(m, ck->offsets[(size_t)i + 1] - ck->offsets[(size_t)i]);
    ck->max_read_len = m;
    return KMC_OK;
}

int KmcFastaIngest::next(uint8_t* out_buf, bool check_alphabet, KmcIngestChunk* ck, std::string* err) {
    if (done_ || next_cut_ + 1 >= cuts_.size()) {
        ck->pieces.clear();
        ck->offsets.assign(1, 0);
        ck->n_reads = ck->n_bases = ck->max_read_len = 0;
        ck->bad_byte = -1;
        ck->eof = true;
        ck->terminated = false;
        done_ = true;
        return KMC_OK;
    }
    const int rc = parse_chunk(next_cut_, threads_, out_buf, check_alphabet, ck, err);
    next_cut_++;
    if (rc || ck->eof) done_ = true;
    return rc;
}

// ---- public streaming form: dense host buffers, one chunk at a time -----------------------------
struct kmc_fasta_stream {
    KmcFastaIngest ing;
    KmcIngestChunk ck;
    std::vector<uint8_t> buf;
};

extern "C" int kmc_fasta_stream_open(const char* path, uint64_t chunk_bytes, kmc_fasta_stream** out, char* errbuf, size_t errbuf_len) {
    if (!path || !out) return KMC_ERR_ARG;
    *out = nullptr;
    kmc_fasta_stream* s = new (std::nothrow) kmc_fasta_stream();
    if (!s) return KMC_ERR_NOMEM;
    try {
        std::string err;
        int rc = s->ing.open(path, chunk_bytes ? chunk_bytes : (256ull << 20), &err);
        if (rc) { set_err(errbuf, errbuf_len, err.c_str()); delete s; return rc; }
        s->buf.resize((size_t)s->ing.chunk_capacity());
    } catch (...) { delete s; set_err(errbuf, errbuf_len, "out of memory"); return KMC_ERR_NOMEM; }
    *out = s;
    return KMC_OK;
}

extern "C" int kmc_fasta_stream_next(kmc_fasta_stream* s, kmc_reads* out, int* eof, char* errbuf, size_t errbuf_len) {
    if (!s || !out) return KMC_ERR_ARG;
    memset(out, 0, sizeof(*out));
    int rc;
    try {
        std::string err;
        rc = s->ing.next(s->buf.data(), false, &s->ck, &err);
        if (rc) { set_err(errbuf, errbuf_len, err.c_str()); return rc; }
    } catch (...) { set_err(errbuf, errbuf_len, "out of memory"); return KMC_ERR_NOMEM; }
    // dense form: the pieces move left, in order (destination never passes a later piece's source)
    for (auto& pc : s->ck.pieces)
        if (pc.dst_off != pc.src_off) memmove(s->buf.data() + pc.dst_off, s->buf.data() + pc.src_off, (size_t)pc.n_bytes);
    out->bases = s->buf.data();
    out->offsets = s->ck.offsets.data();
    out->n_reads = s->ck.n_reads;
    out->n_bases = s->ck.n_bases;
    out->max_read_len = s->ck.max_read_len;
    if (eof) *eof = s->ck.eof ? 1 : 0;
    return KMC_OK;
}

extern "C" void kmc_fasta_stream_close(kmc_fasta_stream* s) { delete s; }

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
    std::vector<uint8_t> pool;
    try { pool.resize((size_t)s->pool * s->line_len); } catch (...) { return KMC_ERR_NOMEM; }
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
    try {
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
    } catch (...) { return KMC_ERR_NOMEM; }
    return KMC_OK;
}
