// Sanitizer harness for the host-only part of libkmc (kmc_host.cpp: whole-file reader, streaming
// reader incl. FASTQ, key decoding, synthetic generator).  Built with -fsanitize=address,undefined
// by tests/test_abi_host.py::test_host_code_under_sanitizers and run on the files given on the
// command line with several chunk sizes; prints a checksum per file so that the test can compare
// the streamed form with the whole-file form.  CPU only (GPU ASan is unavailable on the pool).
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "kmc.h"

static uint64_t fnv(uint64_t h, const void* p, size_t n) {
    const unsigned char* b = (const unsigned char*)p;
    for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 1099511628211ull; }
    return h;
}

int main(int argc, char** argv) {
    char eb[256];
    for (int a = 1; a < argc; ++a) {
        const char* path = argv[a];
        const bool fastq = strstr(path, ".fastq") != nullptr;
        uint64_t whole = 0;
        int whole_rc = 0;
        if (!fastq) {
            kmc_reads rd;
            whole_rc = kmc_parse_fasta(path, &rd, eb, sizeof(eb));
            if (!whole_rc) {
                whole = fnv(1469598103934665603ull, rd.bases, (size_t)rd.n_bases);
                for (uint64_t i = 0; i <= rd.n_reads; ++i) whole = fnv(whole, &rd.offsets[i], 8);
                kmc_free_reads(&rd);
            }
        }
        const uint64_t chunks[] = {1, 7, 64, 4096, 0};
        for (uint64_t cb : chunks) {
            kmc_fasta_stream* s = nullptr;
            int rc = kmc_fasta_stream_open(path, cb, &s, eb, sizeof(eb));
            uint64_t h = 1469598103934665603ull, base = 0, zero = 0;
            std::vector<uint64_t> offs(1, 0);
            int eof = 0;
            while (!rc && !eof) {
                kmc_reads rd;
                rc = kmc_fasta_stream_next(s, &rd, &eof, eb, sizeof(eb));
                if (rc) break;
                h = fnv(h, rd.bases, (size_t)rd.n_bases);
                for (uint64_t i = 1; i <= rd.n_reads; ++i) offs.push_back(base + rd.offsets[i]);
                base += rd.n_bases;
            }
            if (s) kmc_fasta_stream_close(s);
            (void)zero;
            for (uint64_t o : offs) h = fnv(h, &o, 8);
            if (!fastq && rc != whole_rc) { printf("%s chunk %llu: rc %d vs whole-file rc %d\n", path, (unsigned long long)cb, rc, whole_rc); return 1; }
            if (!fastq && !rc && h != whole) { printf("%s chunk %llu: streamed form differs from the whole-file form\n", path, (unsigned long long)cb); return 1; }
            printf("%s chunk %llu rc %d hash %016llx\n", path, (unsigned long long)cb, rc, (unsigned long long)h);
        }
    }
    // key decoding and the generator
    char buf[64];
    kmc_decode_key(0x0123456789abcdefull, 0xfedcba9876543210ull, 63, buf);
    kmc_synth sy = {7, 10, 80, 5, 0};
    std::vector<uint8_t> b(400 * 33);
    std::vector<uint64_t> o(34);
    if (kmc_synth_reads_host(&sy, 5, 33, b.data(), o.data()) != 0) return 1;
    uint64_t exact = 0;
    (void)kmc_synth_records_for_bytes(&sy, 123456789, &exact);
    puts("ok");
    return 0;
}
