// kmc_genfasta.cpp -- `kmc-genfasta`: seeded, size-parameterised generator with the
// distribution of the reference's random_fasta_generator.py:5-15 (which is unseeded and fixed
// at 200 records).  Writes FASTA text to stdout.
//
//   kmc-genfasta --bytes N | --records R  [--seed S] [--pool 10] [--line 80] [--lines 5]
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>

#include "kmc.h"

int main(int argc, char** argv) {
    kmc_synth s;
    memset(&s, 0, sizeof(s));
    s.seed = 1; s.pool = 10; s.line_len = 80; s.lines_per_record = 5;
    unsigned long long bytes = 0, records = 0;
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        auto next = [&]() -> unsigned long long { return i + 1 < argc ? strtoull(argv[++i], nullptr, 10) : 0ull; };
        if (a == "--bytes") bytes = next();
        else if (a == "--records") records = next();
        else if (a == "--seed") s.seed = next();
        else if (a == "--pool") s.pool = (uint32_t)next();
        else if (a == "--line") s.line_len = (uint32_t)next();
        else if (a == "--lines") s.lines_per_record = (uint32_t)next();
        else { fprintf(stderr, "usage: kmc-genfasta --bytes N | --records R [--seed S] [--pool 10] [--line 80] [--lines 5]\n"); return 2; }
    }
    if (!records) records = bytes ? kmc_synth_records_for_bytes(&s, bytes, nullptr) : 200;  // generator :10
    static char obuf[1 << 22];
    setvbuf(stdout, obuf, _IOFBF, sizeof(obuf));
    const unsigned long long STEP = 1 << 16;
    for (unsigned long long r = 0; r < records; r += STEP) {
        unsigned long long n = records - r < STEP ? records - r : STEP;
        if (kmc_synth_write_fasta(&s, r, n, stdout)) { fprintf(stderr, "kmc-genfasta: write failed\n"); return 1; }
    }
    fflush(stdout);
    return 0;
}
