// kmc_cli.cpp -- `k-mer-count`: the reference's process boundary, kept.
//
// Reference: k-mer-count/src/main.rs:43-91 takes no arguments, opens "sample.fasta" in the
// cwd (main.rs:44) and prints one sorted 54-character line per occurrence (main.rs:88-90);
// test.py:15-18 takes the FASTA path as its only positional argument.  This tool keeps both:
//
//   k-mer-count [FASTA] [-k K] [--forward] [--expand] [--device N | --gpus N] [--algo auto|stream|walk|sort] [--stats]
//
//   --gpus N  the file's chunks go round-robin to GPUs 0..N-1 of this process, tables reduced on GPU 0
//             (there is no CPU backend: SURVEY.md's "--backend cpu" is deliberately absent)
//
//   no -k   reference mode: LR-gapped 27+gap+27 for sizes 80..=140, expanded sorted output,
//           byte-identical to main.rs:87-90
//   -k K    count-table mode: contiguous canonical K-mers, "KMER<TAB>COUNT" lines sorted by KMER
//
// Errors: message on stderr, exit code 101 (what a Rust panic exits with), never partial stdout.
#include <errno.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "kmc.h"

static int die(const char* what, const char* msg) {
    fprintf(stderr, "k-mer-count: %s: %s\n", what, msg);
    return 101;
}

// a whole decimal number in [lo, hi], or exit code 2 (atoi turned "-k abc" and "-k 0" into the
// reference's LR mode and printed 195 MB instead of an error)
static bool parse_int(const char* opt, const char* text, long lo, long hi, int* out) {
    char* end = nullptr;
    errno = 0;
    const long v = strtol(text, &end, 10);
    if (errno || end == text || *end != '\0' || v < lo || v > hi) {
        fprintf(stderr, "k-mer-count: %s needs a whole number in %ld..%ld (got '%s')\n", opt, lo, hi, text);
        return false;
    }
    *out = (int)v;
    return true;
}

int main(int argc, char** argv) {
    const char* path = "sample.fasta";  // main.rs:44
    int k = 0, canonical = 1, expand = 0, device = 0, algo = KMC_ALGO_AUTO, stats = 0, gpus = 1;
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        if (a == "-k" && i + 1 < argc) { if (!parse_int("-k", argv[++i], 1, 63, &k)) return 2; }
        else if (a == "--forward") canonical = 0;
        else if (a == "--expand") expand = 1;
        else if (a == "--stats") stats = 1;
        else if (a == "--device" && i + 1 < argc) { if (!parse_int("--device", argv[++i], 0, 1023, &device)) return 2; }
        else if (a == "--gpus" && i + 1 < argc) { if (!parse_int("--gpus", argv[++i], 1, 64, &gpus)) return 2; }
        else if (a == "--algo" && i + 1 < argc) {
            std::string v = argv[++i];
            algo = v == "stream" ? KMC_ALGO_STREAM : v == "walk" ? KMC_ALGO_WALK : v == "sort" ? KMC_ALGO_SORT : KMC_ALGO_AUTO;
        } else if (a == "-h" || a == "--help") {
            fprintf(stderr, "usage: k-mer-count [FASTA] [-k K] [--forward] [--expand] [--device N | --gpus N] [--algo auto|stream|walk|sort] [--stats]\n");
            return 0;
        } else if (a == "-k" || a == "--device" || a == "--gpus" || a == "--algo") {
            fprintf(stderr, "k-mer-count: %s needs a value\n", a.c_str());
            return 2;
        } else if (!a.empty() && a[0] != '-') path = argv[i];
        else { fprintf(stderr, "k-mer-count: unknown option %s\n", a.c_str()); return 2; }
    }
    kmc_config cfg;
    memset(&cfg, 0, sizeof(cfg));
    cfg.struct_size = sizeof(cfg);
    cfg.mode = k ? KMC_MODE_CONTIG : KMC_MODE_LR;
    cfg.k = k ? k : 54;
    cfg.canonical = canonical;
    cfg.device = device;
    cfg.algo = algo;
    if (!k) expand = 1;
    if (gpus < 1 || gpus > 64) { fprintf(stderr, "k-mer-count: --gpus must be 1..64\n"); return 2; }
    // KMC_CLI_SHARE_DEVICE=D (testing on a box with fewer GPUs): all --gpus contexts live on device D
    const char* share = getenv("KMC_CLI_SHARE_DEVICE");
    std::vector<kmc_ctx*> ctxs;
    int rc = 0;
    for (int g = 0; g < gpus && !rc; ++g) {
        cfg.device = gpus == 1 ? device : (share ? atoi(share) : g);
        kmc_ctx* one = nullptr;
        rc = kmc_create(&one, &cfg);
        if (!rc) ctxs.push_back(one);
    }
    auto destroy_all = [&]() { for (kmc_ctx* x : ctxs) kmc_destroy(x); };
    if (rc) { int r = die("kmc_create", kmc_last_error(nullptr)); destroy_all(); return r; }
    kmc_ctx* ctx = ctxs[0];
    uint64_t nd = 0, nt = 0;
    rc = gpus == 1 ? kmc_count_file(ctx, path, &nd, &nt) : kmc_count_file_multi(ctxs.data(), (uint32_t)ctxs.size(), path, &nd, &nt);
    if (rc) { int r = die(path, kmc_last_error(ctx)); destroy_all(); return r; }
    std::vector<uint64_t> hi(nd ? nd : 1), lo(nd ? nd : 1), cnt(nd ? nd : 1);
    rc = kmc_export(ctx, hi.data(), lo.data(), cnt.data(), nd);
    if (rc) { int r = die("kmc_export", kmc_last_error(ctx)); destroy_all(); return r; }
    const int klen = k ? k : 54;
    std::vector<char> obuf(1 << 22);
    setvbuf(stdout, obuf.data(), _IOFBF, obuf.size());
    char line[96];
    for (uint64_t i = 0; i < nd; ++i) {
        kmc_decode_key(hi[i], lo[i], klen, line);
        if (expand) {
            line[klen] = '\n';
            for (uint64_t c = 0; c < cnt[i]; ++c) fwrite(line, 1, (size_t)klen + 1, stdout);
        } else {
            int m = snprintf(line + klen, sizeof(line) - klen, "\t%llu\n", (unsigned long long)cnt[i]);
            fwrite(line, 1, (size_t)(klen + m), stdout);
        }
    }
    fflush(stdout);
    if (stats) {
        kmc_stats s;
        kmc_get_stats(ctx, &s);
        fprintf(stderr, "reads %llu bases %llu kmers %llu distinct %llu table_slots %llu spilled %llu kernel_ms %.3f algo %d\n",
                (unsigned long long)s.n_reads, (unsigned long long)s.n_bases, (unsigned long long)s.n_kmers,
                (unsigned long long)s.n_distinct, (unsigned long long)s.table_capacity, (unsigned long long)s.n_spilled,
                s.kernel_ms_last, s.algo_last);
    }
    destroy_all();
    return 0;
}
