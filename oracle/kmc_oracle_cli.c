/*
 * kmc_oracle_cli.c -- command-line front end of the CPU oracle (TEST INFRASTRUCTURE ONLY).
 *
 *   kmc_oracle_cli lr    FASTA                      reference mode, byte-identical to
 *                                                   k-mer-count/src/main.rs:87-90 output
 *   kmc_oracle_cli count FASTA K [--forward] [--expand] [--hash|--strings]
 *
 * Exit code 101 + message on stderr on error, like a Rust panic (main.rs:23,44,59).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "kmc_oracle.h"

static int die(int code) {
    fprintf(stderr, "kmc_oracle: %s\n", kmo_strerror(code));
    return 101;
}

int main(int argc, char** argv) {
    if (argc < 3) {
        fprintf(stderr, "usage: %s lr FASTA | count FASTA K [--forward] [--expand] [--hash|--strings]\n", argv[0]);
        return 2;
    }
    kmo_reads rd;
    kmo_table tb;
    int rc = kmo_parse_fasta(argv[2], &rd);
    if (rc) return die(rc);
    int expand = 0;
    if (!strcmp(argv[1], "lr")) {
        rc = kmo_count_lr(rd.bases, rd.offsets, rd.n_reads, &tb);
        expand = 1;
    } else if (!strcmp(argv[1], "count") && argc >= 4) {
        int k = atoi(argv[3]), canonical = 1, method = 0, strings = 0;
        for (int i = 4; i < argc; i++) {
            if (!strcmp(argv[i], "--forward")) canonical = 0;
            else if (!strcmp(argv[i], "--expand")) expand = 1;
            else if (!strcmp(argv[i], "--hash")) method = 1;
            else if (!strcmp(argv[i], "--strings")) strings = 1;
        }
        rc = strings ? kmo_count_kmers_strings(rd.bases, rd.offsets, rd.n_reads, k, canonical, &tb)
                     : kmo_count_kmers(rd.bases, rd.offsets, rd.n_reads, k, canonical, method, &tb);
    } else {
        return 2;
    }
    if (rc) return die(rc);
    static char obuf[1 << 20];
    setvbuf(stdout, obuf, _IOFBF, sizeof(obuf));
    if (kmo_write_table(&tb, expand, stdout) < 0) return die(KMO_ERR_IO);
    fflush(stdout);
    kmo_free_table(&tb);
    kmo_free_reads(&rd);
    return 0;
}
