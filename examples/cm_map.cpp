// Minimal C++ caller of the C ABI (no Python, no torch): stage 1 from stock CircMiner files to PAM / SAM + the remain FASTQ
// that stage 2 reads, then stage 2 (circ_detect) to <out>.candidates.pam and <out>.circ_report.  Build:  g++ -std=c++17 -I include examples/cm_map.cpp -L circminer_amd/csrc -lcmhot
//                              -Wl,-rpath,$PWD/circminer_amd/csrc -o cm_map
// Usage:  cm_map <ref>.packed.fa.index <annotation.gtf> <R1.fastq[.gz]> <R2.fastq[.gz]> <out_prefix> [pam|sam|none] [k]
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "circminer_hot.h"

int main(int argc, char **argv) {
    if (argc < 6) {
        fprintf(stderr, "usage: %s index gtf r1 r2 out_prefix [pam|sam|none] [k]\n", argv[0]);
        return 2;
    }
    const std::string info = std::string(argv[1]) + ".info";
    cm_mapping_args a;
    memset(&a, 0, sizeof a);
    a.index_path = argv[1];
    a.index_info_path = info.c_str();
    a.gtf_path = argv[2];
    a.fastq1 = argv[3];
    a.fastq2 = argv[4];
    a.out_prefix = argv[5];
    a.report = argc > 6 ? (strcmp(argv[6], "sam") == 0 ? 2 : (strcmp(argv[6], "none") == 0 ? 0 : 1)) : 1;
    a.n_threads = 8;
    // defaults of the reference's command line (src/commandline_parser.cpp:7-33); kmer 0 = the index file's k
    a.params.kmer = argc > 7 ? atoi(argv[7]) : 0;
    a.params.seed_lim = 500;
    a.params.max_read_len = 300;
    a.params.scan_level = 0;
    a.params.max_ed = 4;
    a.params.max_sc = 7;
    a.params.band = 3;
    a.params.max_tlen = 500;
    a.params.max_intron = 2000000;
    a.params.max_chain_len = 30;
    a.params.device = 0;
    cm_mapping_stats st;
    char err[512];
    const int rc = cm_mapping_run(&a, &st, err, sizeof err);
    if (rc != CM_OK) {
        fprintf(stderr, "cm_mapping_run: %s\n", err);
        return 1;
    }
    printf("%llu pairs, %d round(s), %llu BSJ candidate pairs; load %.2fs, map %.2fs (%.2f M pairs/s; parse %.2fs, device %.2fs, write %.2fs)\n",
           (unsigned long long)st.pairs, st.rounds, (unsigned long long)st.bsj_pairs, st.seconds_load, st.seconds_map,
           st.seconds_map > 0 ? st.pairs / st.seconds_map / 1e6 : 0.0, st.seconds_parse, st.seconds_device, st.seconds_write);
    // stage 2: circ_detect(last_round) of the reference (src/circminer.cpp:347-352)
    cm_circ_args c;
    memset(&c, 0, sizeof c);
    c.index_path = argv[1];
    c.index_info_path = info.c_str();
    c.gtf_path = argv[2];
    c.out_prefix = argv[5];
    c.params = a.params;
    c.last_round = st.rounds;
    c.n_threads = 8;
    cm_circ_stats cs;
    const int rc2 = cm_circ_run(&c, &cs, err, sizeof err);
    if (rc2 != CM_OK) {
        fprintf(stderr, "cm_circ_run: %s\n", err);
        return 1;
    }
    printf("stage 2: %llu pairs, %llu candidate rows, %llu junction calls in %.2fs\n", (unsigned long long)cs.pairs,
           (unsigned long long)cs.candidate_rows, (unsigned long long)cs.calls, cs.seconds);
    return 0;
}
