// Stage 1 of CircMiner end to end, on top of the C ABI (the caller of the hot path).
//
// Mirrors mapping() + map_reads() of the reference (src/circminer.cpp:98-352, :354-400): index info -> GTF ->
// one ROUND per packed contig -> per pair the skip / print / write rules of :386-397.  What changes is the order of
// the loops.  The reference streams ALL reads through one contig, writes the survivors to <out>_<r>_remain_R?.fastq,
// loads the next contig and streams those files again.  Here every packed contig is resident in HBM (hg38: ~40 GB of
// 288 GB) and each batch of pairs goes through all rounds back to back on the device, so the only files written are
// the ones a user of the reference keeps:
//   <out>.mapping.pam | .sam            one row (two SAM lines) per pair, with the state it had when it was retired
//                                       (skip) or after the last round -- the same rows the reference prints, in
//                                       another order (the reference's order depends on its thread schedule);
//   <out>_<R>_remain_R{1,2}.fastq       R = number of packed contigs: the CHIBSJ / CHI2BSJ pairs of the last round with
//                                       their 23-token headers, stage 2's input (it sorts them itself).
// A pair retired in round r is left untouched by later rounds (cm_map_round), so its final state is the one the
// reference printed in round r.
//
// Four batches are in flight: batch k-1's rows are being written by a worker thread, batch k is on the GPU, batch k+1 is staged
// (copied over PCIe on the copy stream out of page-locked parser buffers; its first round is seeded and chained under batch k's
// last pair stage, cm_map_rounds) and batch k+2 is being parsed on another worker thread (cm_fastq_next keeps four generations).
// Only what the rows need leaves the device: with report 0 (the reference's default, src/commandline_parser.cpp:26) that is the
// (pair, state) records of the re-queued pairs and a 14-bin type histogram.
//
// One process per GPU: rank r of w maps the r-th contiguous block of pairs (cm_fastq_open_shard) and writes .part<r> files;
// cm_merge_parts on rank 0 concatenates them in rank order -- the bytes one process would have written.
#include <algorithm>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <mutex>
#include <thread>
#include <vector>

#include "circminer_hot.h"

namespace {

struct Fail {
    char *buf;
    size_t cap;
    int operator()(int code, const char *fmt, ...) const {
        if (buf && cap) {
            va_list ap;
            va_start(ap, fmt);
            vsnprintf(buf, cap, fmt, ap);
            va_end(ap);
        }
        return code;
    }
};

double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

}  // namespace

extern "C" int cm_mapping_run(const cm_mapping_args *a, cm_mapping_stats *stats, char *err, uint64_t err_cap) {
    Fail fail{err, (size_t)err_cap};
    if (err && err_cap) err[0] = 0;
    if (!a || !a->index_path || !a->index_info_path || !a->gtf_path || !a->fastq1 || !a->fastq2 || !a->out_prefix)
        return fail(CM_EINVAL, "cm_mapping_run: null argument");
    if (a->report < 0 || a->report > 2) return fail(CM_EINVAL, "report must be 0 (none), 1 (PAM) or 2 (SAM)");
    cm_mapping_stats st;
    memset(&st, 0, sizeof st);
    const double t0 = now();
    const int n_threads = a->n_threads > 0 ? a->n_threads : 1;
    const uint64_t batch_pairs = a->batch_pairs ? a->batch_pairs : (1ull << 18);
    const int world = a->world > 1 ? a->world : 1, rank = a->world > 1 ? a->rank : 0;
    if (rank < 0 || rank >= world) return fail(CM_EINVAL, "rank %d of %d", a->rank, a->world);
    const std::string part = world > 1 ? ".part" + std::to_string(rank) : "";

    cm_chr_info *chrs = nullptr;
    uint32_t n_chr = 0;
    cm_index_file *idx = nullptr;
    cm_ctx *cm = nullptr;
    cm_fastq *fq = nullptr;
    cm_writer *w_map = nullptr, *w_rem = nullptr;
    std::vector<cm_index_view> views;
    std::vector<cm_annot_view> annots;
    // results of batch k live in set k & 1: the writer thread of batch k reads them while batch k+1 is downloaded
    struct Result {
        std::vector<cm_mapped_read> state;
        std::vector<uint8_t> active;
        std::vector<uint64_t> sel;
        std::vector<cm_record> recs;
        uint64_t n_rec = 0;
        cm_fastq_batch batch;
    } res[2];
    // parser arrays registered with the runtime (cm_host_register).  The parser tells us (release hook, on its own thread) before
    // one of them is freed -- a generation's array that has to grow is allocated anew -- so that no registration outlives its block.
    struct Pinned {
        std::mutex mu;
        std::vector<std::pair<void *, uint64_t>> v;
        cm_ctx *cm = nullptr;
    } pin;
    std::vector<std::pair<void *, uint64_t>> &pinned = pin.v;
    std::thread writer, parser, gtf_thread, free_thread;
    int writer_rc = CM_OK, parser_rc = CM_OK;
    int rc = CM_OK;
    auto cleanup = [&]() {
        if (free_thread.joinable()) free_thread.join();
        if (gtf_thread.joinable()) gtf_thread.join();           // it reads chrs
        if (parser.joinable()) parser.join();
        if (writer.joinable()) writer.join();
        if (w_map) cm_writer_close(w_map);
        if (w_rem) cm_writer_close(w_rem);
        if (cm) (void)cm_sync(cm);                             // drains the copy stream too: no copy out of the parser's arrays is in flight any more
        {
            std::lock_guard<std::mutex> lk(pin.mu);
            for (auto &pr : pinned) (void)cm_host_unregister(cm, pr.first);
            pinned.clear();
        }
        if (fq) cm_fastq_close(fq);
        if (cm) cm_destroy(cm);
        if (!annots.empty()) cm_host_free_annotation(annots.data(), (uint32_t)annots.size());
        for (auto &v : views) cm_host_free_loaded_contig(&v);
        if (idx) cm_host_close_index(idx);
        if (chrs) cm_host_free_index_info(chrs, n_chr);
    };
#define MAP_TRY(call, what)                                                                                  \
    do {                                                                                                     \
        rc = (call);                                                                                         \
        if (rc != CM_OK) {                                                                                   \
            rc = fail(rc, "%s failed (%d)%s%s", what, rc, cm ? ": " : "", cm ? cm_last_error(cm) : "");       \
            cleanup();                                                                                       \
            return rc;                                                                                       \
        }                                                                                                    \
    } while (0)

    // ---- index info, index file, context (genome_packer.load_index_info, checkHashTable, initLoadingHashTableMeta) ----
    MAP_TRY(cm_host_read_index_info(a->index_info_path, &chrs, &n_chr), "cm_host_read_index_info");
    int32_t kmer = 0, full = 0;
    uint32_t n_rec = 0;
    MAP_TRY(cm_host_open_index(a->index_path, &idx, &kmer, &full, &n_rec), "cm_host_open_index");
    cm_params P = a->params;
    if (P.kmer == 0) P.kmer = kmer;
    if (P.kmer != kmer) {
        rc = fail(CM_EINVAL, "index was built for k = %d, params ask for k = %d", kmer, P.kmer);
        cleanup();
        return rc;
    }
    MAP_TRY(cm_create(&P, &cm), "cm_create");
    pin.cm = cm;

    // ---- every packed contig into its own slot (loadHashTable + pac2char_whole_contig per round in the reference) ----
    // Contig c + 1 is read and decoded from the index file (host threads) while contig c goes over PCIe and gets its bucket
    // descriptors built (cm_load_contig), and the GTF is parsed meanwhile on a thread of its own (it needs the contig lengths
    // only at the end: they are in the .index.info rows already).
    const bool trace = getenv("CM_INDEX_TRACE") != nullptr;
    auto lap = [&](const char *what, double since) {
        if (trace) fprintf(stderr, "[load] %s %.3f s (at %.3f s)\n", what, now() - since, now() - t0);
    };
    lap("index info + header + context", t0);
    // the GTF model is built on a thread of its own meanwhile; the contig lengths it needs follow from the .index.info rows
    // (a packed contig ends with its last chromosome) and are checked against the index file's afterwards
    std::vector<uint32_t> clen_guess;
    for (uint32_t i = 0; i < n_chr; ++i) {
        if (chrs[i].contig_id == 0) continue;
        if (clen_guess.size() < chrs[i].contig_id) clen_guess.resize(chrs[i].contig_id, 0);
        clen_guess[chrs[i].contig_id - 1] = std::max(clen_guess[chrs[i].contig_id - 1], chrs[i].start_pos + chrs[i].len);
    }
    std::vector<cm_annot_view> annots_early(clen_guess.size());
    int early_rc = CM_EINVAL;
    if (!clen_guess.empty())
        gtf_thread = std::thread([&]() {
            const double tg = now();
            early_rc = cm_host_build_annotation(a->gtf_path, chrs, n_chr, clen_guess.data(), (uint32_t)clen_guess.size(), P.max_read_len, annots_early.data());
            lap("GTF -> annotation tables (under the contig loads)", tg);
        });
    if (full) {
        // full-format index: the table crosses PCIe as it is in the file and the device flattens it (cm_load_contig_raw); the
        // host only reads and decodes the bucket headers -- of contig c + 1 while contig c uploads
        cm_index_raw nxt_raw;
        int nxt_loaded = 0, nxt_rc = CM_OK;
        double tl = now();
        nxt_rc = cm_host_next_contig_raw(idx, n_threads, &nxt_raw, &nxt_loaded);
        lap("contig record read", tl);
        for (;;) {
            MAP_TRY(nxt_rc, "cm_host_next_contig_raw");
            if (!nxt_loaded) break;
            const cm_index_raw raw = nxt_raw;
            cm_index_view iv;
            memset(&iv, 0, sizeof iv);
            iv.contig_num = raw.contig_num;
            iv.ref_len = raw.ref_len;
            views.push_back(iv);                                  // (lengths only: the arrays stay with the file handle)
            if (iv.contig_num != (int32_t)views.size() - 1) {
                rc = fail(CM_EINVAL, "packed contigs out of order: record %zu is contig %d", views.size(), iv.contig_num + 1);
                cleanup();
                return rc;
            }
            std::thread ahead([&]() {
                const double ta = now();
                nxt_rc = cm_host_next_contig_raw(idx, n_threads, &nxt_raw, &nxt_loaded);
                lap("next contig record read (under the upload)", ta);
            });
            tl = now();
            const int lrc = cm_load_contig_raw(cm, (int)views.size() - 1, &raw);
            lap("contig uploaded, flattened on the device + descriptors", tl);
            ahead.join();
            MAP_TRY(lrc, "cm_load_contig_raw");
        }
    } else {
        cm_index_view nxt_iv;
        int nxt_loaded = 0, nxt_rc = CM_OK;
        double tl = now();
        nxt_rc = cm_host_next_contig(idx, n_threads, &nxt_iv, &nxt_loaded);
        lap("contig decoded", tl);
        for (;;) {
            MAP_TRY(nxt_rc, "cm_host_next_contig");
            if (!nxt_loaded) break;
            cm_index_view iv = nxt_iv;
            views.push_back(iv);
            if (iv.contig_num != (int32_t)views.size() - 1) {
                rc = fail(CM_EINVAL, "packed contigs out of order: record %zu is contig %d", views.size(), iv.contig_num + 1);
                cleanup();
                return rc;
            }
            std::thread ahead([&]() {
                const double ta = now();
                nxt_rc = cm_host_next_contig(idx, n_threads, &nxt_iv, &nxt_loaded);
                lap("next contig decoded (under the upload)", ta);
            });
            tl = now();
            const int lrc = cm_load_contig(cm, (int)views.size() - 1, &iv);
            lap("contig uploaded + descriptors", tl);
            // its host copy is dead now (everything is in HBM): returned to the system here, while the next contig is still being
            // decoded, rather than in one 25-GB sweep next to the FASTQ parser's first page faults
            cm_host_free_loaded_contig(&views.back());
            ahead.join();
            if (nxt_rc == CM_OK && nxt_loaded && lrc != CM_OK) cm_host_free_loaded_contig(&nxt_iv);
            MAP_TRY(lrc, "cm_load_contig");
        }
    }
    const uint32_t n_con = (uint32_t)views.size();
    if (n_con == 0) {
        rc = fail(CM_EINVAL, "index file holds no contig");
        cleanup();
        return rc;
    }
    // ---- GTF (gtf_parser.init / load_gtf) ----
    {
        std::vector<uint32_t> clen(n_con);
        for (uint32_t c = 0; c < n_con; ++c) clen[c] = views[c].ref_len;
        if (gtf_thread.joinable()) gtf_thread.join();
        if (early_rc == CM_OK && clen == clen_guess) {
            annots = annots_early;
            rc = CM_OK;
        } else {                                      // .index.info and the index file disagree on the contig lengths: the file decides
            if (early_rc == CM_OK) cm_host_free_annotation(annots_early.data(), (uint32_t)annots_early.size());
            annots.resize(n_con);
            const double tg = now();
            rc = cm_host_build_annotation(a->gtf_path, chrs, n_chr, clen.data(), n_con, P.max_read_len, annots.data());
            lap("GTF -> annotation tables", tg);
        }
        if (rc != CM_OK) {
            annots.clear();
            rc = fail(rc, "cm_host_build_annotation failed (%d)", rc);
            cleanup();
            return rc;
        }
        const double tu = now();
        for (uint32_t c = 0; c < n_con; ++c) MAP_TRY(cm_load_annotation(cm, (int)c, &annots[c]), "cm_load_annotation");
        lap("annotation uploaded", tu);
        views.clear();                                              // (their arrays went back after each upload)
        // everything is in HBM: the file handle's buffers (two sets of raw records on the full-format path, ~ 19 GB for hg38)
        // are dead weight from here on -- per process, and there is one process per GPU
        // (on a thread of its own: returning that much memory takes about a second, and nothing waits for it)
        free_thread = std::thread([h = idx]() { cm_host_close_index(h); });
        idx = nullptr;
        lap("load done", t0);
    }
    st.rounds = (int32_t)n_con;
    st.seconds_load = now() - t0;

    // ---- outputs (FilterRead::init, src/filter.cpp:34-84; SAMOutput::init) ----
    const std::string out = a->out_prefix;
    if (a->report) {
        const std::string path = out + (a->report == 2 ? ".mapping.sam" : ".mapping.pam") + part;
        MAP_TRY(cm_writer_open(path.c_str(), nullptr, chrs, n_chr, &w_map), "cm_writer_open (mapping)");
        if (a->report == 2 && rank == 0) MAP_TRY(cm_write_sam_header(w_map), "cm_write_sam_header");      // one header: rank 0's part comes first
    }
    {
        const std::string r1 = out + "_" + std::to_string(n_con) + "_remain_R1.fastq" + part, r2 = out + "_" + std::to_string(n_con) + "_remain_R2.fastq" + part;
        MAP_TRY(cm_writer_open(r1.c_str(), r2.c_str(), chrs, n_chr, &w_rem), "cm_writer_open (remain)");
    }
    MAP_TRY(cm_fastq_open_shard(a->fastq1, a->fastq2, chrs, n_chr, P.max_ed, rank, world, n_threads, &fq, nullptr, nullptr), "cm_fastq_open_shard");
    cm_fastq_set_release_hook(fq, [](void *user, const void *ptr, uint64_t bytes) {
        Pinned *pn = (Pinned *)user;
        std::lock_guard<std::mutex> lk(pn->mu);
        const char *lo = (const char *)ptr, *hi = lo + bytes;
        for (auto it = pn->v.begin(); it != pn->v.end();) {
            const char *a0 = (const char *)it->first, *a1 = a0 + it->second;
            if (a0 < hi && a1 > lo) {                         // a registration inside the block that is going
                (void)cm_host_unregister(pn->cm, it->first);
                it = pn->v.erase(it);
            } else ++it;
        }
    }, &pin);

    // ---- batches: write k-1 (worker) | rounds of k (device, driven by this thread) | H2D + first round of k+1 | parse k+2 (worker) ----
    const double t1 = now();
    std::vector<int> all(n_con);
    for (uint32_t c = 0; c < n_con; ++c) all[c] = (int)c;
    // page-lock the parser's arrays of a batch (they are reused every fourth batch; re-registered only when one has moved or grown)
    auto pin_batch = [&](const cm_fastq_batch &b) -> int {
        std::lock_guard<std::mutex> lk(pin.mu);
        const uint64_t n = b.reads.n_pairs;
        const void *ptr[4] = {b.reads.seq1, b.reads.seq2, b.reads.off1, b.reads.off2};
        const uint64_t bytes[4] = {b.reads.off1[n], b.reads.off2[n], (n + 1) * sizeof(uint64_t), (n + 1) * sizeof(uint64_t)};
        for (int j = 0; j < 4; ++j) {
            if (!bytes[j]) continue;
            bool have = false;
            for (auto it = pinned.begin(); it != pinned.end();) {
                const char *lo = (const char *)it->first, *hi = lo + it->second, *p = (const char *)ptr[j];
                if (p >= lo && p + bytes[j] <= hi) {
                    have = true;
                    break;
                }
                if (p < hi && p + bytes[j] > lo) {                    // overlaps a stale registration: drop that one
                    (void)cm_host_unregister(cm, it->first);
                    it = pinned.erase(it);
                } else ++it;
            }
            if (have) continue;
            // (the parser reuses its four generations of arrays, so there are 16 live ranges; if they kept moving -- batches that
            // keep growing -- stale registrations would pile up: past 64 the copies simply go through pageable staging)
            if (pinned.size() >= 64) continue;
            if (cm_host_register(cm, (void *)ptr[j], bytes[j]) == CM_OK) pinned.emplace_back((void *)ptr[j], bytes[j]);
            // (a failed registration is not an error: the copy is then a staged pageable one)
        }
        return CM_OK;
    };
    cm_fastq_batch cur, nxt, nn;
    memset(&nxt, 0, sizeof nxt);
    memset(&nn, 0, sizeof nn);
    double write_s = 0.0;                                       // written by the writer thread, read after its join
    {
        const double tp = now();
        MAP_TRY(cm_fastq_next(fq, batch_pairs, &cur), "cm_fastq_next");
        if (cur.reads.n_pairs) MAP_TRY(cm_fastq_next(fq, batch_pairs, &nxt), "cm_fastq_next");
        st.seconds_parse += now() - tp;
    }
    if (cur.reads.n_pairs) {
        pin_batch(cur);
        MAP_TRY(cm_reads_stage(cm, &cur.reads, cur.prior), "cm_reads_stage");
        MAP_TRY(cm_reads_swap(cm), "cm_reads_swap");
    }
    for (uint64_t k = 0; cur.reads.n_pairs; ++k) {
        const uint64_t n = cur.reads.n_pairs;
        Result &R = res[k & 1];
        double parse_s = 0.0;
        const bool more = nxt.reads.n_pairs != 0;
        parser_rc = CM_OK;
        memset(&nn, 0, sizeof nn);
        if (more)
            parser = std::thread([&]() {
                const double tp = now();
                parser_rc = cm_fastq_next(fq, batch_pairs, &nn);
                parse_s = now() - tp;
            });
        const double td = now();
        if (more) {
            pin_batch(nxt);
            MAP_TRY(cm_reads_stage(cm, &nxt.reads, nxt.prior), "cm_reads_stage");
        }
        MAP_TRY(cm_map_rounds(cm, all.data(), (int)n_con, 1), "cm_map_rounds");          // round r + 1 seeds while r pairs
        // results of batch k: rows of every pair (PAM / SAM) or just the re-queued pairs' records
        uint64_t n_rec = 0;
        if (a->report) {
            R.state.resize(n);
            R.active.resize(n);
            MAP_TRY(cm_reads_download(cm, R.state.data(), nullptr, R.active.data()), "cm_reads_download");
            R.sel.clear();
            for (uint64_t i = 0; i < n; ++i) {
                if (R.active[i]) R.sel.push_back(i);
                const int t = R.state[i].type;
                if (t >= 0 && t < 14) ++st.by_type[t];
            }
            n_rec = R.sel.size();
        } else {
            if (R.recs.size() < n) R.recs.resize(n);
            MAP_TRY(cm_collect_records(cm, 0, n, R.recs.data(), &n_rec), "cm_collect_records");
            uint64_t h[14];
            MAP_TRY(cm_type_histogram(cm, h), "cm_type_histogram");
            for (int t = 0; t < 14; ++t) st.by_type[t] += h[t];
        }
        R.n_rec = n_rec;
        if (more) MAP_TRY(cm_reads_swap(cm), "cm_reads_swap");
        st.seconds_device += now() - td;
        st.pairs += n;
        st.bsj_pairs += n_rec;
        R.batch = cur;
        if (writer.joinable()) writer.join();                    // rows of batch k-1 are out: file order = batch order
        MAP_TRY(writer_rc, "writer");
        // map_reads, src/circminer.cpp:386-397: printed if (skip || last round) = every pair by now;
        // written to the remain files if still active after the last round = CHIBSJ / CHI2BSJ
        writer = std::thread([&R, &writer_rc, &write_s, w_map, w_rem, report = a->report]() {
            const double tw = now();
            int r = CM_OK;
            if (report == 1) r = cm_write_pam(w_map, &R.batch, R.state.data(), nullptr, 0);
            if (report == 2) r = cm_write_sam(w_map, &R.batch, R.state.data(), nullptr, 0);
            if (r == CM_OK && R.n_rec) {
                if (report) r = cm_write_remain(w_rem, &R.batch, R.state.data(), R.sel.data(), R.sel.size());
                else r = cm_write_remain_records(w_rem, &R.batch, R.recs.data(), R.n_rec);
            }
            writer_rc = r;
            write_s += now() - tw;
        });
        if (parser.joinable()) parser.join();
        st.seconds_parse += parse_s;
        MAP_TRY(parser_rc, "cm_fastq_next");
        cur = nxt;
        nxt = nn;
    }
    if (writer.joinable()) writer.join();
    MAP_TRY(writer_rc, "writer");
    if (w_map) MAP_TRY(cm_writer_flush(w_map), "writing the mapping file");          // a full disk is an error, not a short file
    MAP_TRY(cm_writer_flush(w_rem), "writing the remain files");
    st.seconds_map = now() - t1;
    st.seconds_write = write_s;
    cleanup();
    if (stats) *stats = st;
    return CM_OK;
#undef MAP_TRY
}

extern "C" int cm_merge_parts(const char *out_prefix, int32_t rounds, int32_t world, int32_t report) {
    if (!out_prefix || rounds < 1 || world < 1 || report < 0 || report > 2) return CM_EINVAL;
    if (world == 1) return CM_OK;
    const std::string out = out_prefix;
    std::vector<std::string> names = {out + "_" + std::to_string(rounds) + "_remain_R1.fastq", out + "_" + std::to_string(rounds) + "_remain_R2.fastq"};
    if (report) names.push_back(out + (report == 2 ? ".mapping.sam" : ".mapping.pam"));
    std::vector<char> buf(8u << 20);
    for (const std::string &name : names) {
        FILE *dst = fopen(name.c_str(), "wb");
        if (!dst) return CM_EIO;
        for (int r = 0; r < world; ++r) {
            const std::string pn = name + ".part" + std::to_string(r);
            FILE *src = fopen(pn.c_str(), "rb");
            if (!src) {
                fclose(dst);
                return CM_EINVAL;                             // a rank has not written its part
            }
            size_t got;
            while ((got = fread(buf.data(), 1, buf.size(), src)) > 0)
                if (fwrite(buf.data(), 1, got, dst) != got) {
                    fclose(src);
                    fclose(dst);
                    return CM_EIO;
                }
            const bool bad = ferror(src) != 0;
            fclose(src);
            if (bad) {
                fclose(dst);
                return CM_EIO;
            }
        }
        if (fclose(dst) != 0) return CM_EIO;
        for (int r = 0; r < world; ++r) remove((name + ".part" + std::to_string(r)).c_str());
    }
    return CM_OK;
}

// sizeof of every struct that crosses the ABI, in the order of the header: lets a binding (ctypes, cgo, JNI) check its mirrors
extern "C" int cm_abi_sizes(uint32_t *out, uint32_t cap) {
    const uint32_t v[] = {(uint32_t)sizeof(cm_params),       (uint32_t)sizeof(cm_index_view),  (uint32_t)sizeof(cm_annot_view), (uint32_t)sizeof(cm_mapped_read),
                          (uint32_t)sizeof(cm_reads),        (uint32_t)sizeof(cm_record),      (uint32_t)sizeof(cm_chr_info),   (uint32_t)sizeof(cm_fastq_batch),
                          (uint32_t)sizeof(cm_mapping_args), (uint32_t)sizeof(cm_mapping_stats), (uint32_t)sizeof(cm_circ_res), (uint32_t)sizeof(cm_circ_args),
                          (uint32_t)sizeof(cm_circ_stats), (uint32_t)sizeof(cm_index_raw)};
    const uint32_t n = (uint32_t)(sizeof v / sizeof v[0]);
    if (!out || cap < n) return CM_EINVAL;
    for (uint32_t i = 0; i < n; ++i) out[i] = v[i];
    return (int)n;
}
