// ============================================================================================
// cm_oracle_build.cpp — the oracle's OWN construction of the inputs the mapping path reads: the k-mer table of one packed
// contig and the flattened GTF model.  TEST INFRASTRUCTURE ONLY (same rules as cm_oracle.cpp).
//
// Written straight from the reference, function by function, with its containers (std::map keyed by UniqSeg / GeneInfo, vectors
// of IntervalInfo with copied seg_lists), NOT from circminer_amd/csrc/host_index.cpp / host_annot.cpp: the parity tests feed
// the oracle from these builders and the HIP path from the product's, so a bug in a product builder shows up as a parity failure
// instead of being shared by both sides ("What's weak" #1).
//
//   oracle_build_index       src/mrsfast/HashTable.c:769-839 (scatter of every valid k-mer into its 14-mer bucket) and
//                            src/mrsfast/Sort.c:116-117 (per-bucket order: checksum, then position)
//   oracle_build_annotation  GTFParser::load_gtf src/gene_annotation.cpp:191-399 (tokenize :79-100, parse_gtf_rec :103-143,
//                            chrloc2conloc :182-189, add2merged_exons :167-180), FlatIntervalTree::build / handle_overlap /
//                            shift_right / build_trans2seg_table / add_dummy_interval src/interval_tree_impl.h:17-127,186-242,
//                            UniqSeg / GeneInfo order src/common.cpp:74-78,110-118
// PARITY UNPINNED like the rest of the oracle: no reference output exists to compare with.
// ============================================================================================
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <sstream>
#include <string>
#include <vector>

#include "circminer_hot.h"

namespace {

const uint32_t MAXUB = 4294967295u;

// ---------------------------------------------------------------- k-mer table
int code_of(uint8_t ch) {   // hashVal / checkSumVal alphabet (upper case only)
    switch (ch) { case 'A': return 0; case 'C': return 1; case 'G': return 2; case 'T': return 3; default: return -1; }
}

// ---------------------------------------------------------------- GTF model (reference containers)
struct UniqSeg {            // src/common.h:227-251
    uint32_t start = 0, end = 0, next_exon_beg = 0, gene_id = 0;
    std::vector<uint32_t> trans_id;
    bool operator<(const UniqSeg &r) const {      // src/common.cpp:110-118
        if (start != r.start) return start < r.start;
        if (end != r.end) return end < r.end;
        if (gene_id != r.gene_id) return gene_id < r.gene_id;
        return next_exon_beg > r.next_exon_beg;
    }
};
struct GeneInfo {           // src/common.h:174-184
    uint32_t start = 0, end = 0, gene_id = 0;
    bool operator<(const GeneInfo &g) const { return start != g.start ? start < g.start : end < g.end; }   // src/common.cpp:74-78
};
template <class T> struct IntervalInfo {          // src/interval_info.h
    uint32_t spos, epos, max_end = 0, min_end = 0, max_next_exon = 0;
    std::vector<T> seg_list;
    explicit IntervalInfo(const T &seg) : spos(seg.start), epos(seg.end), seg_list(1, seg) {}
    IntervalInfo(uint32_t s, uint32_t e, const std::vector<T> &l) : spos(s), epos(e), seg_list(l) {}
    IntervalInfo(uint32_t s, uint32_t e, const std::vector<T> &l, const T &seg) : spos(s), epos(e), seg_list(l) { seg_list.push_back(seg); }
};
template <class T> struct FlatIntervalTree {      // src/interval_tree_impl.h
    std::vector<IntervalInfo<T>> disjoint_intervals;
    void shift_right(int ind) {                                                   // :27-35
        int size = (int)disjoint_intervals.size();
        disjoint_intervals.push_back(disjoint_intervals[size - 1]);
        for (int i = size - 1; i > ind; i--) disjoint_intervals[i] = disjoint_intervals[i - 1];
    }
    bool handle_overlap(int &cur_ind, const T &fresh) {                           // :40-95
        IntervalInfo<T> *main = &disjoint_intervals[cur_ind];
        int new_interval_ind;
        if (main->spos < fresh.start) {
            uint32_t pre_main_epos = main->epos;
            main->epos = fresh.start - 1;
            new_interval_ind = cur_ind + 1;
            shift_right(new_interval_ind);
            main = &disjoint_intervals[cur_ind];
            uint32_t end = (pre_main_epos < fresh.end) ? pre_main_epos : fresh.end;
            disjoint_intervals[new_interval_ind] = IntervalInfo<T>(fresh.start, end, main->seg_list, fresh);
            if (pre_main_epos < fresh.end) { cur_ind += 2; return true; }
            else if (pre_main_epos == fresh.end) return false;
            else {
                new_interval_ind++;
                shift_right(new_interval_ind);
                main = &disjoint_intervals[cur_ind];
                disjoint_intervals[new_interval_ind] = IntervalInfo<T>(fresh.end + 1, pre_main_epos, main->seg_list);
                return false;
            }
        } else {
            if (main->epos < fresh.end) { main->seg_list.push_back(fresh); cur_ind++; return true; }
            else if (main->epos == fresh.end) { main->seg_list.push_back(fresh); return false; }
            else {
                uint32_t pre_main_spos = main->spos;
                main->spos = fresh.end + 1;
                new_interval_ind = cur_ind;
                shift_right(new_interval_ind);
                main = &disjoint_intervals[cur_ind];
                disjoint_intervals[new_interval_ind] = IntervalInfo<T>(pre_main_spos, fresh.end, main->seg_list, fresh);
                return false;
            }
        }
    }
    template <class V> void build(std::map<T, V> &sorted_list) {                   // :97-127
        size_t j = 0;
        for (auto it = sorted_list.begin(); it != sorted_list.end(); it++) {
            while (j < disjoint_intervals.size() && it->first.start > disjoint_intervals[j].epos) j++;
            if (j == disjoint_intervals.size()) disjoint_intervals.push_back(IntervalInfo<T>(it->first));
            else {
                int curr = (int)j;
                bool overlap_remained = false;
                while (curr < (int)disjoint_intervals.size()) {
                    overlap_remained = handle_overlap(curr, it->first);
                    if (!overlap_remained) break;
                }
                if (curr == (int)disjoint_intervals.size() && overlap_remained) {
                    IntervalInfo<T> t(it->first);
                    t.spos = disjoint_intervals[curr - 1].epos + 1;
                    disjoint_intervals.push_back(t);
                }
            }
        }
    }
};

// tokenize with len == 0 (the strip loop never runs), src/gene_annotation.cpp:79-100
void tokenize(const char *line, const std::string &delim, std::vector<std::string> &f) {
    std::string cur;
    size_t k = 0;
    for (const char *c = line; *c; ++c) {
        if (delim.find(*c) != std::string::npos) {
            if (k < f.size()) f[k] = cur;
            if (cur != "") k++;
            cur = "";
        } else cur += *c;
    }
    if (cur != "" && k < f.size()) f[k++] = cur;
}

struct GTFRecord {
    std::string chr, type;
    uint32_t start = 0, end = 0, next_start = 0, prev_end = 0;
    int gene_id_int = 0, trans_id_int = 0, chr_id = 0;
    bool forward_strand = true;
};

struct Holder {             // keeps the flattened arrays of one cm_annot_view alive
    std::vector<uint32_t> iv_spos, iv_epos, iv_max_end, iv_min_end, iv_max_next, iv_seg_off, iv_seg, seg_start, seg_end, seg_next, seg_gene, seg_tid_off,
        seg_tid, t2s_off, gene_start, gene_end, chr_shift, giv_spos, giv_epos, giv_gene_off, giv_gene;
    std::vector<int32_t> trans_start, chr_id;
    std::vector<uint8_t> t2s;
    std::vector<uint64_t> near, intr;
};

}  // namespace

extern "C" {

// generateHashTableOnDisk's table for one contig, flattened like cm_index_view (single thread, simple on purpose)
int oracle_build_index(const uint8_t *genome, uint32_t ref_len, int32_t kmer, int32_t contig_num, cm_index_view *out) {
    const int W = CM_WINDOW_SIZE, c = kmer - W;
    if (c < 0 || c > 8) return CM_EINVAL;
    const uint64_t nb = 1ull << (2 * W);
    std::vector<uint32_t> cnt(nb + 1, 0);
    auto kmer_at = [&](uint32_t i, uint32_t &hv, uint32_t &ck) {      // k-mer starting at 0-based i; false when a base is not ACGT
        if ((uint64_t)i + (uint32_t)kmer > ref_len) return false;
        uint64_t v = 0;
        for (int x = 0; x < kmer; ++x) {
            int b = code_of(genome[i + x]);
            if (b < 0) return false;
            v = (v << 2) | (uint64_t)b;
        }
        hv = (uint32_t)(v >> (2 * c));
        ck = (uint32_t)(v & ((1ull << (2 * c)) - 1));
        return true;
    };
    uint32_t hv, ck;
    for (uint32_t i = 0; i < ref_len; ++i) if (kmer_at(i, hv, ck)) ++cnt[hv + 1];
    for (uint64_t h = 0; h < nb; ++h) cnt[h + 1] += cnt[h];
    const uint64_t total = cnt[nb];
    uint32_t *off = (uint32_t *)malloc((nb + 1) * sizeof(uint32_t));
    uint16_t *cs = (uint16_t *)malloc((total ? total : 1) * sizeof(uint16_t));
    uint32_t *ps = (uint32_t *)malloc((total ? total : 1) * sizeof(uint32_t));
    memcpy(off, cnt.data(), (nb + 1) * sizeof(uint32_t));
    std::vector<uint32_t> cur(cnt.begin(), cnt.end() - 1);
    for (uint32_t i = 0; i < ref_len; ++i)
        if (kmer_at(i, hv, ck)) {
            const uint32_t w = cur[hv]++;
            cs[w] = (uint16_t)ck;
            ps[w] = i + 1;                                            // 1-based start, HashTable.c:806
        }
    std::vector<std::pair<uint16_t, uint32_t>> tmp;
    for (uint64_t h = 0; h < nb; ++h) {
        const uint32_t a = off[h], b = off[h + 1];
        if (b - a < 2) continue;
        tmp.clear();
        for (uint32_t i = a; i < b; ++i) tmp.push_back({cs[i], ps[i]});
        std::sort(tmp.begin(), tmp.end());                            // (checksum, info), Sort.c:116-117
        for (uint32_t i = a; i < b; ++i) { cs[i] = tmp[i - a].first; ps[i] = tmp[i - a].second; }
    }
    out->contig_num = contig_num;
    out->ref_len = ref_len;
    out->genome = genome;
    out->bucket_off = off;
    out->checksum = cs;
    out->pos = ps;
    out->n_entries = total;
    return 0;
}
void oracle_free_index(cm_index_view *iv) {
    free((void *)iv->bucket_off);
    free((void *)iv->checksum);
    free((void *)iv->pos);
    iv->bucket_off = nullptr; iv->checksum = nullptr; iv->pos = nullptr;
}

// load_gtf for every packed contig.  `holders` (opaque, n_contigs entries) own the arrays; free with oracle_free_annotation.
int oracle_build_annotation(const char *gtf_path, const cm_chr_info *chrs, uint32_t n_chr, const uint32_t *contig_len, uint32_t n_contigs,
                            int32_t max_read_len, cm_annot_view *out, void **holders) {
    FILE *fp = fopen(gtf_path, "r");
    if (!fp) return CM_EINVAL;
    struct ConShift { int contig; uint32_t shift; };
    std::map<std::string, ConShift> chr2con;                                      // set_contig_shift, :424-449
    for (uint32_t i = 0; i < n_chr; ++i) chr2con[chrs[i].name] = ConShift{(int)chrs[i].contig_id, chrs[i].start_pos};
    const uint32_t contig_cnt = n_contigs;
    std::vector<int> n_gene(contig_cnt, 0), n_trans(contig_cnt, 0);
    std::vector<std::vector<GeneInfo>> gid2ginfo(contig_cnt);
    std::vector<std::map<UniqSeg, int>> merged_exons(contig_cnt);
    std::vector<std::map<GeneInfo, int>> merged_genes(contig_cnt);
    std::vector<std::vector<uint64_t>> near_bs(contig_cnt), intr_bs(contig_cnt);
    for (uint32_t c = 0; c < contig_cnt; ++c) {
        const size_t words = ((size_t)contig_len[c] + 64 + 63) / 64;             // layout of cm_annot_view (the reference: bitset<DEF_CONTIG_MAX_SIZE>)
        near_bs[c].assign(words, 0);
        intr_bs[c].assign(words, 0);
    }
    auto setbit = [](std::vector<uint64_t> &b, uint64_t k, bool v) {
        if (k / 64 >= b.size()) return;
        if (v) b[k / 64] |= 1ull << (k & 63);
        else b[k / 64] &= ~(1ull << (k & 63));
    };
    auto add2merged_exons = [](std::map<UniqSeg, int> &m, UniqSeg seg, const GTFRecord &rec) {     // :167-180
        auto it = m.find(seg);
        if (it != m.end()) {
            seg = it->first;
            seg.trans_id.push_back((uint32_t)rec.trans_id_int);
            m.erase(it);
            m[seg] = 0;
        } else {
            seg.trans_id.assign(1, (uint32_t)rec.trans_id_int);
            m[seg] = 0;
        }
    };
    GTFRecord cur, prev;
    prev.type = "";
    UniqSeg seg;
    char *line = nullptr;
    size_t cap = 0;
    const uint32_t maxReadLength = (uint32_t)max_read_len;
    auto flush_prev = [&]() {
        seg.start = prev.start; seg.end = prev.end; seg.gene_id = (uint32_t)prev.gene_id_int; seg.next_exon_beg = prev.next_start;
        add2merged_exons(merged_exons[prev.chr_id], seg, prev);
    };
    while (getline(&line, &cap, fp) != -1) {
        if (line[0] == '#') continue;                                            // read_next :59-69
        std::vector<std::string> f(10);
        tokenize(line, "\t", f);                                                 // parse_gtf_rec :103-143 (attributes are not queried by the mapping path)
        if (!(f[2] == "gene" || f[2] == "transcript" || f[2] == "exon")) continue;
        cur.chr = f[0]; cur.type = f[2];
        cur.start = (uint32_t)atoi(f[3].c_str()); cur.end = (uint32_t)atoi(f[4].c_str());
        cur.forward_strand = (f[6] == "+");
        auto cc = chr2con.find(cur.chr);                                         // chrloc2conloc :182-189
        int tmp_chr = -1;
        if (cc != chr2con.end()) { cur.start += cc->second.shift; cur.end += cc->second.shift; tmp_chr = cc->second.contig - 1; }
        if (tmp_chr < 0 || (uint32_t)tmp_chr >= contig_cnt) continue;
        cur.chr_id = tmp_chr;
        const int C = cur.chr_id;
        if (cur.type == "gene") {
            ++n_gene[C];
            for (uint64_t k = cur.start; k <= cur.end; k++) setbit(intr_bs[C], k, true);
            GeneInfo g;
            g.start = cur.start; g.end = cur.end; g.gene_id = (uint32_t)gid2ginfo[C].size();
            gid2ginfo[C].push_back(g);
            if (merged_genes[C].find(g) == merged_genes[C].end()) merged_genes[C][g] = 0;   // a second gene with the same span keeps the first key
        }
        if (cur.type == "transcript") ++n_trans[C];
        if (cur.type == "exon") {
            for (uint64_t k = cur.start; k <= cur.end; k++) setbit(intr_bs[C], k, false);
            // uint32 arithmetic as in the reference: maxM(0, start - maxReadLength) wraps for exons near the contig start (no flank then)
            for (uint32_t k = std::max<uint32_t>(0u, cur.start - maxReadLength); k < cur.start; k++) setbit(near_bs[C], k, true);
            for (uint32_t k = std::max<uint32_t>(0u, cur.end - maxReadLength + 1); k <= cur.end; k++) setbit(near_bs[C], k, true);
            cur.trans_id_int = n_trans[C] - 1;
            cur.gene_id_int = n_gene[C] - 1;
            if (prev.type != "exon") {
                prev = cur;
                prev.next_start = 0;
                prev.prev_end = 0;
                continue;
            } else {
                if (prev.forward_strand) { prev.next_start = cur.start; cur.prev_end = prev.end; }
                else { prev.prev_end = cur.end; cur.next_start = prev.start; }
                flush_prev();
                prev = cur;
            }
        } else if (prev.type == "exon") {
            if (prev.forward_strand) prev.next_start = 0;
            else prev.prev_end = 0;
            flush_prev();
            prev.type = "";
        }
    }
    if (prev.type == "exon") {
        if (prev.forward_strand) prev.next_start = 0;
        else prev.prev_end = 0;
        flush_prev();
    }
    free(line);
    fclose(fp);

    for (uint32_t con = 0; con < contig_cnt; ++con) {
        FlatIntervalTree<UniqSeg> exons;
        exons.build(merged_exons[con]);
        FlatIntervalTree<GeneInfo> genes;
        genes.build(merged_genes[con]);
        // build_trans2seg_table :186-242
        const int trans_cnt = n_trans[con];
        std::vector<int> starts(trans_cnt, 1000000000), ends(trans_cnt, 0);
        auto &D = exons.disjoint_intervals;
        for (int i = 0; i < (int)D.size(); i++) {
            uint32_t max_end = 0, min_end = 1000000000u, max_next = 0;
            for (auto &s : D[i].seg_list) {
                max_end = std::max(max_end, s.end); min_end = std::min(min_end, s.end); max_next = std::max(max_next, s.next_exon_beg);
                D[i].max_end = max_end; D[i].min_end = min_end; D[i].max_next_exon = max_next;
                for (uint32_t tid : s.trans_id) { if (i < starts[tid]) starts[tid] = i; if (i > ends[tid]) ends[tid] = i; }
            }
        }
        std::vector<std::vector<uint8_t>> trans2seg(trans_cnt);
        for (int i = 0; i < trans_cnt; i++) {
            int s = ends[i] - starts[i] + 1;
            if (s < 0) s = 0;               // a transcript row without exon rows: resize(negative) in the reference; defined as empty here
            trans2seg[i].assign((size_t)s, 0);
        }
        for (int i = 0; i < (int)D.size(); i++)
            for (auto &s : D[i].seg_list) {
                uint8_t state = (D[i].spos == s.start) ? 1 : ((D[i].epos == s.end) ? 3 : 2);
                for (uint32_t tid : s.trans_id) trans2seg[tid][i - starts[tid]] = state;
            }
        // add_dummy_interval :17-24 with the records of gene_annotation.cpp:368-372
        if (D.empty()) {
            UniqSeg t; t.start = MAXUB; t.end = MAXUB;
            D.push_back(IntervalInfo<UniqSeg>(t));
        }
        auto &G = genes.disjoint_intervals;
        if (G.empty()) {
            GeneInfo t; t.start = MAXUB; t.end = MAXUB; t.gene_id = 0;
            G.push_back(IntervalInfo<GeneInfo>(t));
        }
        // ---- flatten into cm_annot_view: unique segments numbered in merged_exons (map) order, the dummy last
        Holder *H = new Holder();
        holders[con] = H;
        std::map<UniqSeg, uint32_t> seg_no;
        for (auto &kv : merged_exons[con]) {
            const uint32_t id = (uint32_t)seg_no.size();
            seg_no[kv.first] = id;
            H->seg_start.push_back(kv.first.start); H->seg_end.push_back(kv.first.end); H->seg_next.push_back(kv.first.next_exon_beg);
            H->seg_gene.push_back(kv.first.gene_id);
            H->seg_tid_off.push_back((uint32_t)H->seg_tid.size());
            for (uint32_t t : kv.first.trans_id) H->seg_tid.push_back(t);
        }
        if (merged_exons[con].empty()) {
            H->seg_start.push_back(MAXUB); H->seg_end.push_back(MAXUB); H->seg_next.push_back(0); H->seg_gene.push_back(0);
            H->seg_tid_off.push_back(0);
        }
        H->seg_tid_off.push_back((uint32_t)H->seg_tid.size());
        H->iv_seg_off.push_back(0);
        for (auto &iv : D) {
            H->iv_spos.push_back(iv.spos); H->iv_epos.push_back(iv.epos);
            H->iv_max_end.push_back(iv.max_end); H->iv_min_end.push_back(iv.min_end); H->iv_max_next.push_back(iv.max_next_exon);
            for (auto &s : iv.seg_list) H->iv_seg.push_back(merged_exons[con].empty() ? 0u : seg_no[s]);
            H->iv_seg_off.push_back((uint32_t)H->iv_seg.size());
        }
        H->t2s_off.push_back(0);
        for (int i = 0; i < trans_cnt; i++) {
            H->trans_start.push_back(starts[i]);
            for (uint8_t v : trans2seg[i]) H->t2s.push_back(v);
            H->t2s_off.push_back((uint32_t)H->t2s.size());
        }
        for (auto &g : gid2ginfo[con]) { H->gene_start.push_back(g.start); H->gene_end.push_back(g.end); }
        H->giv_gene_off.push_back(0);
        for (auto &iv : G) {
            H->giv_spos.push_back(iv.spos); H->giv_epos.push_back(iv.epos);
            for (auto &g : iv.seg_list) H->giv_gene.push_back(g.gene_id);
            H->giv_gene_off.push_back((uint32_t)H->giv_gene.size());
        }
        for (uint32_t i = 0; i < n_chr; ++i)
            if (chrs[i].contig_id == con + 1) { H->chr_shift.push_back(chrs[i].start_pos); H->chr_id.push_back((int32_t)i); }
        H->near = near_bs[con];
        H->intr = intr_bs[con];
        cm_annot_view &A = out[con];
        memset(&A, 0, sizeof A);
        auto P = [](std::vector<uint32_t> &v) { if (v.empty()) v.push_back(0); return v.data(); };
        A.n_iv = (uint32_t)D.size();
        A.iv_spos = P(H->iv_spos); A.iv_epos = P(H->iv_epos); A.iv_max_end = P(H->iv_max_end); A.iv_min_end = P(H->iv_min_end);
        A.iv_max_next_exon = P(H->iv_max_next); A.iv_seg_off = P(H->iv_seg_off); A.iv_seg = P(H->iv_seg);
        A.n_seg = (uint32_t)H->seg_start.size();
        A.seg_start = P(H->seg_start); A.seg_end = P(H->seg_end); A.seg_next_exon_beg = P(H->seg_next); A.seg_gene_id = P(H->seg_gene);
        A.seg_tid_off = P(H->seg_tid_off); A.seg_tid = P(H->seg_tid);
        A.n_trans = (uint32_t)trans_cnt;
        if (H->trans_start.empty()) H->trans_start.push_back(0);
        if (H->t2s.empty()) H->t2s.push_back(0);
        A.trans_start_ind = H->trans_start.data(); A.t2s_off = P(H->t2s_off); A.t2s = H->t2s.data();
        A.n_gene = (uint32_t)gid2ginfo[con].size();
        A.gene_start = P(H->gene_start); A.gene_end = P(H->gene_end);
        A.n_bits = (uint64_t)H->near.size() * 64;
        A.near_border_bits = H->near.data(); A.intronic_bits = H->intr.data();
        A.n_chr = (uint32_t)H->chr_shift.size();
        if (H->chr_id.empty()) H->chr_id.push_back(0);
        A.chr_shift = P(H->chr_shift); A.chr_id = H->chr_id.data();
        A.iv_bucket = nullptr; A.iv_bucket_shift = 0; A.n_iv_bucket = 0;          // the accelerator table is the product's; the oracle searches plainly
        A.n_giv = (uint32_t)G.size();
        A.giv_spos = P(H->giv_spos); A.giv_epos = P(H->giv_epos); A.giv_gene_off = P(H->giv_gene_off); A.giv_gene = P(H->giv_gene);
    }
    return 0;
}
void oracle_free_annotation(void **holders, uint32_t n) {
    for (uint32_t i = 0; i < n; ++i) { delete (Holder *)holders[i]; holders[i] = nullptr; }
}

}  // extern "C"
