// Host-side GTF -> flattened annotation builder (stays on host).
//
// Restates the parts of GTFParser::load_gtf that the mapping hot path queries
// (reference src/gene_annotation.cpp:79-143 tokenizer, :182-189 chrloc2conloc, :191-399 load_gtf,
// src/interval_tree_impl.h:40-127 FlatIntervalTree::build / handle_overlap, :186-242
// build_trans2seg_table) and writes them as the CSR arrays of cm_annot_view.
// The gene interval tree (genes_int_map) is stage-2 only and is not built.
//
// Quirks kept on purpose:
//  * the CR/LF strip in tokenize() never runs because load_gtf passes a never-set (zero) member
//    `len` (gene_annotation.cpp:207, gene_annotation.h:51,61-62): the trailing "\n" stays in the
//    last attribute token;
//  * attribute tokens are split on ' ', ';' and '"' with empty tokens dropped, then read as
//    (key, value) pairs at even/odd positions (gene_annotation.cpp:106-138);
//  * near-border flanks use uint32 arithmetic: `maxM(0, start - maxReadLength)` wraps for exons
//    closer than maxReadLength to the contig start, so those exons get no left flank
//    (gene_annotation.cpp:273,276);
//  * transcript / gene ids are running per-contig counters taken when an exon row is seen
//    (gene_annotation.cpp:281-282).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <utility>
#include <vector>

#include "circminer_hot.h"

namespace {

struct Seg {  // UniqSeg, reference src/common.h:227-251, order src/common.cpp:110-118
    uint32_t start, end, next_exon_beg, gene_id;
    bool operator<(const Seg &r) const {
        if (start != r.start) return start < r.start;
        if (end != r.end) return end < r.end;
        if (gene_id != r.gene_id) return gene_id < r.gene_id;
        return next_exon_beg > r.next_exon_beg;
    }
};

struct Interval {
    uint32_t spos, epos;
    std::vector<uint32_t> segs;  // indices into the contig's unique-segment table
};

struct ExonRec {
    uint32_t start = 0, end = 0, next_start = 0, prev_end = 0;
    uint32_t gene_id_int = 0, trans_id_int = 0;
    int chr_id = 0;
    bool forward = true;
    bool is_exon = false;  // prev_record->type == "exon"
};

struct ContigBuild {
    uint32_t n_gene = 0, n_trans = 0;
    std::vector<uint32_t> gene_start, gene_end;
    std::map<Seg, std::vector<uint32_t>> merged;  // merged_exons[con]: seg -> trans ids
    std::map<std::pair<uint32_t, uint32_t>, uint32_t> merged_genes;   // (start, end) -> gene id of the first gene with that span
    std::vector<uint64_t> near, intr;
    bool has_gene = false;
};

// GTFParser::tokenize with len == 0 on the tab-separated columns: consecutive delimiters collapse (empty fields are
// dropped), no CR/LF strip.  Fields are (pointer, length) views into the line; only the first `cap` are kept.
struct Field {
    const char *p = nullptr;
    size_t n = 0;
    bool is(const char *lit) const { return strlen(lit) == n && memcmp(p, lit, n) == 0; }
};
size_t split_tabs(const char *line, Field *out, size_t cap) {
    size_t nf = 0;
    const char *q = line;
    while (*q && nf < cap) {
        while (*q == '\t') ++q;
        if (!*q) break;
        const char *b = q;
        while (*q && *q != '\t') ++q;
        out[nf].p = b;
        out[nf].n = (size_t)(q - b);
        ++nf;
    }
    for (size_t k = nf; k < cap; ++k) out[k] = Field();
    return nf;
}
// atoi on a field view (leading blanks, optional sign, digits; stops at the first other byte)
inline int field_atoi(const Field &f) {
    char buf[24];
    const size_t n = f.n < sizeof(buf) - 1 ? f.n : sizeof(buf) - 1;
    memcpy(buf, f.p, n);
    buf[n] = 0;
    return atoi(buf);
}

// bits [lo, hi_incl] of a little-endian word bitset; whole words at a time (the reference's per-position loops,
// gene_annotation.cpp:236-238,269-278, are what makes its GTF load slow on long genes)
inline void set_bits(std::vector<uint64_t> &bs, uint64_t lo, uint64_t hi_incl, bool v) {
    const uint64_t nbits = bs.size() * 64;
    if (lo >= nbits || hi_incl < lo) return;
    if (hi_incl >= nbits) hi_incl = nbits - 1;
    const uint64_t w0 = lo >> 6, w1 = hi_incl >> 6;
    const uint64_t m0 = ~0ull << (lo & 63), m1 = ~0ull >> (63 - (hi_incl & 63));
    auto apply = [&](uint64_t w, uint64_t m) {
        if (v) bs[w] |= m;
        else bs[w] &= ~m;
    };
    if (w0 == w1) {
        apply(w0, m0 & m1);
        return;
    }
    apply(w0, m0);
    for (uint64_t w = w0 + 1; w < w1; ++w) bs[w] = v ? ~0ull : 0ull;
    apply(w1, m1);
}

// FlatIntervalTree::handle_overlap, reference src/interval_tree_impl.h:40-95
struct Span {
    uint32_t start, end;
};
bool handle_overlap(std::vector<Interval> &iv, int &cur, uint32_t fresh_idx, const Span &fresh) {
    if (iv[cur].spos < fresh.start) {
        uint32_t pre_epos = iv[cur].epos;
        iv[cur].epos = fresh.start - 1;
        Interval ov;
        ov.spos = fresh.start;
        ov.epos = pre_epos < fresh.end ? pre_epos : fresh.end;
        ov.segs = iv[cur].segs;
        ov.segs.push_back(fresh_idx);
        iv.insert(iv.begin() + cur + 1, ov);
        if (pre_epos < fresh.end) {
            cur += 2;
            return true;
        }
        if (pre_epos == fresh.end) return false;
        Interval tail;
        tail.spos = fresh.end + 1;
        tail.epos = pre_epos;
        tail.segs = iv[cur].segs;
        iv.insert(iv.begin() + cur + 2, tail);
        return false;
    }
    if (iv[cur].epos < fresh.end) {
        iv[cur].segs.push_back(fresh_idx);
        ++cur;
        return true;
    }
    if (iv[cur].epos == fresh.end) {
        iv[cur].segs.push_back(fresh_idx);
        return false;
    }
    Interval ov;
    ov.spos = iv[cur].spos;
    ov.epos = fresh.end;
    ov.segs = iv[cur].segs;
    ov.segs.push_back(fresh_idx);
    iv[cur].spos = fresh.end + 1;
    iv.insert(iv.begin() + cur, ov);
    return false;
}

// FlatIntervalTree::build, interval_tree_impl.h:98-127: spans in the order of the reference's std::map
std::vector<Interval> build_intervals(const std::vector<Span> &spans) {
    std::vector<Interval> iv;
    size_t j = 0;
    for (uint32_t si = 0; si < spans.size(); ++si) {
        const Span &s = spans[si];
        while (j < iv.size() && s.start > iv[j].epos) ++j;
        if (j == iv.size()) {
            iv.push_back(Interval{s.start, s.end, {si}});
        } else {
            int curi = (int)j;
            bool rem = false;
            while (curi < (int)iv.size()) {
                rem = handle_overlap(iv, curi, si, s);
                if (!rem) break;
            }
            if (curi == (int)iv.size() && rem) iv.push_back(Interval{iv[curi - 1].epos + 1, s.end, {si}});
        }
    }
    return iv;
}

template <class T>
T *dup(const std::vector<T> &v) {
    T *p = (T *)malloc((v.size() ? v.size() : 1) * sizeof(T));
    if (p && !v.empty()) memcpy(p, v.data(), v.size() * sizeof(T));
    return p;
}

}  // namespace

extern "C" int cm_host_build_annotation(const char *gtf_path, const cm_chr_info *chrs, uint32_t n_chr,
                                        const uint32_t *contig_len, uint32_t n_contigs, int32_t max_read_len,
                                        cm_annot_view *out) {
    if (!gtf_path || !chrs || !contig_len || !out || n_contigs == 0) return CM_EINVAL;
    FILE *fp = fopen(gtf_path, "r");
    if (!fp) return CM_EINVAL;

    std::map<std::string, std::pair<int, uint32_t>> chr2con;  // name -> (contig idx, shift)
    for (uint32_t i = 0; i < n_chr; ++i) chr2con[chrs[i].name] = {(int)chrs[i].contig_id - 1, chrs[i].start_pos};

    std::vector<ContigBuild> cb(n_contigs);
    for (uint32_t c = 0; c < n_contigs; ++c) {
        size_t words = ((size_t)contig_len[c] + 64 + 63) / 64;
        cb[c].near.assign(words, 0);
        cb[c].intr.assign(words, 0);
    }

    ExonRec prev, cur;
    auto flush_prev = [&]() {  // add2merged_exons, gene_annotation.cpp:167-180
        Seg s{prev.start, prev.end, prev.next_start, prev.gene_id_int};
        cb[prev.chr_id].merged[s].push_back(prev.trans_id_int);
    };

    char *line = nullptr;
    size_t cap = 0;
    Field f[8];
    std::string chr_key;
    auto it = chr2con.end();
    const uint32_t mrl = (uint32_t)max_read_len;
    while (getline(&line, &cap, fp) != -1) {
        if (line[0] == '#') continue;
        split_tabs(line, f, 8);
        const bool is_gene = f[2].is("gene"), is_trans = f[2].is("transcript"), is_exon = f[2].is("exon");
        if (!is_gene && !is_trans && !is_exon) continue;
        if (it == chr2con.end() || chr_key.size() != f[0].n || memcmp(chr_key.data(), f[0].p, f[0].n) != 0) {
            chr_key.assign(f[0].p ? f[0].p : "", f[0].n);
            it = chr2con.find(chr_key);
        }
        if (it == chr2con.end()) continue;  // chr = "0" -> tmp_chr < 0
        int con = it->second.first;
        if (con < 0 || con >= (int)n_contigs) continue;
        uint32_t start = (uint32_t)field_atoi(f[3]) + it->second.second;
        uint32_t end = (uint32_t)field_atoi(f[4]) + it->second.second;
        bool fwd = f[6].is("+");
        ContigBuild &B = cb[con];

        if (is_gene) {
            ++B.n_gene;
            B.has_gene = true;
            set_bits(B.intr, start, end, true);
            B.gene_start.push_back(start);
            B.gene_end.push_back(end);
            // merged_genes, gene_annotation.cpp:243-256: GeneInfo orders by (start, end) only, so a second gene with the same
            // span is folded into the first one's key (its id is not kept)
            B.merged_genes.emplace(std::make_pair(start, end), B.n_gene - 1);
        }
        if (is_trans) ++B.n_trans;

        if (is_exon) {
            set_bits(B.intr, start, end, false);
            // uint32 wrap quirk: no flank when the subtraction underflows
            uint32_t lo1 = start - mrl;
            if (lo1 < start) set_bits(B.near, lo1, (uint64_t)start - 1, true);
            uint32_t lo2 = end - mrl + 1;
            if (lo2 <= end) set_bits(B.near, lo2, end, true);

            cur.start = start;
            cur.end = end;
            cur.chr_id = con;
            cur.forward = fwd;
            cur.trans_id_int = B.n_trans - 1;
            cur.gene_id_int = B.n_gene - 1;
            if (!prev.is_exon) {
                prev = cur;
                prev.is_exon = true;
                prev.next_start = 0;
                prev.prev_end = 0;
                continue;
            }
            if (prev.forward) {
                prev.next_start = cur.start;
                cur.prev_end = prev.end;
            } else {
                prev.prev_end = cur.end;
                cur.next_start = prev.start;
            }
            flush_prev();
            prev = cur;  // copy_seg: next_start / prev_end travel with the record
            prev.is_exon = true;
        } else if (prev.is_exon) {
            if (prev.forward) prev.next_start = 0;
            else prev.prev_end = 0;
            flush_prev();
            prev.is_exon = false;
        }
    }
    if (prev.is_exon) {
        if (prev.forward) prev.next_start = 0;
        else prev.prev_end = 0;
        flush_prev();
    }
    free(line);
    fclose(fp);

    for (uint32_t c = 0; c < n_contigs; ++c) {
        ContigBuild &B = cb[c];
        // unique segments in map order
        std::vector<Seg> segs;
        std::vector<std::vector<uint32_t>> seg_tids;
        for (auto &kv : B.merged) {
            segs.push_back(kv.first);
            seg_tids.push_back(kv.second);
        }
        std::vector<Span> spans;
        for (const Seg &g : segs) spans.push_back(Span{g.start, g.end});
        std::vector<Interval> iv = build_intervals(spans);
        // build_trans2seg_table, interval_tree_impl.h:186-242
        const uint32_t nt = B.n_trans;
        std::vector<int32_t> starts(nt, 1000000000), ends(nt, 0);
        std::vector<uint32_t> mx_end(iv.size(), 0), mn_end(iv.size(), 1000000000u), mx_next(iv.size(), 0);
        for (size_t i = 0; i < iv.size(); ++i)
            for (uint32_t si : iv[i].segs) {
                if (segs[si].end > mx_end[i]) mx_end[i] = segs[si].end;
                if (segs[si].end < mn_end[i]) mn_end[i] = segs[si].end;
                if (segs[si].next_exon_beg > mx_next[i]) mx_next[i] = segs[si].next_exon_beg;
                for (uint32_t t : seg_tids[si]) {
                    if ((int32_t)i < starts[t]) starts[t] = (int32_t)i;
                    if ((int32_t)i > ends[t]) ends[t] = (int32_t)i;
                }
            }
        std::vector<uint32_t> t2s_off(nt + 1, 0);
        for (uint32_t t = 0; t < nt; ++t) {
            int32_t s = ends[t] - starts[t] + 1;
            if (s < 0) s = 0;  // transcript without exon rows (the reference would crash here)
            t2s_off[t + 1] = t2s_off[t] + (uint32_t)s;
        }
        std::vector<uint8_t> t2s(t2s_off[nt], 0);
        for (size_t i = 0; i < iv.size(); ++i)
            for (uint32_t si : iv[i].segs) {
                uint8_t st = (iv[i].spos == segs[si].start) ? 1 : ((iv[i].epos == segs[si].end) ? 3 : 2);
                for (uint32_t t : seg_tids[si]) t2s[t2s_off[t] + ((int32_t)i - starts[t])] = st;
            }
        // dummy interval for annotation-less contigs, gene_annotation.cpp:368-382
        if (iv.empty()) {
            segs.push_back(Seg{0xffffffffu, 0xffffffffu, 0, 0});
            seg_tids.push_back({});
            iv.push_back(Interval{0xffffffffu, 0xffffffffu, {(uint32_t)segs.size() - 1}});
            mx_end.push_back(0);
            mn_end.push_back(0);
            mx_next.push_back(0);
        }
        // ---- flatten ----
        std::vector<uint32_t> iv_spos, iv_epos, iv_seg_off{0}, iv_seg;
        for (auto &x : iv) {
            iv_spos.push_back(x.spos);
            iv_epos.push_back(x.epos);
            for (uint32_t si : x.segs) iv_seg.push_back(si);
            iv_seg_off.push_back((uint32_t)iv_seg.size());
        }
        std::vector<uint32_t> s_start, s_end, s_next, s_gene, s_toff{0}, s_tid;
        for (size_t si = 0; si < segs.size(); ++si) {
            s_start.push_back(segs[si].start);
            s_end.push_back(segs[si].end);
            s_next.push_back(segs[si].next_exon_beg);
            s_gene.push_back(segs[si].gene_id);
            for (uint32_t t : seg_tids[si]) s_tid.push_back(t);
            s_toff.push_back((uint32_t)s_tid.size());
        }
        std::vector<uint32_t> cshift;
        std::vector<int32_t> cid;
        for (uint32_t i = 0; i < n_chr; ++i)
            if (chrs[i].contig_id == c + 1) {
                cshift.push_back(chrs[i].start_pos);
                cid.push_back((int32_t)i);
            }
        // genes_int_map (stage 2: get_gene_overlap): the same interval construction over the gene spans
        std::vector<uint32_t> giv_spos, giv_epos, giv_off{0}, giv_gene;
        {
            std::vector<Span> gspans;
            std::vector<uint32_t> gid;
            for (auto &kv : B.merged_genes) {
                gspans.push_back(Span{kv.first.first, kv.first.second});
                gid.push_back(kv.second);
            }
            std::vector<Interval> giv = build_intervals(gspans);
            for (auto &x : giv) {
                giv_spos.push_back(x.spos);
                giv_epos.push_back(x.epos);
                for (uint32_t k : x.segs) giv_gene.push_back(gid[k]);
                giv_off.push_back((uint32_t)giv_gene.size());
            }
            if (giv.empty()) {                     // add_dummy_interval(temp_gene), gene_annotation.cpp:371-378: {MAXUB, MAXUB, gene 0}
                giv_spos.push_back(0xffffffffu);
                giv_epos.push_back(0xffffffffu);
                giv_gene.push_back(0);
                giv_off.push_back(1);
            }
        }
        cm_annot_view &A = out[c];
        memset(&A, 0, sizeof(A));
        A.n_giv = (uint32_t)giv_spos.size();
        A.giv_spos = dup(giv_spos);
        A.giv_epos = dup(giv_epos);
        A.giv_gene_off = dup(giv_off);
        A.giv_gene = dup(giv_gene);
        A.n_iv = (uint32_t)iv.size();
        A.iv_spos = dup(iv_spos);
        A.iv_epos = dup(iv_epos);
        A.iv_max_end = dup(mx_end);
        A.iv_min_end = dup(mn_end);
        A.iv_max_next_exon = dup(mx_next);
        A.iv_seg_off = dup(iv_seg_off);
        A.iv_seg = dup(iv_seg);
        A.n_seg = (uint32_t)segs.size();
        A.seg_start = dup(s_start);
        A.seg_end = dup(s_end);
        A.seg_next_exon_beg = dup(s_next);
        A.seg_gene_id = dup(s_gene);
        A.seg_tid_off = dup(s_toff);
        A.seg_tid = dup(s_tid);
        A.n_trans = nt;
        A.trans_start_ind = dup(starts);
        A.t2s_off = dup(t2s_off);
        A.t2s = dup(t2s);
        A.n_gene = B.n_gene;
        A.gene_start = dup(B.gene_start);
        A.gene_end = dup(B.gene_end);
        A.n_bits = (uint64_t)B.near.size() * 64;
        A.near_border_bits = dup(B.near);
        A.intronic_bits = dup(B.intr);
        A.n_chr = (uint32_t)cshift.size();
        A.chr_shift = dup(cshift);
        A.chr_id = dup(cid);
        {   // bucketed index over the interval starts (query accelerator, see circminer_hot.h)
            const uint32_t shift = 10;
            const uint64_t nbk = (A.n_bits >> shift) + 2;
            std::vector<uint32_t> bk(nbk);
            size_t k = 0;
            for (uint64_t b = 0; b < nbk; ++b) {
                const uint64_t lim = b << shift;
                while (k < iv.size() && (uint64_t)iv[k].spos < lim) ++k;
                bk[b] = (uint32_t)k;
            }
            A.iv_bucket = dup(bk);
            A.iv_bucket_shift = shift;
            A.n_iv_bucket = (uint32_t)nbk;
        }
    }
    return CM_OK;
}

// GTFParser::get_gene_overlap(loc, false) + FlatIntervalTree::find, gene_annotation.cpp:572-585, interval_tree_impl.h:136-162
extern "C" int cm_host_gene_overlap(const cm_annot_view *av, uint32_t pos, const uint32_t **genes, uint32_t *n_genes) {
    if (!av || !genes || !n_genes) return CM_EINVAL;
    *genes = nullptr;
    *n_genes = 0;
    if (av->n_giv == 0 || pos < av->giv_spos[0]) return CM_OK;
    uint32_t lo = 0, hi = av->n_giv;                 // number of intervals with spos <= pos
    while (lo < hi) {
        const uint32_t mid = (lo + hi) / 2;
        if (av->giv_spos[mid] <= pos) lo = mid + 1;
        else hi = mid;
    }
    if (lo == 0 || av->giv_epos[lo - 1] < pos) return CM_OK;
    *genes = av->giv_gene + av->giv_gene_off[lo - 1];
    *n_genes = av->giv_gene_off[lo] - av->giv_gene_off[lo - 1];
    return CM_OK;
}

extern "C" void cm_host_free_annotation(cm_annot_view *av, uint32_t n_contigs) {
    if (!av) return;
    for (uint32_t c = 0; c < n_contigs; ++c) {
        cm_annot_view &A = av[c];
        const void *ptrs[] = {A.iv_spos, A.iv_epos, A.iv_max_end, A.iv_min_end, A.iv_max_next_exon, A.iv_seg_off,
                              A.iv_seg, A.seg_start, A.seg_end, A.seg_next_exon_beg, A.seg_gene_id, A.seg_tid_off,
                              A.seg_tid, A.trans_start_ind, A.t2s_off, A.t2s, A.gene_start, A.gene_end,
                              A.near_border_bits, A.intronic_bits, A.chr_shift, A.chr_id, A.iv_bucket, A.giv_spos, A.giv_epos,
                              A.giv_gene_off, A.giv_gene};
        for (const void *p : ptrs) free((void *)p);
        memset(&A, 0, sizeof(A));
    }
}
