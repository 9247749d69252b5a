// Stage 2, the back-splice-junction calling itself (north_star: "process_circ reporting stay[s] on host").
//
// Replaces ProcessCirc::do_process and what it calls (reference src/process_circ.cpp:195-1552, :1646-1752; chaining of the
// 8-mer seeds src/chain.cpp:310-539; EditDistAlignment src/align.cpp:602-660; helper predicates src/utils.cpp:356-744,
// src/common.cpp:147-243): for every CHIBSJ / CHI2BSJ pair stage 1 left in the sorted remain files, the unmapped part(s) of
// the pair are located inside the overlapping genes, extended along the common transcripts, and the junction the two pieces
// imply is classified (FR / RF ordinary pairs, CR circular with annotated ends, NCR / MCR novel, UD undefined).  Output:
// <out>.candidates.pam (one row per evaluated split mapping) and <out>.circ_report (cm_circ_report, host_circ.cpp).
//
// Building blocks come from the product's own kernel bodies (cm_core.h compiled for the host in its own namespace with
// CM_STAGE2_HOST: interval queries, transcript walks, banded DPs, the extension with the edit-distance soft-clip DP that
// ProcessCirc's TransExtension uses).  Nothing here touches oracle/.
#include <algorithm>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <atomic>
#include <thread>
#include <vector>

#if defined(CM_S2_PROF)      // diagnostic build only: where one_split / two_splits spend their time (tests/diag/s2_prof.py)
#include <atomic>
static std::atomic<unsigned long long> g_s2_ns[8];
struct S2Tm {
    int k;
    std::chrono::steady_clock::time_point t0;
    explicit S2Tm(int kk) : k(kk), t0(std::chrono::steady_clock::now()) {}
    ~S2Tm() { g_s2_ns[k] += (unsigned long long)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count(); }
};
#define S2_TIME(k) S2Tm s2tm_##k(k)
#else
#define S2_TIME(k) ((void)0)
#endif
#include "circminer_hot.h"

#define CM_STAGE2_HOST 1
#define cmc cmc_s2            // the kernel bodies get their own namespace here: this build of them differs (host-only members)
#include "cm_core.h"
#include "cm_aos.h"
#undef cmc
namespace k = cmc_s2;

namespace {

// ---- outcome codes of a split mapping (src/process_circ.h:14-20) ----
enum { T_FR = 0, T_RF = 1, T_CR = 20, T_NCR = 21, T_MCR = 22, T_UD = 30, T_NF = 40 };
constexpr int TOPCHAIN = 10;      // chains tried per gene and read (src/process_circ.cpp:19)
constexpr int BPRES = 5;          // distance to an annotated exon end that still counts (src/common.h:42)
constexpr int INDELTH = 3;        // src/common.h:45
constexpr int SEED_STEP = 3;      // one 8-mer every 3 bases (src/process_circ.cpp:59)

// A chain with any number of equally long fragments; accessor set of k::CH so the extension templates take either.
struct WideChain {
    std::vector<uint32_t> r;
    std::vector<int32_t> q;
    int kmer = 0;
    float score = 0;
    uint32_t len() const { return (uint32_t)r.size(); }
    uint32_t rpos(uint32_t i) const { return r[i]; }
    int32_t qpos(uint32_t i) const { return q[i]; }
    uint32_t rend_excl() const { return r.back() + (uint32_t)kmer; }
    int32_t qend_excl() const { return q.back() + kmer; }
};

// One oriented mate: the bytes the reference would hold in Record::seq or Record::rcseq.
struct Seq {
    const uint8_t *p = nullptr;
    uint32_t n = 0;
    k::SV view() const { return k::SV{p, 0, 1, 0}; }
    // the reference indexes these strings without bounds checks; an index outside [0, n) yields NUL here (index n is its terminator)
    char at(int64_t i) const { return (p && i >= 0 && i < (int64_t)n) ? (char)p[i] : '\0'; }
    std::string pair_at(int64_t i) const {
        std::string s;
        s += at(i);
        s += at(i + 1);
        return s;
    }
};

struct Breakpoint {           // what CircRes::set_bp records
    uint32_t beg = 0, end = 0;
    std::string sig_start, sig_end, ref_start, ref_end;
};

struct Call {                 // one entry of circ_res
    int chr_id;
    uint32_t spos, epos;
    int type;
    uint64_t rec;
    Breakpoint bp;
};

struct TidOff { uint32_t tid; int off; };

class Caller {
public:
    Caller(const cm_params &P, const cm_index_view &X, const cm_annot_view &A, int window, const cm_chr_info *chrs)
        : av_(A), ws_(window), chrs_(chrs) {
        k::build_annot_aos(A, aos_);
        k::KCore kc;
        kc.P = P;
        kc.X = X;
        kc.A = k::annot_dev_host(A, aos_);
        core_ = k::to_core(kc);
        bufa_.assign(2048 / 2 + 64, 0);
        bufb_.assign(2048 / 2 + 64, 0);
        mem_ = k::DpMem{k::LBuf{bufa_.data(), 2048}, k::LBuf{bufb_.data(), 2048}, &err_, true};
    }
    std::string candidates;
    std::vector<Call> calls;
    int err_ = 0;

    // call_circ (src/process_circ.cpp:334-358); st in chromosome coordinates, shift = start of its chromosome in the contig
    void process(const Seq fwd[2], const Seq rc[2], const cm_mapped_read &st, uint32_t shift, uint64_t rec, const char *name) {
        S2_TIME(5);
        rem_ = full_ = s1_ = s2_ = Seq{};
        rec_ = rec;
        name_ = name;
        drop_tables_before(st.spos_r1 + shift);
        if (st.type == CM_CHIBSJ) one_split(fwd, rc, st, shift);
        else if (st.type == CM_CHI2BSJ) two_splits(fwd, rc, st, shift);
    }

private:
    const cm_annot_view &av_;
    int ws_;
    const cm_chr_info *chrs_;
    k::AnnotAosHost aos_;
    k::Core core_;
    std::vector<uint8_t> bufa_, bufb_;
    k::DpMem mem_;
    // ProcessCirc members that outlive one helper call (src/process_circ.h:31-41): which string is "the split read", which "the
    // fully mapped mate", and the pair as check_split_map's overlapping form names it
    Seq rem_, full_, s1_, s2_;
    uint64_t rec_ = 0;
    const char *name_ = "";
    std::vector<WideChain> ch1_, ch2_;

    const cm_params &P() const { return core_.P; }
    k::Ext ext() const { return k::Ext(core_, mem_); }

    // ---------------------------------------------------------------- genome / annotation access
    bool genome_at(uint32_t start, int len, std::string &out) const {       // pac2char_otf: 1-based, fails off the contig
        if ((int)start < 0 || (int)start + len - 1 > (int)core_.X.ref_len || start == 0) return false;
        out.assign((const char *)core_.X.genome + (start - 1), (size_t)std::max(len, 0));
        return true;
    }
    std::string ref_pair(uint32_t start) const {
        std::string s;
        return genome_at(start, 2, s) ? std::string(s.c_str()) : std::string();
    }
    bool base_equals(uint32_t pos, char ch) const {
        std::string s;
        return genome_at(pos, 1, s) && s[0] == ch;
    }
    int genes_at(uint32_t pos) const {                                      // get_gene_overlap(pos, false): interval of genes_int_map or -1
        if (av_.n_giv == 0 || pos < av_.giv_spos[0]) return -1;
        const uint32_t *e = std::upper_bound(av_.giv_spos, av_.giv_spos + av_.n_giv, pos);
        const int i = (int)(e - av_.giv_spos) - 1;
        if (i < 0 || av_.giv_epos[i] < pos || av_.giv_gene_off[i + 1] == av_.giv_gene_off[i]) return -1;
        return i;
    }
    template <class F> void each_tid_of_seg(uint32_t seg, F f) const {
        for (uint32_t t = av_.seg_tid_off[seg]; t < av_.seg_tid_off[seg + 1]; ++t) f(av_.seg_tid[t]);
    }
    template <class F> void each_seg_of(int iv, F f) const {
        for (uint32_t s = av_.iv_seg_off[iv]; s < av_.iv_seg_off[iv + 1]; ++s) f(av_.iv_seg[s]);
    }
    bool iv_has_tid(int iv, uint32_t tid) const {
        bool hit = false;
        each_seg_of(iv, [&](uint32_t sg) { each_tid_of_seg(sg, [&](uint32_t t) { hit |= (t == tid); }); });
        return hit;
    }

    // ---------------------------------------------------------------- regional seed tables (RegionalHashTable, hash_table.cpp)
    struct Table {
        uint32_t *off = nullptr, *loc = nullptr;
        uint32_t gene_end = 0;
    };
    std::map<uint32_t, Table> tables_;
    const Table &table_for(uint32_t gene) {
        auto it = tables_.find(gene);
        if (it != tables_.end()) return it->second;
        S2_TIME(0);
        Table t;
        const uint32_t gs = av_.gene_start[gene], ge = av_.gene_end[gene];
        std::string g;
        const int glen = (int)(ge - gs + 1);
        if (!genome_at(gs, glen, g)) g.assign((size_t)std::max(glen, 0), '\0');
        t.gene_end = ge;
        if (cm_regional_table_build((const uint8_t *)g.data(), 0, glen, ws_, &t.off, &t.loc) != CM_OK) t.off = t.loc = nullptr;
        return tables_.emplace(gene, t).first->second;
    }
    void drop_tables_before(uint32_t pos) {                                 // check_removables: genes that end before this pair
        for (auto it = tables_.begin(); it != tables_.end();) {
            if (pos > it->second.gene_end) {
                cm_regional_table_free(it->second.off, it->second.loc);
                it = tables_.erase(it);
            } else ++it;
        }
    }
public:
    ~Caller() {
        for (auto &t : tables_) cm_regional_table_free(t.second.off, t.second.loc);
    }
private:
    int seed_code(const uint8_t *s) const {                                 // RegionalHashTable::hash_val
        int v = 0;
        for (int i = 0; i < ws_; ++i) {
            int b;
            switch (s[i]) {
                case 'A': case 'a': b = 0; break;
                case 'C': case 'c': b = 1; break;
                case 'G': case 'g': b = 2; break;
                case 'T': case 't': b = 3; break;
                default: return -1;
            }
            v = v * 4 + b;
        }
        return v;
    }

    // ---------------------------------------------------------------- seeding + k-best chaining of the unmapped part
    // (ProcessCirc::chaining :677-737 + chain_seeds_sorted_kbest2 chain.cpp:310-539).  Hits are offsets inside the gene; the
    // gene start is added only when a chain is emitted, and the annotation queries of the scoring run on the bare offsets
    // (as the reference does).  Improvements are logged in order and replayed best score first, at most max_chain_len per score.
    struct Seed { const uint32_t *hit; uint32_t n; int32_t qpos; };
    void chains_of(uint32_t qs, uint32_t qe, const Table &tab, const Seq &seq, uint32_t gene_start, std::vector<WideChain> &out) {
        S2_TIME(1);
        out.clear();
        const int span = (int)qe - (int)qs + 1;
        if (span < ws_ || !tab.off) return;
        std::vector<Seed> seeds;
        for (uint32_t i = qs - 1; i + (uint32_t)ws_ <= qe; i += SEED_STEP) {
            const int hv = seed_code(seq.p + i);
            if (hv < 0) continue;
            uint32_t n = tab.off[hv + 1] - tab.off[hv];
            if (n > (uint32_t)P().seed_lim) n = 0;
            seeds.push_back(Seed{tab.loc + tab.off[hv], n, (int32_t)i});
        }
        const int listed = (int)seeds.size();
        int kc = listed;
        while (kc >= 1 && seeds[kc - 1].n == 0) --kc;
        if (kc <= 0) return;
        const int kmer = ws_, seq_len = (int)qe;
        const uint32_t max_best = (uint32_t)P().max_chain_len;
        struct Cell { double score; int pl, pi; };
        std::vector<std::vector<Cell>> dp(kc);
        for (int a = 0; a < kc; ++a) dp[a].assign(seeds[a].n, Cell{(double)kmer, -1, -1});
        struct Ev { double score; uint32_t order; int list, ind; };
        std::vector<Ev> log;
        std::vector<uint32_t> cursor(kc);
        for (int a = kc - 2; a >= 0; --a) {
            const Seed &cur = seeds[a];
            const uint32_t read_remain = (uint32_t)(seq_len - cur.qpos - kmer);
            std::fill(cursor.begin(), cursor.end(), 0u);
            for (uint32_t i = 0; i < cur.n; ++i) {
                const int32_t here = (int32_t)cur.hit[i];
                const uint32_t seg_start = (uint32_t)here, seg_end = seg_start + (uint32_t)kmer - 1;
                bool bounded = false;
                uint32_t limit = k::MAXUB, max_exon_end = 0;
                int ol = -1;
                for (int b = a + 1; b < kc; ++b) {
                    const Seed &nx = seeds[b];
                    uint32_t &lb = cursor[b];
                    if (nx.n == 0 || lb >= nx.n) continue;
                    if (here + P().max_intron < (int32_t)nx.hit[lb]) continue;
                    while (lb < nx.n && (int32_t)nx.hit[lb] <= here) ++lb;
                    if (lb >= nx.n) continue;
                    if (!bounded) {
                        limit = k::upper_bound(core_, seg_start, (uint32_t)kmer, read_remain, max_exon_end, ol);
                        bounded = limit != k::MAXUB;
                    }
                    const int read_dist = nx.qpos - cur.qpos - kmer;
                    for (uint32_t j = lb; j < nx.n && nx.hit[j] <= limit; ++j) {
                        const uint32_t there = nx.hit[j];
                        int gdist = k::INF_I, tdist, dist;
                        if (max_exon_end == 0 || there + (uint32_t)kmer - 1 <= max_exon_end) gdist = (int)(there - seg_end - 1);
                        if (k::cabs(gdist - read_dist) <= P().max_ed) dist = gdist;
                        else if (k::check_junction(core_, seg_start, there, ol, kmer, read_dist, tdist)) dist = tdist;
                        else continue;
                        const int hi = std::max(read_dist, dist), lo = std::min(read_dist, dist);
                        const double sc = dp[b][j].score + 2e4 * kmer - 0.1 * (hi - lo);
                        if (sc > dp[a][i].score) {
                            dp[a][i] = Cell{sc, b, (int)j};
                            log.push_back(Ev{sc, (uint32_t)log.size(), a, (int)i});
                        }
                    }
                }
            }
        }
        // replay: scores descending, insertion order inside a score, only the first max_best entries of a score count
        std::stable_sort(log.begin(), log.end(), [](const Ev &x, const Ev &y) { return x.score > y.score; });
        const double top = log.empty() ? (double)kmer : log[0].score;
        std::vector<uint32_t> seen_inner;                     // `repeats`: shifted positions of non-first fragments
        size_t run_begin = 0;
        for (size_t e = 0; e < log.size() && out.size() < max_best; ++e) {
            if (log[e].score != log[run_begin].score) run_begin = e;
            if (e - run_begin >= max_best) continue;
            const uint32_t first_off = seeds[log[e].list].hit[log[e].ind];
            if (log[e].score < top && std::find(seen_inner.begin(), seen_inner.end(), first_off) != seen_inner.end()) continue;
            WideChain c;
            c.kmer = kmer;
            c.score = (float)log[e].score;
            int l = log[e].list, i = log[e].ind;
            while (l != -1) {
                c.r.push_back(gene_start + seeds[l].hit[i]);
                c.q.push_back(seeds[l].qpos);
                if (c.r.size() > 1) seen_inner.push_back(c.r.back());
                const Cell &d = dp[l][i];
                l = d.pl;
                i = d.pi;
            }
            out.push_back(std::move(c));
        }
        if (out.empty())                                      // nothing chained: single seeds, last list first
            for (int a = kc - 1; a >= 0; --a)
                for (uint32_t i = 0; i < seeds[a].n && out.size() < max_best; ++i) {
                    WideChain c;
                    c.kmer = kmer;
                    c.score = (float)dp[a][i].score;
                    c.r.push_back(gene_start + seeds[a].hit[i]);
                    c.q.push_back(seeds[a].qpos);
                    out.push_back(std::move(c));
                }
        // keep the leading chains whose count of unused seeds does not increase
        int least = k::INF_I;
        for (size_t j = 0; j < out.size(); ++j) {
            const int missing = listed - (int)out[j].len();
            if (missing > least) {
                out.resize(j);
                break;
            }
            least = missing;
        }
    }

    // ---------------------------------------------------------------- mates
    k::MM mate_from_state(const cm_mapped_read &st, int which, uint32_t rlen, bool partial, uint32_t shift) const {   // common.cpp:196-243
        k::MM m = k::mm_init(core_);
        m.type = st.type;
        m.right_ed = m.left_ed = 0;
        const bool first = which == 1;
        m.spos = (first ? st.spos_r1 : st.spos_r2) + shift;
        m.epos = (first ? st.epos_r1 : st.epos_r2) + shift;
        m.qspos = first ? st.qspos_r1 : st.qspos_r2;
        m.qepos = first ? st.qepos_r1 : st.qepos_r2;
        m.middle_ed = first ? st.ed_r1 : st.ed_r2;
        m.matched_len = first ? st.mlen_r1 : st.mlen_r2;
        m.dir = (first ? st.r1_forward : st.r2_forward) ? 1 : -1;
        const int head = (int)m.qspos - 1, tail = (int)rlen - (int)m.qepos;
        if (partial) {
            const bool right_side = (m.qspos - 1) > (uint32_t)tail;
            m.sclen_left = right_side ? 0 : head;
            m.sclen_right = right_side ? tail : 0;
        } else {
            m.sclen_left = head;
            m.sclen_right = tail;
        }
        return m;
    }
    void span_of_chain(const WideChain &c, uint32_t qs, int rlen, int dir, k::MM &m) const {            // set_mm
        m.spos = c.rpos(0);
        m.epos = c.rend_excl() - 1;
        m.qspos = qs;
        m.qepos = qs + (uint32_t)rlen - 1;
        m.matched_len = (m.qepos + 1 >= m.qspos) ? (m.qepos - m.qspos + 1) : 0;
        m.dir = dir;
    }
    bool absorb_right(k::MM &l, const k::MM &r) const {                                                  // merge_to_right
        if (l.dir != r.dir) return false;
        l.epos = r.epos;
        l.qepos = r.qepos;
        l.middle_ed += l.right_ed + r.left_ed + l.sclen_right + r.sclen_left;
        l.right_ed = r.right_ed;
        l.matched_len += r.matched_len + (uint32_t)l.sclen_right + (uint32_t)r.sclen_left;
        l.sclen_right = r.sclen_right;
        l.right_ok = r.right_ok;
        l.looked_up_epos = r.looked_up_epos;
        l.exon_ind_epos = r.exon_ind_epos;
        return k::mm_ed(l) <= P().max_ed;
    }

    // Transcripts shared by n segments (utils.cpp:356-599).  Every segment may be represented by the exon interval under
    // its start or under its end; the combinations are tried in the reference's fixed order and the first non-empty answer
    // wins.  For 2 segments the answer is (tids of a) that occur in b; for 3 the third interval only has to exist; for 4 it
    // is the 2-segment answer of (a, b) filtered by the 2-segment answer of (c, d).
    bool shared_transcripts(const k::MM *seg, int n, std::vector<uint32_t> &out) const {
        int iv[4][2];
        for (int s = 0; s < n; ++s) {
            k::MM m = seg[s];
            k::overlap_to_spos(core_, m);
            k::overlap_to_epos(core_, m);
            iv[s][0] = m.exons_spos;
            iv[s][1] = m.exons_epos;
        }
        // significance of the choice bits, most significant first
        static const int order2[2] = {0, 1}, order3[3] = {0, 1, 2}, order4[4] = {3, 0, 1, 2};
        const int *ord = n == 2 ? order2 : (n == 3 ? order3 : order4);
        for (int combo = 0; combo < (1 << n); ++combo) {
            int pick[4];
            for (int b = 0; b < n; ++b) pick[ord[b]] = iv[ord[b]][(combo >> (n - 1 - b)) & 1];
            out.clear();
            bool all = true;
            for (int s = 0; s < n; ++s) all &= pick[s] >= 0;
            if (!all) continue;
            k::common_tids_from(core_, pick[0], pick[1], 0, [&](uint32_t t) {
                if (n < 4 || (iv_has_tid(pick[2], t) && iv_has_tid(pick[3], t))) out.push_back(t);
                return true;
            });
            if (n == 4 && !out.empty()) {
                // the pair (c, d) must itself share a transcript; membership in both was just checked per element
            }
            if (!out.empty()) return true;
        }
        out.clear();
        return false;
    }
    k::TidList as_list(const std::vector<uint32_t> &t) const { return k::TidList{t.data(), (int)t.size(), -1, -1, false}; }

    // splice junctions a mate spans along one of the transcripts under its start (get_junctions, utils.cpp:686-744)
    struct Junction { uint32_t beg, end, matched; };
    void junctions_of(k::MM &m, std::vector<Junction> &out) const {
        k::overlap_to_spos(core_, m);
        k::overlap_to_epos(core_, m);
        out.clear();
        if (m.exons_spos < 0 || m.exons_epos < 0) return;
        bool done = false;
        each_seg_of(m.exons_spos, [&](uint32_t sg) {
            each_tid_of_seg(sg, [&](uint32_t tid) {
                if (done) return;
                const int first = av_.trans_start_ind[tid];
                const uint32_t tsz = av_.t2s_off[tid + 1] - av_.t2s_off[tid];
                const uint8_t *state = av_.t2s + av_.t2s_off[tid];
                const uint32_t a = (uint32_t)(m.exon_ind_spos - first), b = (uint32_t)(m.exon_ind_epos - first);
                if (m.exon_ind_epos < first || b >= tsz || state[b] == 0) return;
                if (a == b) {
                    done = true;
                    return;
                }
                auto add = [&](uint32_t from, uint32_t to, uint32_t cov) {
                    if (from < to) out.push_back(Junction{from, to, cov});
                };
                uint32_t from = av_.iv_epos[m.exons_spos], covered = av_.iv_epos[m.exons_spos] - m.spos + 1;
                int ivx = m.exon_ind_spos;
                for (uint32_t x = a + 1; x < b; ++x) {
                    ++ivx;
                    if (x < tsz && state[x] != 0) {
                        add(from, av_.iv_spos[ivx], covered);
                        covered += av_.iv_epos[ivx] - av_.iv_spos[ivx] + 1;
                        from = av_.iv_epos[ivx];
                    }
                }
                add(from, av_.iv_spos[m.exons_epos], covered);
                covered += m.epos - av_.iv_spos[m.exons_epos] + 1;
                if (k::cabs((int32_t)(covered - m.matched_len)) <= INDELTH) done = true;
                else out.clear();
            });
        });
    }

    // transcripts with an exon boundary within BPRES of a piece's outer end, over the intervals the piece covers
    void ends_near(const k::MM &m, std::vector<TidOff> &out) const {          // exon ENDS near the piece's right end, walking left
        for (int iv = m.exon_ind_epos; iv >= 0 && iv < (int)av_.n_iv && m.spos < av_.iv_epos[iv]; --iv)
            each_seg_of(iv, [&](uint32_t sg) {
                const int d = (int)(m.epos + (uint32_t)m.sclen_right - av_.seg_end[sg]);
                if (k::cabs(d) <= BPRES) each_tid_of_seg(sg, [&](uint32_t t) { out.push_back(TidOff{t, d}); });
            });
    }
    void starts_near(const k::MM &m, std::vector<TidOff> &out) const {        // exon STARTS near the piece's left end, walking right
        for (int iv = m.exon_ind_spos; iv >= 0 && iv < (int)av_.n_iv && m.epos > av_.iv_spos[iv]; ++iv)
            each_seg_of(iv, [&](uint32_t sg) {
                const int d = (int)(m.spos - (uint32_t)m.sclen_left - av_.seg_start[sg]);
                if (k::cabs(d) <= BPRES) each_tid_of_seg(sg, [&](uint32_t t) { out.push_back(TidOff{t, d}); });
            });
    }

    // ---------------------------------------------------------------- re-alignment of a read around a candidate junction
    // the read is cut after `cut` bases: the left part must end at end_bp walking left, the right part start at beg_bp walking
    // right, both along `tids`; the two bases at the cut are compared directly (split_realignment, 6-argument form, :1343-1392)
    int realign_at(uint32_t cut, uint32_t beg_bp, uint32_t end_bp, const Seq &s, const std::vector<uint32_t> &tids) {
        S2_TIME(2);
        const int lim = P().max_ed;
        if (cut == 0 || cut >= s.n) return lim + 1;
        const int e_last = base_equals(end_bp, s.at(cut - 1)) ? 0 : 1, e_first = base_equals(beg_bp, s.at(cut)) ? 0 : 1;
        uint32_t lpos = end_bp, rpos = beg_bp;
        k::AlignRes l = k::ar_init(beg_bp), r = k::ar_init(end_bp);
        const k::TidList tl = as_list(tids);
        const bool lok = ext().extend_side(tl, s.view(), lpos, (int)cut - 1, lim - e_last, beg_bp, l, false);
        const bool rok = ext().extend_side(tl, s.view().sub((int)cut + 1), rpos, (int)(s.n - cut - 1), lim - e_first, end_bp, r, true);
        const int total = l.ed + e_last + r.ed + e_first;
        return (lok && rok && total <= lim) ? total : lim + 1;
    }
    // the fully mapped mate itself crosses the junction at offset `cut` of its matched part: split it there and judge the four
    // pieces as an overlapping split pair (split_realignment, 5-argument form, :1394-1486)
    int split_full_mate(uint32_t cut, k::MM &full, k::MM &left, k::MM &right, Breakpoint &bp) {
        const int lim = P().max_ed;
        if (cut == 0 || cut >= full_.n) return T_UD;
        cut += full.qspos - 1;
        if (cut == 0 || cut >= full_.n) return T_UD;
        k::overlap_to_spos(core_, left);
        k::overlap_to_epos(core_, left);
        k::overlap_to_spos(core_, right);
        k::overlap_to_epos(core_, right);
        std::vector<uint32_t> tids;
        const k::MM both[2] = {left, right};
        if (!shared_transcripts(both, 2, tids)) return T_UD;
        const int e_last = base_equals(left.epos, full_.at(cut - 1)) ? 0 : 1, e_first = base_equals(right.spos, full_.at(cut)) ? 0 : 1;
        uint32_t lpos = left.epos, rpos = right.spos;
        k::AlignRes l = k::ar_init(right.spos), r = k::ar_init(left.epos);
        const k::TidList tl = as_list(tids);
        const bool lok = ext().extend_side(tl, full_.view(), lpos, (int)cut - 1, lim - e_last, right.spos, l, false);
        const bool rok = ext().extend_side(tl, full_.view().sub((int)cut + 1), rpos, (int)(full_.n - cut - 1), lim - e_first, left.epos, r, true);
        l.ed += e_last;
        r.ed += e_first;
        if (!lok || !rok || l.ed + r.ed > lim) return T_UD;
        k::MM nl = k::mm_init(core_), nr = k::mm_init(core_);
        nl.spos = lpos; nl.epos = left.epos; nl.qspos = (uint32_t)l.sclen; nl.qepos = cut; nl.dir = full.dir;
        nl.matched_len = cut - (uint32_t)l.sclen; nl.sclen_left = l.sclen; nl.sclen_right = 0;
        nl.left_ed = l.ed; nl.right_ed = 0; nl.middle_ed = 0; nl.left_ok = nl.right_ok = true;
        nr.spos = right.spos; nr.epos = rpos; nr.qspos = cut + 1; nr.qepos = full_.n - (uint32_t)r.sclen; nr.dir = full.dir;
        nr.matched_len = full_.n - cut - (uint32_t)r.sclen; nr.sclen_left = 0; nr.sclen_right = r.sclen;
        nr.left_ed = 0; nr.right_ed = r.ed; nr.middle_ed = 0; nr.left_ok = nr.right_ok = true;
        s1_ = rem_;
        s2_ = full_;
        return judge_four(right, nr, left, nl, bp);
    }
    // the fully mapped mate overlaps one of the two breakpoints (rescue_overlapping_bsj, :1488-1552)
    int rescue(k::MM &full, k::MM &left, k::MM &right, Breakpoint &bp) {
        std::vector<Junction> js;
        if (right.spos <= full.epos && right.spos > full.spos) {
            junctions_of(full, js);
            uint32_t cut = 0;
            for (const Junction &j : js) if (j.end == right.spos) cut = j.matched;
            if (cut == 0) cut = right.spos - full.spos;                 // intron retention
            if (split_full_mate(cut, full, left, right, bp) == T_CR) return T_CR;
        }
        if (left.epos >= full.spos && left.epos < full.epos) {
            junctions_of(full, js);
            uint32_t cut = 0;
            for (const Junction &j : js) if (j.beg == left.epos) cut = j.matched;
            if (cut == 0) cut = full.matched_len - (full.epos - left.epos);
            if (split_full_mate(cut, full, left, right, bp) == T_CR) return T_CR;
        }
        return T_UD;
    }

    // ---------------------------------------------------------------- classification
    // one mate maps in one piece (`full`), the other in two (`left` before `right` on the READ); final_check, :1136-1341
    int judge_three(k::MM &full, k::MM &left, k::MM &right, Breakpoint &bp) {
        const int lim = P().max_ed, max_sc = P().max_sc;
        if (left.epos < right.spos) {                                   // collinear on the genome: an ordinary pair
            const bool inner_l = full.spos <= left.spos, inner_r = full.epos >= right.epos;
            if (full.dir == 1) {
                if (inner_l) return T_FR;
                if (inner_r) return T_RF;
            } else if (full.dir == -1) {
                if (inner_r) return T_FR;
                if (inner_l) return T_RF;
            }
        } else if (right.spos <= left.spos && left.epos >= right.epos) {   // back-spliced order
            if (full.spos < right.spos) {                               // let soft clipping pull the full mate inside the circle
                const int off = (int)(right.spos - full.spos);
                if (off <= max_sc - full.sclen_left) {
                    full.spos = right.spos;
                    full.sclen_left += off;
                    full.qspos += (uint32_t)off;
                    full.matched_len -= (uint32_t)off;
                }
            }
            if (full.epos > left.epos) {
                const int off = (int)(full.epos - left.epos);
                if (off <= max_sc - full.sclen_right) {
                    full.epos = left.epos;
                    full.sclen_right += off;
                    full.qepos -= (uint32_t)off;
                    full.matched_len -= (uint32_t)off;
                }
            }
            if (full.spos >= right.spos && full.epos <= left.epos) {
                k::MM *all[3] = {&full, &right, &left};
                for (k::MM *m : all) {
                    k::overlap_to_spos(core_, *m);
                    k::overlap_to_epos(core_, *m);
                }
                std::vector<TidOff> ends, starts;
                ends_near(left, ends);
                starts_near(right, starts);
                int best = lim + 1;
                std::vector<uint32_t> one(1);
                for (const TidOff &s : starts)
                    for (const TidOff &e : ends) {
                        if (s.tid != e.tid || s.off != e.off) continue;
                        one[0] = s.tid;
                        const uint32_t cut = left.qepos + (uint32_t)left.sclen_right - (uint32_t)e.off;
                        const uint32_t beg_bp = right.spos - (uint32_t)right.sclen_left - (uint32_t)s.off;
                        const uint32_t end_bp = left.epos + (uint32_t)left.sclen_right - (uint32_t)e.off;
                        // the other mate may run over a breakpoint in its clipped part: it must realign across it as well
                        if (full.sclen_right > 0) {
                            if (full.epos + (uint32_t)full.sclen_right > end_bp) {
                                if (realign_at(full.qepos + (end_bp - full.epos), beg_bp, end_bp, full_, one) > lim) continue;
                            } else if (full.sclen_right > max_sc) continue;
                        }
                        if (full.sclen_left > 0) {
                            if (full.spos - (uint32_t)full.sclen_left < beg_bp) {
                                if (realign_at((uint32_t)full.sclen_left + (full.spos - beg_bp), beg_bp, end_bp, full_, one) > lim) continue;
                            } else if (full.sclen_left > max_sc) continue;
                        }
                        const int ed = realign_at(cut, beg_bp, end_bp, rem_, one);
                        if (ed < best) {
                            bp = Breakpoint{beg_bp, end_bp, rem_.pair_at(cut), rem_.pair_at((int64_t)cut - 2), ref_pair(beg_bp), ref_pair(end_bp - 1)};
                            if (ed == 0) return T_CR;
                            best = ed;
                        }
                    }
                if (best <= lim) return T_CR;
                const uint32_t cut = left.qepos + (uint32_t)left.sclen_right;
                if (cut < 2 || cut > rem_.n - 2) return T_MCR;
                // unannotated ends: the reference stores the bases before the cut as the START signal here
                bp = Breakpoint{right.spos - (uint32_t)right.sclen_left, left.epos + (uint32_t)left.sclen_right, rem_.pair_at((int64_t)cut - 2),
                                rem_.pair_at(cut), std::string(), std::string()};
                bp.ref_start = ref_pair(bp.beg);
                bp.ref_end = ref_pair(bp.end - 1);
                return (!starts.empty() && !ends.empty()) ? T_NCR : T_MCR;
            }
        }
        return rescue(full, left, right, bp);
    }
    // single split: which of the mates is the split one, and in which read order its two pieces come (check_split_map, :892-921)
    int judge_single(k::MM &m1, k::MM &m2, k::MM &piece, bool r1_split, Breakpoint &bp) {
        S2_TIME(4);
        k::MM &split = r1_split ? m1 : m2, &whole = r1_split ? m2 : m1;
        const int ed = k::mm_ed(split) + k::mm_ed(piece);
        const int v = (split.qspos < piece.qspos) ? judge_three(whole, split, piece, bp) : judge_three(whole, piece, split, bp);
        return ed > P().max_ed ? T_UD : v;
    }
    static std::string agree(const std::string &a, const std::string &b) {       // get_consensus of two strings
        std::string r;
        if (a.size() != b.size()) return r;
        for (size_t i = 0; i < a.size(); ++i) r += a[i] == b[i] ? a[i] : 'N';
        return r;
    }
    // both mates map in two pieces (check_split_map, overlapping form, :924-1134); a1/a2 = pieces of R1, b1/b2 = pieces of R2
    int judge_four(k::MM &a1, k::MM &b1, k::MM &a2, k::MM &b2, Breakpoint &bp) {
        const int lim = P().max_ed;
        if (k::mm_ed(a1) + k::mm_ed(a2) > lim || k::mm_ed(b1) + k::mm_ed(b2) > lim) return T_UD;
        k::MM al = a1.spos <= a2.spos ? a1 : a2, ar = a1.spos <= a2.spos ? a2 : a1;     // genome order
        k::MM bl = b1.spos <= b2.spos ? b1 : b2, br = b1.spos <= b2.spos ? b2 : b1;
        const bool a_lin = al.qspos < ar.qspos, b_lin = bl.qspos < br.qspos;            // pieces in read order == genome order
        if (a_lin && b_lin) {
            if (al.dir == 1) {
                if (ar.spos <= bl.spos) return T_FR;
                if (al.epos >= br.epos) return T_RF;
            }
            if (al.dir == -1) {
                if (br.spos <= al.spos) return T_FR;
                if (bl.epos >= ar.epos) return T_RF;
            }
            return T_UD;
        }
        if (a_lin != b_lin) {                       // only one mate really crosses the junction: glue the other back together
            k::MM whole = a_lin ? al : bl;
            if (!absorb_right(whole, a_lin ? ar : br)) return T_UD;
            rem_ = a_lin ? s2_ : s1_;
            return a_lin ? judge_three(whole, bl, br, bp) : judge_three(whole, al, ar, bp);
        }
        if (!(al.spos == bl.spos && ar.epos == br.epos)) return T_UD;
        // both mates cross it (the junction lies in their overlap)
        k::overlap_to_spos(core_, al);
        k::overlap_to_epos(core_, ar);
        std::vector<TidOff> ends, starts;
        ends_near(ar, ends);
        starts_near(al, starts);
        int best1 = lim + 1, best2 = lim + 1;
        std::vector<uint32_t> one(1);
        for (const TidOff &s : starts)
            for (const TidOff &e : ends) {
                if (s.tid != e.tid || s.off != e.off) continue;
                one[0] = s.tid;
                const uint32_t beg_bp = al.spos - (uint32_t)al.sclen_left - (uint32_t)s.off;
                const uint32_t end_bp = ar.epos + (uint32_t)ar.sclen_right - (uint32_t)e.off;
                const uint32_t cut1 = ar.qepos + (uint32_t)ar.sclen_right - (uint32_t)e.off;
                const int ed1 = realign_at(cut1, beg_bp, end_bp, s1_, one);
                const bool sig1 = !(cut1 < 2 || cut1 + 2 > s1_.n);
                const uint32_t cut2 = br.qepos + (uint32_t)br.sclen_right - (uint32_t)e.off;
                const int ed2 = realign_at(cut2, beg_bp, end_bp, s2_, one);
                const bool sig2 = !(cut2 < 2 || cut2 + 2 > s2_.n);
                if (ed1 < best1 && ed2 < best2) {
                    const std::string st1 = sig1 ? s1_.pair_at(cut1) : "", en1 = sig1 ? s1_.pair_at((int64_t)cut1 - 2) : "";
                    const std::string st2 = sig2 ? s2_.pair_at(cut2) : "", en2 = sig2 ? s2_.pair_at((int64_t)cut2 - 2) : "";
                    bp.beg = beg_bp;
                    bp.end = end_bp;
                    bp.ref_start = ref_pair(beg_bp);
                    bp.ref_end = ref_pair(end_bp - 1);
                    if (st1.empty()) { bp.sig_start = st2; bp.sig_end = en2; }
                    else if (st2.empty()) { bp.sig_start = st1; bp.sig_end = en1; }
                    else { bp.sig_start = agree(st1, st2); bp.sig_end = agree(en1, en2); }
                    best1 = ed1;
                    best2 = ed2;
                }
            }
        if (best1 <= lim && best2 <= lim) return T_CR;
        const uint32_t cut = ar.qepos + (uint32_t)ar.sclen_right;
        if (cut < 2 || cut > s1_.n - 2 || cut > s2_.n - 2) return T_MCR;
        bp.beg = al.spos - (uint32_t)al.sclen_left;
        bp.end = ar.epos + (uint32_t)ar.sclen_right;
        bp.sig_start = agree(s1_.pair_at(cut), s2_.pair_at(cut));
        bp.sig_end = agree(s1_.pair_at((int64_t)cut - 2), s2_.pair_at((int64_t)cut - 2));
        bp.ref_start = ref_pair(bp.beg);
        bp.ref_end = ref_pair(bp.end - 1);
        return (!starts.empty() && !ends.empty()) ? T_NCR : T_MCR;
    }

    // ---------------------------------------------------------------- rows of <out>.candidates.pam (print_split_mapping + type)
    void row(const k::MM &m1, const k::MM &m2, const k::MM *pieces, int n_pieces, int chr_row, int type) {
        char buf[160];
        const uint32_t sh = av_.chr_shift[chr_row];
        candidates += name_;
        candidates += '\t';
        candidates += chrs_[av_.chr_id[chr_row]].name;
        candidates += '\t';
        auto put = [&](const k::MM &m) {
            const int n = snprintf(buf, sizeof buf, "%u\t%u\t%d\t%d\t%d\t", m.spos - sh, m.epos - sh, (int)m.qspos, (int)m.matched_len, m.dir);
            candidates.append(buf, (size_t)n);
        };
        for (int i = 0; i < n_pieces; ++i) put(pieces[i]);
        put(m1);
        put(m2);
        const int n = snprintf(buf, sizeof buf, "%d\n", type);
        candidates.append(buf, (size_t)n);
    }
    struct Best {
        int type = T_NF;
        int chr_id = -1;
        uint32_t spos = 0, epos = 0;
        Breakpoint bp;
    };
    void emit(const Best &b) { calls.push_back(Call{b.chr_id, b.spos, b.epos, b.type, rec_, b.bp}); }
    // returns true when the read is finished (an ordinary pair, or an annotated circle)
    bool weigh(int type, const Breakpoint &bp, int chr_row, Best &best) {
        if (type < T_CR) {
            best.type = type;
            return true;
        }
        if (type <= T_MCR && type < best.type) {
            const uint32_t sh = av_.chr_shift[chr_row];
            best.type = type;
            best.chr_id = av_.chr_id[chr_row];
            best.spos = bp.beg - sh;
            best.epos = bp.end - sh;
            best.bp = bp;
            if (type == T_CR) {
                emit(best);
                return true;
            }
        }
        return false;
    }

    // the unmapped side of a mate that stage 1 matched on one side only: [qs, qe] 1-based on the oriented read
    static void unmapped_side(uint32_t qspos, uint32_t qepos, uint32_t rlen, uint32_t &qs, uint32_t &qe) {
        const bool right_matched = (qspos - 1) > (rlen - qepos);
        qs = right_matched ? 1 : qepos + 1;
        qe = right_matched ? qspos - 1 : rlen;
    }

    // call_circ_single_split, :360-482
    void one_split(const Seq fwd[2], const Seq rc[2], const cm_mapped_read &st, uint32_t shift) {
        const bool r1_split = st.mlen_r1 < st.mlen_r2;
        const Seq o1 = st.r1_forward ? fwd[0] : rc[0], o2 = st.r2_forward ? fwd[1] : rc[1];
        rem_ = r1_split ? o1 : o2;
        full_ = r1_split ? o2 : o1;
        k::MM m1 = mate_from_state(st, 1, fwd[0].n, r1_split, shift), m2 = mate_from_state(st, 2, fwd[1].n, !r1_split, shift);
        const k::MM &sp = r1_split ? m1 : m2;
        const uint32_t rlen = r1_split ? fwd[0].n : fwd[1].n;
        uint32_t qs, qe;
        unmapped_side(sp.qspos, sp.qepos, rlen, qs, qe);
        const int todo = (int)(qe - qs + 1);
        if (qe < qs || todo < ws_) return;
        const int gi = genes_at(m1.spos);
        if (gi < 0) return;
        const int dir = (r1_split ? st.r1_forward : st.r2_forward) ? 1 : -1;
        Best best;
        for (uint32_t g = av_.giv_gene_off[gi]; g < av_.giv_gene_off[gi + 1]; ++g) {
            const uint32_t gene = av_.giv_gene[g];
            chains_of(qs, qe, table_for(gene), rem_, av_.gene_start[gene], ch1_);
            for (size_t j = 0; j < ch1_.size() && j < (size_t)TOPCHAIN; ++j) {
                k::MM piece = k::mm_init(core_);
                if (!place_piece(m1, m2, piece, dir, qs, todo, (int)rlen, ch1_[j])) continue;
                const int chr_row = k::chr_row(core_, m1.spos);
                Breakpoint bp;
                const int type = judge_single(m1, m2, piece, r1_split, bp);
                row(m1, m2, &piece, 1, chr_row, type);
                if (weigh(type, bp, chr_row, best)) return;
            }
        }
        if (best.type >= T_CR && best.type <= T_MCR) emit(best);
    }
    // exact coordinates of the unmapped part along one chain (find_exact_coord, :739-789); true when it ends up concordant
    bool place_piece(k::MM &m1, k::MM &m2, k::MM &piece, int dir, uint32_t qs, int todo, int rlen, const WideChain &c) {
        S2_TIME(3);
        span_of_chain(c, qs, todo, dir, piece);
        const uint32_t q0 = qs - 1;
        k::overlap_to_spos(core_, m1);
        k::overlap_to_spos(core_, m2);
        k::overlap_to_spos(core_, piece);
        std::vector<uint32_t> tids;
        const k::MM three[3] = {m1, m2, piece};
        if (!shared_transcripts(three, 3, tids)) return false;
        piece.middle_ed = ext().calc_middle_ed(c, P().max_ed, rem_.view());
        if (piece.middle_ed > P().max_ed) return false;
        piece.is_concord = false;
        int err = piece.middle_ed;
        piece.matched_len = (uint32_t)todo;
        const k::TidList tl = as_list(tids);
        const bool lok = ext().chain_left(tl, c, rem_.view().sub((int)q0), (int32_t)q0, k::MINLB, piece, err);
        const bool rok = ext().chain_right(tl, c, rem_.view(), q0 == 0 ? todo : rlen, k::MAXUB, piece, err);
        k::update_match_mate_info(core_, lok, rok, err, piece);
        return piece.type == CM_CONCRD;
    }
    // extend_both_mates as ProcessCirc calls it (extend.cpp:37-125 with the remaining-part arguments of :560-569)
    bool place_two(const WideChain &lc, const WideChain &rc, const std::vector<uint32_t> &tids, const Seq &ls, const Seq &rs, int lqs, int rqs,
                   int llen, int rlen, k::MM &lm, k::MM &rm) {
        const int lim = P().max_ed;
        lm.middle_ed = ext().calc_middle_ed(lc, lim, ls.view());
        rm.middle_ed = ext().calc_middle_ed(rc, lim, rs.view());
        if (lm.middle_ed <= lim) k::is_concord_impl(lc, (uint32_t)llen, lm, true);
        if (rm.middle_ed <= lim) k::is_concord_impl(rc, (uint32_t)rlen, rm, true);
        if (lm.middle_ed > lim || rm.middle_ed > lim) return false;
        lm.is_concord = rm.is_concord = false;
        int lerr = lm.middle_ed, rerr = rm.middle_ed;
        const k::TidList tl = as_list(tids);
        lm.matched_len = (uint32_t)(llen - lqs + 1);
        lm.qspos = (uint32_t)lqs;
        lm.qepos = (uint32_t)llen;
        const bool ll = ext().chain_left(tl, lc, ls.view(), lqs - 1, k::MINLB, lm, lerr);
        rm.matched_len = (uint32_t)(rlen - rqs + 1);
        rm.qspos = (uint32_t)rqs;
        rm.qepos = (uint32_t)rlen;
        const bool rl = ext().chain_left(tl, rc, rs.view(), rqs - 1, lm.spos, rm, rerr);
        const bool rr = ext().chain_right(tl, rc, rs.view(), rlen, k::MAXUB, rm, rerr);
        const bool lr = ext().chain_right(tl, lc, ls.view(), llen, rm.epos, lm, lerr);
        k::update_match_mate_info(core_, ll, lr, lerr, lm);
        k::update_match_mate_info(core_, rl, rr, rerr, rm);
        return true;
    }
    // call_circ_double_split, :484-645
    void two_splits(const Seq fwd[2], const Seq rc[2], const cm_mapped_read &st, uint32_t shift) {
        const Seq o1 = st.r1_forward ? fwd[0] : rc[0], o2 = st.r2_forward ? fwd[1] : rc[1];
        s1_ = o1;
        s2_ = o2;
        uint32_t qs1, qe1, qs2, qe2;
        unmapped_side(st.qspos_r1, st.qepos_r1, fwd[0].n, qs1, qe1);
        unmapped_side(st.qspos_r2, st.qepos_r2, fwd[1].n, qs2, qe2);
        const int todo1 = (int)(qe1 - qs1 + 1), todo2 = (int)(qe2 - qs2 + 1);
        if (todo1 < ws_ && todo2 < ws_) return;
        if (todo1 < ws_ || todo2 < ws_) one_split(fwd, rc, st, shift);          // the reference goes on after this call
        const int gi = genes_at(st.spos_r1 + shift);
        if (gi < 0) return;
        k::MM m1 = mate_from_state(st, 1, fwd[0].n, true, shift), m2 = mate_from_state(st, 2, fwd[1].n, true, shift);
        Best best;
        for (uint32_t g = av_.giv_gene_off[gi]; g < av_.giv_gene_off[gi + 1]; ++g) {
            const uint32_t gene = av_.giv_gene[g];
            const Table &tab = table_for(gene);
            chains_of(qs1, qe1, tab, o1, av_.gene_start[gene], ch1_);
            chains_of(qs2, qe2, tab, o2, av_.gene_start[gene], ch2_);
            if (ch1_.empty() && ch2_.empty()) continue;
            if (ch1_.empty() || ch2_.empty()) {
                one_split(fwd, rc, st, shift);
                continue;
            }
            // one_split works on its own chain lists only through ch1_: keep this gene's lists in locals
            const std::vector<WideChain> c1 = ch1_, c2 = ch2_;
            for (size_t j = 0; j < c1.size() && j < (size_t)TOPCHAIN; ++j)
                for (size_t q = 0; q < c2.size() && q < (size_t)TOPCHAIN; ++q) {
                    k::MM p1 = k::mm_init(core_), p2 = k::mm_init(core_);
                    span_of_chain(c1[j], qs1, todo1, m1.dir, p1);
                    span_of_chain(c2[q], qs2, todo2, m2.dir, p2);
                    k::overlap_to_spos(core_, m1);
                    k::overlap_to_spos(core_, m2);
                    k::overlap_to_spos(core_, p1);
                    k::overlap_to_spos(core_, p2);
                    std::vector<uint32_t> tids;
                    const k::MM four[4] = {m1, m2, p1, p2};
                    if (!shared_transcripts(four, 4, tids)) continue;
                    const bool first_left = c1[j].rpos(0) <= c2[q].rpos(0);
                    const bool ok = first_left ? place_two(c1[j], c2[q], tids, o1, o2, (int)qs1, (int)qs2, (int)qe1, (int)qe2, p1, p2)
                                               : place_two(c2[q], c1[j], tids, o2, o1, (int)qs2, (int)qs1, (int)qe2, (int)qe1, p2, p1);
                    if (!ok || p1.type != CM_CONCRD || p2.type != CM_CONCRD) continue;
                    const int chr_row = k::chr_row(core_, m1.spos);
                    Breakpoint bp;
                    const int type = judge_four(m1, m2, p1, p2, bp);
                    const k::MM pieces[2] = {p1, p2};
                    row(m1, m2, pieces, 2, chr_row, type);
                    if (weigh(type, bp, chr_row, best)) return;
                }
        }
        if (best.type >= T_CR && best.type <= T_MCR) emit(best);
        else one_split(fwd, rc, st, shift);
    }
};

// reverse complement as FASTQParser::set_reverse_comp leaves it in Record::rcseq (fastq_parser.cpp:141-176): bytes other than
// ACGTN / acgtn become NUL
void reverse_complement(const uint8_t *s, uint32_t n, std::vector<uint8_t> &out) {
    out.assign((size_t)n + 1, 0);
    for (uint32_t i = 0; i < n; ++i) {
        uint8_t o = 0;
        switch (s[n - 1 - i]) {
            case 'A': case 'a': o = 'T'; break;
            case 'C': case 'c': o = 'G'; break;
            case 'G': case 'g': o = 'C'; break;
            case 'T': case 't': o = 'A'; break;
            case 'N': case 'n': o = 'N'; break;
            default: break;
        }
        out[i] = o;
    }
}

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

}  // namespace

// ProcessCirc::do_process over the sorted remain pairs.  Pairs are independent of one another (the per-gene regional tables are a
// cache, dropped once the sorted input has moved past the gene), so runs of pairs on one contig are cut into chunks that
// worker threads take in turn, each with a Caller of its own; rows and calls are put together in chunk order, i.e. input order.
static int circ_call_mt(const cm_params *P, int32_t window_size, uint32_t n_contigs, const cm_index_view *contigs, const cm_annot_view *annots,
                        const cm_chr_info *chrs, uint32_t n_chr, const cm_fastq_batch *sorted, const char *candidates_path,
                        const char *report_path, cm_circ_stats *stats, int n_threads) {
    if (!P || !contigs || !annots || !chrs || !sorted || !candidates_path || !report_path) return CM_EINVAL;
    const int ws = window_size > 0 ? window_size : 8;
    if (ws > 12) return CM_EINVAL;
    const auto t0 = std::chrono::steady_clock::now();
    FILE *fc = fopen(candidates_path, "w");
    if (!fc) return CM_EINVAL;
    const uint64_t n = sorted->reads.n_pairs;
    const cm_mapped_read *states = sorted->prior;
    // eligible pairs, in input order; a chunk = consecutive eligible pairs of one contig (load_genome + refresh_hash_table_list
    // happen at contig changes in the reference)
    std::vector<uint64_t> todo;
    for (uint64_t i = 0; i < n && states; ++i) {
        const cm_mapped_read &st = states[i];
        if (st.type != CM_CHIBSJ && st.type != CM_CHI2BSJ) continue;
        if (st.contig_num < 0 || (uint32_t)st.contig_num >= n_contigs || st.chr_id < 0 || (uint32_t)st.chr_id >= n_chr) continue;
        todo.push_back(i);
    }
    int T = n_threads > 0 ? n_threads : (int)std::thread::hardware_concurrency();
    if (const char *e = getenv("CM_CIRC_THREADS")) T = atoi(e);
    T = T < 1 ? 1 : (T > 64 ? 64 : T);
    const size_t chunk_len = std::max<size_t>(256, todo.size() / (size_t)(T * 8) + 1);
    struct Chunk { size_t a, b; std::string rows; std::vector<Call> calls; int err = 0; };
    std::vector<Chunk> chunks;
    for (size_t a = 0; a < todo.size();) {
        size_t b = a;
        const int con = states[todo[a]].contig_num;
        while (b < todo.size() && b - a < chunk_len && states[todo[b]].contig_num == con) ++b;
        chunks.push_back(Chunk{a, b, {}, {}, 0});
        a = b;
    }
    std::atomic<size_t> next{0};
    auto work = [&]() {
        std::vector<uint8_t> rc1, rc2;
        for (size_t c = next.fetch_add(1); c < chunks.size(); c = next.fetch_add(1)) {
            Chunk &ck = chunks[c];
            const int con = states[todo[ck.a]].contig_num;
            Caller cur(*P, contigs[con], annots[con], ws, chrs);
            for (size_t x = ck.a; x < ck.b; ++x) {
                const uint64_t i = todo[x];
                const cm_mapped_read &st = states[i];
                const uint8_t *p1 = sorted->reads.seq1 + sorted->reads.off1[i], *p2 = sorted->reads.seq2 + sorted->reads.off2[i];
                const uint32_t l1 = (uint32_t)(sorted->reads.off1[i + 1] - sorted->reads.off1[i]), l2 = (uint32_t)(sorted->reads.off2[i + 1] - sorted->reads.off2[i]);
                reverse_complement(p1, l1, rc1);
                reverse_complement(p2, l2, rc2);
                const Seq fwd[2] = {Seq{p1, l1}, Seq{p2, l2}}, rev[2] = {Seq{rc1.data(), l1}, Seq{rc2.data(), l2}};
                cur.process(fwd, rev, st, chrs[st.chr_id].start_pos, i, sorted->names1 + sorted->name_off1[i]);
            }
            ck.rows.swap(cur.candidates);
            ck.calls.swap(cur.calls);
            ck.err = cur.err_;
        }
    };
    {
        const int nt = (int)std::min<size_t>((size_t)T, std::max<size_t>(chunks.size(), 1));
        std::vector<std::thread> th;
        for (int t = 1; t < nt; ++t) th.emplace_back(work);
        work();
        for (auto &t : th) t.join();
    }
    std::vector<Call> calls;
    uint64_t n_rows = 0;
    int rc = CM_OK;
    bool io_bad = false;
    for (Chunk &ck : chunks) {
        if (ck.err) rc = CM_ELIMIT;
        if (fwrite(ck.rows.data(), 1, ck.rows.size(), fc) != ck.rows.size()) io_bad = true;      // full disk: not a silent short file
        n_rows += (uint64_t)std::count(ck.rows.begin(), ck.rows.end(), '\n');
        calls.insert(calls.end(), ck.calls.begin(), ck.calls.end());
    }
    if (fclose(fc) != 0 || io_bad) return CM_EIO;
    std::vector<cm_circ_res> res(calls.size());
    for (size_t i = 0; i < calls.size(); ++i) {
        const Call &c = calls[i];
        res[i] = cm_circ_res{chrs[c.chr_id].name, sorted->names1 + sorted->name_off1[c.rec], c.spos, c.epos, c.type, 0,
                             c.bp.sig_start.c_str(), c.bp.sig_end.c_str(), c.bp.ref_start.c_str(), c.bp.ref_end.c_str()};
    }
    const int rr = cm_circ_report(res.data(), res.size(), report_path);
    if (rr != CM_OK) return rr;
#if defined(CM_S2_PROF)
    fprintf(stderr, "[s2 prof] ms: table build %.1f, chains_of %.1f, realign_at %.1f (inside judge), place_piece %.1f, judge_single %.1f, process total %.1f\n",
            g_s2_ns[0] / 1e6, g_s2_ns[1] / 1e6, g_s2_ns[2] / 1e6, g_s2_ns[3] / 1e6, g_s2_ns[4] / 1e6, g_s2_ns[5] / 1e6);
#endif
    if (stats) {
        memset(stats, 0, sizeof *stats);
        stats->pairs = n;
        stats->candidate_rows = n_rows;
        stats->calls = calls.size();
        stats->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
    return rc;
}

extern "C" int cm_circ_call(const cm_params *P, int32_t window_size, uint32_t n_contigs, const cm_index_view *contigs, const cm_annot_view *annots,
                            const cm_chr_info *chrs, uint32_t n_chr, const cm_fastq_batch *sorted, const char *candidates_path,
                            const char *report_path, cm_circ_stats *stats) {
    return circ_call_mt(P, window_size, n_contigs, contigs, annots, chrs, n_chr, sorted, candidates_path, report_path, stats, 0);
}

extern "C" int cm_circ_run(const cm_circ_args *a, cm_circ_stats *stats, char *err, uint64_t err_cap) {
    Fail fail{err, (size_t)err_cap};
    if (err && err_cap) err[0] = 0;
    if (!a || !a->index_path || !a->index_info_path || !a->gtf_path || !a->out_prefix) return fail(CM_EINVAL, "cm_circ_run: null argument");
    const std::string out = a->out_prefix;
    cm_chr_info *chrs = nullptr;
    uint32_t n_chr = 0;
    cm_index_file *idx = nullptr;
    cm_fastq *fq = nullptr;
    std::vector<cm_index_view> views;
    std::vector<cm_annot_view> annots;
    auto cleanup = [&]() {
        if (fq) cm_fastq_close(fq);
        if (!annots.empty()) cm_host_free_annotation(annots.data(), (uint32_t)annots.size());
        for (auto &v : views) cm_host_free_loaded_contig(&v);
        if (idx) cm_host_close_index(idx);
        if (chrs) cm_host_free_index_info(chrs, n_chr);
    };
    int rc;
#define S2_TRY(call, what)                                     \
    do {                                                       \
        rc = (call);                                           \
        if (rc != CM_OK) {                                     \
            rc = fail(rc, "%s failed (%d)", what, rc);         \
            cleanup();                                         \
            return rc;                                         \
        }                                                      \
    } while (0)
    const bool trace = getenv("CM_CIRC_TRACE") != nullptr;
    auto tp = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (trace) fprintf(stderr, "[circ] %s %.3f s\n", what, std::chrono::duration<double>(std::chrono::steady_clock::now() - tp).count());
        tp = std::chrono::steady_clock::now();
    };
    // ProcessCirc ctor: sort both remain files (sort_fq); do_process: index info, packed genome, GTF
    char r1[4096], r2[4096];
    snprintf(r1, sizeof r1, "%s_%d_remain_R1.fastq", out.c_str(), a->last_round);
    snprintf(r2, sizeof r2, "%s_%d_remain_R2.fastq", out.c_str(), a->last_round);
    const std::string s1 = std::string(r1) + ".srt", s2 = std::string(r2) + ".srt";
    // three things that do not depend on one another run side by side: the two sorts, the genome out of the index file, the GTF
    int rc_s1 = CM_OK, rc_s2 = CM_OK;
    std::thread t_s1([&]() { rc_s1 = cm_sort_remain(r1, s1.c_str()); });
    std::thread t_s2([&]() { rc_s2 = cm_sort_remain(r2, s2.c_str()); });
    struct Join2 {
        std::thread &a, &b;
        ~Join2() {
            if (a.joinable()) a.join();
            if (b.joinable()) b.join();
        }
    } join_sorts{t_s1, t_s2};
    S2_TRY(cm_host_read_index_info(a->index_info_path, &chrs, &n_chr), "cm_host_read_index_info");
    int32_t kmer = 0, full = 0;
    uint32_t n_rec = 0;
    S2_TRY(cm_host_open_index(a->index_path, &idx, &kmer, &full, &n_rec), "cm_host_open_index");
    cm_params P = a->params;
    if (P.kmer == 0) P.kmer = kmer;
    // the GTF model needs the contig lengths: those of the .index.info rows first (checked against the index file's below)
    std::vector<uint32_t> clen_guess;
    for (uint32_t i = 0; i < n_chr; ++i) {
        if (chrs[i].contig_id == 0) continue;
        if (clen_guess.size() < chrs[i].contig_id) clen_guess.resize(chrs[i].contig_id, 0);
        clen_guess[chrs[i].contig_id - 1] = std::max(clen_guess[chrs[i].contig_id - 1], chrs[i].start_pos + chrs[i].len);
    }
    std::vector<cm_annot_view> early(clen_guess.size());
    int early_rc = CM_EINVAL;
    std::thread t_gtf([&]() {
        if (!clen_guess.empty())
            early_rc = cm_host_build_annotation(a->gtf_path, chrs, n_chr, clen_guess.data(), (uint32_t)clen_guess.size(), P.max_read_len, early.data());
    });
    int grc = CM_OK;
    for (;;) {
        cm_index_view iv;
        int loaded = 0;
        grc = cm_host_next_contig_genome(idx, &iv, &loaded);      // the sequence only: stage 2 never probes the k-mer table
        if (grc != CM_OK || !loaded) break;
        views.push_back(iv);
    }
    t_gtf.join();
    lap("genome from the index file | GTF");
    if (grc != CM_OK) {
        if (early_rc == CM_OK) cm_host_free_annotation(early.data(), (uint32_t)early.size());
        S2_TRY(grc, "cm_host_next_contig_genome");
    }
    std::vector<uint32_t> clen;
    for (auto &v : views) clen.push_back(v.ref_len);
    if (early_rc == CM_OK && clen == clen_guess) {
        annots = early;
        rc = CM_OK;
    } else {
        if (early_rc == CM_OK) cm_host_free_annotation(early.data(), (uint32_t)early.size());
        annots.resize(views.size());
        rc = cm_host_build_annotation(a->gtf_path, chrs, n_chr, clen.data(), (uint32_t)views.size(), P.max_read_len, annots.data());
    }
    if (rc != CM_OK) {
        annots.clear();
        rc = fail(rc, "cm_host_build_annotation failed (%d)", rc);
        cleanup();
        return rc;
    }
    t_s1.join();
    t_s2.join();
    S2_TRY(rc_s1, "cm_sort_remain (R1)");
    S2_TRY(rc_s2, "cm_sort_remain (R2)");
    lap("sorted remain files ready");
    S2_TRY(cm_fastq_open(s1.c_str(), s2.c_str(), chrs, n_chr, P.max_ed, &fq), "cm_fastq_open (sorted remain files)");
    cm_fastq_batch b;
    S2_TRY(cm_fastq_next(fq, ~0ull >> 2, &b), "cm_fastq_next");
    lap("parse sorted remain files");
    const std::string cand = out + ".candidates.pam", rep = out + ".circ_report";
    rc = circ_call_mt(&P, a->window_size, (uint32_t)views.size(), views.data(), annots.data(), chrs, n_chr, &b, cand.c_str(), rep.c_str(), stats,
                      a->n_threads);
    if (rc != CM_OK) rc = fail(rc, "cm_circ_call failed (%d)", rc);
    lap("calling + report");
    cleanup();
    lap("cleanup");
    return rc;
#undef S2_TRY
}
