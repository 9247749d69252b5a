// ============================================================================================
// cm_oracle.cpp — CPU restatement of CircMiner 0.4.5's per-pair mapping hot path.
//
// THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke() and the
// cpu_baseline leg of bench.py may build, load or call it, and only as the checker / the reported
// CPU baseline.  The product path (circminer_amd/csrc, libcmhot.so) never links or calls it.
//
// PARITY UNPINNED: the reference cannot be built in this image (every source includes the absent
// submodule headers logger.h / mrsfast/Common.h) and it ships no tests, golden
// vectors or fixtures, so this restatement is checked only against planted-truth
// properties of synthetic data, never against reference outputs.
//
// Each function cites the reference file:line it follows.  The data it runs on is the flattened
// index / annotation of include/circminer_hot.h (pure data layout, shared with the product).
// Compile with -ffp-contract=off: chain scores are fp64 sums whose rounding order matters.
// ============================================================================================
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <set>
#include <string>
#include <vector>

#include "circminer_hot.h"

namespace {

// ---- constants, reference src/common.h:34-53 ----
const int INF_I = 1000000000;         // (int)INF, INF = 1e9
const uint32_t MINLB = 0;
const uint32_t MAXUB = 4294967295u;
const int MAXDISCRDTLEN = 20000;
const uint32_t LARIAT2BEGTH = 1000;
const int DPTINF = 10000000;          // src/align.cpp:12

struct Ctx {
    cm_params P;
    const cm_index_view *X;
    const cm_annot_view *A;
    int kmer() const { return P.kmer; }
};

// ============================ A1: hashVal / checkSumVal ============================
// Declared in the absent mrsfast/Common.h; semantics fixed by the in-tree index builder
// (src/mrsfast/HashTable.c:271-279, 799-806): A0 C1 G2 T3 MSB-first, -1 on anything else.
int pack2(const uint8_t *s, int n) {
    int v = 0;
    for (int i = 0; i < n; ++i) {
        int b;
        switch (s[i]) {
            case 'A': b = 0; break;
            case 'C': b = 1; break;
            case 'G': b = 2; break;
            case 'T': b = 3; break;
            default: return -1;
        }
        v = (v << 2) | b;
    }
    return v;
}

// ============================ A2-A4: seeding ============================
struct MatchedKmer {        // GIMatchedKmer, src/common.h:165-170
    int64_t frags;          // index of first hit in the entry arrays, -1 == NULL
    uint32_t frag_count;
    int32_t qpos;
    uint32_t raw;           // occurrences before the seedLim rule (diagnostic only)
};

inline uint32_t frag_info(const Ctx &c, const MatchedKmer &mk, uint32_t i) { return c.X->pos[mk.frags + i]; }

// GenomeSeeder::get_exact_locs_hash, src/match_read.cpp:54-110 (+ getCandidates, HashTable.c:1093-1098)
int get_exact_locs_hash(const Ctx &c, const uint8_t *seq, int32_t qpos, MatchedKmer *mk) {
    mk->frag_count = 0;
    mk->frags = -1;
    mk->qpos = qpos;
    mk->raw = 0;
    int hv = pack2(seq, CM_WINDOW_SIZE);
    if (hv < 0) return 0;
    int cl = c.kmer() - CM_WINDOW_SIZE;
    int cv = pack2(seq + CM_WINDOW_SIZE, cl);
    if (cv < 0) return 0;
    const uint32_t b0 = c.X->bucket_off[hv], b1 = c.X->bucket_off[hv + 1];
    if (b1 == b0) return 0;                      // getCandidates: NULL list or [0].info == 0
    const uint16_t *it = c.X->checksum + b0;     // it[m-1] here == it[m].checksum in the reference
    uint32_t lb = 1, ub = b1 - b0, mid;
    int16_t target = (int16_t)cv;                // quirk: int16 vs uint16 field (match_read.cpp:77)
    uint32_t LB = 0, UB = 0;
    while (lb < ub) {
        mid = (lb + ub) / 2;
        if ((int)target <= (int)it[mid - 1]) ub = mid;
        else lb = mid + 1;
    }
    if (ub < lb || (int)target != (int)it[lb - 1]) return 0;
    UB = LB = lb;
    lb = LB;
    ub = b1 - b0;
    while (lb < ub) {
        mid = (lb + ub + 1) / 2;
        if ((int)target < (int)it[mid - 1]) ub = mid - 1;
        else lb = mid;
    }
    if ((int)target == (int)it[lb - 1]) UB = lb;
    mk->frag_count = UB - LB + 1;
    mk->raw = mk->frag_count;
    mk->frags = (int64_t)b0 + (LB - 1);
    return (int)(UB - LB + 1);
}

int max_seg_cnt(const Ctx &c) { return 2 * (int)std::ceil(1.0 * c.P.max_read_len / c.kmer()) - 1; }

// GenomeSeeder::kmer_match_skip_hash(shift 0, skip k, ll_step 2) via split_match_hash,
// src/match_read.cpp:180-286
int split_match_hash(const Ctx &c, const uint8_t *rseq, int rseq_len, MatchedKmer *mk_res) {
    const int k = c.kmer();
    int msc = max_seg_cnt(c);
    for (int j = 0; j < msc; ++j) {
        mk_res[j].frag_count = 0;
        mk_res[j].frags = -1;
        mk_res[j].raw = 0;
    }
    int em = 0, invalid = 0;
    MatchedKmer *cur = mk_res;
    for (int i = 0; i < rseq_len; i += k) {
        if (rseq_len - i < k) break;
        if (i != 0) cur += 2;
        cur->qpos = i;
        int occ = get_exact_locs_hash(c, rseq + i, i, cur);
        if (occ <= 0) invalid++;
        else if ((uint32_t)occ > (uint32_t)c.P.seed_lim) invalid++;
        em++;
    }
    for (int i = 0; i < em * 2; i += 2)
        if (mk_res[i].frag_count > (uint32_t)c.P.seed_lim) mk_res[i].frag_count = 0;   // frags stays non-NULL
    return em - invalid;
}

// ============================ A7: annotation queries ============================
// FlatIntervalTree::search, src/interval_tree_impl.h:136-150
int iv_search(const cm_annot_view *A, uint32_t target) {
    int beg = 0, end = (int)A->n_iv, mid;
    while (end - beg > 1) {
        mid = (beg + end) / 2;
        if (target < A->iv_spos[mid]) end = mid;
        else beg = mid;
    }
    return end;
}
// FlatIntervalTree::find_ind, :165-175 — returns interval index or -1 (NULL); ind as in the reference
int iv_find_ind(const cm_annot_view *A, uint32_t pos, int &ind) {
    ind = -1;
    if (pos < A->iv_spos[0]) return -1;
    ind = iv_search(A, pos) - 1;
    if (ind < 0 || A->iv_epos[ind] < pos) return -1;
    return ind;
}
int iv_get_node(const cm_annot_view *A, int ind) { return (ind < 0 || ind >= (int)A->n_iv) ? -1 : ind; }   // :178-182
inline uint32_t iv_nseg(const cm_annot_view *A, int iv) { return A->iv_seg_off[iv + 1] - A->iv_seg_off[iv]; }
inline uint32_t iv_segid(const cm_annot_view *A, int iv, uint32_t i) { return A->iv_seg[A->iv_seg_off[iv] + i]; }
inline bool bit(const uint64_t *b, uint64_t n, uint64_t p) { return p < n && ((b[p >> 6] >> (p & 63)) & 1); }

// GTFParser::get_location_overlap[_ind], src/gene_annotation.cpp:538-568
int get_location_overlap_ind(const Ctx &c, uint32_t loc, int &ind) {
    int r = iv_find_ind(c.A, loc, ind);
    if (r < 0 || iv_nseg(c.A, r) == 0) return -1;
    return r;
}
int get_location_overlap(const Ctx &c, uint32_t loc) { int ind; return get_location_overlap_ind(c, loc, ind); }

// GTFParser::get_upper_bound_lookup, src/gene_annotation.cpp:464-533
uint32_t get_upper_bound_lookup(const Ctx &c, uint32_t spos, uint32_t mlen, uint32_t rlen, uint32_t &max_end,
                                int &ol_exons) {
    const cm_annot_view *A = c.A;
    max_end = 0;
    int it_ind = -1;
    int ov = iv_find_ind(A, spos, it_ind);
    uint32_t epos = spos + mlen - 1;
    if (ov < 0 || iv_nseg(A, ov) == 0) {
        ol_exons = -1;
        int nx = iv_get_node(A, it_ind + 1);
        // the reference dereferences NULL here when spos lies beyond the last interval; that is
        // unreachable through get_upper_bound because such positions are never near a border.
        max_end = (nx < 0 ? 0u : A->iv_spos[nx]) - 1;
        if (max_end < epos) return 0;
        uint32_t a = spos + rlen + (uint32_t)c.P.max_ed, b = max_end - mlen + 1;
        return a < b ? a : b;
    }
    ol_exons = -1;
    uint32_t min_end = 1000000000u, max_next_exon = 0;
    if (epos > A->iv_epos[ov]) {
        for (uint32_t i = 0; i < iv_nseg(A, ov); ++i) {
            uint32_t s = iv_segid(A, ov, i);
            if (A->seg_end[s] >= epos) {
                max_end = std::max(max_end, A->seg_end[s]);
                min_end = std::min(min_end, A->seg_end[s]);
                max_next_exon = std::max(max_next_exon, A->seg_next_exon_beg[s]);
            }
        }
    } else {
        max_end = A->iv_max_end[ov];
        min_end = A->iv_min_end[ov];
        max_next_exon = A->iv_max_next_exon[ov];
    }
    if (max_end > 0 && max_end >= epos) {
        ol_exons = ov;
        if (min_end < rlen + epos && max_next_exon != 0) return max_next_exon + mlen - 1;
        return max_end - mlen + 1;
    }
    max_end = 0;
    ol_exons = -1;
    return 0;
}

// GTFParser::get_upper_bound, src/gene_annotation.h:123-133
uint32_t get_upper_bound(const Ctx &c, uint32_t spos, uint32_t mlen, uint32_t rlen, uint32_t &max_end, int &ol_exons) {
    if (bit(c.A->near_border_bits, c.A->n_bits, spos)) return get_upper_bound_lookup(c, spos, mlen, rlen, max_end, ol_exons);
    max_end = 0;
    ol_exons = -1;
    return spos + rlen + (uint32_t)c.P.max_ed;
}

// GTFParser::get_shift, src/gene_annotation.cpp:451-457 — returns row in the contig's chr table
int get_shift(const Ctx &c, uint32_t loc) {
    uint32_t i;
    for (i = 1; i < c.A->n_chr; ++i)
        if (loc < c.A->chr_shift[i]) return (int)i - 1;
    return (int)i - 1;
}

// ============================ A6: chaining ============================
struct Frag { uint32_t rpos; int32_t qpos; uint32_t len; };            // fragment_t, common.h:138-146
struct Chain { std::vector<Frag> frags; uint32_t chain_len = 0; float score = 0; };   // chain_t
struct ChainList { std::vector<Chain> chains; int best_chain_count = 0; };            // chain_list
struct Cell { double score; int prev_list; int prev_ind; };            // chain_cell, chain.h:8-12
struct CellList { Cell chain_list[CM_BESTCHAINLIM]; uint32_t count; }; // chain_cell_list, chain.h:14-17

// check_junction, src/chain.cpp:28-64
bool check_junction(const Ctx &c, uint32_t s1, uint32_t s2, int ol_exons, int kmer, int read_dist, int &trans_dist) {
    trans_dist = INF_I;
    if (ol_exons < 0) return false;
    uint32_t e1 = s1 + kmer - 1;
    if (s2 <= e1) return false;
    int e12end, beg2s2, trans_dist2intron = -1;
    const cm_annot_view *A = c.A;
    for (uint32_t i = 0; i < iv_nseg(A, ol_exons); ++i) {
        uint32_t s = iv_segid(A, ol_exons, i);
        e12end = (int)(A->seg_end[s] - e1);
        beg2s2 = (int)(s2 - A->seg_next_exon_beg[s]);
        if (e12end >= 0 && e12end < read_dist && beg2s2 + kmer < 0) trans_dist2intron = (int)(s2 - e1 - 1);
        if (e12end < 0 || beg2s2 < 0) continue;
        trans_dist = e12end + beg2s2;
        if (std::abs(trans_dist - read_dist) <= c.P.max_ed) return true;
    }
    if (trans_dist2intron != -1) {
        trans_dist = trans_dist2intron;
        return true;
    }
    trans_dist = INF_I;
    return false;
}

// chain_seeds_sorted_kbest, src/chain.cpp:73-301
void chain_seeds_sorted_kbest(const Ctx &c, int seq_len, MatchedKmer *fl, ChainList &best_chain) {
    const int kmer = c.kmer();
    best_chain.best_chain_count = 0;
    int kmer_cnt = 2 * (int)std::ceil(1.0 * seq_len / kmer) - 1;
    const uint32_t max_best = (uint32_t)c.P.max_chain_len;
    while (kmer_cnt >= 1 && fl[kmer_cnt - 1].frag_count <= 0) kmer_cnt--;
    if (kmer_cnt <= 0) return;

    std::vector<std::vector<Cell>> dp(kmer_cnt);
    for (int ii = kmer_cnt - 1; ii >= 0; ii--) {
        dp[ii].resize(fl[ii].frag_count);
        for (uint32_t i = 0; i < fl[ii].frag_count; i++) dp[ii][i] = Cell{(double)kmer, -1, -1};
    }
    std::map<double, CellList> score2chain;
    std::vector<uint32_t> lb_ind(kmer_cnt);
    uint32_t max_exon_end = 0;
    int ol_exons = -1;

    for (int ii = kmer_cnt - 2; ii >= 0; ii--) {
        MatchedKmer *cur_mk = fl + ii;
        uint32_t read_remain = (uint32_t)(seq_len - cur_mk->qpos - kmer);
        for (int k = 0; k < kmer_cnt; k++) lb_ind[k] = 0;
        for (uint32_t i = 0; i < cur_mk->frag_count; i++) {
            const int32_t cur_info = (int32_t)frag_info(c, *cur_mk, i);      // GeneralIndex.info is int
            uint32_t seg_start = (uint32_t)cur_info;
            uint32_t seg_end = (uint32_t)cur_info + kmer - 1;
            uint32_t max_lpos_lim = MAXUB;
            for (int jj = ii + 1; jj < kmer_cnt; jj++) {
                MatchedKmer *pc = fl + jj;
                if (pc->frag_count <= 0 || lb_ind[jj] >= pc->frag_count) continue;
                if (cur_info + c.P.max_intron < (int32_t)frag_info(c, *pc, lb_ind[jj])) continue;
                while (lb_ind[jj] < pc->frag_count && (int32_t)frag_info(c, *pc, lb_ind[jj]) <= cur_info) lb_ind[jj]++;
                if (lb_ind[jj] >= pc->frag_count) continue;
                if (max_lpos_lim == MAXUB)
                    max_lpos_lim = get_upper_bound(c, seg_start, (uint32_t)kmer, read_remain, max_exon_end, ol_exons);
                int distr = pc->qpos - cur_mk->qpos - kmer;
                int read_dist = distr;
                uint32_t j = lb_ind[jj];
                while (j < pc->frag_count && frag_info(c, *pc, j) <= max_lpos_lim) {
                    uint32_t pinfo = frag_info(c, *pc, j);
                    int genome_dist, distt, trans_dist;
                    if (max_exon_end == 0 || (pinfo + kmer - 1) <= max_exon_end) genome_dist = (int)(pinfo - seg_end - 1);
                    else genome_dist = INF_I;
                    if (std::abs(genome_dist - read_dist) <= c.P.max_ed) {
                        distt = genome_dist;
                    } else if (check_junction(c, seg_start, pinfo, ol_exons, kmer, read_dist, trans_dist)) {
                        distt = trans_dist;
                    } else {
                        j++;
                        continue;
                    }
                    // score_alpha / score_beta, chain.cpp:13-22: (prev + 2e4*k) - 0.1*|distr-distt|
                    int maxd = distr < distt ? distt : distr, mind = distr < distt ? distr : distt;
                    double beta = 0.1 * (maxd - mind);
                    double alpha = 2e4 * kmer;
                    double temp_score = dp[jj][j].score + alpha - beta;
                    if (temp_score > dp[ii][i].score) {
                        dp[ii][i].score = temp_score;
                        dp[ii][i].prev_list = jj;
                        dp[ii][i].prev_ind = (int)j;
                        auto it = score2chain.find(temp_score);
                        if (it == score2chain.end()) {
                            CellList e;
                            e.count = 0;
                            it = score2chain.insert(std::make_pair(temp_score, e)).first;
                        }
                        if (it->second.count < max_best) {
                            Cell t{dp[ii][i].score, ii, (int)i};
                            it->second.chain_list[it->second.count++] = t;
                        }
                    }
                    j++;
                }
            }
        }
    }

    uint32_t best_count = 0;
    double best_score = score2chain.empty() ? (double)kmer : score2chain.rbegin()->first;
    std::set<uint32_t> repeats;
    for (auto it = score2chain.rbegin(); it != score2chain.rend(); ++it) {
        for (uint32_t l = 0; l < it->second.count; l++) {
            if (best_count >= max_best) break;
            Cell bi = it->second.chain_list[l];
            uint32_t spos = frag_info(c, fl[bi.prev_list], (uint32_t)bi.prev_ind);
            if (bi.score < best_score && repeats.find(spos) != repeats.end()) continue;
            uint32_t i = 0, j = best_count++;
            Chain &ch = best_chain.chains[j];
            ch.frags.clear();
            while (bi.prev_list != -1) {
                Frag f{frag_info(c, fl[bi.prev_list], (uint32_t)bi.prev_ind), fl[bi.prev_list].qpos, (uint32_t)kmer};
                ch.frags.push_back(f);
                if (i != 0) repeats.insert(f.rpos);
                int tl = bi.prev_list;
                bi.prev_list = dp[tl][bi.prev_ind].prev_list;
                bi.prev_ind = dp[tl][bi.prev_ind].prev_ind;
                i++;
            }
            ch.score = (float)bi.score;
            ch.chain_len = i;
        }
    }
    if (best_count == 0) {
        for (int ii = kmer_cnt - 1; ii >= 0; ii--) {
            for (uint32_t i = 0; i < fl[ii].frag_count; i++) {
                if (best_count >= max_best) break;
                uint32_t j = best_count++;
                Chain &ch = best_chain.chains[j];
                ch.frags.assign(1, Frag{frag_info(c, fl[ii], i), fl[ii].qpos, (uint32_t)kmer});
                ch.score = (float)dp[ii][i].score;
                ch.chain_len = 1;
            }
        }
    }
    best_chain.best_chain_count = (int)best_count;
}

// FilterRead::get_best_chains, src/filter.cpp:469-482
void get_best_chains(const Ctx &c, const uint8_t *seq, int seq_len, ChainList &bc, MatchedKmer *fl, int &high_hits) {
    int msc = max_seg_cnt(c);
    split_match_hash(c, seq, seq_len, fl);
    chain_seeds_sorted_kbest(c, seq_len, fl, bc);
    high_hits = 0;
    for (int i = 0; i < msc; i += 2)
        if (fl[i].frags >= 0 && fl[i].frag_count == 0) high_hits++;
}

// ============================ A14/A15/A16: alignment ============================
// ScoreMatrix::init, src/align.cpp:739-760: case-insensitive ACGT identity, everything else
// (including N vs N) is a mismatch.
inline bool same_base(uint8_t a, uint8_t b) {
    uint8_t x = a & 0xDF, y = b & 0xDF;
    return x == y && (x == 'A' || x == 'C' || x == 'G' || x == 'T') &&
           ((a >= 'A' && a <= 'Z') || (a >= 'a' && a <= 'z')) && ((b >= 'A' && b <= 'Z') || (b >= 'a' && b <= 'z'));
}
inline int diff_ch(uint8_t a, uint8_t b) { return same_base(a, b) ? 0 : 1; }        // edit_mat.init(0,1,1,..)
inline int score_ch(uint8_t a, uint8_t b) { return same_base(a, b) ? 1 : -3; }      // score_mat.init(1,-3,-3,8)
const int SC_IND = -3, SC_XD = 8, SC_MAT = 1, SC_MIS = -3;

struct AlignRes {      // src/align.h:12-121
    uint32_t pos; int ed, sclen, indel, qcovlen, rcovlen, score;
    explicit AlignRes(uint32_t p) : pos(p), ed(0), sclen(0), indel(0), qcovlen(0), rcovlen(0), score(-INF_I) {}
    void set(uint32_t p, int e, int s, int i, int qc, int scr) { pos = p; ed = e; sclen = s; indel = i; qcovlen = qc; rcovlen = qc - i; score = scr; }
    void update(int e, int s, uint32_t np, int i, int qc, int scr) { pos = np; ed += e; sclen = s; indel += i; qcovlen += qc; rcovlen += qc - i; score = scr; }
};
bool update_by_score_right(AlignRes &b, const AlignRes &r) {
    if (b.score < r.score || (b.score == r.score && r.pos < b.pos)) { b.set(r.pos, r.ed, r.sclen, r.indel, r.qcovlen, r.score); return true; }
    return false;
}
bool update_by_score_left(AlignRes &b, const AlignRes &r) {
    if (b.score < r.score || (b.score == r.score && r.pos > b.pos)) { b.set(r.pos, r.ed, r.sclen, r.indel, r.qcovlen, r.score); return true; }
    return false;
}
void update_side(const Ctx &c, AlignRes &b, const AlignRes &r, bool right) {   // update_right / update_left
    if (r.qcovlen > b.qcovlen) {
        int pre_ed = b.ed;
        if (r.ed <= c.P.max_ed && r.sclen <= c.P.max_sc && 2 * (r.ed - pre_ed) < (r.qcovlen - b.qcovlen))
            b.set(r.pos, r.ed, r.sclen, r.indel, r.qcovlen, r.score);
    } else if (r.qcovlen < b.qcovlen) {
        if (r.ed <= c.P.max_ed && r.sclen <= c.P.max_sc && 2 * (b.ed - r.ed) >= (b.qcovlen - r.qcovlen))
            b.set(r.pos, r.ed, r.sclen, r.indel, r.qcovlen, r.score);
    } else {
        bool pos_better = right ? (r.pos < b.pos) : (r.pos > b.pos);
        if ((r.ed < b.ed) || (r.ed == b.ed && r.sclen < b.sclen) || (r.ed == b.ed && r.sclen == b.sclen && pos_better))
            b.set(r.pos, r.ed, r.sclen, r.indel, r.qcovlen, r.score);
    }
}
struct AlignCandid {   // src/align.h:123-153
    int ed, sclen, indel, score;
    AlignCandid(int e, int s, int i) : ed(e), sclen(s), indel(i), score(-1 * s - 2 * e) {}
    AlignCandid(int e, int s, int i, int scr) : ed(e), sclen(s), indel(i), score(scr) {}
    bool operator<(const AlignCandid &r) const {
        if (score != r.score) return score > r.score;
        if (ed != r.ed) return ed < r.ed;
        return std::abs(indel) < std::abs(r.indel);
    }
    void update(const AlignCandid &r) { if (r < *this) *this = r; }
};

// The reference keeps 600x600 member matrices that are never cleared and relies on sentinels
// written per call.  Here every call gets its own matrix pre-filled with a poison value and any
// read of a poisoned cell aborts: that is the proof that no call depends on stale cells.
struct Mat {
    int n1, m1;
    std::vector<int64_t> v;
    static int64_t poison() { return INT64_MIN; }
    Mat(int n, int m) : n1(n + 8), m1(m + 8), v((size_t)(n + 8) * (m + 8), poison()) {}
    int64_t &w(int i, int j) { return v[(size_t)i * m1 + j]; }
    int64_t r(int i, int j) const {
        int64_t x = v[(size_t)i * m1 + j];
        if (x == poison()) { fprintf(stderr, "oracle: stale DP cell read (%d,%d)\n", i, j); abort(); }
        return x;
    }
};
inline int64_t min3(int64_t a, int64_t b, int64_t c) { return std::min(std::min(a, b), c); }
inline int64_t max3(int64_t a, int64_t b, int64_t c) { return std::max(std::max(a, b), c); }

// Alignment::global_alignment / _reverse, src/align.cpp:166-212
void global_alignment(Mat &dp, const uint8_t *s, int n, const uint8_t *t, int m, bool rev) {
    for (int i = 0; i <= n; i++) dp.w(i, 0) = i;
    for (int j = 0; j <= m; j++) dp.w(0, j) = j;
    for (int i = 1; i <= n; i++)
        for (int j = 1; j <= m; j++) {
            int d = rev ? diff_ch(s[n - i], t[m - j]) : diff_ch(s[i - 1], t[j - 1]);
            dp.w(i, j) = min3(dp.r(i - 1, j - 1) + d, dp.r(i - 1, j) + 1, dp.r(i, j - 1) + 1);
        }
}
// Alignment::global_one_side_banded_alignment, src/align.cpp:219-252
int global_one_side_banded_alignment(const uint8_t *s, int n, const uint8_t *t, int m, int w) {
    Mat dp(std::max(n, 0) + w + 2, std::max(m, 0) + w + 2);
    if (w < 0 || n <= w) {
        global_alignment(dp, s, n, t, m, false);
        return (int)dp.r(n, m);
    }
    int i, j;
    j = 0;
    for (i = 1; i <= n; i++) dp.w(i, j++) = DPTINF;
    i = 0;
    for (j = w + 1; j <= m; j++) dp.w(i++, j) = DPTINF;
    for (j = 0; j <= w; j++) dp.w(0, j) = j;
    for (i = 1; i <= n; i++)
        for (j = i; j <= i + w; j++)
            dp.w(i, j) = min3(dp.r(i - 1, j - 1) + diff_ch(s[i - 1], t[j - 1]), dp.r(i - 1, j) + 1, dp.r(i, j - 1) + 1);
    return (int)dp.r(n, m);
}
// Alignment::global_banded_alignment / _reverse, src/align.cpp:395-509
void global_banded_alignment(Mat &dp, const uint8_t *s, int n, const uint8_t *t, int m, int w, bool rev) {
    if (w < 0 || n <= 2 * w || m <= w) {
        global_alignment(dp, s, n, t, m, rev);
        return;
    }
    auto D = [&](int i, int j) { return rev ? diff_ch(s[n - i], t[m - j]) : diff_ch(s[i - 1], t[j - 1]); };
    int i, j;
    j = 0;
    for (i = w + 1; i <= n; i++) dp.w(i, j++) = DPTINF;
    i = 0;
    for (j = w + 1; j <= m; j++) dp.w(i++, j) = DPTINF;
    for (i = 0; i <= w; i++) { dp.w(i, 0) = i; dp.w(0, i) = i; }
    for (j = 1; j <= w; j++)
        for (i = 1; i <= j + w; i++) dp.w(i, j) = min3(dp.r(i - 1, j - 1) + D(i, j), dp.r(i - 1, j) + 1, dp.r(i, j - 1) + 1);
    // (the reference runs this loop to n - w even when that is beyond column m; those columns read an unset border cell and
    // feed nothing back into columns <= m, so they are not computed here)
    for (j = w + 1; j <= std::min(n - w, m); j++)
        for (i = j - w; i <= j + w; i++) dp.w(i, j) = min3(dp.r(i - 1, j - 1) + D(i, j), dp.r(i - 1, j) + 1, dp.r(i, j - 1) + 1);
    for (j = n - w + 1; j <= m; j++)
        for (i = j - w; i <= n; i++) dp.w(i, j) = min3(dp.r(i - 1, j - 1) + D(i, j), dp.r(i - 1, j) + 1, dp.r(i, j - 1) + 1);
}
// Alignment::local_alignment_right / _left, src/align.cpp:556-600
int local_alignment_side(const Ctx &c, const uint8_t *s, int n, const uint8_t *t, int m, int &indel, int &align_score, bool rev) {
    const int max_indel = c.P.band;
    const uint32_t max_edit = (uint32_t)c.P.max_ed;
    Mat dp(std::max(n, m) + 2 * c.P.band + 4, std::max(n, m) + 2 * c.P.band + 4);
    global_banded_alignment(dp, s, n, t, m, c.P.band, rev);
    AlignCandid best((int)max_edit + 1, c.P.max_sc + 1, max_indel + 1);
    for (int i = std::max(0, m - max_indel); i <= std::min(m + max_indel, n); i++) {
        int64_t v = dp.r(i, m);
        if ((uint32_t)v <= max_edit) best.update(AlignCandid((int)v, 0, m - i));
    }
    align_score = -1 * best.ed;
    indel = best.indel;
    return best.ed;
}
// Alignment::global_banded_alignment_drop, src/align.cpp:254-390
void global_banded_alignment_drop(Mat &dpx, const uint8_t *s, int n, const uint8_t *t, int m, int w, int &on_s, int &on_t) {
    int32_t pre_optimum = 0, cur_optimum = 0;
    int i, j, k;
    j = 0;
    for (i = w + 1; i <= n; i++) dpx.w(i, j++) = -DPTINF;
    i = 0;
    for (j = w + 1; j <= m; j++) dpx.w(i++, j) = -DPTINF;
    for (i = 0; i <= w; i++) { dpx.w(i, 0) = i * SC_IND; dpx.w(0, i) = i * SC_IND; }
    on_s = 0;
    on_t = 0;
    if (m <= 0 || n <= 0) return;
    int lb = 1, ub = 1;
    int new_ub, pre_ub = 0, best_i = 0, best_j = 0;
    for (k = 2; k <= m + n; ++k) {
        new_ub = -1;
        for (i = lb; i <= ub; ++i) {
            j = k - i;
            int64_t v = max3(dpx.r(i - 1, j - 1) + score_ch(s[i - 1], t[j - 1]), dpx.r(i - 1, j) + SC_IND, dpx.r(i, j - 1) + SC_IND);
            dpx.w(i, j) = v;
            cur_optimum = std::max<int32_t>(cur_optimum, (int32_t)v);
            if (v >= cur_optimum) { cur_optimum = (int32_t)v; best_i = i; best_j = j; }
            if (v + SC_XD < pre_optimum) dpx.w(i, j) = -DPTINF;
            if (dpx.r(i, j) > -DPTINF) new_ub = i;
        }
        int lb_t = k - lb;
        if (lb_t == m || (k > w && ((k - w) % 2 == 0))) ++lb;
        if (ub < n && (k <= w || (k > w && ((k - w) % 2 == 1)))) ++ub;
        if ((pre_ub == -1 && new_ub == -1) || lb > ub) break;
        pre_ub = new_ub;
        pre_optimum = std::max(pre_optimum, cur_optimum);
    }
    on_s = best_i;
    on_t = best_j;
}
// DropAlignment::local_alignment_right_sc / _left_sc, src/align.cpp:669-723
int local_alignment_sc(const Ctx &c, const uint8_t *s, int n, const uint8_t *t, int m, int &sc_len, int &indel, int &align_score, bool left) {
    const int max_indel = c.P.band;
    const uint32_t max_edit = (uint32_t)c.P.max_ed;
    std::vector<uint8_t> rs, rt;
    if (left) {   // reverse_str, src/utils.cpp:819-824
        rs.assign(s, s + n); std::reverse(rs.begin(), rs.end());
        rt.assign(t, t + m); std::reverse(rt.begin(), rt.end());
        s = rs.data(); t = rt.data();
    }
    int on_s, on_t;
    Mat dpx(std::max(std::max(n, m), 0) + c.P.band + 4, std::max(std::max(n, m), 0) + c.P.band + 4);
    global_banded_alignment_drop(dpx, s, n, t, m, c.P.band, on_s, on_t);
    int32_t score = (int32_t)dpx.r(on_s, on_t);
    uint32_t ed = (uint32_t)((SC_MAT * std::max(on_s, on_t) - score) / (SC_MAT - SC_MIS));
    int indel_cnt = on_t - on_s, clip = m - on_t;
    AlignCandid best((int)max_edit + 1, std::max(c.P.max_sc, m) + 1, max_indel + 1, 0);
    if (ed <= max_edit) {
        AlignCandid cand((int)ed, clip, indel_cnt, score);
        if (left) best = cand;          // best.set(...)   align.cpp:714
        else best.update(cand);         // best.update(...) align.cpp:683
    }
    align_score = score;
    sc_len = best.sclen;
    indel = best.indel;
    return best.ed;
}

// EditDistAlignment::local_alignment_right_sc / _left_sc, src/align.cpp:602-660 (stage 2 only): full banded edit DP, then the
// best (soft clip, indel) end cell under AlignCandid's order; align_score = m - sclen - 2 ed.
int local_alignment_sc_edit(const Ctx &c, const uint8_t *s, int n, const uint8_t *t, int m, int &sc_len, int &indel, int &align_score, bool left) {
    const int max_sclen = std::min(c.P.max_sc, m);
    const int max_indel = c.P.band;
    const uint32_t max_edit = (uint32_t)c.P.max_ed;
    Mat dp(std::max(std::max(n, m), 0) + 2 * c.P.band + 4, std::max(std::max(n, m), 0) + 2 * c.P.band + 4);
    global_banded_alignment(dp, s, n, t, m, c.P.band, left);
    AlignCandid best((int)max_edit + 1, c.P.max_sc + 1, max_indel + 1);
    for (int j = m; j >= m - max_sclen; j--)
        for (int i = std::max(0, j - max_indel); i <= std::min(j + max_indel, n); i++) {
            int64_t v = dp.r(i, j);
            if ((uint32_t)v <= max_edit) best.update(AlignCandid((int)v, m - j, j - i));
        }
    if (m <= c.P.max_ed) best.update(AlignCandid(m, 0, 0));
    align_score = m - best.sclen - 2 * best.ed;
    sc_len = best.sclen;
    indel = best.indel;
    return best.ed;
}

// GenomeSeeder::pac2char, src/match_read.cpp:288-299.  start == 0 reads one byte in front of the
// malloc'd contig string in the reference; with glibc that byte is the (zero) top byte of the
// mmap chunk size, so strncpy yields an all-NUL window — reproduced here as zeros.
bool pac2char(const Ctx &c, uint32_t start, int len, std::vector<uint8_t> &out) {
    int ref_len = (int)c.X->ref_len;
    if ((int)start < 0 || (int)start + len - 1 > ref_len) return false;
    out.assign((size_t)std::max(len, 0) + 1, 0);
    if (start == 0) return true;
    for (int i = 0; i < len; ++i) out[i] = c.X->genome[start - 1 + i];
    return true;
}

// ============================ mates / reads ============================
struct MatchedMate {     // src/common.h:260-307, ctor src/common.cpp:147-152
    uint32_t spos = 0, epos = 0, qspos = 0, qepos = 0;
    int right_ed, left_ed, middle_ed;
    int sclen_right = 0, sclen_left = 0;
    uint32_t matched_len = 0;
    int dir = 0;
    int type = CM_ORPHAN;
    uint16_t junc_num = 0;
    bool is_concord = false, left_ok = false, right_ok = false;
    bool looked_up_spos = false, looked_up_epos = false;
    int exon_ind_spos = -1, exon_ind_epos = -1;
    int exons_spos = -1, exons_epos = -1;    // interval index, -1 == NULL
    explicit MatchedMate(const Ctx &c) : right_ed(c.P.max_ed + 1), left_ed(c.P.max_ed + 1), middle_ed(c.P.max_ed + 1) {}
};

void default_mr(const Ctx &c, cm_mapped_read &m) {   // fill_map_info else-branch, fastq_parser.cpp:243-267
    memset(&m, 0, sizeof(m));
    m.r1_forward = 1; m.r2_forward = 1;
    m.ed_r1 = c.P.max_ed + 1; m.ed_r2 = c.P.max_ed + 1;
    m.type = CM_NOPROC_NOMATCH; m.tlen = INF_I; m.chr_id = -1;
}
inline int ed_of(const MatchedMate &m) { return m.left_ed + m.middle_ed + m.right_ed; }

// MatchedRead::go_for_update, src/common.cpp:362-411
bool go_for_update(const cm_mapped_read &t, const MatchedMate &r1, const MatchedMate &r2, int32_t tlen, bool gm, int type) {
    if (type < t.type) return true;
    if (type > t.type) return false;
    if (gm && !t.gm_compatible) return true;
    if (!gm && t.gm_compatible) return false;
    int edit_dist = ed_of(r1) + ed_of(r2);
    uint32_t ml = r1.matched_len + r2.matched_len;
    if (type < CM_CHIBSJ) {
        if ((t.ed_r1 + t.ed_r2) > edit_dist) return true;
        if ((t.ed_r1 + t.ed_r2) < edit_dist) return false;
        if (t.tlen > tlen) return true;
        if (t.tlen < tlen) return false;
        if ((t.mlen_r1 + t.mlen_r2) < ml) return true;
        if ((t.mlen_r1 + t.mlen_r2) > ml) return false;
    } else {
        if ((t.mlen_r1 + t.mlen_r2) < ml) return true;
        if ((t.mlen_r1 + t.mlen_r2) > ml) return false;
        if ((t.ed_r1 + t.ed_r2) > edit_dist) return true;
        if ((t.ed_r1 + t.ed_r2) < edit_dist) return false;
    }
    return false;
}
// MatchedRead::update, src/common.cpp:286-351
bool mr_update(const Ctx &c, cm_mapped_read &t, const MatchedMate &r1, const MatchedMate &r2, int chr_row, int32_t tlen,
               uint16_t jun_between, bool gm, int type, bool r1_first) {
    if (!go_for_update(t, r1, r2, tlen, gm, type)) return false;
    uint32_t shift = c.A->chr_shift[chr_row];
    t.type = type;
    t.chr_id = c.A->chr_id[chr_row];
    const MatchedMate &a = r1_first ? r1 : r2, &b = r1_first ? r2 : r1;
    t.spos_r1 = a.spos - shift; t.epos_r1 = a.epos - shift; t.qspos_r1 = a.qspos; t.qepos_r1 = a.qepos;
    t.mlen_r1 = a.matched_len; t.ed_r1 = ed_of(a);
    t.spos_r2 = b.spos - shift; t.epos_r2 = b.epos - shift; t.qspos_r2 = b.qspos; t.qepos_r2 = b.qepos;
    t.mlen_r2 = b.matched_len; t.ed_r2 = ed_of(b);
    t.r1_forward = a.dir > 0; t.r2_forward = b.dir > 0;
    t.tlen = tlen;
    t.junc_num = (uint16_t)(jun_between + r1.junc_num + r2.junc_num);
    t.gm_compatible = gm;
    t.contig_num = c.X->contig_num;
    return true;
}
bool mr_update_type(cm_mapped_read &t, int type) { if (type < t.type) { t.type = type; return true; } return false; }

// ============================ A17 helpers (utils.cpp) ============================
void update_match_mate_info(const Ctx &c, bool lok, bool rok, int err, MatchedMate &mm) {   // utils.cpp:22-32
    mm.left_ok = lok && (mm.sclen_left <= c.P.max_sc);
    mm.right_ok = rok && (mm.sclen_right <= c.P.max_sc);
    if (lok && rok && (err <= c.P.max_ed) && (mm.sclen_right <= c.P.max_sc) && (mm.sclen_left <= c.P.max_sc)) {
        mm.is_concord = true;
        mm.type = CM_CONCRD;
    } else if (lok || rok) mm.type = CM_CANDID;
    else mm.type = CM_ORPHAN;
}
int estimate_middle_error(const Ctx &c, const Chain &ch) {   // utils.cpp:35-49
    int mid_err = 0;
    for (uint32_t i = 0; i + 1 < ch.chain_len; i++) {
        if (ch.frags[i + 1].qpos > (int32_t)(ch.frags[i].qpos + ch.frags[i].len)) {
            int diff = (int)(ch.frags[i + 1].rpos - ch.frags[i].rpos) - (ch.frags[i + 1].qpos - ch.frags[i].qpos);
            if (diff == 0) mid_err++;
            else if (diff > 0 && diff <= c.P.band) mid_err += diff;
            else if (diff < 0 && diff >= -c.P.band) mid_err -= diff;
        }
    }
    return mid_err;
}
// is_concord / is_concord2, utils.cpp:116-153
bool is_concord_impl(const Chain &a, uint32_t seq_len, MatchedMate &mr, bool v2) {
    if (a.chain_len < 2) {
        mr.is_concord = false;
    } else {
        const Frag &l = a.frags[a.chain_len - 1];
        if ((uint32_t)(l.qpos + l.len - a.frags[0].qpos) >= seq_len) {
            mr.is_concord = true;
            mr.type = CM_CONCRD;
            mr.spos = a.frags[0].rpos;
            mr.epos = l.rpos + l.len - 1;
            mr.matched_len = l.qpos + l.len - a.frags[0].qpos;
            mr.qspos = a.frags[0].qpos;
            mr.qepos = l.qpos + l.len - 1;
        } else {
            mr.is_concord = false;
            if (v2 && (a.frags[0].qpos == 0 || (uint32_t)(l.qpos + l.len) == seq_len)) mr.type = CM_CANDID;
        }
    }
    return mr.is_concord;
}
void overlap_to_epos(const Ctx &c, MatchedMate &mr) {   // utils.cpp:667-674
    if (mr.looked_up_epos || mr.exons_epos >= 0) return;
    mr.exons_epos = get_location_overlap_ind(c, mr.epos, mr.exon_ind_epos);
    mr.looked_up_epos = true;
}
void overlap_to_spos(const Ctx &c, MatchedMate &mr) {   // utils.cpp:676-683
    if (mr.looked_up_spos || mr.exons_spos >= 0) return;
    mr.exons_spos = get_location_overlap_ind(c, mr.spos, mr.exon_ind_spos);
    mr.looked_up_spos = true;
}
// calc_tlen, utils.cpp:53-113
int calc_tlen(const Ctx &c, const MatchedMate &sm, const MatchedMate &lm, int &intron_num) {
    const cm_annot_view *A = c.A;
    int min_tlen = INF_I;
    for (uint32_t i = 0; i < iv_nseg(A, sm.exons_epos); i++) {
        uint32_t sg = iv_segid(A, sm.exons_epos, i);
        for (uint32_t j = A->seg_tid_off[sg]; j < A->seg_tid_off[sg + 1]; j++) {
            uint32_t tid = A->seg_tid[j];
            int start_ind = A->trans_start_ind[tid];
            uint32_t start_table_ind = (uint32_t)(sm.exon_ind_epos - start_ind);   // unsigned: "< 0" never true
            uint32_t end_table_ind = (uint32_t)(lm.exon_ind_spos - start_ind);
            uint32_t tsz = A->t2s_off[tid + 1] - A->t2s_off[tid];
            const uint8_t *t2s = A->t2s + A->t2s_off[tid];
            if (lm.exon_ind_spos < start_ind || end_table_ind >= tsz || t2s[end_table_ind] == 0) continue;
            int in, tlen;
            if (start_table_ind == end_table_ind) {
                in = 0;
                tlen = (int)(lm.spos - sm.epos + 1);
            } else {
                bool pre_zero = false;
                in = 0;
                tlen = (int)(A->iv_epos[sm.exons_epos] - sm.epos + 1);
                int this_it_ind = sm.exon_ind_epos;
                for (uint32_t k = start_table_ind + 1; k < end_table_ind; k++) {
                    this_it_ind++;
                    if (t2s[k] != 0) {
                        tlen += (int)(A->iv_epos[this_it_ind] - A->iv_spos[this_it_ind] + 1);
                        pre_zero = false;
                    } else {
                        if (!pre_zero) in++;
                        pre_zero = true;
                    }
                }
                tlen += (int)(lm.spos - A->iv_spos[lm.exons_spos] + 1);
            }
            if (tlen < min_tlen) { intron_num = in; min_tlen = tlen; }
        }
    }
    return (min_tlen == INF_I) ? -1 : (int)(min_tlen + sm.matched_len - 1 + lm.matched_len - 1);
}
bool seg_same_exon(const cm_annot_view *A, uint32_t a, uint32_t b) { return A->seg_start[a] == A->seg_start[b] && A->seg_end[a] == A->seg_end[b]; }
bool same_gene_mm(const Ctx &c, const MatchedMate &mm, const MatchedMate &other) {   // utils.cpp:629-639
    const cm_annot_view *A = c.A;
    for (uint32_t i = 0; i < iv_nseg(A, mm.exons_spos); i++) {
        uint32_t g = A->seg_gene_id[iv_segid(A, mm.exons_spos, i)];
        if (A->gene_start[g] <= other.spos && other.epos <= A->gene_end[g]) return true;
    }
    return false;
}
bool same_gene_iv(const Ctx &c, int mate_iv, uint32_t s, uint32_t e) {   // utils.cpp:617-627
    const cm_annot_view *A = c.A;
    for (uint32_t i = 0; i < iv_nseg(A, mate_iv); i++) {
        uint32_t g = A->seg_gene_id[iv_segid(A, mate_iv, i)];
        if (A->gene_start[g] <= s && e <= A->gene_end[g]) return true;
    }
    return false;
}
// same_transcript (2-interval form) + intersect_trans, utils.cpp:322-354
bool same_transcript(const Ctx &c, int s, int r, std::vector<uint32_t> &common_tid) {
    common_tid.clear();
    if (s < 0 || r < 0) return false;
    const cm_annot_view *A = c.A;
    std::vector<uint32_t> t1, t2;
    for (uint32_t i = 0; i < iv_nseg(A, s); i++) { uint32_t g = iv_segid(A, s, i); for (uint32_t k = A->seg_tid_off[g]; k < A->seg_tid_off[g + 1]; k++) t1.push_back(A->seg_tid[k]); }
    for (uint32_t i = 0; i < iv_nseg(A, r); i++) { uint32_t g = iv_segid(A, r, i); for (uint32_t k = A->seg_tid_off[g]; k < A->seg_tid_off[g + 1]; k++) t2.push_back(A->seg_tid[k]); }
    for (uint32_t a : t1)
        for (uint32_t b : t2)
            if (a == b) { common_tid.push_back(a); break; }
    return !common_tid.empty();
}
// concordant_explanation, utils.cpp:157-213
bool concordant_explanation(const Ctx &c, const MatchedMate &sm, const MatchedMate &lm, cm_mapped_read &mr, int chr_row, bool r1_sm, int pair_type) {
    if (sm.spos > lm.spos) return false;
    const cm_annot_view *A = c.A;
    int32_t tlen;
    bool on_cdna = (sm.exons_spos >= 0) && (sm.exons_epos >= 0) && (lm.exons_spos >= 0) && (lm.exons_epos >= 0);
    if (sm.exons_spos < 0 || lm.exons_spos < 0) {
        tlen = (int32_t)(lm.spos - sm.epos - 1 + lm.matched_len + sm.matched_len);
        if (tlen <= c.P.max_tlen) mr_update(c, mr, sm, lm, chr_row, tlen, 0, false, CM_CONGNM, r1_sm);
        else if (tlen <= MAXDISCRDTLEN) mr_update(c, mr, sm, lm, chr_row, tlen, 0, false, CM_CONGNM, r1_sm);
    } else {
        for (uint32_t i = 0; i < iv_nseg(A, sm.exons_spos); i++)
            for (uint32_t j = 0; j < iv_nseg(A, lm.exons_spos); j++)
                if (seg_same_exon(A, iv_segid(A, sm.exons_spos, i), iv_segid(A, lm.exons_spos, j))) {
                    tlen = (int32_t)(lm.spos + lm.matched_len - sm.spos);
                    if (tlen <= c.P.max_tlen) mr_update(c, mr, sm, lm, chr_row, tlen, 0, on_cdna, (pair_type == 0) ? CM_CONCRD : CM_CONGEN, r1_sm);
                    else mr_update(c, mr, sm, lm, chr_row, tlen, 0, on_cdna, CM_DISCRD, r1_sm);
                }
    }
    if (sm.exons_epos < 0 || lm.exons_spos < 0) {
        tlen = (int32_t)(lm.spos - sm.epos - 1 + sm.matched_len + lm.matched_len);
        if (tlen <= c.P.max_tlen) mr_update(c, mr, sm, lm, chr_row, tlen, 0, false, CM_CONGNM, r1_sm);
        else if (tlen <= MAXDISCRDTLEN) mr_update(c, mr, sm, lm, chr_row, tlen, 0, false, CM_CONGNM, r1_sm);
    } else {
        int intron_num = 0;   // always assigned before use: by calc_tlen when tlen >= 0, else by the tlen < 0 branch
        tlen = calc_tlen(c, sm, lm, intron_num);
        if (tlen >= 0 && tlen <= c.P.max_tlen) {
            mr_update(c, mr, sm, lm, chr_row, tlen, (uint16_t)intron_num, on_cdna, (pair_type == 0) ? CM_CONCRD : CM_CONGEN, r1_sm);
        } else {
            if (tlen < 0) {
                tlen = (int32_t)(lm.spos - sm.epos - 1 + sm.matched_len + lm.matched_len);
                intron_num = 0;
            }
            mr_update(c, mr, sm, lm, chr_row, tlen, (uint16_t)intron_num, on_cdna, CM_DISCRD, r1_sm);
        }
    }
    return mr.type == CM_CONCRD;
}
// check_chimeric, utils.cpp:215-231
bool check_chimeric(const Ctx &c, const MatchedMate &sm, const MatchedMate &lm, cm_mapped_read &mr, int chr_row, bool r1_sm) {
    if (mr.type == CM_CONCRD) return false;
    if (sm.exons_spos < 0 || lm.exons_spos < 0) return false;
    const cm_annot_view *A = c.A;
    for (uint32_t i = 0; i < iv_nseg(A, sm.exons_spos); i++)
        for (uint32_t j = 0; j < iv_nseg(A, lm.exons_spos); j++)
            if (A->seg_gene_id[iv_segid(A, sm.exons_spos, i)] == A->seg_gene_id[iv_segid(A, lm.exons_spos, j)] && sm.spos < lm.spos) {
                mr_update(c, mr, sm, lm, chr_row, (int32_t)(lm.epos - sm.spos + 1), 0, false, CM_CHIORF, r1_sm);
                return true;
            }
    return false;
}
// shared tail of check_bsj / check_2bsj, utils.cpp:242-265 / :296-319
bool bsj_tail(const Ctx &c, MatchedMate &sm, MatchedMate &lm, cm_mapped_read &mr, int chr_row, bool r1_sm, int type) {
    const cm_annot_view *A = c.A;
    int32_t tl = (int32_t)(lm.epos - sm.spos + 1);
    if (sm.exons_spos < 0 || lm.exons_spos < 0) {
        if ((sm.exons_spos >= 0 && same_gene_mm(c, sm, lm)) || (lm.exons_spos >= 0 && same_gene_mm(c, lm, sm))) {
            mr_update(c, mr, sm, lm, chr_row, tl, 0, false, type, r1_sm);
            return true;
        }
        // ciRNA / lariat
        if (bit(A->intronic_bits, A->n_bits, sm.spos) && bit(A->intronic_bits, A->n_bits, lm.spos) &&
            sm.exon_ind_spos >= 0 && lm.exon_ind_epos >= 0 && sm.exon_ind_spos == lm.exon_ind_epos &&
            (uint32_t)(sm.spos - A->iv_epos[sm.exon_ind_spos]) <= LARIAT2BEGTH) {
            mr_update(c, mr, sm, lm, chr_row, tl, 0, false, type, r1_sm);
            return true;
        }
        return false;
    }
    for (uint32_t i = 0; i < iv_nseg(A, sm.exons_spos); i++)
        for (uint32_t j = 0; j < iv_nseg(A, lm.exons_spos); j++)
            if (A->seg_gene_id[iv_segid(A, sm.exons_spos, i)] == A->seg_gene_id[iv_segid(A, lm.exons_spos, j)]) {
                mr_update(c, mr, sm, lm, chr_row, tl, 0, false, type, r1_sm);
                return true;
            }
    return false;
}
bool check_bsj(const Ctx &c, MatchedMate &sm, MatchedMate &lm, cm_mapped_read &mr, int chr_row, bool r1_sm) {   // utils.cpp:235-266
    if (mr.type == CM_CONCRD || mr.type == CM_DISCRD) return false;
    if (!sm.right_ok || !lm.left_ok) return false;
    return bsj_tail(c, sm, lm, mr, chr_row, r1_sm, CM_CHIBSJ);
}
bool check_2bsj(const Ctx &c, MatchedMate &sm, MatchedMate &lm, cm_mapped_read &mr, int chr_row, bool r1_sm) {  // utils.cpp:270-320
    if (mr.type < CM_CHI2BSJ) return false;
    if (sm.spos > lm.spos) return false;
    if (sm.right_ok && lm.right_ok && sm.spos != lm.spos) return false;
    if (sm.left_ok && lm.left_ok && sm.epos != lm.epos) return false;
    if (sm.left_ok && lm.right_ok) return false;
    return bsj_tail(c, sm, lm, mr, chr_row, r1_sm, CM_CHI2BSJ);
}
// is_left_chain, utils.cpp:827-887
bool is_left_chain(const Chain &a, const Chain &b, int read_length) {
    uint32_t a_beg = a.frags[0].rpos, b_beg = b.frags[0].rpos;
    uint32_t a_end = a.frags[a.chain_len - 1].rpos + a.frags[a.chain_len - 1].len - 1;
    uint32_t b_end = b.frags[b.chain_len - 1].rpos + b.frags[b.chain_len - 1].len - 1;
    bool non_overlapping = (b_beg > a_end) || (a_beg > b_end);
    if (non_overlapping) return a_beg < b_beg;
    uint32_t i = 0, j = 0;
    int best_distance = INF_I, best_i = -1, best_j = -1;
    while (i < a.chain_len && j < b.chain_len) {
        uint32_t bj_beg = b.frags[j].rpos, ai_end = a.frags[i].rpos + a.frags[i].len - 1;
        if (ai_end < bj_beg) {
            int d = (int)(bj_beg - ai_end);
            if (d < best_distance) { best_distance = d; best_i = (int)i; best_j = (int)j; }
            ++i;
            continue;
        }
        uint32_t ai_beg = a.frags[i].rpos, bj_end = b.frags[j].rpos + b.frags[j].len - 1;
        if (bj_end < ai_beg) {
            int d = (int)(ai_beg - bj_end);
            if (d < best_distance) { best_distance = d; best_i = (int)i; best_j = (int)j; }
            ++j;
            continue;
        }
        best_i = (int)i;
        best_j = (int)j;
        break;
    }
    uint32_t common_bp = std::max(a.frags[best_i].rpos, b.frags[best_j].rpos);
    int32_t a_ov = a.frags[best_i].qpos + (int32_t)(common_bp - a.frags[best_i].rpos);
    int32_t b_ov = b.frags[best_j].qpos + (int32_t)(common_bp - b.frags[best_j].rpos);
    if (a_ov < read_length && b_ov < read_length) return a_ov >= b_ov;
    return a_beg < b_beg;
}

// ============================ A11-A13, A19: extension (extend.cpp) ============================
struct AllCoord {   // src/common.h:393-402, order src/common.cpp:456-464
    uint32_t rspos, rlen, qspos, qlen;
    bool operator<(const AllCoord &r) const {
        if (rspos != r.rspos) return rspos < r.rspos;
        if (qspos != r.qspos) return qspos < r.qspos;
        if (rlen != r.rlen) return rlen < r.rlen;
        return qlen < r.qlen;
    }
};
typedef std::map<AllCoord, AlignRes> Memo;

struct Ext {
    const Ctx &c;
    bool edit;      // false: DropAlignment (FilterRead's extensions), true: EditDistAlignment (ProcessCirc, process_circ.cpp:25)
    explicit Ext(const Ctx &cc, bool edit_alignment = false) : c(cc), edit(edit_alignment) {}
    int band() const { return c.P.band; }
    int local_sc(const uint8_t *s, int n, const uint8_t *t, int m, int &sc_len, int &indel, int &align_score, bool left) const {
        return edit ? local_alignment_sc_edit(c, s, n, t, m, sc_len, indel, align_score, left)
                    : local_alignment_sc(c, s, n, t, m, sc_len, indel, align_score, left);
    }

    // extend_right_middle / extend_left_middle, extend.cpp:435-461 / :653-679
    bool extend_middle(uint32_t pos, uint32_t exon_len, const uint8_t *qseq, uint32_t qseq_len, int ed_th, AlignRes &best,
                       AlignRes &curr, AlignRes &exon_res, bool right) {
        std::vector<uint8_t> ref;
        if (!pac2char(c, right ? pos + 1 : pos - exon_len, (int)exon_len, ref)) return false;
        int indel, sc;
        uint32_t seq_remain = std::min<uint32_t>(exon_len + band(), qseq_len);
        int edit_dist = local_alignment_side(c, qseq, (int)seq_remain, ref.data(), (int)exon_len, indel, sc, !right);
        uint32_t new_pos = right ? pos + exon_len : pos - exon_len;
        exon_res.set(new_pos, edit_dist, 0, -1 * indel, (int)exon_len - indel, sc);
        if (curr.ed + edit_dist <= ed_th) {
            curr.update(edit_dist, 0, new_pos, -1 * indel, (int)exon_len - indel, sc);
            update_side(c, best, curr, right);
            return true;
        }
        return false;
    }
    // extend_right_end / extend_left_end, extend.cpp:463-487 / :681-705
    void extend_end(uint32_t pos, uint32_t ref_len, const uint8_t *qseq, int qseq_len, int ed_th, AlignRes &best, AlignRes &curr,
                    AlignRes &exon_res, bool right) {
        std::vector<uint8_t> ref;
        if (!pac2char(c, right ? pos + 1 : pos - ref_len, (int)ref_len, ref)) return;
        int sclen, indel, sc;
        int edit_dist = local_sc(ref.data(), (int)ref_len, qseq, qseq_len, sclen, indel, sc, !right);
        uint32_t new_pos = right ? pos + qseq_len - indel : pos - qseq_len + indel;
        exon_res.set(new_pos, edit_dist, sclen, indel, qseq_len, sc);
        int actual_mapped_bp = qseq_len - sclen;
        if ((curr.ed + edit_dist <= ed_th) && (sclen <= c.P.max_sc) && (actual_mapped_bp >= sclen)) {
            curr.update(edit_dist, sclen, new_pos, indel, qseq_len, sc);
            if (right) update_by_score_right(best, curr);
            else update_by_score_left(best, curr);
        }
    }
    // shared "found in memo / compute" step for the middle pieces; returns false on the
    // reference's early `return`
    bool middle_step(Memo &memo, const AllCoord &key, uint32_t pos, uint32_t exon_len, const uint8_t *q, uint32_t qlen, int ed_th,
                     AlignRes &best, AlignRes &curr, AlignRes &exon_res, bool right, int &indel) {
        auto it = memo.find(key);
        if (it != memo.end()) {
            if (curr.ed + it->second.ed > ed_th) return false;
            curr.update(it->second.ed, it->second.sclen, it->second.pos, it->second.indel, it->second.qcovlen, it->second.score);
            update_side(c, best, curr, right);
            indel = it->second.indel;
            return true;
        }
        bool ok = extend_middle(pos, exon_len, q, qlen, ed_th, best, curr, exon_res, right);
        memo.insert(std::make_pair(key, exon_res));
        if (!ok) return false;
        indel = exon_res.indel;
        return true;
    }
    void end_step(Memo &memo, const AllCoord &key, uint32_t pos, uint32_t ref_len, const uint8_t *q, int qlen, int ed_th,
                  AlignRes &best, AlignRes &curr, AlignRes &exon_res, bool right) {
        auto it = memo.find(key);
        if (it != memo.end()) {
            int actual = it->second.qcovlen - it->second.sclen;
            if ((curr.ed + it->second.ed > ed_th) || (it->second.sclen > c.P.max_sc) || (actual < it->second.sclen)) return;
            curr.update(it->second.ed, it->second.sclen, it->second.pos, it->second.indel, it->second.qcovlen, it->second.score);
            if (right) update_by_score_right(best, curr);
            else update_by_score_left(best, curr);
        } else {
            extend_end(pos, ref_len, q, qlen, ed_th, best, curr, exon_res, right);
            memo.insert(std::make_pair(key, exon_res));
        }
    }

    // extend_right_trans, extend.cpp:490-650
    void extend_right_trans(uint32_t tid, uint32_t pos, int ref_len, const uint8_t *qseq, int qseq_len, int ed_th, uint32_t ub,
                            AlignRes &best, bool &consecutive, Memo &memo) {
        const cm_annot_view *A = c.A;
        consecutive = false;
        AlignRes curr(ub), exon_res(ub);
        int it_ind;
        int it_seg = get_location_overlap_ind(c, pos, it_ind);
        int covered = 0;
        if (it_seg < 0) return;
        int it_ind_start = A->trans_start_ind[tid];
        int rel_ind = it_ind - it_ind_start;
        uint32_t tsz = A->t2s_off[tid + 1] - A->t2s_off[tid];
        const uint8_t *t2s = A->t2s + A->t2s_off[tid];
        uint32_t rspos = pos;
        int exon_len = (int)(A->iv_epos[it_seg] - pos);
        int remain_ref_len = ref_len;
        int indel;
        for (unsigned int i = (unsigned int)(rel_ind + 1); i < tsz; i++) {
            if (exon_len >= qseq_len - covered) break;
            if (t2s[i] == 1) {
                indel = 0;
                if (exon_len > 0) {
                    if (rspos + exon_len > ub) return;
                    uint32_t rq = (uint32_t)std::min(exon_len + band(), qseq_len - covered);
                    AllCoord key{rspos, (uint32_t)exon_len, (uint32_t)covered, rq};
                    if (!middle_step(memo, key, rspos, (uint32_t)exon_len, qseq + covered, rq, ed_th, best, curr, exon_res, true, indel)) return;
                }
                remain_ref_len -= exon_len;
                covered += exon_len + indel;
                exon_len = 0;
                it_seg = iv_get_node(A, (int)i + it_ind_start);
                rspos = A->iv_spos[it_seg] - 1;
            }
            if (t2s[i] != 0) {
                it_seg = iv_get_node(A, (int)i + it_ind_start);
                exon_len += (int)(A->iv_epos[it_seg] - A->iv_spos[it_seg] + 1);
            }
        }
        if ((exon_len > 0) && (exon_len < qseq_len - covered) && (rspos + exon_len <= ub)) {
            uint32_t rq = (uint32_t)std::min(exon_len + band(), qseq_len - covered);
            AllCoord key{rspos, (uint32_t)exon_len, (uint32_t)covered, rq};
            middle_step(memo, key, rspos, (uint32_t)exon_len, qseq + covered, rq, ed_th, best, curr, exon_res, true, indel);
            return;
        }
        if (covered >= qseq_len || (rspos + qseq_len - covered > ub) || (exon_len < qseq_len - covered)) return;
        consecutive = (rspos == pos);
        remain_ref_len = std::min(remain_ref_len, exon_len);
        AllCoord key{rspos, (uint32_t)remain_ref_len, (uint32_t)covered, (uint32_t)(qseq_len - covered)};
        end_step(memo, key, rspos, (uint32_t)remain_ref_len, qseq + covered, qseq_len - covered, ed_th, best, curr, exon_res, true);
    }

    // extend_left_trans, extend.cpp:707-875
    void extend_left_trans(uint32_t tid, uint32_t pos, int ref_len, const uint8_t *qseq, int qseq_len, int ed_th, uint32_t lb,
                           AlignRes &best, bool &consecutive, Memo &memo) {
        const cm_annot_view *A = c.A;
        consecutive = false;
        AlignRes curr(lb), exon_res(lb);
        int it_ind;
        int covered = 0;
        int it_seg = get_location_overlap_ind(c, pos, it_ind);
        if (it_seg < 0) return;
        int it_ind_start = A->trans_start_ind[tid];
        int rel_ind = it_ind - it_ind_start;
        const uint8_t *t2s = A->t2s + A->t2s_off[tid];
        // NOTE: the reference indexes trans2seg[tid][i] for i = rel_ind..0 without a size check; a
        // rel_ind beyond the table is an out-of-bounds vector read there.  Clamp = treat as 0.
        uint32_t tsz = A->t2s_off[tid + 1] - A->t2s_off[tid];
        uint32_t lepos = pos;
        int exon_len = 0;
        int remain_ref_len = ref_len;
        int indel;
        bool first_seg = true;
        for (int i = rel_ind; i >= 0; i--) {
            uint8_t st = ((uint32_t)i < tsz) ? t2s[i] : 0;
            if (st != 0) {
                it_seg = iv_get_node(A, i + it_ind_start);
                if (first_seg) {
                    exon_len = (int)(pos - A->iv_spos[it_seg]);
                    first_seg = false;
                } else {
                    if (exon_len == 0) lepos = A->iv_epos[it_seg] + 1;
                    exon_len += (int)(A->iv_epos[it_seg] - A->iv_spos[it_seg] + 1);
                }
            }
            if (exon_len >= qseq_len - covered) break;
            if (st == 1) {
                indel = 0;
                if (exon_len > 0) {
                    if (lepos < lb + exon_len) return;
                    uint32_t rq = (uint32_t)std::min(exon_len + band(), qseq_len - covered);
                    AllCoord key{lepos, (uint32_t)exon_len, (uint32_t)covered, rq};
                    if (!middle_step(memo, key, lepos, (uint32_t)exon_len, qseq + qseq_len - covered - rq, rq, ed_th, best, curr, exon_res, false, indel)) return;
                }
                remain_ref_len -= exon_len;
                covered += exon_len + indel;
                exon_len = 0;
            }
        }
        if ((exon_len > 0) && (exon_len < qseq_len - covered) && (lepos >= lb + exon_len)) {
            uint32_t rq = (uint32_t)std::min(exon_len + band(), qseq_len - covered);
            AllCoord key{lepos, (uint32_t)exon_len, (uint32_t)covered, rq};
            middle_step(memo, key, lepos, (uint32_t)exon_len, qseq + qseq_len - covered - rq, rq, ed_th, best, curr, exon_res, false, indel);
            return;
        }
        if (covered >= qseq_len || (lepos < lb + qseq_len - covered) || (exon_len < qseq_len - covered)) return;
        consecutive = (lepos == pos);
        remain_ref_len = std::min(remain_ref_len, exon_len);
        AllCoord key{lepos, (uint32_t)remain_ref_len, (uint32_t)covered, (uint32_t)(qseq_len - covered)};
        end_step(memo, key, lepos, (uint32_t)remain_ref_len, qseq, qseq_len - covered, ed_th, best, curr, exon_res, false);
    }

    // extend_right / extend_left, extend.cpp:285-432
    bool extend_side(const std::vector<uint32_t> &common_tid, const uint8_t *seq, uint32_t &pos, int len, int ed_th, uint32_t bound,
                     AlignRes &best, bool right) {
        int seq_len = len, ref_len = len + band();
        uint32_t orig_pos = pos;
        int indel;
        bool consecutive = false;
        AlignRes curr(bound);
        best.set(pos, ed_th + 1, len + 1, band() + 1, 0, 0);
        Memo memo;
        for (uint32_t tid : common_tid) {
            if (right) extend_right_trans(tid, pos, ref_len, seq, seq_len, ed_th, bound, best, consecutive, memo);
            else extend_left_trans(tid, pos, ref_len, seq, seq_len, ed_th, bound, best, consecutive, memo);
        }
        uint32_t best_pos = best.pos;
        int min_ed = best.ed, sclen_best = best.sclen;
        if (min_ed <= ed_th) {
            pos = right ? best_pos - sclen_best : best_pos + sclen_best;
            if (best.qcovlen >= seq_len && sclen_best <= c.P.max_sc) return true;
        }
        std::vector<uint8_t> ref;
        if (!consecutive && pac2char(c, right ? orig_pos + 1 : orig_pos - ref_len, ref_len, ref)) {   // intron retention
            int sc;
            min_ed = local_sc(ref.data(), ref_len, seq, seq_len, sclen_best, indel, sc, !right);
            if (min_ed <= ed_th && sclen_best <= c.P.max_sc) {
                uint32_t np = right ? orig_pos + seq_len - indel : orig_pos - seq_len + indel;
                curr.set(np, min_ed, sclen_best, indel, seq_len, sc);
                bool upd = right ? update_by_score_right(best, curr) : update_by_score_left(best, curr);
                if (upd) {
                    pos = right ? np - sclen_best : np + sclen_best;
                    return true;
                }
            }
        }
        if (best.qcovlen <= 0) {
            pos = orig_pos;
            best.set(pos, 0, 0, 0, 0, -INF_I);
        }
        int qremain = seq_len - best.qcovlen;
        if (qremain + best.sclen <= c.P.max_sc) {
            best.set(pos, best.ed, best.sclen + qremain, best.indel, seq_len, best.score);
            return true;
        }
        return (best.qcovlen >= seq_len && best.ed <= ed_th);
    }

    // extend_chain_right / extend_chain_left, extend.cpp:215-280
    bool extend_chain_right(const std::vector<uint32_t> &tids, const Chain &ch, const uint8_t *seq, int seq_len, uint32_t ub, MatchedMate &mr, int &err) {
        const Frag &l = ch.frags[ch.chain_len - 1];
        uint32_t rm_pos = l.rpos + l.len - 1;
        int remain_end = seq_len - (int)(l.qpos + l.len);
        bool right_ok = (remain_end <= 0);
        AlignRes best(ub);
        if (remain_end > 0) right_ok = extend_side(tids, seq + seq_len - remain_end, rm_pos, remain_end, c.P.max_ed - err, ub, best, true);
        int sclen_right = best.sclen, err_right = best.ed;
        remain_end -= best.qcovlen;
        mr.epos = rm_pos;
        mr.matched_len -= right_ok ? sclen_right : remain_end;
        mr.qepos -= right_ok ? sclen_right : remain_end;
        mr.sclen_right = sclen_right;
        mr.right_ed = best.ed;
        err += err_right;
        return right_ok;
    }
    bool extend_chain_left(const std::vector<uint32_t> &tids, const Chain &ch, const uint8_t *seq, int32_t qspos, uint32_t lb, MatchedMate &mr, int &err) {
        uint32_t lm_pos = ch.frags[0].rpos;
        int remain_beg = ch.frags[0].qpos - qspos;
        bool left_ok = (remain_beg <= 0);
        AlignRes best(lb);
        if (remain_beg > 0) left_ok = extend_side(tids, seq, lm_pos, remain_beg, c.P.max_ed - err, lb, best, false);
        int sclen_left = best.sclen, err_left = best.ed;
        remain_beg -= best.qcovlen;
        mr.spos = lm_pos;
        mr.matched_len -= left_ok ? sclen_left : remain_beg;
        mr.qspos += left_ok ? sclen_left : remain_beg;
        mr.sclen_left = sclen_left;
        mr.left_ed = best.ed;
        err += err_left;
        return left_ok;
    }
    // calc_middle_ed, extend.cpp:878-920
    int calc_middle_ed(const Chain &ch, int edth, const uint8_t *qseq, int qseq_len) {
        std::vector<uint8_t> tmp;
        int mid_err = 0;
        if (ch.chain_len == 0) return 0;
        for (uint32_t i = 0; i + 1 < ch.chain_len; i++) {
            if (ch.frags[i + 1].qpos > (int32_t)(ch.frags[i].qpos + ch.frags[i].len)) {
                int diff = (int)(ch.frags[i + 1].rpos - ch.frags[i].rpos) - (ch.frags[i + 1].qpos - ch.frags[i].qpos);
                int32_t qspos = ch.frags[i].qpos + (int32_t)ch.frags[i].len;
                int qlen = ch.frags[i + 1].qpos - qspos;
                uint32_t rspos = ch.frags[i].rpos + ch.frags[i].len;
                int rlen = qlen + diff;
                if (rlen < 0) rlen = 0;
                if ((diff >= 0 && diff <= band()) || (diff < 0 && diff >= -band())) {
                    // The reference ignores a pac2char failure here and aligns against whatever its
                    // uninitialised stack buffer holds; this restatement defines that case as an all-NUL
                    // window (only reachable for chains that run off the contig end).
                    if (!pac2char(c, rspos, rlen, tmp)) tmp.assign((size_t)rlen + 1, 0);
                    if (diff >= 0) mid_err += global_one_side_banded_alignment(qseq + qspos, qlen, tmp.data(), rlen, diff);
                    else mid_err += global_one_side_banded_alignment(tmp.data(), rlen, qseq + qspos, qlen, -diff);
                }
                if (mid_err > edth) return edth + 1;
            }
        }
        return mid_err;
    }
    // extend_both_mates, extend.cpp:37-125
    bool extend_both_mates(const Chain &lch, const Chain &rch, const std::vector<uint32_t> &tids, const uint8_t *lseq, const uint8_t *rseq,
                           int lqspos, int rqspos, int lseq_len, int rseq_len, MatchedMate &lmm, MatchedMate &rmm) {
        const int maxEd = c.P.max_ed;
        lmm.middle_ed = calc_middle_ed(lch, maxEd, lseq, lseq_len);
        rmm.middle_ed = calc_middle_ed(rch, maxEd, rseq, rseq_len);
        if (lmm.middle_ed <= maxEd) is_concord_impl(lch, (uint32_t)lseq_len, lmm, true);
        if (rmm.middle_ed <= maxEd) is_concord_impl(rch, (uint32_t)rseq_len, rmm, true);
        if (lmm.middle_ed > maxEd || rmm.middle_ed > maxEd) return false;
        bool l_extend = true, r_extend = true;
        lmm.is_concord = false;
        if (lch.chain_len <= 0) { lmm.type = CM_ORPHAN; lmm.matched_len = 0; l_extend = false; }
        rmm.is_concord = false;
        if (rch.chain_len <= 0) { rmm.type = CM_ORPHAN; rmm.matched_len = 0; r_extend = false; }
        bool llok = false, lrok = false, rlok = false, rrok = false;
        int lerr = lmm.middle_ed, rerr = rmm.middle_ed;
        if (l_extend) {
            lmm.matched_len = (uint32_t)(lseq_len - lqspos + 1);
            lmm.qspos = (uint32_t)lqspos;
            lmm.qepos = (uint32_t)lseq_len;
            llok = extend_chain_left(tids, lch, lseq, lqspos - 1, MINLB, lmm, lerr);
        }
        if (r_extend) {
            rmm.matched_len = (uint32_t)(rseq_len - rqspos + 1);
            rmm.qspos = (uint32_t)rqspos;
            rmm.qepos = (uint32_t)rseq_len;
            rlok = extend_chain_left(tids, rch, rseq, rqspos - 1, l_extend ? lmm.spos : MINLB, rmm, rerr);
        }
        if (r_extend) rrok = extend_chain_right(tids, rch, rseq, rseq_len, MAXUB, rmm, rerr);
        if (l_extend) lrok = extend_chain_right(tids, lch, lseq, lseq_len, r_extend ? rmm.epos : MAXUB, lmm, lerr);
        if (l_extend) update_match_mate_info(c, llok, lrok, lerr, lmm);
        if (r_extend) update_match_mate_info(c, rlok, rrok, rerr, rmm);
        return true;
    }
    // extend_chain_both_sides, extend.cpp:131-213
    int extend_chain_both_sides(const Chain &ch, const uint8_t *seq, int seq_len, MatchedMate &mr, int dir) {
        const int maxEd = c.P.max_ed;
        mr.is_concord = false;
        if (ch.chain_len <= 0) { mr.type = CM_ORPHAN; return mr.type; }
        mr.middle_ed = estimate_middle_error(c, ch);
        if (is_concord_impl(ch, (uint32_t)seq_len, mr, false)) { mr.dir = dir; return mr.type; }
        uint32_t lm_pos = ch.frags[0].rpos;
        int remain_beg = ch.frags[0].qpos;
        bool left_ok = (remain_beg <= 0);
        AlignRes bl(MINLB);
        std::vector<uint32_t> empty;
        if (remain_beg > 0) left_ok = extend_side(empty, seq, lm_pos, remain_beg, maxEd - mr.middle_ed, MINLB, bl, false);
        int err_left = bl.ed, sclen_left = bl.sclen;
        remain_beg -= bl.qcovlen;
        const Frag &l = ch.frags[ch.chain_len - 1];
        uint32_t rm_pos = l.rpos + l.len - 1;
        int remain_end = seq_len - (int)(l.qpos + l.len);
        bool right_ok = (remain_end <= 0);
        AlignRes br(MAXUB);
        if (remain_end > 0) right_ok = extend_side(empty, seq + seq_len - remain_end, rm_pos, remain_end, maxEd - mr.middle_ed - err_left, MAXUB, br, true);
        int err_right = br.ed, sclen_right = br.sclen;
        remain_end -= br.qcovlen;
        mr.spos = lm_pos;
        mr.epos = rm_pos;
        mr.matched_len = (uint32_t)seq_len;
        mr.matched_len -= left_ok ? sclen_left : remain_beg;
        mr.matched_len -= right_ok ? sclen_right : remain_end;
        mr.qspos = (uint32_t)(1 + (left_ok ? sclen_left : remain_beg));
        mr.qepos = (uint32_t)(seq_len - (right_ok ? sclen_right : remain_end));
        mr.right_ed = br.ed;
        mr.left_ed = bl.ed;
        mr.dir = dir;
        if (left_ok && right_ok && (err_left + err_right <= maxEd) && sclen_left <= c.P.max_sc && sclen_right <= c.P.max_sc) {
            mr.is_concord = true;
            mr.type = CM_CONCRD;
        } else if (left_ok || right_ok) mr.type = CM_CANDID;
        else mr.type = CM_ORPHAN;
        return mr.type;
    }
};

// ============================ A8-A10: the per-pair controller (filter.cpp) ============================
struct Rec { const uint8_t *seq; std::vector<uint8_t> rc; int seq_len; };

struct MatePair { int type; int fi, rj; std::vector<uint32_t> common_tid; };

// FilterRead::pair_chains, filter.cpp:484-551
void pair_chains(const Ctx &c, const ChainList &fwd, const ChainList &rev, std::vector<MatePair> &mate_pairs, std::vector<char> &fp,
                 std::vector<char> &rp, int saved_type) {
    std::vector<int> fe_list(fwd.best_chain_count), re_list(rev.best_chain_count);
    for (int i = 0; i < fwd.best_chain_count; i++) fe_list[i] = get_location_overlap(c, fwd.chains[i].frags[0].rpos);
    for (int j = 0; j < rev.best_chain_count; j++) re_list[j] = get_location_overlap(c, rev.chains[j].frags[0].rpos);
    mate_pairs.clear();
    fp.assign(c.P.max_chain_len, 0);
    rp.assign(c.P.max_chain_len, 0);
    for (int i = 0; i < fwd.best_chain_count; i++)
        for (int j = 0; j < rev.best_chain_count; j++) {
            const Chain &F = fwd.chains[i], &R = rev.chains[j];
            uint32_t fs = F.frags[0].rpos, rs = R.frags[0].rpos;
            uint32_t fe = F.frags[F.chain_len - 1].rpos + F.frags[F.chain_len - 1].len;
            uint32_t re = R.frags[R.chain_len - 1].rpos + R.frags[R.chain_len - 1].len;
            int tlen = (int)((fs < rs) ? (re - fs) : (fe - rs));
            MatePair temp;
            bool same_tr = false, same_gen = false;
            if (fe_list[i] >= 0 && re_list[j] >= 0) same_tr = same_transcript(c, fe_list[i], re_list[j], temp.common_tid);
            if (!same_tr && fe_list[i] >= 0 &&
                ((c.P.scan_level == 0 && saved_type > CM_CONGEN) || (c.P.scan_level > 0 && saved_type >= CM_CONGEN)))
                same_gen = same_gene_iv(c, fe_list[i], rs, re);
            if (!same_gen && re_list[j] >= 0 && saved_type >= CM_CONGEN) same_gen = same_gene_iv(c, re_list[j], fs, fe);
            if (same_tr || same_gen || ((tlen <= MAXDISCRDTLEN) && (saved_type >= CM_CONGNM))) {
                temp.fi = i;
                temp.rj = j;
                temp.type = same_tr ? 0 : (same_gen ? 1 : 2);
                mate_pairs.push_back(temp);
                fp[i] = 1;
                rp[j] = 1;
            }
        }
}

// FilterRead::process_mates, filter.cpp:244-395
int process_mates(const Ctx &c, Ext &ext, const ChainList &fwd, const Rec &frec, const ChainList &bwd, const Rec &brec, cm_mapped_read &mr,
                  bool r1_forward) {
    std::vector<MatePair> mate_pairs;
    std::vector<char> fpaired, bpaired;
    pair_chains(c, fwd, bwd, mate_pairs, fpaired, bpaired, mr.type);
    int min_ret1 = CM_ORPHAN, min_ret2 = CM_ORPHAN;
    bool r1_genic = false, r2_genic = false;
    for (size_t i = 0; i < mate_pairs.size(); i++) {
        MatchedMate r1_mm(c), r2_mm(c);
        r1_mm.dir = 1;
        r2_mm.dir = -1;
        const Chain &F = fwd.chains[mate_pairs[i].fi], &R = bwd.chains[mate_pairs[i].rj];
        bool success;
        bool is_forward_left = is_left_chain(F, R, frec.seq_len);
        if (is_forward_left) {
            success = ext.extend_both_mates(F, R, mate_pairs[i].common_tid, frec.seq, brec.rc.data(), 1, 1, frec.seq_len, brec.seq_len, r1_mm, r2_mm);
            if (success) {
                int row = get_shift(c, r1_mm.spos);
                overlap_to_epos(c, r1_mm); overlap_to_spos(c, r1_mm);
                overlap_to_epos(c, r2_mm); overlap_to_spos(c, r2_mm);
                if (r1_mm.type == CM_CONCRD && r2_mm.type == CM_CONCRD) {
                    if (concordant_explanation(c, r1_mm, r2_mm, mr, row, r1_forward, mate_pairs[i].type) && c.P.scan_level == 0) return CM_CONCRD;
                } else if ((r1_mm.type == CM_CANDID && r2_mm.type == CM_CONCRD) || (r1_mm.type == CM_CONCRD && r2_mm.type == CM_CANDID)) {
                    check_bsj(c, r1_mm, r2_mm, mr, row, r1_forward);
                } else if (r1_mm.type == CM_CANDID && r2_mm.type == CM_CANDID) {
                    check_2bsj(c, r1_mm, r2_mm, mr, row, r1_forward);
                }
            }
        } else {
            success = ext.extend_both_mates(R, F, mate_pairs[i].common_tid, brec.rc.data(), frec.seq, 1, 1, brec.seq_len, frec.seq_len, r2_mm, r1_mm);
            if (success) {
                int row = get_shift(c, r2_mm.spos);
                overlap_to_epos(c, r1_mm); overlap_to_spos(c, r1_mm);
                overlap_to_epos(c, r2_mm); overlap_to_spos(c, r2_mm);
                if (r1_mm.type == CM_CONCRD && r2_mm.type == CM_CONCRD) {
                    check_chimeric(c, r2_mm, r1_mm, mr, row, !r1_forward);
                } else if ((r1_mm.type == CM_CANDID && r2_mm.type == CM_CONCRD) || (r1_mm.type == CM_CONCRD && r2_mm.type == CM_CANDID)) {
                    check_bsj(c, r2_mm, r1_mm, mr, row, !r1_forward);
                } else if (r1_mm.type == CM_CANDID && r2_mm.type == CM_CANDID) {
                    check_2bsj(c, r2_mm, r1_mm, mr, row, !r1_forward);
                }
            }
        }
        min_ret1 = std::min(r1_mm.type, min_ret1);
        min_ret2 = std::min(r2_mm.type, min_ret2);
        r1_genic = (r1_mm.exons_spos >= 0) || (r1_mm.exons_epos >= 0);
        r2_genic = (r2_mm.exons_spos >= 0) || (r2_mm.exons_epos >= 0);
    }
    if (mr.type == CM_CONCRD || mr.type == CM_DISCRD || mr.type == CM_CHIORF || mr.type == CM_CHIBSJ || mr.type == CM_CHI2BSJ) return mr.type;
    MatchedMate mm1(c);     // NOT reset between chains: the looked_up_* caches go stale exactly as in the reference
    if (min_ret1 != CM_CONCRD)
        for (int i = 0; i < fwd.best_chain_count; i++)
            if (!fpaired[i]) {
                int ex_ret = ext.extend_chain_both_sides(fwd.chains[i], frec.seq, frec.seq_len, mm1, 1);
                min_ret1 = std::min(ex_ret, min_ret1);
                overlap_to_spos(c, mm1); overlap_to_epos(c, mm1);
                r1_genic = (mm1.exons_spos >= 0) || (mm1.exons_epos >= 0);
            }
    MatchedMate mm2(c);
    if (min_ret2 != CM_CONCRD)
        for (int i = 0; i < bwd.best_chain_count; i++)
            if (!bpaired[i]) {
                int ex_ret = ext.extend_chain_both_sides(bwd.chains[i], brec.rc.data(), brec.seq_len, mm2, -1);
                min_ret2 = std::min(ex_ret, min_ret2);
                overlap_to_spos(c, mm2); overlap_to_epos(c, mm2);
                r2_genic = (mm2.exons_spos >= 0) || (mm2.exons_epos >= 0);
            }
    int new_type = (((min_ret1 == CM_ORPHAN) && (min_ret2 == CM_CONCRD)) || ((min_ret1 == CM_CONCRD) && (min_ret2 == CM_ORPHAN))) ? CM_OEANCH
                 : ((min_ret1 == CM_ORPHAN) || (min_ret2 == CM_ORPHAN)) ? CM_ORPHAN
                 : ((min_ret1 == CM_CONCRD) && (min_ret2 == CM_CONCRD) && (r1_genic && r2_genic)) ? CM_CHIFUS
                 : ((min_ret1 == CM_CONCRD) && (min_ret2 == CM_CONCRD)) ? CM_OEA2 : CM_CANDID;
    mr_update_type(mr, new_type);
    return mr.type;
}

void make_rec(Rec &r, const uint8_t *seq, int len) {   // FASTQParser::set_reverse_comp + set_comp, fastq_parser.cpp:141-162
    r.seq = seq;
    r.seq_len = len;
    r.rc.assign((size_t)len + 1, 0);
    for (int i = 0; i < len; ++i) {
        uint8_t ch = seq[len - 1 - i], o;
        switch (ch) {
            case 'A': case 'a': o = 'T'; break;
            case 'C': case 'c': o = 'G'; break;
            case 'G': case 'g': o = 'C'; break;
            case 'T': case 't': o = 'A'; break;
            case 'N': case 'n': o = 'N'; break;
            default: o = 0; break;     // comp[] is zero-initialised static storage for other bytes
        }
        r.rc[i] = o;
    }
}

struct Scratch {
    std::vector<MatchedKmer> fl, bl;
    ChainList fbc_r1, bbc_r1, fbc_r2, bbc_r2;
    int fhh_r1 = 0, bhh_r1 = 0, fhh_r2 = 0, bhh_r2 = 0;
    explicit Scratch(const Ctx &c) : fl(max_seg_cnt(c) + 2), bl(max_seg_cnt(c) + 2) {
        for (ChainList *l : {&fbc_r1, &bbc_r1, &fbc_r2, &bbc_r2}) l->chains.resize(CM_BESTCHAINLIM);
    }
};

// FilterRead::process_read (PE), filter.cpp:124-241
int process_read(const Ctx &c, Scratch &S, const Rec &r1, const Rec &r2, cm_mapped_read &mr) {
    Ext ext(c);
    get_best_chains(c, r1.seq, r1.seq_len, S.fbc_r1, S.fl.data(), S.fhh_r1);
    get_best_chains(c, r1.rc.data(), r1.seq_len, S.bbc_r1, S.bl.data(), S.bhh_r1);
    get_best_chains(c, r2.seq, r2.seq_len, S.fbc_r2, S.fl.data(), S.fhh_r2);
    get_best_chains(c, r2.rc.data(), r2.seq_len, S.bbc_r2, S.bl.data(), S.bhh_r2);
    int n1 = S.fbc_r1.best_chain_count + S.bbc_r1.best_chain_count, n2 = S.fbc_r2.best_chain_count + S.bbc_r2.best_chain_count;
    if (n1 + n2 <= 0) {
        if ((S.fhh_r1 + S.bhh_r1 > 0) && (S.fhh_r2 + S.bhh_r2 > 0)) { mr_update_type(mr, CM_NOPROC_MANYHIT); return CM_NOPROC_MANYHIT; }
        mr_update_type(mr, CM_NOPROC_NOMATCH);
        return CM_NOPROC_NOMATCH;
    }
    if (n1 <= 0 || n2 <= 0) { mr_update_type(mr, CM_OEANCH); return CM_OEANCH; }
    float fc1 = S.fbc_r1.best_chain_count > 0 ? S.fbc_r1.chains[0].score : 0;
    float bc1 = S.bbc_r1.best_chain_count > 0 ? S.bbc_r1.chains[0].score : 0;
    float fc2 = S.fbc_r2.best_chain_count > 0 ? S.fbc_r2.chains[0].score : 0;
    float bc2 = S.bbc_r2.best_chain_count > 0 ? S.bbc_r2.chains[0].score : 0;
    // float + float evaluated in float (x86-64 SSE; FLT_EVAL_METHOD 0)
    volatile float lhs = fc1 + bc2, rhs = fc2 + bc1;
    int a1, a2;
    if (lhs >= rhs) {
        a1 = process_mates(c, ext, S.fbc_r1, r1, S.bbc_r2, r2, mr, true);
        if (c.P.scan_level == 0 && a1 == CM_CONCRD) return CM_CONCRD;
        a2 = process_mates(c, ext, S.fbc_r2, r2, S.bbc_r1, r1, mr, false);
        if (c.P.scan_level == 0 && a2 == CM_CONCRD) return CM_CONCRD;
        return mr.type;
    }
    a1 = process_mates(c, ext, S.fbc_r2, r2, S.bbc_r1, r1, mr, false);
    if (c.P.scan_level == 0 && a1 == CM_CONCRD) return CM_CONCRD;
    a2 = process_mates(c, ext, S.fbc_r1, r1, S.bbc_r2, r2, mr, true);
    if (c.P.scan_level == 0 && a2 == CM_CONCRD) return CM_CONCRD;
    return mr.type;
}

bool mapped_type(int t) {   // filter.cpp:422-423, fastq_parser.cpp:216-219
    return t == CM_CONCRD || t == CM_DISCRD || t == CM_CHIORF || t == CM_CHIBSJ || t == CM_CHI2BSJ || t == CM_CONGNM || t == CM_CONGEN;
}

void copy_chain(const Chain &s, cm_chain &d) {
    memset(&d, 0, sizeof(d));
    d.score = s.score;
    d.chain_len = s.chain_len;
    for (uint32_t i = 0; i < s.chain_len && i < CM_MAX_CHAIN_FRAGS; ++i) { d.rpos[i] = s.frags[i].rpos; d.qpos[i] = s.frags[i].qpos; }
}


// ==================================================================================================
// Stage 2: ProcessCirc, src/process_circ.cpp -- back-splice-junction calling on the
// CHIBSJ / CHI2BSJ pairs stage 1 left in the last round's remain files.  Restated function by function.
// ==================================================================================================
const int S2_FR = 0, S2_RF = 1, S2_CR = 20, S2_NCR = 21, S2_MCR = 22, S2_UD = 30, S2_NF = 40;   // process_circ.h:14-20
const int S2_TOPCHAIN = 10;                 // process_circ.cpp:19
const int S2_BPRES = 5, S2_INDELTH = 3;     // common.h:42,45
const int S2_MAXHIT = 1000;                 // hash_table.cpp:6

// RegionalHashTable, src/hash_table.cpp:28-119: window_size-mers of one gene's sequence, location = offset in the gene
struct RegionalHT {
    int ws = 0, size = 0;
    uint32_t gene_spos = 0, gene_epos = 0;
    std::vector<std::vector<uint32_t>> loc;   // kept up to MAXHIT entries per bucket
    std::vector<uint32_t> cnt;                // frag_count (0 when it went above MAXHIT)
    static int nuc(uint8_t ch) {
        switch (ch) { case 'a': case 'A': return 0; case 'c': case 'C': return 1; case 'g': case 'G': return 2; case 't': case 'T': return 3; default: return -1; }
    }
    int hash_val(const uint8_t *seq) const {   // :99-109 (nuc_hval is indexed by uint8_t: bytes >= 128 read past the table there; -1 here)
        int val = 0;
        for (int i = 0; i < ws; ++i) {
            int b = nuc(seq[i]);
            if (b == -1) return -1;
            val = (val << 2) | b;
        }
        return val;
    }
    void create(int window, uint32_t gs, uint32_t ge, const uint8_t *seq, uint32_t start, int len) {   // init + create_table :28-78
        ws = window; size = 1 << (2 * ws); gene_spos = gs; gene_epos = ge;
        loc.assign(size, std::vector<uint32_t>());
        cnt.assign(size, 0);
        if (len < ws) return;
        uint32_t l = start;
        for (int i = 0; i <= len - ws; i++) {
            int hv = hash_val(seq + i);
            if (hv >= 0 && hv < size) {
                if (cnt[hv] < (uint32_t)S2_MAXHIT) loc[hv].push_back(l);
                ++cnt[hv];
            }
            ++l;
        }
        for (int hv = 0; hv < size; ++hv)
            if (cnt[hv] > (uint32_t)S2_MAXHIT) cnt[hv] = 0;
    }
};
struct RKmer { const uint32_t *frags; uint32_t frag_count; int32_t qpos; };      // GIMatchedKmer copy made by ProcessCirc::chaining

// chain_seeds_sorted_kbest2, src/chain.cpp:310-539.  Seed positions are offsets in the gene sequence; `shift` (the gene start)
// is added only to the emitted fragments -- the annotation queries (get_upper_bound, check_junction) and the `repeats`
// test are made with the UNSHIFTED offsets, as in the reference.
void chain_seeds_sorted_kbest2(const Ctx &c, int seq_len, RKmer *fl, ChainList &best_chain, int kmer, int kmer_cnt, uint32_t shift) {
    best_chain.best_chain_count = 0;
    if (kmer_cnt <= 0) return;
    const uint32_t max_best = (uint32_t)c.P.max_chain_len;
    while (kmer_cnt >= 1 && fl[kmer_cnt - 1].frag_count <= 0) kmer_cnt--;
    if (kmer_cnt <= 0) return;
    std::vector<std::vector<Cell>> dp(kmer_cnt);
    for (int ii = kmer_cnt - 1; ii >= 0; ii--) dp[ii].assign(fl[ii].frag_count, Cell{(double)kmer, -1, -1});
    std::map<double, CellList> score2chain;
    std::vector<uint32_t> lb_ind(kmer_cnt);
    uint32_t max_exon_end = 0;
    int ol_exons = -1;
    for (int ii = kmer_cnt - 2; ii >= 0; ii--) {
        RKmer *cur_mk = fl + ii;
        uint32_t read_remain = (uint32_t)(seq_len - cur_mk->qpos - kmer);
        for (int k = 0; k < kmer_cnt; k++) lb_ind[k] = 0;
        for (uint32_t i = 0; i < cur_mk->frag_count; i++) {
            const int32_t cur_info = (int32_t)cur_mk->frags[i];
            uint32_t seg_start = (uint32_t)cur_info, seg_end = (uint32_t)cur_info + kmer - 1;
            uint32_t max_lpos_lim = MAXUB;
            for (int jj = ii + 1; jj < kmer_cnt; jj++) {
                RKmer *pc = fl + jj;
                if (pc->frag_count <= 0 || lb_ind[jj] >= pc->frag_count) continue;
                if (cur_info + c.P.max_intron < (int32_t)pc->frags[lb_ind[jj]]) continue;
                while (lb_ind[jj] < pc->frag_count && (int32_t)pc->frags[lb_ind[jj]] <= cur_info) lb_ind[jj]++;
                if (lb_ind[jj] >= pc->frag_count) continue;
                if (max_lpos_lim == MAXUB) max_lpos_lim = get_upper_bound(c, seg_start, (uint32_t)kmer, read_remain, max_exon_end, ol_exons);
                int distr = pc->qpos - cur_mk->qpos - kmer, read_dist = distr;
                uint32_t j = lb_ind[jj];
                while (j < pc->frag_count && pc->frags[j] <= max_lpos_lim) {
                    uint32_t pinfo = pc->frags[j];
                    int genome_dist, distt, trans_dist;
                    if (max_exon_end == 0 || (pinfo + kmer - 1) <= max_exon_end) genome_dist = (int)(pinfo - seg_end - 1);
                    else genome_dist = INF_I;
                    if (std::abs(genome_dist - read_dist) <= c.P.max_ed) distt = genome_dist;
                    else if (check_junction(c, seg_start, pinfo, ol_exons, kmer, read_dist, trans_dist)) distt = trans_dist;
                    else { j++; continue; }
                    int maxd = distr < distt ? distt : distr, mind = distr < distt ? distr : distt;
                    double temp_score = dp[jj][j].score + 2e4 * kmer - 0.1 * (maxd - mind);     // score_alpha - score_beta, chain.cpp:13-22
                    if (temp_score > dp[ii][i].score) {
                        dp[ii][i] = Cell{temp_score, jj, (int)j};
                        auto it = score2chain.find(temp_score);
                        if (it == score2chain.end()) { CellList e; e.count = 0; it = score2chain.insert(std::make_pair(temp_score, e)).first; }
                        if (it->second.count < max_best) it->second.chain_list[it->second.count++] = Cell{temp_score, ii, (int)i};
                    }
                    j++;
                }
            }
        }
    }
    uint32_t best_count = 0;
    double best_score = score2chain.empty() ? (double)kmer : score2chain.rbegin()->first;
    std::set<uint32_t> repeats;
    for (auto it = score2chain.rbegin(); it != score2chain.rend(); ++it)
        for (uint32_t l = 0; l < it->second.count; l++) {
            if (best_count >= max_best) break;
            Cell bi = it->second.chain_list[l];
            uint32_t spos = fl[bi.prev_list].frags[bi.prev_ind];
            if (bi.score < best_score && repeats.find(spos) != repeats.end()) continue;
            uint32_t i = 0, j = best_count++;
            Chain &ch = best_chain.chains[j];
            ch.frags.clear();
            while (bi.prev_list != -1) {
                Frag f{shift + fl[bi.prev_list].frags[bi.prev_ind], fl[bi.prev_list].qpos, (uint32_t)kmer};
                ch.frags.push_back(f);
                if (i != 0) repeats.insert(f.rpos);
                int tl = bi.prev_list;
                bi.prev_list = dp[tl][bi.prev_ind].prev_list;
                bi.prev_ind = dp[tl][bi.prev_ind].prev_ind;
                i++;
            }
            ch.score = (float)bi.score;
            ch.chain_len = i;
        }
    if (best_count == 0)
        for (int ii = kmer_cnt - 1; ii >= 0; ii--)
            for (uint32_t i = 0; i < fl[ii].frag_count; i++) {
                if (best_count >= max_best) break;
                Chain &ch = best_chain.chains[best_count++];
                ch.frags.assign(1, Frag{shift + fl[ii].frags[i], fl[ii].qpos, (uint32_t)kmer});
                ch.score = (float)dp[ii][i].score;
                ch.chain_len = 1;
            }
    best_chain.best_chain_count = (int)best_count;
}

struct Junc { uint32_t beg, end, bp_matched; };
struct CircRes2 {      // CircRes, src/common.h:406-423
    int chr_id = -1;
    uint32_t spos = 0, epos = 0;
    int type = S2_NF;
    std::string start_signal, end_signal, start_bp_ref, end_bp_ref;
    void set_bp(uint32_t sp, uint32_t ep, const std::string &ss, const std::string &es, const std::string &sb, const std::string &eb) {
        spos = sp; epos = ep; start_signal = ss; end_signal = es; start_bp_ref = sb; end_bp_ref = eb;
    }
};
std::string consensus2(const std::string &a, const std::string &b) {     // get_consensus(s1, s2), utils.cpp:746-756
    std::string r;
    if (a.length() != b.length()) return r;
    for (size_t i = 0; i < a.length(); ++i) r += (a[i] == b[i]) ? a[i] : 'N';
    return r;
}

struct CircCaller {
    Ctx c;                    // current contig
    Ext ext;
    int ws;                   // window_size of the regional tables (ProcessCirc ctor argument)
    const int step = 3;       // process_circ.cpp:59
    // member state of ProcessCirc that lives across the calls made for one read (process_circ.h:31-41)
    const uint8_t *fullmap_seq = nullptr, *remain_seq = nullptr, *r1_seq = nullptr, *r2_seq = nullptr;
    uint32_t fullmap_seq_len = 0, remain_seq_len = 0, r1_seq_len = 0, r2_seq_len = 0;
    ChainList bc1, bc2;
    // outputs
    std::string candid;       // <out>.candidates.pam
    struct Call { int chr_id; uint32_t spos, epos; int type; uint64_t rec; std::string ss, es, sb, eb; };
    std::vector<Call> calls;  // circ_res
    const char *const *chr_names;
    uint64_t cur_rec = 0;
    const char *cur_name = "";

    CircCaller(const Ctx &cc, int window, const char *const *names) : c(cc), ext(c, true), ws(window), chr_names(names) {
        int max_kmer_cnt = (c.P.max_read_len - ws) / step + 1;
        bc1.chains.resize(c.P.max_chain_len); bc2.chains.resize(c.P.max_chain_len);
        for (auto &ch : bc1.chains) ch.frags.reserve(max_kmer_cnt);
        for (auto &ch : bc2.chains) ch.frags.reserve(max_kmer_cnt);
    }

    // a character of a read the reference indexes without a bounds check (process_circ.cpp:1295-1296 and friends): out of
    // range is defined as NUL here (index seq_len IS the terminator in the reference)
    static char at(const uint8_t *s, uint32_t len, int64_t i) { return (s && i >= 0 && i < (int64_t)len) ? (char)s[i] : '\0'; }
    static std::string two(const uint8_t *s, uint32_t len, int64_t i) { std::string r; r += at(s, len, i); r += at(s, len, i + 1); return r; }
    std::string ref_bp(uint32_t start, int len) const {   // pac2char_otf(start, len, buf) + string(buf); failure leaves buf unset there, "" here
        std::vector<uint8_t> b;
        if (!pac2char_otf(start, len, b)) return std::string();
        return std::string((const char *)b.data());
    }
    // GenomeSeeder::pac2char_otf, src/match_read.cpp:301-332: like pac2char but decodes the packed words, so position 0 has no meaning
    // (start - 1 underflows): defined as failure here
    bool pac2char_otf(uint32_t start, int len, std::vector<uint8_t> &out) const {
        int ref_len = (int)c.X->ref_len;
        if ((int)start < 0 || (int)start + len - 1 > ref_len || start == 0) return false;
        out.assign((size_t)std::max(len, 0) + 1, 0);
        for (int i = 0; i < len; ++i) out[i] = c.X->genome[start - 1 + i];
        return true;
    }
    int gene_overlap(uint32_t pos) const {     // GTFParser::get_gene_overlap(pos, false), gene_annotation.cpp:572-585 -> interval index or -1
        const cm_annot_view *A = c.A;
        if (A->n_giv == 0 || pos < A->giv_spos[0]) return -1;
        int beg = 0, end = (int)A->n_giv;
        while (end - beg > 1) { int mid = (beg + end) / 2; if (pos < A->giv_spos[mid]) end = mid; else beg = mid; }
        int ind = end - 1;
        if (ind < 0 || A->giv_epos[ind] < pos) return -1;
        if (A->giv_gene_off[ind + 1] == A->giv_gene_off[ind]) return -1;
        return ind;
    }

    // set_mm, process_circ.cpp:1710-1752
    void set_mm(const Chain &ch, uint32_t qspos, int rlen, int dir, MatchedMate &mm) const {
        uint32_t spos = ch.frags[0].rpos, epos = ch.frags[ch.chain_len - 1].rpos + ch.frags[ch.chain_len - 1].len - 1;
        uint32_t qepos = qspos + rlen - 1;
        mm.spos = spos; mm.epos = epos; mm.qspos = qspos; mm.qepos = qepos;            // MatchedMate::set, common.cpp:154-161
        mm.matched_len = (qepos + 1 >= qspos) ? (qepos - qspos + 1) : 0;
        mm.dir = dir;
    }
    // MatchedMate(const MatchedRead&, r1_2, rlen, partial), common.cpp:196-243
    MatchedMate mate_of(const cm_mapped_read &mr, int r1_2, int rlen, bool partial) const {
        MatchedMate m(c);
        m.type = mr.type; m.right_ed = 0; m.left_ed = 0;
        if (r1_2 == 1) { m.spos = mr.spos_r1; m.epos = mr.epos_r1; m.qspos = mr.qspos_r1; m.qepos = mr.qepos_r1; m.middle_ed = mr.ed_r1; m.matched_len = mr.mlen_r1; m.dir = mr.r1_forward ? 1 : -1; }
        else { m.spos = mr.spos_r2; m.epos = mr.epos_r2; m.qspos = mr.qspos_r2; m.qepos = mr.qepos_r2; m.middle_ed = mr.ed_r2; m.matched_len = mr.mlen_r2; m.dir = mr.r2_forward ? 1 : -1; }
        if (partial) {
            if ((m.qspos - 1) > (uint32_t)(rlen - (int)m.qepos)) { m.sclen_left = 0; m.sclen_right = rlen - (int)m.qepos; }
            else { m.sclen_left = (int)m.qspos - 1; m.sclen_right = 0; }
        } else { m.sclen_left = (int)m.qspos - 1; m.sclen_right = rlen - (int)m.qepos; }
        // the ctor leaves junc_num / is_concord / left_ok / right_ok unset in the reference; nothing downstream reads them
        return m;
    }
    // MatchedMate::merge_to_right, common.cpp:163-193
    bool merge_to_right(MatchedMate &l, const MatchedMate &r) const {
        if (l.dir != r.dir) return false;
        l.epos = r.epos; l.qepos = r.qepos;
        l.middle_ed += l.right_ed + r.left_ed;
        l.right_ed = r.right_ed;
        l.matched_len += r.matched_len + l.sclen_right + r.sclen_left;
        l.middle_ed += l.sclen_right + r.sclen_left;
        l.sclen_right = r.sclen_right;
        l.right_ok = r.right_ok;
        l.looked_up_epos = r.looked_up_epos;
        l.exon_ind_epos = r.exon_ind_epos;
        return !(l.left_ed + l.middle_ed + l.right_ed > c.P.max_ed);
    }
    void tids_of(int iv, std::vector<uint32_t> &out) const {
        const cm_annot_view *A = c.A;
        for (uint32_t i = 0; i < iv_nseg(A, iv); i++) { uint32_t g = iv_segid(A, iv, i); for (uint32_t k = A->seg_tid_off[g]; k < A->seg_tid_off[g + 1]; k++) out.push_back(A->seg_tid[k]); }
    }
    static void intersect_trans(const std::vector<uint32_t> &a, const std::vector<uint32_t> &b, std::vector<uint32_t> &out) {   // utils.cpp:322-333
        for (uint32_t x : a) for (uint32_t y : b) if (x == y) { out.push_back(x); break; }
    }
    // the 3- and 4-interval forms, utils.cpp:356-398 (the 3-interval form intersects (s ^ r) with s again, not with q)
    bool same_tr3(int s, int r, int q, std::vector<uint32_t> &common) const {
        common.clear();
        if (s < 0 || r < 0 || q < 0) return false;
        std::vector<uint32_t> sr;
        if (!same_transcript(c, s, r, sr)) return false;
        std::vector<uint32_t> st;
        tids_of(s, st);
        intersect_trans(sr, st, common);
        return !common.empty();
    }
    bool same_tr4(int s, int r, int q, int p, std::vector<uint32_t> &common) const {
        common.clear();
        if (s < 0 || r < 0 || q < 0 || p < 0) return false;
        std::vector<uint32_t> sr, qp;
        if (!same_transcript(c, s, r, sr)) return false;
        if (!same_transcript(c, q, p, qp)) return false;
        intersect_trans(sr, qp, common);
        return !common.empty();
    }
    // same_transcript(vector<MatchedMate>&, size, common_tid), utils.cpp:419-599: tries start/end combinations in a fixed order
    bool same_transcript_n(std::vector<MatchedMate> &g, int size, std::vector<uint32_t> &common) const {
        auto S = [&](int i) { return g[i].exons_spos; };
        auto E = [&](int i) { return g[i].exons_epos; };
        if (size == 2) {
            overlap_to_spos(c, g[0]); overlap_to_spos(c, g[1]);
            if (same_transcript(c, S(0), S(1), common)) return true;
            overlap_to_epos(c, g[1]);
            if (same_transcript(c, S(0), E(1), common)) return true;
            overlap_to_epos(c, g[0]);
            if (same_transcript(c, E(0), S(1), common)) return true;
            if (same_transcript(c, E(0), E(1), common)) return true;
        }
        if (size == 3) {
            overlap_to_spos(c, g[0]); overlap_to_spos(c, g[1]); overlap_to_spos(c, g[2]);
            if (same_tr3(S(0), S(1), S(2), common)) return true;
            overlap_to_epos(c, g[2]);
            if (same_tr3(S(0), S(1), E(2), common)) return true;
            overlap_to_epos(c, g[1]);
            if (same_tr3(S(0), E(1), S(2), common)) return true;
            if (same_tr3(S(0), E(1), E(2), common)) return true;
            overlap_to_epos(c, g[0]);
            if (same_tr3(E(0), S(1), S(2), common)) return true;
            if (same_tr3(E(0), S(1), E(2), common)) return true;
            if (same_tr3(E(0), E(1), S(2), common)) return true;
            if (same_tr3(E(0), E(1), E(2), common)) return true;
        }
        if (size == 4) {
            overlap_to_spos(c, g[0]); overlap_to_spos(c, g[1]); overlap_to_spos(c, g[2]); overlap_to_spos(c, g[3]);
            if (same_tr4(S(0), S(1), S(2), S(3), common)) return true;
            overlap_to_epos(c, g[2]);
            if (same_tr4(S(0), S(1), E(2), S(3), common)) return true;
            overlap_to_epos(c, g[1]);
            if (same_tr4(S(0), E(1), S(2), S(3), common)) return true;
            if (same_tr4(S(0), E(1), E(2), S(3), common)) return true;
            overlap_to_epos(c, g[0]);
            if (same_tr4(E(0), S(1), S(2), S(3), common)) return true;
            if (same_tr4(E(0), S(1), E(2), S(3), common)) return true;
            if (same_tr4(E(0), E(1), S(2), S(3), common)) return true;
            if (same_tr4(E(0), E(1), E(2), S(3), common)) return true;
            overlap_to_epos(c, g[3]);
            if (same_tr4(S(0), S(1), S(2), E(3), common)) return true;
            if (same_tr4(S(0), S(1), E(2), E(3), common)) return true;
            if (same_tr4(S(0), E(1), S(2), E(3), common)) return true;
            if (same_tr4(S(0), E(1), E(2), E(3), common)) return true;
            if (same_tr4(E(0), S(1), S(2), E(3), common)) return true;
            if (same_tr4(E(0), S(1), E(2), E(3), common)) return true;
            if (same_tr4(E(0), E(1), S(2), E(3), common)) return true;
            if (same_tr4(E(0), E(1), E(2), E(3), common)) return true;
        }
        return false;
    }
    // get_junctions, utils.cpp:686-744
    void get_junctions(MatchedMate &mm, std::vector<Junc> &ji) const {
        const cm_annot_view *A = c.A;
        overlap_to_spos(c, mm);
        overlap_to_epos(c, mm);
        ji.clear();
        if (mm.exons_spos < 0 || mm.exons_epos < 0) return;
        auto push = [&](uint32_t b, uint32_t e, uint32_t m) { if (b >= e) return; ji.push_back(Junc{b, e, m}); };
        for (uint32_t i = 0; i < iv_nseg(A, mm.exons_spos); ++i) {
            uint32_t sg = iv_segid(A, mm.exons_spos, i);
            for (uint32_t k = A->seg_tid_off[sg]; k < A->seg_tid_off[sg + 1]; ++k) {
                uint32_t tid = A->seg_tid[k];
                int start_ind = A->trans_start_ind[tid];
                uint32_t start_table_ind = (uint32_t)(mm.exon_ind_spos - start_ind);
                uint32_t end_table_ind = (uint32_t)(mm.exon_ind_epos - start_ind);
                uint32_t tsz = A->t2s_off[tid + 1] - A->t2s_off[tid];
                const uint8_t *t2s = A->t2s + A->t2s_off[tid];
                if (mm.exon_ind_epos < start_ind || end_table_ind >= tsz || t2s[end_table_ind] == 0) continue;
                if (start_table_ind == end_table_ind) return;
                uint32_t junc_start = A->iv_epos[mm.exons_spos];
                uint32_t covered = A->iv_epos[mm.exons_spos] - mm.spos + 1;
                int this_it_ind = mm.exon_ind_spos;
                for (uint32_t q = start_table_ind + 1; q < end_table_ind; q++) {
                    this_it_ind++;
                    if (q < tsz && t2s[q] != 0) {       // q >= tsz (start index beyond the table) is an out-of-bounds vector read there
                        push(junc_start, A->iv_spos[this_it_ind], covered);
                        covered += A->iv_epos[this_it_ind] - A->iv_spos[this_it_ind] + 1;
                        junc_start = A->iv_epos[this_it_ind];
                    }
                }
                push(junc_start, A->iv_spos[mm.exons_epos], covered);
                covered += mm.epos - A->iv_spos[mm.exons_epos] + 1;
                if (std::abs((int32_t)(covered - mm.matched_len)) <= S2_INDELTH) return;
                ji.clear();
            }
        }
    }

    // ProcessCirc::chaining, process_circ.cpp:677-737
    void chaining(uint32_t qspos, uint32_t qepos, const RegionalHT &ht, const uint8_t *seq, uint32_t shift, ChainList &bc) {
        int seq_len = (int)qepos - (int)qspos + 1;
        if (seq_len < ws) { bc.best_chain_count = 0; return; }
        int kmer_cnt = (seq_len - ws) / step + 1;
        std::vector<RKmer> fl(kmer_cnt + 1);
        int l = 0;
        for (uint32_t i = qspos - 1; i <= qepos - ws; i += step) {
            int hv = ht.hash_val(seq + i);
            if (hv < 0 || hv >= ht.size) continue;       // find_hash == NULL: an N inside the k-mer
            fl[l] = RKmer{ht.loc[hv].data(), ht.cnt[hv], (int32_t)i};
            if (fl[l].frag_count > (uint32_t)c.P.seed_lim) fl[l].frag_count = 0;
            l++;
        }
        kmer_cnt = l;
        chain_seeds_sorted_kbest2(c, (int)qepos, fl.data(), bc, ws, kmer_cnt, shift);
        int least_miss = INF_I;
        for (int j = 0; j < bc.best_chain_count; j++) {
            int missing = kmer_cnt - (int)bc.chains[j].chain_len;
            if (missing > least_miss) { bc.best_chain_count = j; break; }
            least_miss = missing;
        }
    }

    // ProcessCirc::find_exact_coord, process_circ.cpp:739-789
    bool find_exact_coord(MatchedMate &mm_r1, MatchedMate &mm_r2, MatchedMate &partial_mm, int dir, uint32_t qspos, const uint8_t *rseq,
                          int rlen, int whole_len, const Chain &bc) {
        set_mm(bc, qspos, rlen, dir, partial_mm);
        --qspos;
        overlap_to_spos(c, mm_r1); overlap_to_spos(c, mm_r2); overlap_to_spos(c, partial_mm);
        std::vector<uint32_t> common_tid;
        std::vector<MatchedMate> segments{mm_r1, mm_r2, partial_mm};
        if (!same_transcript_n(segments, 3, common_tid)) return false;
        partial_mm.middle_ed = ext.calc_middle_ed(bc, c.P.max_ed, rseq, rlen);
        if (partial_mm.middle_ed > c.P.max_ed) return false;
        partial_mm.is_concord = false;
        if (bc.chain_len <= 0) { partial_mm.type = CM_ORPHAN; partial_mm.matched_len = 0; return false; }
        int err = partial_mm.middle_ed;
        partial_mm.matched_len = (uint32_t)rlen;
        bool lok = ext.extend_chain_left(common_tid, bc, rseq + qspos, (int32_t)qspos, MINLB, partial_mm, err);
        bool rok = ext.extend_chain_right(common_tid, bc, rseq, qspos == 0 ? rlen : whole_len, MAXUB, partial_mm, err);
        update_match_mate_info(c, lok, rok, err, partial_mm);
        return partial_mm.type == CM_CONCRD;
    }

    // the (transcript, offset) lists of final_check / check_split_map: exon ends within BPRES of the left piece's end, exon starts
    // within BPRES of the right piece's start, walking the intervals the piece covers (process_circ.cpp:973-1009, 1193-1243).
    // get_interval(ind) == NULL (ind off either end of the table) is dereferenced by the reference; the walk stops there here.
    typedef std::pair<uint32_t, int> pu32i;
    void end_tids_of(const MatchedMate &m, std::vector<pu32i> &out) const {
        const cm_annot_view *A = c.A;
        int ind = m.exon_ind_epos;
        while (ind >= 0 && ind < (int)A->n_iv && m.spos < A->iv_epos[ind]) {
            for (uint32_t i = 0; i < iv_nseg(A, ind); ++i) {
                uint32_t sg = iv_segid(A, ind, i);
                int diff = (int)(m.epos + m.sclen_right - A->seg_end[sg]);
                if (std::abs(diff) <= S2_BPRES) for (uint32_t k = A->seg_tid_off[sg]; k < A->seg_tid_off[sg + 1]; ++k) out.push_back(pu32i(A->seg_tid[k], diff));
            }
            --ind;
        }
    }
    void start_tids_of(const MatchedMate &m, std::vector<pu32i> &out) const {
        const cm_annot_view *A = c.A;
        int ind = m.exon_ind_spos;
        while (ind >= 0 && ind < (int)A->n_iv && m.epos > A->iv_spos[ind]) {
            for (uint32_t i = 0; i < iv_nseg(A, ind); ++i) {
                uint32_t sg = iv_segid(A, ind, i);
                int diff = (int)(m.spos - m.sclen_left - A->seg_start[sg]);
                if (std::abs(diff) <= S2_BPRES) for (uint32_t k = A->seg_tid_off[sg]; k < A->seg_tid_off[sg + 1]; ++k) out.push_back(pu32i(A->seg_tid[k], diff));
            }
            ++ind;
        }
    }

    // ProcessCirc::split_realignment (6-argument form), process_circ.cpp:1343-1392
    int split_realignment6(uint32_t qcutpos, uint32_t beg_bp, uint32_t end_bp, const uint8_t *seq, uint32_t seq_len, const std::vector<uint32_t> &common_tid) {
        const int maxEd = c.P.max_ed;
        if (qcutpos <= 0 || qcutpos >= seq_len) return maxEd + 1;
        std::vector<uint8_t> bp;
        int last_bp_err = (pac2char_otf(end_bp, 1, bp) && seq[qcutpos - 1] == bp[0]) ? 0 : 1;
        int first_bp_err = (pac2char_otf(beg_bp, 1, bp) && seq[qcutpos] == bp[0]) ? 0 : 1;
        uint32_t lm_pos = end_bp, rm_pos = beg_bp, lb = beg_bp, ub = end_bp;
        AlignRes bl(lb), br(ub);
        bool lok = ext.extend_side(common_tid, seq, lm_pos, (int)qcutpos - 1, maxEd - last_bp_err, lb, bl, false);
        bool rok = ext.extend_side(common_tid, seq + qcutpos + 1, rm_pos, (int)(seq_len - qcutpos - 1), maxEd - first_bp_err, ub, br, true);
        bl.ed += last_bp_err;
        br.ed += first_bp_err;
        if (lok && rok && (bl.ed + br.ed) <= maxEd) return bl.ed + br.ed;
        return maxEd + 1;
    }
    // ProcessCirc::split_realignment (5-argument form), process_circ.cpp:1394-1486
    int split_realignment5(uint32_t qcutpos, MatchedMate &full_mm, MatchedMate &split_mm_left, MatchedMate &split_mm_right, CircRes2 &cr) {
        const int maxEd = c.P.max_ed;
        if (qcutpos <= 0 || qcutpos >= fullmap_seq_len) return S2_UD;
        qcutpos += full_mm.qspos - 1;
        if (qcutpos <= 0 || qcutpos >= fullmap_seq_len) return S2_UD;
        overlap_to_spos(c, split_mm_left); overlap_to_epos(c, split_mm_left);
        overlap_to_spos(c, split_mm_right); overlap_to_epos(c, split_mm_right);
        std::vector<uint32_t> common_tid;
        std::vector<MatchedMate> segments{split_mm_left, split_mm_right};
        if (!same_transcript_n(segments, 2, common_tid)) return S2_UD;
        std::vector<uint8_t> bp;
        int last_bp_err = (pac2char_otf(split_mm_left.epos, 1, bp) && fullmap_seq[qcutpos - 1] == bp[0]) ? 0 : 1;
        int first_bp_err = (pac2char_otf(split_mm_right.spos, 1, bp) && fullmap_seq[qcutpos] == bp[0]) ? 0 : 1;
        uint32_t lm_pos = split_mm_left.epos, rm_pos = split_mm_right.spos, lb = split_mm_right.spos, ub = split_mm_left.epos;
        AlignRes bl(lb), br(ub);
        bool lok = ext.extend_side(common_tid, fullmap_seq, lm_pos, (int)qcutpos - 1, maxEd - last_bp_err, lb, bl, false);
        bool rok = ext.extend_side(common_tid, fullmap_seq + qcutpos + 1, rm_pos, (int)(fullmap_seq_len - qcutpos - 1), maxEd - first_bp_err, ub, br, true);
        bl.ed += last_bp_err;
        br.ed += first_bp_err;
        if (!lok || !rok || (bl.ed + br.ed) > maxEd) return S2_UD;
        MatchedMate nl(c), nr(c);       // default-constructed there: type ORPHAN, *_ed = maxEd + 1 until overwritten, lookups unset
        nl.spos = lm_pos; nl.epos = split_mm_left.epos; nl.qspos = (uint32_t)bl.sclen; nl.qepos = qcutpos; nl.dir = full_mm.dir;
        nl.matched_len = qcutpos - (uint32_t)bl.sclen; nl.sclen_left = bl.sclen; nl.sclen_right = 0; nl.left_ed = bl.ed; nl.right_ed = 0;
        nl.middle_ed = 0; nl.left_ok = true; nl.right_ok = true;
        nr.spos = split_mm_right.spos; nr.epos = rm_pos; nr.qspos = qcutpos + 1; nr.qepos = fullmap_seq_len - (uint32_t)br.sclen; nr.dir = full_mm.dir;
        nr.matched_len = fullmap_seq_len - qcutpos - (uint32_t)br.sclen; nr.sclen_left = 0; nr.sclen_right = br.sclen; nr.left_ed = 0;
        nr.right_ed = br.ed; nr.middle_ed = 0; nr.left_ok = true; nr.right_ok = true;
        r1_seq = remain_seq; r2_seq = fullmap_seq; r1_seq_len = remain_seq_len; r2_seq_len = fullmap_seq_len;
        return check_split_map4(split_mm_right, nr, split_mm_left, nl, cr);
    }
    // ProcessCirc::rescue_overlapping_bsj, process_circ.cpp:1488-1552
    int rescue_overlapping_bsj(MatchedMate &full_mm, MatchedMate &split_mm_left, MatchedMate &split_mm_right, CircRes2 &cr) {
        std::vector<Junc> ji;
        if (split_mm_right.spos <= full_mm.epos && split_mm_right.spos > full_mm.spos) {
            get_junctions(full_mm, ji);
            uint32_t qcutpos = 0;
            for (auto &j : ji) if (j.end == split_mm_right.spos) qcutpos = j.bp_matched;
            if (qcutpos == 0) qcutpos = split_mm_right.spos - full_mm.spos;
            if (split_realignment5(qcutpos, full_mm, split_mm_left, split_mm_right, cr) == S2_CR) return S2_CR;
        }
        if (split_mm_left.epos >= full_mm.spos && split_mm_left.epos < full_mm.epos) {
            get_junctions(full_mm, ji);
            uint32_t qcutpos = 0;
            for (auto &j : ji) if (j.beg == split_mm_left.epos) qcutpos = j.bp_matched;
            if (qcutpos == 0) qcutpos = full_mm.matched_len - (full_mm.epos - split_mm_left.epos);
            if (split_realignment5(qcutpos, full_mm, split_mm_left, split_mm_right, cr) == S2_CR) return S2_CR;
        }
        return S2_UD;
    }
    // ProcessCirc::final_check, process_circ.cpp:1136-1341
    int final_check(MatchedMate &full_mm, MatchedMate &sl, MatchedMate &sr, CircRes2 &cr) {
        const int maxEd = c.P.max_ed, maxSc = c.P.max_sc;
        if (sl.epos < sr.spos) {
            if (full_mm.dir == 1) {
                if (full_mm.spos <= sl.spos) return S2_FR;
                else if (full_mm.epos >= sr.epos) return S2_RF;
            }
            if (full_mm.dir == -1) {
                if (full_mm.epos >= sr.epos) return S2_FR;
                else if (full_mm.spos <= sl.spos) return S2_RF;
            }
        } else if (sr.spos <= sl.spos && sl.epos >= sr.epos) {
            if (full_mm.spos < sr.spos) {
                int off = (int)(sr.spos - full_mm.spos), sc_remained = maxSc - full_mm.sclen_left;
                if (off <= sc_remained) { full_mm.spos = sr.spos; full_mm.sclen_left += off; full_mm.qspos += off; full_mm.matched_len -= off; }
            }
            if (full_mm.epos > sl.epos) {
                int off = (int)(full_mm.epos - sl.epos), sc_remained = maxSc - full_mm.sclen_right;
                if (off <= sc_remained) { full_mm.epos = sl.epos; full_mm.sclen_right += off; full_mm.qepos -= off; full_mm.matched_len -= off; }
            }
            if (full_mm.spos >= sr.spos && full_mm.epos <= sl.epos) {
                overlap_to_spos(c, full_mm); overlap_to_epos(c, full_mm);
                overlap_to_spos(c, sr); overlap_to_epos(c, sr);
                overlap_to_spos(c, sl); overlap_to_epos(c, sl);
                std::vector<pu32i> end_tids, start_tids;
                end_tids_of(sl, end_tids);
                start_tids_of(sr, start_tids);
                int best_ed = maxEd + 1;
                std::vector<uint32_t> common_tid;
                for (size_t i = 0; i < start_tids.size(); ++i)
                    for (size_t j = 0; j < end_tids.size(); ++j) {
                        int sdiff = start_tids[i].second, ediff = end_tids[j].second;
                        if (!(start_tids[i].first == end_tids[j].first && sdiff == ediff)) continue;
                        common_tid.assign(1, start_tids[i].first);
                        uint32_t qcutpos = sl.qepos + sl.sclen_right - ediff;
                        uint32_t beg_bp = sr.spos - sr.sclen_left - sdiff;
                        uint32_t end_bp = sl.epos + sl.sclen_right - ediff;
                        if (full_mm.sclen_right > 0) {
                            if (full_mm.epos + full_mm.sclen_right > end_bp) {
                                uint32_t fm_qcutpos = full_mm.qepos + (end_bp - full_mm.epos);
                                if (split_realignment6(fm_qcutpos, beg_bp, end_bp, fullmap_seq, fullmap_seq_len, common_tid) > maxEd) continue;
                            } else if (full_mm.sclen_right > maxSc) continue;
                        }
                        if (full_mm.sclen_left > 0) {
                            if (full_mm.spos - full_mm.sclen_left < beg_bp) {
                                uint32_t fm_qcutpos = full_mm.sclen_left + (full_mm.spos - beg_bp);
                                if (split_realignment6(fm_qcutpos, beg_bp, end_bp, fullmap_seq, fullmap_seq_len, common_tid) > maxEd) continue;
                            } else if (full_mm.sclen_left > maxSc) continue;
                        }
                        int ed = split_realignment6(qcutpos, beg_bp, end_bp, remain_seq, remain_seq_len, common_tid);
                        if (ed < best_ed) {
                            std::string esignal = two(remain_seq, remain_seq_len, (int64_t)qcutpos - 2), ssignal = two(remain_seq, remain_seq_len, qcutpos);
                            cr.set_bp(beg_bp, end_bp, ssignal, esignal, ref_bp(beg_bp, 2), ref_bp(end_bp - 1, 2));
                            if (ed == 0) return S2_CR;
                            best_ed = ed;
                        }
                    }
                if (best_ed <= maxEd) return S2_CR;
                uint32_t qcutpos = sl.qepos + sl.sclen_right, beg_bp = sr.spos - sr.sclen_left, end_bp = sl.epos + sl.sclen_right;
                if (qcutpos < 2 || qcutpos > (remain_seq_len - 2)) return S2_MCR;
                // (the reference swaps the two names here: "ssignal" holds the two bases before the cut)
                std::string ssignal = two(remain_seq, remain_seq_len, (int64_t)qcutpos - 2), esignal = two(remain_seq, remain_seq_len, qcutpos);
                cr.set_bp(beg_bp, end_bp, ssignal, esignal, ref_bp(beg_bp, 2), ref_bp(end_bp - 1, 2));
                if (!start_tids.empty() && !end_tids.empty()) return S2_NCR;
                return S2_MCR;
            }
        }
        return rescue_overlapping_bsj(full_mm, sl, sr, cr);
    }
    // ProcessCirc::check_split_map (non-overlapping split mates), process_circ.cpp:892-921
    int check_split_map3(MatchedMate &mm_r1, MatchedMate &mm_r2, MatchedMate &partial_mm, bool r1_partial, CircRes2 &cr) {
        int valid, split_read_ed;
        if (r1_partial) {
            split_read_ed = mm_r1.right_ed + mm_r1.left_ed + mm_r1.middle_ed + partial_mm.right_ed + partial_mm.left_ed + partial_mm.middle_ed;
            valid = (mm_r1.qspos < partial_mm.qspos) ? final_check(mm_r2, mm_r1, partial_mm, cr) : final_check(mm_r2, partial_mm, mm_r1, cr);
        } else {
            split_read_ed = mm_r2.right_ed + mm_r2.left_ed + mm_r2.middle_ed + partial_mm.right_ed + partial_mm.left_ed + partial_mm.middle_ed;
            valid = (mm_r2.qspos < partial_mm.qspos) ? final_check(mm_r1, mm_r2, partial_mm, cr) : final_check(mm_r1, partial_mm, mm_r2, cr);
        }
        if (split_read_ed > c.P.max_ed) valid = S2_UD;
        return valid;
    }
    // ProcessCirc::check_split_map (overlapping split mates), process_circ.cpp:924-1134
    int check_split_map4(MatchedMate &mm_r1_1, MatchedMate &mm_r2_1, MatchedMate &mm_r1_2, MatchedMate &mm_r2_2, CircRes2 &cr) {
        const int maxEd = c.P.max_ed;
        int r1_ed = mm_r1_1.right_ed + mm_r1_1.left_ed + mm_r1_1.middle_ed + mm_r1_2.right_ed + mm_r1_2.left_ed + mm_r1_2.middle_ed;
        int r2_ed = mm_r2_1.right_ed + mm_r2_1.left_ed + mm_r2_1.middle_ed + mm_r2_2.right_ed + mm_r2_2.left_ed + mm_r2_2.middle_ed;
        if (r1_ed > maxEd || r2_ed > maxEd) return S2_UD;
        MatchedMate mm_r1_l = (mm_r1_1.spos <= mm_r1_2.spos) ? mm_r1_1 : mm_r1_2, mm_r1_r = (mm_r1_1.spos <= mm_r1_2.spos) ? mm_r1_2 : mm_r1_1;
        MatchedMate mm_r2_l = (mm_r2_1.spos <= mm_r2_2.spos) ? mm_r2_1 : mm_r2_2, mm_r2_r = (mm_r2_1.spos <= mm_r2_2.spos) ? mm_r2_2 : mm_r2_1;
        bool r1_regular_bsj = (mm_r1_l.qspos < mm_r1_r.qspos), r2_regular_bsj = (mm_r2_l.qspos < mm_r2_r.qspos);
        if (r1_regular_bsj && r2_regular_bsj) {
            if (mm_r1_l.dir == 1) {
                if (mm_r1_r.spos <= mm_r2_l.spos) return S2_FR;
                else if (mm_r1_l.epos >= mm_r2_r.epos) return S2_RF;
            }
            if (mm_r1_l.dir == -1) {
                if (mm_r2_r.spos <= mm_r1_l.spos) return S2_FR;
                else if (mm_r2_l.epos >= mm_r1_r.epos) return S2_RF;
            }
        } else if (r1_regular_bsj && !r2_regular_bsj) {
            MatchedMate full_mm = mm_r1_l;
            if (!merge_to_right(full_mm, mm_r1_r)) return S2_UD;
            remain_seq = r2_seq; remain_seq_len = r2_seq_len;
            return final_check(full_mm, mm_r2_l, mm_r2_r, cr);
        } else if (!r1_regular_bsj && r2_regular_bsj) {
            MatchedMate full_mm = mm_r2_l;
            if (!merge_to_right(full_mm, mm_r2_r)) return S2_UD;
            remain_seq = r1_seq; remain_seq_len = r1_seq_len;
            return final_check(full_mm, mm_r1_l, mm_r1_r, cr);
        } else {
            if (mm_r1_l.spos == mm_r2_l.spos && mm_r1_r.epos == mm_r2_r.epos) {
                overlap_to_spos(c, mm_r1_l);
                overlap_to_epos(c, mm_r1_r);
                std::vector<pu32i> end_tids, start_tids;
                end_tids_of(mm_r1_r, end_tids);
                start_tids_of(mm_r1_l, start_tids);
                int best_ed1 = maxEd + 1, best_ed2 = maxEd + 1;
                uint32_t qcutpos, beg_bp, end_bp;
                std::vector<uint32_t> common_tid;
                std::string ssignal1, esignal1, ssignal2, esignal2;
                for (size_t i = 0; i < start_tids.size(); ++i)
                    for (size_t j = 0; j < end_tids.size(); ++j) {
                        int sdiff = start_tids[i].second, ediff = end_tids[j].second;
                        if (!(start_tids[i].first == end_tids[j].first && sdiff == ediff)) continue;
                        common_tid.assign(1, start_tids[i].first);
                        beg_bp = mm_r1_l.spos - mm_r1_l.sclen_left - sdiff;
                        end_bp = mm_r1_r.epos + mm_r1_r.sclen_right - ediff;
                        qcutpos = mm_r1_r.qepos + mm_r1_r.sclen_right - ediff;
                        int ed1 = split_realignment6(qcutpos, beg_bp, end_bp, r1_seq, r1_seq_len, common_tid);
                        if (qcutpos < 2 || qcutpos + 2 > r1_seq_len) { esignal1 = ""; ssignal1 = ""; }
                        else { esignal1 = two(r1_seq, r1_seq_len, (int64_t)qcutpos - 2); ssignal1 = two(r1_seq, r1_seq_len, qcutpos); }
                        qcutpos = mm_r2_r.qepos + mm_r2_r.sclen_right - ediff;
                        int ed2 = split_realignment6(qcutpos, beg_bp, end_bp, r2_seq, r2_seq_len, common_tid);
                        if (qcutpos < 2 || qcutpos + 2 > r2_seq_len) { ssignal2 = ""; esignal2 = ""; }
                        else { esignal2 = two(r2_seq, r2_seq_len, (int64_t)qcutpos - 2); ssignal2 = two(r2_seq, r2_seq_len, qcutpos); }
                        if (ed1 < best_ed1 && ed2 < best_ed2) {
                            std::string nsb = ref_bp(beg_bp, 2), neb = ref_bp(end_bp - 1, 2);
                            if (ssignal1 == "") cr.set_bp(beg_bp, end_bp, ssignal2, esignal2, nsb, neb);
                            else if (ssignal2 == "") cr.set_bp(beg_bp, end_bp, ssignal1, esignal1, nsb, neb);
                            else cr.set_bp(beg_bp, end_bp, consensus2(ssignal1, ssignal2), consensus2(esignal1, esignal2), nsb, neb);
                            best_ed1 = ed1;
                            best_ed2 = ed2;
                        }
                    }
                if (best_ed1 <= maxEd && best_ed2 <= maxEd) return S2_CR;
                qcutpos = mm_r1_r.qepos + mm_r1_r.sclen_right;
                beg_bp = mm_r1_l.spos - mm_r1_l.sclen_left;
                end_bp = mm_r1_r.epos + mm_r1_r.sclen_right;
                if (qcutpos < 2 || qcutpos > (r1_seq_len - 2) || qcutpos > (r2_seq_len - 2)) return S2_MCR;
                esignal1 = two(r1_seq, r1_seq_len, (int64_t)qcutpos - 2); ssignal1 = two(r1_seq, r1_seq_len, qcutpos);
                esignal2 = two(r2_seq, r2_seq_len, (int64_t)qcutpos - 2); ssignal2 = two(r2_seq, r2_seq_len, qcutpos);
                cr.set_bp(beg_bp, end_bp, consensus2(ssignal1, ssignal2), consensus2(esignal1, esignal2), ref_bp(beg_bp, 2), ref_bp(end_bp - 1, 2));
                if (!start_tids.empty() && !end_tids.empty()) return S2_NCR;
                return S2_MCR;
            }
        }
        return S2_UD;
    }

    // print_split_mapping (both forms), process_circ.cpp:1670-1708, followed by fprintf(candid_file, "%d\n", type)
    void print_split(const MatchedMate &mm_r1, const MatchedMate &mm_r2, const MatchedMate *parts, int n_parts, int chr_row, int type) {
        char buf[512];
        const uint32_t sh = c.A->chr_shift[chr_row];
        int k = snprintf(buf, sizeof buf, "%s\t%s\t", cur_name, chr_names[c.A->chr_id[chr_row]]);
        candid.append(buf, k);
        auto one = [&](const MatchedMate &m) {
            int q = snprintf(buf, sizeof buf, "%u\t%u\t%d\t%d\t%d\t", m.spos - sh, m.epos - sh, m.qspos, m.matched_len, m.dir);
            candid.append(buf, q);
        };
        for (int i = 0; i < n_parts; ++i) one(parts[i]);
        one(mm_r1);
        one(mm_r2);
        k = snprintf(buf, sizeof buf, "%d\n", type);
        candid.append(buf, k);
    }
    void push_call(const CircRes2 &b) { calls.push_back(Call{b.chr_id, b.spos, b.epos, b.type, cur_rec, b.start_signal, b.end_signal, b.start_bp_ref, b.end_bp_ref}); }
    // the best_cr bookkeeping shared by both callers (process_circ.cpp:456-478, 614-634): returns true when the caller must return
    bool consider(int type, const CircRes2 &cr, int chr_row, CircRes2 &best_cr) {
        if (type < S2_CR) { best_cr.type = type; return true; }
        if (type >= S2_CR && type <= S2_MCR && type < best_cr.type) {
            const uint32_t sh = c.A->chr_shift[chr_row];
            best_cr.chr_id = c.A->chr_id[chr_row];
            best_cr.spos = cr.spos - sh; best_cr.epos = cr.epos - sh; best_cr.type = type;
            best_cr.start_signal = cr.start_signal; best_cr.end_signal = cr.end_signal;
            best_cr.start_bp_ref = cr.start_bp_ref; best_cr.end_bp_ref = cr.end_bp_ref;
            if (type == S2_CR) { push_call(best_cr); return true; }
        }
        return false;
    }

    struct Rd { const uint8_t *seq, *rcseq; uint32_t len; };
    std::map<uint32_t, RegionalHT> tables;       // get_hash_table_smart's pool, keyed by gene (the pool only saves rebuilding)
    const RegionalHT &table_of(uint32_t gene) {
        auto it = tables.find(gene);
        if (it != tables.end()) return it->second;
        const cm_annot_view *A = c.A;
        const uint32_t gs = A->gene_start[gene], ge = A->gene_end[gene];
        const int gene_len = (int)(ge - gs + 1);
        std::vector<uint8_t> gseq;
        RegionalHT &t = tables[gene];
        if (!pac2char_otf(gs, gene_len, gseq)) gseq.assign((size_t)gene_len + 1, 0);   // buffer left unset by a failed pac2char_otf there
        t.create(ws, gs, ge, gseq.data(), 0, gene_len);
        return t;
    }

    // ProcessCirc::call_circ_single_split, process_circ.cpp:360-482; mr in chromosome coordinates (as parsed from the header)
    void single_split(const Rd &rec1, const Rd &rec2, const cm_mapped_read &mr_in, uint32_t chr_shift) {
        cm_mapped_read mr = mr_in;
        bool r1_partial = mr.mlen_r1 < mr.mlen_r2;
        remain_seq = r1_partial ? (mr.r1_forward ? rec1.seq : rec1.rcseq) : (mr.r2_forward ? rec2.seq : rec2.rcseq);
        fullmap_seq = !r1_partial ? (mr.r1_forward ? rec1.seq : rec1.rcseq) : (mr.r2_forward ? rec2.seq : rec2.rcseq);
        remain_seq_len = r1_partial ? rec1.len : rec2.len;
        fullmap_seq_len = !r1_partial ? rec1.len : rec2.len;
        mr.spos_r1 += chr_shift; mr.epos_r1 += chr_shift; mr.spos_r2 += chr_shift; mr.epos_r2 += chr_shift;      // chrloc2conloc
        MatchedMate mm_r1 = mate_of(mr, 1, (int)rec1.len, r1_partial), mm_r2 = mate_of(mr, 2, (int)rec2.len, !r1_partial);
        const MatchedMate &pm = r1_partial ? mm_r1 : mm_r2;
        const uint32_t plen = r1_partial ? rec1.len : rec2.len;
        const bool right_matched = (pm.qspos - 1) > (plen - pm.qepos);
        uint32_t qspos = right_matched ? 1 : pm.qepos + 1;
        uint32_t qepos = right_matched ? pm.qspos - 1 : plen;
        int whole_seq_len = (int)plen;
        int remain_len = (int)(qepos - qspos + 1);
        if (qepos < qspos || remain_len < ws) return;
        int gi = gene_overlap(mm_r1.spos);
        if (gi < 0) return;
        const cm_annot_view *A = c.A;
        CircRes2 best_cr;
        best_cr.type = S2_NF;
        for (uint32_t g = A->giv_gene_off[gi]; g < A->giv_gene_off[gi + 1]; ++g) {
            const uint32_t gene = A->giv_gene[g];
            const RegionalHT &ht = table_of(gene);
            chaining(qspos, qepos, ht, remain_seq, A->gene_start[gene], bc1);
            if (bc1.best_chain_count <= 0) continue;
            bool forward = r1_partial ? mr.r1_forward : mr.r2_forward;
            int dir = forward ? 1 : -1;
            for (int j = 0; j < std::min(bc1.best_chain_count, S2_TOPCHAIN); ++j) {
                MatchedMate partial_mm(c);
                find_exact_coord(mm_r1, mm_r2, partial_mm, dir, qspos, remain_seq, remain_len, whole_seq_len, bc1.chains[j]);
                if (partial_mm.type != CM_CONCRD) continue;
                int chr_row = get_shift(c, mm_r1.spos);
                CircRes2 cr;
                int type = check_split_map3(mm_r1, mm_r2, partial_mm, r1_partial, cr);
                print_split(mm_r1, mm_r2, &partial_mm, 1, chr_row, type);
                if (consider(type, cr, chr_row, best_cr)) return;
            }
        }
        if (best_cr.type >= S2_CR && best_cr.type <= S2_MCR) push_call(best_cr);
    }
    // ProcessCirc::call_circ_double_split, process_circ.cpp:484-645
    void double_split(const Rd &rec1, const Rd &rec2, const cm_mapped_read &mr_in, uint32_t chr_shift) {
        cm_mapped_read mr = mr_in;
        const uint8_t *r1_remain_seq = mr.r1_forward ? rec1.seq : rec1.rcseq, *r2_remain_seq = mr.r2_forward ? rec2.seq : rec2.rcseq;
        r1_seq = r1_remain_seq; r2_seq = r2_remain_seq; r1_seq_len = rec1.len; r2_seq_len = rec2.len;
        const bool r1_right = (mr.qspos_r1 - 1) > (rec1.len - mr.qepos_r1), r2_right = (mr.qspos_r2 - 1) > (rec2.len - mr.qepos_r2);
        uint32_t r1_qspos = r1_right ? 1 : mr.qepos_r1 + 1, r2_qspos = r2_right ? 1 : mr.qepos_r2 + 1;
        uint32_t r1_qepos = r1_right ? mr.qspos_r1 - 1 : rec1.len, r2_qepos = r2_right ? mr.qspos_r2 - 1 : rec2.len;
        int r1_remain_len = (int)(r1_qepos - r1_qspos + 1), r2_remain_len = (int)(r2_qepos - r2_qspos + 1);
        if (r1_remain_len < ws && r2_remain_len < ws) return;
        if (r1_remain_len < ws || r2_remain_len < ws) single_split(rec1, rec2, mr_in, chr_shift);      // ... and carries on (no return there)
        mr.spos_r1 += chr_shift; mr.epos_r1 += chr_shift; mr.spos_r2 += chr_shift; mr.epos_r2 += chr_shift;
        int gi = gene_overlap(mr.spos_r1);
        if (gi < 0) return;
        MatchedMate mm_r1 = mate_of(mr, 1, (int)rec1.len, true), mm_r2 = mate_of(mr, 2, (int)rec2.len, true);
        const cm_annot_view *A = c.A;
        CircRes2 best_cr;
        best_cr.type = S2_NF;
        for (uint32_t g = A->giv_gene_off[gi]; g < A->giv_gene_off[gi + 1]; ++g) {
            const uint32_t gene = A->giv_gene[g];
            const RegionalHT &ht = table_of(gene);
            chaining(r1_qspos, r1_qepos, ht, r1_remain_seq, A->gene_start[gene], bc1);
            chaining(r2_qspos, r2_qepos, ht, r2_remain_seq, A->gene_start[gene], bc2);
            if (bc1.best_chain_count <= 0 && bc2.best_chain_count <= 0) continue;
            if (bc1.best_chain_count <= 0 || bc2.best_chain_count <= 0) { single_split(rec1, rec2, mr_in, chr_shift); continue; }
            for (int j = 0; j < std::min(bc1.best_chain_count, S2_TOPCHAIN); ++j)
                for (int k = 0; k < std::min(bc2.best_chain_count, S2_TOPCHAIN); ++k) {
                    MatchedMate r1_partial_mm(c), r2_partial_mm(c);
                    set_mm(bc1.chains[j], r1_qspos, r1_remain_len, mm_r1.dir, r1_partial_mm);
                    set_mm(bc2.chains[k], r2_qspos, r2_remain_len, mm_r2.dir, r2_partial_mm);
                    overlap_to_spos(c, mm_r1); overlap_to_spos(c, mm_r2); overlap_to_spos(c, r1_partial_mm); overlap_to_spos(c, r2_partial_mm);
                    std::vector<uint32_t> common_tid;
                    std::vector<MatchedMate> segments{mm_r1, mm_r2, r1_partial_mm, r2_partial_mm};
                    if (!same_transcript_n(segments, 4, common_tid)) continue;
                    bool success;
                    if (bc1.chains[j].frags[0].rpos <= bc2.chains[k].frags[0].rpos)
                        success = ext.extend_both_mates(bc1.chains[j], bc2.chains[k], common_tid, r1_remain_seq, r2_remain_seq, (int)r1_qspos, (int)r2_qspos,
                                                        (int)r1_qepos, (int)r2_qepos, r1_partial_mm, r2_partial_mm);
                    else
                        success = ext.extend_both_mates(bc2.chains[k], bc1.chains[j], common_tid, r2_remain_seq, r1_remain_seq, (int)r2_qspos, (int)r1_qspos,
                                                        (int)r2_qepos, (int)r1_qepos, r2_partial_mm, r1_partial_mm);
                    if (!success) continue;
                    if (!(r1_partial_mm.type == CM_CONCRD && r2_partial_mm.type == CM_CONCRD)) continue;
                    int chr_row = get_shift(c, mm_r1.spos);
                    CircRes2 cr;
                    int type = check_split_map4(mm_r1, mm_r2, r1_partial_mm, r2_partial_mm, cr);
                    MatchedMate parts[2] = {r1_partial_mm, r2_partial_mm};
                    print_split(mm_r1, mm_r2, parts, 2, chr_row, type);
                    if (consider(type, cr, chr_row, best_cr)) return;
                }
        }
        if (best_cr.type >= S2_CR && best_cr.type <= S2_MCR) push_call(best_cr);
        else single_split(rec1, rec2, mr_in, chr_shift);
    }
    // ProcessCirc::call_circ, process_circ.cpp:334-358
    void call_circ(const Rd &rec1, const Rd &rec2, const cm_mapped_read &mr, uint32_t chr_shift, uint64_t rec_index, const char *name) {
        fullmap_seq = remain_seq = r1_seq = r2_seq = nullptr;
        fullmap_seq_len = remain_seq_len = r1_seq_len = r2_seq_len = 0;
        cur_rec = rec_index;
        cur_name = name;
        if (mr.type == CM_CHIBSJ) single_split(rec1, rec2, mr, chr_shift);
        else if (mr.type == CM_CHI2BSJ) double_split(rec1, rec2, mr, chr_shift);
    }
};


// ProcessCirc::report_events + both_side_consensus + get_consensus(vector), process_circ.cpp:1554-1631, utils.cpp:758-816;
// CircRes::operator< / ==, common.cpp:479-493 (chr compared as strings; std::sort, unstable, like the reference)
std::string consensus_n(const std::vector<std::string> &v) {
    std::string res;
    if (v.empty()) return res;
    for (size_t i = 1; i < v.size(); ++i) if (v[i].length() != v[i - 1].length()) return res;
    const char nuc[4] = {'A', 'C', 'G', 'T'};
    for (size_t i = 0; i < v[0].length(); ++i) {
        unsigned counts[256];
        memset(counts, 0, sizeof counts);
        for (auto &x : v) ++counts[(uint8_t)x[i]];
        counts['A'] += counts['a']; counts['C'] += counts['c']; counts['G'] += counts['g']; counts['T'] += counts['t'];
        unsigned max_cnt = 0;
        char ch = 'N';
        for (int k = 0; k < 4; ++k) if (counts[(uint8_t)nuc[k]] > max_cnt) { max_cnt = counts[(uint8_t)nuc[k]]; ch = nuc[k]; }
        res += (max_cnt >= (v.size() / 2)) ? ch : 'N';
    }
    return res;
}
struct RepRow { std::string chr, rname; uint32_t spos, epos; int type; std::string ss, es, sb, eb; };
bool rep_less(const RepRow &a, const RepRow &b) {
    if (a.chr != b.chr) return a.chr < b.chr;
    if (a.spos != b.spos) return a.spos < b.spos;
    if (a.epos != b.epos) return a.epos < b.epos;
    return a.type < b.type;
}
void report_events(std::vector<RepRow> &res, FILE *f) {
    if (res.empty()) return;
    std::sort(res.begin(), res.end(), rep_less);
    auto same = [](const RepRow &a, const RepRow &b) { return a.chr == b.chr && a.spos == b.spos && a.epos == b.epos; };
    auto flush = [&](const RepRow &last, const std::vector<RepRow> &grp, int cnt) {
        if (last.type != S2_CR) return;
        std::vector<std::string> ss, es;
        for (auto &g : grp) { ss.push_back(g.ss); es.push_back(g.es); }
        std::string ssc = consensus_n(ss), esc = consensus_n(es);
        // compare() against start_bp_ref: both are C-string based in the reference (set_bp assigns char* to string)
        bool pass = (ssc.compare(last.sb) == 0) && (esc.compare(last.eb) == 0);
        fprintf(f, "%s\t%u\t%u\t%d\t%s\t%s-%s\t%s-%s\t%s\t", last.chr.c_str(), last.spos, last.epos, cnt, "STC", ssc.c_str(), esc.c_str(),
                last.sb.c_str(), last.eb.c_str(), pass ? "Pass" : "Fail");
        for (size_t j = 0; j + 1 < grp.size(); ++j) fprintf(f, "%s,", grp[j].rname.c_str());
        fprintf(f, "%s\n", grp.back().rname.c_str());
    };
    int cnt = 1;
    RepRow last = res[0];
    std::vector<RepRow> grp{res[0]};
    for (size_t i = 1; i < res.size(); ++i) {
        if (same(res[i], last)) { cnt++; grp.push_back(res[i]); }
        else { flush(last, grp, cnt); cnt = 1; last = res[i]; grp.assign(1, res[i]); }
    }
    flush(last, grp, cnt);
}
}  // namespace

// ============================ C entry points for tests / bench ============================
extern "C" {

// Seeds of every (pair, mate, orientation): same layout as cm_seed_batch (include/circminer_hot.h).
int oracle_seed_batch(const cm_params *P, const cm_index_view *X, const cm_reads *R, uint32_t n_slots, uint32_t *out_start,
                      uint32_t *out_cnt, uint32_t *out_raw) {
    Ctx c{*P, X, nullptr};
    std::vector<MatchedKmer> fl(max_seg_cnt(c) + 2);
    for (uint64_t p = 0; p < R->n_pairs; ++p)
        for (int mate = 0; mate < 2; ++mate) {
            const uint8_t *s = mate ? R->seq2 + R->off2[p] : R->seq1 + R->off1[p];
            int len = (int)(mate ? R->off2[p + 1] - R->off2[p] : R->off1[p + 1] - R->off1[p]);
            Rec r;
            make_rec(r, s, len);
            for (int o = 0; o < 2; ++o) {
                split_match_hash(c, o ? r.rc.data() : r.seq, len, fl.data());
                uint64_t base = (((uint64_t)p * 2 + mate) * 2 + o) * n_slots;
                for (uint32_t sl = 0; sl < n_slots; ++sl) {
                    bool used = (int)(sl * 2) < max_seg_cnt(c) && (int)(sl + 1) * c.kmer() <= len;
                    const MatchedKmer &mk = fl[sl * 2];
                    out_start[base + sl] = (used && mk.frags >= 0) ? (uint32_t)mk.frags : 0;
                    out_cnt[base + sl] = used ? mk.frag_count : 0;
                    out_raw[base + sl] = used ? mk.raw : 0;
                }
            }
        }
    return 0;
}

int oracle_chain_batch(const cm_params *P, const cm_index_view *X, const cm_annot_view *A, const cm_reads *R, cm_chain *out_chains,
                       int32_t *out_nchain, int32_t *out_high) {
    Ctx c{*P, X, A};
    Scratch S(c);
    for (uint64_t p = 0; p < R->n_pairs; ++p)
        for (int mate = 0; mate < 2; ++mate) {
            const uint8_t *s = mate ? R->seq2 + R->off2[p] : R->seq1 + R->off1[p];
            int len = (int)(mate ? R->off2[p + 1] - R->off2[p] : R->off1[p + 1] - R->off1[p]);
            Rec r;
            make_rec(r, s, len);
            for (int o = 0; o < 2; ++o) {
                int hh;
                get_best_chains(c, o ? r.rc.data() : r.seq, len, S.fbc_r1, S.fl.data(), hh);
                uint64_t q = ((uint64_t)p * 2 + mate) * 2 + o;
                out_nchain[q] = S.fbc_r1.best_chain_count;
                out_high[q] = hh;
                for (int k = 0; k < CM_BESTCHAINLIM; ++k) {
                    if (k < S.fbc_r1.best_chain_count) copy_chain(S.fbc_r1.chains[k], out_chains[q * CM_BESTCHAINLIM + k]);
                    else memset(&out_chains[q * CM_BESTCHAINLIM + k], 0, sizeof(cm_chain));
                }
            }
        }
    return 0;
}

// One mapping round (map_reads, src/circminer.cpp:354-406) over pairs [p0, p1).
// state[] is the carried MatchedRead (in/out), active[] in/out, category[] out (-1 if inactive).
int oracle_map_round(const cm_params *P, const cm_index_view *X, const cm_annot_view *A, const cm_reads *R, int is_last,
                     cm_mapped_read *state, uint8_t *active, int32_t *category, uint64_t p0, uint64_t p1) {
    Ctx c{*P, X, A};
    Scratch S(c);
    if (P->max_chain_len > CM_BESTCHAINLIM) return CM_EINVAL;
    for (uint64_t p = p0; p < p1 && p < R->n_pairs; ++p) {
        if (!active[p]) { category[p] = -1; continue; }
        Rec r1, r2;
        make_rec(r1, R->seq1 + R->off1[p], (int)(R->off1[p + 1] - R->off1[p]));
        make_rec(r2, R->seq2 + R->off2[p], (int)(R->off2[p + 1] - R->off2[p]));
        cm_mapped_read &mr = state[p];
        int st = process_read(c, S, r1, r2, mr);
        category[p] = st;
        bool skip = (P->scan_level == 0 && st == CM_CONCRD) ||
                    (P->scan_level == 1 && st == CM_CONCRD && mr.gm_compatible && (mr.ed_r1 + mr.ed_r2 == 0) &&
                     (mr.mlen_r1 + mr.mlen_r2 == (uint32_t)(r1.seq_len + r2.seq_len)));
        bool requeue = (!is_last && !skip) || (is_last && (mr.type == CM_CHIBSJ || mr.type == CM_CHI2BSJ));
        active[p] = requeue ? 1 : 0;
        // what the next round's fill_map_info would read back from the remain FASTQ header
        // (filter.cpp:422-444, fastq_parser.cpp:214-267): unmapped types keep only `type`.
        if (requeue && !is_last && !mapped_type(mr.type)) {
            int t = mr.type;
            default_mr(c, mr);
            mr.type = t;
        }
    }
    return 0;
}

void oracle_default_state(const cm_params *P, cm_mapped_read *state, uint8_t *active, uint64_t n) {
    Ctx c{*P, nullptr, nullptr};
    for (uint64_t i = 0; i < n; ++i) { default_mr(c, state[i]); active[i] = 1; }
}

// Stand-alone DP entry points (property tests of A14).
int oracle_edit_side(const cm_params *P, const uint8_t *s, int n, const uint8_t *t, int m, int left, int *indel, int *score) {
    Ctx c{*P, nullptr, nullptr};
    return local_alignment_side(c, s, n, t, m, *indel, *score, left != 0);
}
int oracle_drop_sc(const cm_params *P, const uint8_t *s, int n, const uint8_t *t, int m, int left, int *sclen, int *indel, int *score) {
    Ctx c{*P, nullptr, nullptr};
    return local_alignment_sc(c, s, n, t, m, *sclen, *indel, *score, left != 0);
}
int oracle_one_side(const uint8_t *s, int n, const uint8_t *t, int m, int w) { return global_one_side_banded_alignment(s, n, t, m, w); }


// Stage 2 over records already in the order of the sorted remain files (ProcessCirc::do_process, process_circ.cpp:195-331):
// states[i] is the carried MatchedRead of pair i as fill_map_info parses it (chromosome coordinates, chr_id = row of the
// chromosome table, contig_num); X / A are the packed contigs and their annotation; chr_contig / chr_shift the chromosome
// table.  Writes <out>.candidates.pam and <out>.circ_report.
int oracle_circ_run(const cm_params *P, int window, uint32_t n_contigs, const cm_index_view *X, const cm_annot_view *A, uint32_t n_chr,
                    const char *const *chr_names, const uint32_t *chr_contig, const uint32_t *chr_shift, uint64_t n_rec,
                    const char *const *names, const cm_reads *R, const cm_mapped_read *states, const char *candid_path,
                    const char *report_path) {
    FILE *fc = fopen(candid_path, "w");
    if (!fc) return CM_EINVAL;
    std::vector<RepRow> rows;
    int cur_contig = -1;
    CircCaller *cc = nullptr;
    for (uint64_t i = 0; i < n_rec; ++i) {
        const cm_mapped_read &mr = states[i];
        if (mr.contig_num < 0 || (uint32_t)mr.contig_num >= n_contigs || mr.chr_id < 0 || (uint32_t)mr.chr_id >= n_chr) continue;
        if (cur_contig != mr.contig_num) {        // load_genome + refresh_hash_table_list
            if (cc) { fwrite(cc->candid.data(), 1, cc->candid.size(), fc); for (auto &c : cc->calls) rows.push_back(RepRow{chr_names[c.chr_id], names[c.rec], c.spos, c.epos, c.type, c.ss, c.es, c.sb, c.eb}); delete cc; }
            cur_contig = mr.contig_num;
            Ctx c{*P, X + cur_contig, A + cur_contig};
            cc = new CircCaller(c, window, chr_names);
        }
        Rec r1, r2;
        make_rec(r1, R->seq1 + R->off1[i], (int)(R->off1[i + 1] - R->off1[i]));
        make_rec(r2, R->seq2 + R->off2[i], (int)(R->off2[i + 1] - R->off2[i]));
        r1.rc.push_back(0); r2.rc.push_back(0);
        CircCaller::Rd a{r1.seq, r1.rc.data(), (uint32_t)r1.seq_len}, b{r2.seq, r2.rc.data(), (uint32_t)r2.seq_len};
        // check_removables: tables of genes that end before this pair (the pool only saves rebuilding; results do not depend on it)
        for (auto it = cc->tables.begin(); it != cc->tables.end();) it = (mr.spos_r1 + chr_shift[mr.chr_id] > it->second.gene_epos) ? cc->tables.erase(it) : ++it;
        cc->call_circ(a, b, mr, chr_shift[mr.chr_id], i, names[i]);
    }
    if (cc) { fwrite(cc->candid.data(), 1, cc->candid.size(), fc); for (auto &c : cc->calls) rows.push_back(RepRow{chr_names[c.chr_id], names[c.rec], c.spos, c.epos, c.type, c.ss, c.es, c.sb, c.eb}); delete cc; }
    fclose(fc);
    (void)chr_contig;
    FILE *fr = fopen(report_path, "w");
    if (!fr) return CM_EINVAL;
    report_events(rows, fr);
    fclose(fr);
    return 0;
}

}  // extern "C"
