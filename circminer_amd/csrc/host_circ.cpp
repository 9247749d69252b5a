// Stage 2 (ProcessCirc), first and last step only -- SURVEY.md §8(f) row N3 is NOT complete: the per-candidate
// back-splice-junction calling in between (call_circ_single_split / call_circ_double_split, check_split_map,
// split_realignment, src/process_circ.cpp:334-1552) is not built yet.  What is here is host code a later round keeps:
//
//   * cm_sort_remain  : ProcessCirc::sort_fq (src/process_circ.cpp:179-193), the
//                       `cat f | paste - - - - | sort -k2,2n | tr "\t" "\n" > f.srt` pipeline on the last round's remain
//                       FASTQ: records ordered by the numeric value of the second blank-separated field of the header
//                       (genome_spos; a non-numeric field counts as 0), ties by the bytes of the whole pasted line
//                       (GNU sort's last-resort comparison in the C locale -- the reference does not set a locale, see
//                       SURVEY §8(f) N3; tests compare with `LC_ALL=C sort`);
//   * cm_circ_report  : ProcessCirc::report_events + both_side_consensus (src/process_circ.cpp:1554-1631) with
//                       CircRes::operator< / operator== (src/common.cpp:479-493) and get_consensus (src/utils.cpp:771-816):
//                       std::sort of the calls, one <out>.circ_report row per (chr, spos, epos) group whose first
//                       element is of type CR.  The order of the read names inside a row is whatever libstdc++'s
//                       (unstable) std::sort leaves for the input order given, exactly as in the reference.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "circminer_hot.h"

namespace {

constexpr int CR = 20;                    // src/process_circ.h:16
const char *const CIRC_TYPE[3] = {"STC", "MTC", "NC"};   // src/process_circ.cpp:86-88

// numeric key of `sort -k2,2n`: field 2 = from the end of field 1 to the end of the next run of non-blanks; -n skips
// leading blanks, takes an optional '-' and digits (no overflow in GNU sort; genome_spos fits 63 bits)
struct NumKey {
    bool neg = false;
    std::string digits;                   // without leading zeros; empty = 0
};
NumKey key_of(const std::string &line) {
    size_t i = 0;
    const size_t n = line.size();
    auto blank = [](char c) { return c == ' ' || c == '\t'; };
    while (i < n && blank(line[i])) ++i;           // field 1: leading blanks + non-blanks
    while (i < n && !blank(line[i])) ++i;
    size_t j = i;                                  // field 2 starts here (leading blanks belong to it)
    while (j < n && blank(line[j])) ++j;
    NumKey k;
    if (j < n && line[j] == '-') {
        k.neg = true;
        ++j;
    }
    size_t d = j;
    while (d < n && line[d] >= '0' && line[d] <= '9') ++d;
    while (j < d && line[j] == '0') ++j;
    k.digits.assign(line, j, d - j);
    if (k.digits.empty()) k.neg = false;           // "-0", "-", "*" ... compare as 0
    return k;
}
int cmp_key(const NumKey &a, const NumKey &b) {
    if (a.neg != b.neg) return a.neg ? -1 : 1;
    int c;
    if (a.digits.size() != b.digits.size()) c = a.digits.size() < b.digits.size() ? -1 : 1;
    else c = a.digits.compare(b.digits);
    c = c < 0 ? -1 : (c > 0 ? 1 : 0);
    return a.neg ? -c : c;
}

struct Call {                             // CircRes, src/common.h:406-423
    std::string chr, rname, start_signal, end_signal, start_bp_ref, end_bp_ref;
    uint32_t spos = 0, epos = 0;
    int type = 0;
    bool operator<(const Call &r) const {
        if (chr != r.chr) return chr < r.chr;
        if (spos != r.spos) return spos < r.spos;
        if (epos != r.epos) return epos < r.epos;
        return type < r.type;
    }
    bool same_event(const Call &r) const { return chr == r.chr && spos == r.spos && epos == r.epos; }
};

std::string consensus(const std::vector<const std::string *> &v) {          // get_consensus(vector<string>)
    std::string res;
    if (v.empty()) return res;
    for (size_t i = 1; i < v.size(); ++i)
        if (v[i]->size() != v[i - 1]->size()) return res;
    const char nuc[4] = {'A', 'C', 'G', 'T'};
    for (size_t i = 0; i < v[0]->size(); ++i) {
        unsigned cnt[4] = {0, 0, 0, 0};
        for (const std::string *s : v) {
            switch ((*s)[i]) {
                case 'A': case 'a': ++cnt[0]; break;
                case 'C': case 'c': ++cnt[1]; break;
                case 'G': case 'g': ++cnt[2]; break;
                case 'T': case 't': ++cnt[3]; break;
                default: break;
            }
        }
        unsigned mx = 0;
        char ch = 'N';
        for (int k = 0; k < 4; ++k)
            if (cnt[k] > mx) {
                mx = cnt[k];
                ch = nuc[k];
            }
        res += (mx >= v.size() / 2) ? ch : 'N';
    }
    return res;
}

void report_group(FILE *f, const std::vector<const Call *> &g) {
    const Call &last = *g[0];
    if (last.type != CR) return;                       // "won't print novel events"
    std::vector<const std::string *> ss, es;
    for (const Call *c : g) {
        ss.push_back(&c->start_signal);
        es.push_back(&c->end_signal);
    }
    const std::string ss_con = consensus(ss), es_con = consensus(es);
    const bool pass = ss_con == last.start_bp_ref && es_con == last.end_bp_ref;
    fprintf(f, "%s\t%u\t%u\t%d\t%s\t%s-%s\t%s-%s\t%s\t", last.chr.c_str(), last.spos, last.epos, (int)g.size(), CIRC_TYPE[last.type - CR], ss_con.c_str(),
            es_con.c_str(), last.start_bp_ref.c_str(), last.end_bp_ref.c_str(), pass ? "Pass" : "Fail");
    for (size_t j = 0; j + 1 < g.size(); ++j) fprintf(f, "%s,", g[j]->rname.c_str());
    fprintf(f, "%s\n", g.back()->rname.c_str());
}

}  // namespace

// RegionalHashTable::create_table / hash_val / add_loc (src/hash_table.cpp:58-112): every window_size-mer of the gene region
// seq[0..len), location = start + offset; stage 2 uses ws = 8 (src/circminer.cpp:348) and probes it every 3 bases of the
// unmapped part of a read.  Flattened: bucket hv owns loc[off[hv] .. off[hv + 1]) in ascending location.  A bucket that
// would hold more than MAXHIT = 1000 locations is emptied, a window with a byte outside ACGTacgt has no hash value.
extern "C" int cm_regional_table_build(const uint8_t *seq, uint32_t start, int32_t len, int32_t window_size, uint32_t **off_out, uint32_t **loc_out) {
    if (!off_out || !loc_out || window_size < 1 || window_size > 12 || (len > 0 && !seq)) return CM_EINVAL;
    constexpr uint32_t MAXHIT = 1000;
    const size_t size = (size_t)1 << (2 * window_size);
    std::vector<uint32_t> cnt(size + 1, 0);
    std::vector<int32_t> hv_at;
    const int n_win = len >= window_size ? len - window_size + 1 : 0;
    hv_at.resize((size_t)n_win);
    auto code = [](uint8_t ch) -> int {
        switch (ch) {
            case 'A': case 'a': return 0;
            case 'C': case 'c': return 1;
            case 'G': case 'g': return 2;
            case 'T': case 't': return 3;
            default: return -1;
        }
    };
    for (int i = 0; i < n_win; ++i) {
        int32_t hv = 0;
        for (int k = 0; k < window_size; ++k) {
            const int c = code(seq[i + k]);
            if (c < 0) {
                hv = -1;
                break;
            }
            hv = (hv << 2) | c;
        }
        hv_at[(size_t)i] = hv;
        if (hv >= 0) ++cnt[(size_t)hv];
    }
    uint32_t *off = (uint32_t *)malloc((size + 1) * sizeof(uint32_t));
    if (!off) return CM_ENOMEM;
    uint32_t total = 0;
    for (size_t h = 0; h < size; ++h) {
        off[h] = total;
        if (cnt[h] <= MAXHIT) total += cnt[h];
        else cnt[h] = 0;                                   // frag_count > MAXHIT -> frag_count = 0
    }
    off[size] = total;
    uint32_t *loc = (uint32_t *)malloc((total ? total : 1) * sizeof(uint32_t));
    if (!loc) {
        free(off);
        return CM_ENOMEM;
    }
    std::vector<uint32_t> fill(size, 0);
    for (int i = 0; i < n_win; ++i) {
        const int32_t hv = hv_at[(size_t)i];
        if (hv < 0 || cnt[(size_t)hv] == 0) continue;
        loc[off[hv] + fill[(size_t)hv]++] = start + (uint32_t)i;
    }
    *off_out = off;
    *loc_out = loc;
    return CM_OK;
}
extern "C" void cm_regional_table_free(uint32_t *off, uint32_t *loc) {
    free(off);
    free(loc);
}

extern "C" int cm_sort_remain(const char *in_path, const char *out_path) {
    if (!in_path || !out_path) return CM_EINVAL;
    FILE *in = fopen(in_path, "rb");
    if (!in) return CM_EINVAL;
    struct Rec {
        std::string pasted;                            // the four lines joined by TABs, as `paste - - - -` emits them
        NumKey key;
    };
    std::vector<Rec> recs;
    char *line = nullptr;
    size_t cap = 0;
    int part = 0;
    std::string cur;
    for (;;) {
        const ssize_t got = getline(&line, &cap, in);
        if (got < 0) break;
        size_t len = (size_t)got;
        if (len && line[len - 1] == '\n') --len;
        if (part) cur += '\t';
        cur.append(line, len);
        if (++part == 4) {
            recs.push_back(Rec{cur, key_of(cur)});
            cur.clear();
            part = 0;
        }
    }
    if (part) {                                        // paste pads an incomplete last group with empty fields
        for (; part < 4; ++part) cur += '\t';
        recs.push_back(Rec{cur, key_of(cur)});
    }
    free(line);
    fclose(in);
    std::sort(recs.begin(), recs.end(), [](const Rec &a, const Rec &b) {
        const int c = cmp_key(a.key, b.key);
        if (c) return c < 0;
        const size_t m = std::min(a.pasted.size(), b.pasted.size());
        const int d = memcmp(a.pasted.data(), b.pasted.data(), m);          // C locale: unsigned bytes
        if (d) return d < 0;
        return a.pasted.size() < b.pasted.size();
    });
    FILE *out = fopen(out_path, "wb");
    if (!out) return CM_EINVAL;
    for (Rec &r : recs) {
        for (char &c : r.pasted)
            if (c == '\t') c = '\n';                    // tr "\t" "\n"
        if (fwrite(r.pasted.data(), 1, r.pasted.size(), out) != r.pasted.size() || fputc('\n', out) == EOF) {
            fclose(out);
            return CM_EIO;
        }
    }
    return fclose(out) == 0 ? CM_OK : CM_EIO;
}

extern "C" int cm_circ_report(const cm_circ_res *res, uint64_t n, const char *report_path) {
    if (!report_path || (n && !res)) return CM_EINVAL;
    FILE *f = fopen(report_path, "w");                  // open_report_file: the file exists even without events
    if (!f) return CM_EINVAL;
    if (n == 0) {
        fclose(f);
        return CM_OK;
    }
    std::vector<Call> calls(n);
    for (uint64_t i = 0; i < n; ++i) {
        const cm_circ_res &r = res[i];
        Call &c = calls[i];
        c.chr = r.chr ? r.chr : "";
        c.rname = r.rname ? r.rname : "";
        c.start_signal = r.start_signal ? r.start_signal : "";
        c.end_signal = r.end_signal ? r.end_signal : "";
        c.start_bp_ref = r.start_bp_ref ? r.start_bp_ref : "";
        c.end_bp_ref = r.end_bp_ref ? r.end_bp_ref : "";
        c.spos = r.spos;
        c.epos = r.epos;
        c.type = r.type;
    }
    std::sort(calls.begin(), calls.end());
    std::vector<const Call *> group{&calls[0]};
    for (uint64_t i = 1; i < n; ++i) {
        if (calls[i].same_event(*group[0])) {
            group.push_back(&calls[i]);
        } else {
            report_group(f, group);
            group.assign(1, &calls[i]);
        }
    }
    report_group(f, group);
    const bool bad = ferror(f) != 0;
    return (fclose(f) == 0 && !bad) ? CM_OK : CM_EIO;
}
