// FASTQ ingest, carry-over header and PAM / remain-FASTQ writers on the host.
//
//   * parser  : FASTQParser::get_next_read / extract_map_info / fill_map_info (reference
//               src/fastq_parser.cpp:100-269, fastq_parser.h:57-67): '@' + header line split on spaces, token 0 is
//               the read name (a trailing "/x" is cut), 23 tokens = the state a previous round carried over;
//               plain or gzip input (gzread handles both, as in the reference);
//   * remain  : FilterRead::write_read_category PE (src/filter.cpp:413-455): "<out>_<round>_remain_R{1,2}.fastq",
//               header "@name gspos type chr spos epos mlen qspos qepos dir ed chr ... tlen junc gm contig";
//   * PAM     : SAMOutput::write_pam_rec_pe (src/output.cpp:279-299);
//   * SAM     : SAMOutput::print_header / set_flag_pe / set_output_pe / write_sam_rec_pe (src/output.cpp:118-277, 301-333).
// Reads are delivered in the cm_reads layout (concatenated bytes + offsets), whole batches at a time, so a
// batch goes to cm_reads_upload without another copy; with cm_host_alloc'ed staging the copy is one DMA.
// Deliberately defined where the reference has undefined behaviour: names shorter than 2 characters are not
// inspected for the "/x" suffix, header lines with more than 23 tokens are treated like fresh reads.
#include <zlib.h>

#include <cinttypes>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "circminer_hot.h"

namespace {

constexpr int FQCOMMENTCNT = 23;         // src/fastq_parser.h:12
constexpr int INF_I = 1000000000;        // INF, src/common.h:34
constexpr size_t BLOCK = 16u << 20;

bool mapped_type(int t) {
    return t == CM_CONCRD || t == CM_DISCRD || t == CM_CHIORF || t == CM_CHIBSJ || t == CM_CHI2BSJ || t == CM_CONGNM || t == CM_CONGEN;
}

void unmapped_state(cm_mapped_read &m, int type, int max_ed) {        // fill_map_info else-branch (:246-267)
    memset(&m, 0, sizeof m);
    m.ed_r1 = m.ed_r2 = max_ed + 1;
    m.type = type;
    m.tlen = INF_I;
    m.chr_id = -1;
    m.r1_forward = m.r2_forward = 1;
}

struct Stream {
    gzFile gz = nullptr;
    std::vector<char> buf;
    size_t pos = 0, end = 0;
    bool eof = false;
    bool open(const char *path) {
        gz = gzopen(path, "r");
        if (!gz) return false;
        gzbuffer(gz, 1u << 20);
        buf.resize(BLOCK);
        return true;
    }
    void close() {
        if (gz) gzclose(gz);
        gz = nullptr;
    }
    // next line as [ptr, ptr + len) without the '\n'; false at end of input
    bool line(const char *&p, size_t &len) {
        for (;;) {
            const char *nl = (const char *)memchr(buf.data() + pos, '\n', end - pos);
            if (nl) {
                p = buf.data() + pos;
                len = (size_t)(nl - p);
                pos += len + 1;
                return true;
            }
            if (eof) {
                if (pos >= end) return false;
                p = buf.data() + pos;             // last line without a newline
                len = end - pos;
                pos = end;
                return true;
            }
            if (pos > 0) {                        // keep the partial line, refill behind it
                memmove(buf.data(), buf.data() + pos, end - pos);
                end -= pos;
                pos = 0;
            }
            if (end == buf.size()) buf.resize(buf.size() * 2);
            const int got = gzread(gz, buf.data() + end, (unsigned)(buf.size() - end));
            if (got <= 0) eof = true;
            else end += (size_t)got;
        }
    }
};

struct Side {
    std::vector<uint8_t> seq, qual;
    std::vector<uint64_t> off;
    std::vector<char> names;
    std::vector<uint64_t> name_off;
    void clear() {
        seq.clear();
        qual.clear();
        off.assign(1, 0);
        names.clear();
        name_off.assign(1, 0);
    }
};

}  // namespace

struct cm_fastq {
    Stream s1, s2;
    // two generations of batch storage, used alternately: the batch a call returns stays valid while the NEXT call fills
    // the other one, so a caller can parse batch k+1 while the GPU maps batch k and still write batch k's records after
    struct Gen {
        Side a, b;
        std::vector<cm_mapped_read> prior;
    } gen[2];
    int cur = 1;
    std::vector<std::string> chr_names;
    int max_ed = 4;
    bool any_prior = false;
    std::string err;
};

namespace {

int chr_lookup(const cm_fastq *f, const char *tok, size_t len) {
    for (size_t i = 0; i < f->chr_names.size(); ++i)
        if (f->chr_names[i].size() == len && memcmp(f->chr_names[i].data(), tok, len) == 0) return (int)i;
    return -1;
}

// one record of one stream; fills `st` (state carried in the header) when want_state
// returns 1 = record, 0 = end of input, -1 = format error
int parse_record(cm_fastq *f, Stream &s, Side &side, bool want_state, cm_mapped_read *st, bool *carried) {
    const char *p;
    size_t len;
    if (!s.line(p, len)) return 0;
    if (len == 0 && s.eof && s.pos >= s.end) return 0;
    if (len == 0 || p[0] != '@') return -1;                       // has_next asserts the '@'
    // header: tokens separated by runs of spaces (strtok)
    const char *tok[FQCOMMENTCNT + 1];
    size_t tl[FQCOMMENTCNT + 1];
    int nt = 0;
    {
        size_t i = 1;
        while (i < len) {
            while (i < len && p[i] == ' ') ++i;
            if (i >= len) break;
            size_t j = i;
            while (j < len && p[j] != ' ') ++j;
            if (nt <= FQCOMMENTCNT) {
                tok[nt] = p + i;
                tl[nt] = j - i;
            }
            ++nt;
            i = j;
        }
    }
    size_t nlen = nt ? tl[0] : 0;
    if (nlen >= 2 && tok[0][nlen - 2] == '/') nlen -= 2;          // extract_map_info :193-194
    if (nt) side.names.insert(side.names.end(), tok[0], tok[0] + nlen);
    side.names.push_back('\0');
    side.name_off.push_back(side.names.size());
    if (want_state) {
        *carried = nt == FQCOMMENTCNT;
        if (nt != FQCOMMENTCNT) {
            unmapped_state(*st, CM_NOPROC_NOMATCH, f->max_ed);         // what cm_reads_upload uses for prior == NULL
        } else {
            auto num = [&](int k) { return strtoull(std::string(tok[k], tl[k]).c_str(), nullptr, 10); };
            auto inum = [&](int k) { return atoi(std::string(tok[k], tl[k]).c_str()); };
            const int type = inum(2);
            if (mapped_type(type)) {
                cm_mapped_read &m = *st;
                memset(&m, 0, sizeof m);
                m.type = type;
                m.chr_id = chr_lookup(f, tok[3], tl[3]);
                m.spos_r1 = (uint32_t)num(4);
                m.epos_r1 = (uint32_t)num(5);
                m.mlen_r1 = (uint32_t)inum(6);
                m.qspos_r1 = (uint32_t)num(7);
                m.qepos_r1 = (uint32_t)num(8);
                m.r1_forward = tok[9][0] == '+';
                m.ed_r1 = inum(10);
                m.spos_r2 = (uint32_t)num(12);
                m.epos_r2 = (uint32_t)num(13);
                m.mlen_r2 = (uint32_t)inum(14);
                m.qspos_r2 = (uint32_t)num(15);
                m.qepos_r2 = (uint32_t)num(16);
                m.r2_forward = tok[17][0] == '+';
                m.ed_r2 = inum(18);
                m.tlen = inum(19);
                m.junc_num = (uint16_t)num(20);
                m.gm_compatible = tok[21][0] == '1';
                m.contig_num = inum(22);
            } else {
                unmapped_state(*st, type, f->max_ed);
            }
        }
    }
    if (!s.line(p, len)) return -1;
    side.seq.insert(side.seq.end(), (const uint8_t *)p, (const uint8_t *)p + len);
    side.off.push_back(side.seq.size());
    const size_t slen = len;
    if (!s.line(p, len) || len == 0 || p[0] != '+') return -1;
    if (!s.line(p, len) || len != slen) return -1;                  // set_reverse_comp aborts on a length mismatch
    side.qual.insert(side.qual.end(), (const uint8_t *)p, (const uint8_t *)p + len);
    return 1;
}

}  // namespace

struct cm_writer {
    FILE *f1 = nullptr, *f2 = nullptr;
    std::vector<std::string> chr_names;
    std::vector<uint32_t> chr_shift, chr_len;
};

extern "C" {

int cm_fastq_open(const char *r1_path, const char *r2_path, const cm_chr_info *chrs, uint32_t n_chr, int32_t max_ed, cm_fastq **out) {
    if (!r1_path || !r2_path || !out || (n_chr && !chrs)) return CM_EINVAL;
    *out = nullptr;
    cm_fastq *f = new cm_fastq();
    if (!f->s1.open(r1_path) || !f->s2.open(r2_path)) {
        f->s1.close();
        f->s2.close();
        delete f;
        return CM_EINVAL;
    }
    for (uint32_t i = 0; i < n_chr; ++i) f->chr_names.emplace_back(chrs[i].name ? chrs[i].name : "");
    f->max_ed = max_ed;
    *out = f;
    return CM_OK;
}

int cm_fastq_next(cm_fastq *f, uint64_t max_pairs, cm_fastq_batch *out) {
    if (!f || !out) return CM_EINVAL;
    f->cur ^= 1;
    cm_fastq::Gen &G = f->gen[f->cur];
    G.a.clear();
    G.b.clear();
    G.prior.clear();
    f->any_prior = false;
    // The two files are independent streams until the records are paired up: R2 is parsed (and, for .gz input, inflated)
    // on a second thread while this one does R1 and its carried state.  The reference does both inside one lock-protected
    // serial section (src/circminer.cpp:373-379), which is its ingest ceiling.
    uint64_t n = 0, n2 = 0;
    int bad2 = 0;
    std::thread side_b([&]() {
        while (n2 < max_pairs) {
            const int r2 = parse_record(f, f->s2, G.b, false, nullptr, nullptr);
            if (r2 == 0) break;
            if (r2 < 0) {
                bad2 = 1;
                break;
            }
            ++n2;
        }
    });
    int bad1 = 0;
    while (n < max_pairs) {
        cm_mapped_read st;
        bool carried = false;
        const int r1 = parse_record(f, f->s1, G.a, true, &st, &carried);
        if (r1 == 0) break;
        if (r1 < 0) {
            bad1 = 1;
            break;
        }
        G.prior.push_back(st);
        f->any_prior = f->any_prior || carried;
        ++n;
    }
    side_b.join();
    if (bad1 || bad2 || n2 < n) return CM_EINVAL;                   // malformed record, or R2 ends before R1
    if (n2 > n) {        // R1 ended first: like the reference, which stops at R1's end, the surplus R2 records are not paired
        G.b.off.resize(n + 1);
        G.b.seq.resize(G.b.off[n]);
        G.b.qual.resize(G.b.off[n]);
        G.b.name_off.resize(n + 1);
        G.b.names.resize(G.b.name_off[n]);
    }
    memset(out, 0, sizeof *out);
    out->reads.n_pairs = n;
    out->reads.seq1 = G.a.seq.data();
    out->reads.off1 = G.a.off.data();
    out->reads.seq2 = G.b.seq.data();
    out->reads.off2 = G.b.off.data();
    out->qual1 = G.a.qual.data();
    out->qual2 = G.b.qual.data();
    out->names1 = G.a.names.data();
    out->name_off1 = G.a.name_off.data();
    out->names2 = G.b.names.data();
    out->name_off2 = G.b.name_off.data();
    out->prior = f->any_prior ? G.prior.data() : nullptr;
    return CM_OK;
}

void cm_fastq_close(cm_fastq *f) {
    if (!f) return;
    f->s1.close();
    f->s2.close();
    delete f;
}

int cm_writer_open(const char *path1, const char *path2, const cm_chr_info *chrs, uint32_t n_chr, cm_writer **out) {
    if (!path1 || !out || (n_chr && !chrs)) return CM_EINVAL;
    *out = nullptr;
    cm_writer *w = new cm_writer();
    w->f1 = fopen(path1, "wb");
    w->f2 = path2 ? fopen(path2, "wb") : nullptr;
    if (!w->f1 || (path2 && !w->f2)) {
        if (w->f1) fclose(w->f1);
        if (w->f2) fclose(w->f2);
        delete w;
        return CM_EINVAL;
    }
    for (uint32_t i = 0; i < n_chr; ++i) {
        w->chr_names.emplace_back(chrs[i].name ? chrs[i].name : "");
        w->chr_shift.push_back(chrs[i].start_pos);
        w->chr_len.push_back(chrs[i].len);
    }
    *out = w;
    return CM_OK;
}

static const char *chr_name(const cm_writer *w, int id) { return (id >= 0 && (size_t)id < w->chr_names.size()) ? w->chr_names[(size_t)id].c_str() : "-"; }

// write_read_category (PE) for the selected pairs of a batch; sel == NULL selects every pair
int cm_write_remain(cm_writer *w, const cm_fastq_batch *b, const cm_mapped_read *states, const uint64_t *sel, uint64_t n_sel) {
    if (!w || !w->f2 || !b || !states) return CM_EINVAL;
    const uint64_t n = sel ? n_sel : b->reads.n_pairs;
    for (uint64_t k = 0; k < n; ++k) {
        const uint64_t i = sel ? sel[k] : k;
        if (i >= b->reads.n_pairs) return CM_EINVAL;
        const cm_mapped_read &m = states[i];
        FILE *fs[2] = {w->f1, w->f2};
        for (int s = 0; s < 2; ++s) {
            FILE *f = fs[s];
            fprintf(f, "@%s", (s ? b->names2 : b->names1) + (s ? b->name_off2 : b->name_off1)[i]);
            if (mapped_type(m.type)) {
                const char *cn = chr_name(w, m.chr_id);
                const uint32_t shift = (m.chr_id >= 0 && (size_t)m.chr_id < w->chr_shift.size()) ? w->chr_shift[(size_t)m.chr_id] : 0u;
                const uint64_t gspos = (uint64_t)(int64_t)m.contig_num * CM_CONTIG_SIZE + (uint32_t)(m.spos_r1 + shift);   // chrloc2conloc
                fprintf(f, " %" PRId64 " %d %s %u %u %d %u %u %c %d %s %u %u %d %u %u %c %d %d %d %d %d", (int64_t)gspos, m.type, cn, m.spos_r1,
                        m.epos_r1, (int)m.mlen_r1, m.qspos_r1, m.qepos_r1, m.r1_forward ? '+' : '-', m.ed_r1, cn, m.spos_r2, m.epos_r2,
                        (int)m.mlen_r2, m.qspos_r2, m.qepos_r2, m.r2_forward ? '+' : '-', m.ed_r2, m.tlen, (int)m.junc_num,
                        (int)(m.gm_compatible != 0), m.contig_num);
            } else {
                fprintf(f, " * %d * * * * * * * * * * * * * * * * * * * *", m.type);
            }
            const uint8_t *seq = s ? b->reads.seq2 : b->reads.seq1, *q = s ? b->qual2 : b->qual1;
            const uint64_t *off = s ? b->reads.off2 : b->reads.off1;
            const int len = (int)(off[i + 1] - off[i]);
            fprintf(f, "\n%.*s\n+\n%.*s\n", len, (const char *)seq + off[i], len, (const char *)q + off[i]);
        }
    }
    return CM_OK;
}

// write_pam_rec_pe for the selected pairs (names of R1)
int cm_write_pam(cm_writer *w, const cm_fastq_batch *b, const cm_mapped_read *states, const uint64_t *sel, uint64_t n_sel) {
    if (!w || !b || !states) return CM_EINVAL;
    const uint64_t n = sel ? n_sel : b->reads.n_pairs;
    for (uint64_t k = 0; k < n; ++k) {
        const uint64_t i = sel ? sel[k] : k;
        if (i >= b->reads.n_pairs) return CM_EINVAL;
        const cm_mapped_read &m = states[i];
        const char *nm = b->names1 + b->name_off1[i];
        if (mapped_type(m.type)) {
            const char *cn = chr_name(w, m.chr_id);
            fprintf(w->f1, "%s\t%s\t%u\t%u\t%d\t%u\t%u\t%c\t%d\t%s\t%u\t%u\t%d\t%u\t%u\t%c\t%d\t%d\t%d\t%d\t%d\n", nm, cn, m.spos_r1, m.epos_r1,
                    (int)m.mlen_r1, m.qspos_r1, m.qepos_r1, m.r1_forward ? '+' : '-', m.ed_r1, cn, m.spos_r2, m.epos_r2, (int)m.mlen_r2, m.qspos_r2,
                    m.qepos_r2, m.r2_forward ? '+' : '-', m.ed_r2, m.tlen, (int)m.junc_num, (int)(m.gm_compatible != 0), m.type);
        } else {
            fprintf(w->f1, "%s\t*\t*\t*\t*\t*\t*\t*\t*\t*\t*\t*\t*\t*\t*\t*\t*\t*\t*\t*\t*\t*\t%d\n", nm, m.type);
        }
    }
    return CM_OK;
}

// SAMOutput::print_header (src/output.cpp:301-311)
int cm_write_sam_header(cm_writer *w) {
    if (!w) return CM_EINVAL;
    fprintf(w->f1, "@HD\tVN:1.4\tSO:unsorted\n");
    for (size_t i = 0; i < w->chr_names.size(); ++i) fprintf(w->f1, "@SQ\tSN:%s\tLN:%u\n", w->chr_names[i].c_str(), w->chr_len[i]);
    return CM_OK;
}

namespace {
constexpr unsigned PAIRED = 1u << 0, PROPER = 1u << 1, RUNMAP = 1u << 2, MUNMAP = 1u << 3, RREVER = 1u << 4, MREVER = 1u << 5, FIPAIR = 1u << 6,
                   SIPAIR = 1u << 7;
unsigned sam_flag(const cm_mapped_read &m, bool first) {                 // set_flag_pe, src/output.cpp:118-149
    unsigned flag = PAIRED;
    if (m.type == CM_CONCRD) flag |= PROPER;
    if (!(m.type <= CM_CHIORF || m.type == CM_CONGEN || m.type == CM_CONGNM)) flag |= RUNMAP | MUNMAP;
    if (first) {
        if (!(flag & RUNMAP) && !m.r1_forward) flag |= RREVER;
        if (!(flag & MUNMAP) && !m.r2_forward) flag |= MREVER;
        flag |= FIPAIR;
    } else {
        if (!(flag & MUNMAP) && !m.r1_forward) flag |= MREVER;
        if (!(flag & RUNMAP) && !m.r2_forward) flag |= RREVER;
        flag |= SIPAIR;
    }
    return flag;
}
// FASTQParser::set_comp / set_reverse_comp (src/fastq_parser.cpp:141-176): bytes outside ACGTN / acgtn map to NUL, which
// ends the %s the reference prints
char comp_of(unsigned char ch) {
    switch (ch) {
        case 'A': case 'a': return 'T';
        case 'C': case 'c': return 'G';
        case 'G': case 'g': return 'C';
        case 'T': case 't': return 'A';
        case 'N': case 'n': return 'N';
        default: return '\0';
    }
}
}  // namespace

// write_sam_rec_pe for the selected pairs (two records per pair)
int cm_write_sam(cm_writer *w, const cm_fastq_batch *b, const cm_mapped_read *states, const uint64_t *sel, uint64_t n_sel) {
    if (!w || !b || !states) return CM_EINVAL;
    const uint64_t n = sel ? n_sel : b->reads.n_pairs;
    std::string seq[2], qual[2];
    for (uint64_t k = 0; k < n; ++k) {
        const uint64_t i = sel ? sel[k] : k;
        if (i >= b->reads.n_pairs) return CM_EINVAL;
        const cm_mapped_read &m = states[i];
        const char *qname = b->names1 + b->name_off1[i];
        const unsigned flag[2] = {sam_flag(m, true), sam_flag(m, false)};
        const char *cn = chr_name(w, m.chr_id);
        int32_t tlen[2];
        if (m.spos_r1 < m.spos_r2) { tlen[0] = m.tlen; tlen[1] = m.tlen * -1; }
        else { tlen[0] = m.tlen * -1; tlen[1] = m.tlen; }
        const char *rname[2], *rnext[2];
        uint32_t pos[2], pnext[2];
        // set_output_pe, src/output.cpp:151-222 (chr_r1 == chr_r2 always, so rnext of a mapped mate is "=")
        if (flag[0] & RUNMAP) { rname[0] = "*"; rnext[1] = "*"; pos[0] = 0; pnext[1] = 0; tlen[0] = tlen[1] = 0; }
        else { rname[0] = cn; rnext[1] = "="; pos[0] = m.spos_r1; pnext[1] = m.spos_r1; }
        if (flag[1] & RUNMAP) { rname[1] = "*"; rnext[0] = "*"; pos[1] = 0; pnext[0] = 0; tlen[0] = tlen[1] = 0; }
        else { rname[1] = cn; rnext[0] = "="; pos[1] = m.spos_r2; pnext[0] = m.spos_r2; }
        for (int s = 0; s < 2; ++s) {
            const uint8_t *sq = s ? b->reads.seq2 : b->reads.seq1, *ql = s ? b->qual2 : b->qual1;
            const uint64_t *off = s ? b->reads.off2 : b->reads.off1;
            const size_t len = (size_t)(off[i + 1] - off[i]);
            seq[s].assign((const char *)sq + off[i], len);
            qual[s].assign((const char *)ql + off[i], len);
            if (flag[s] & RREVER) {
                std::string rc(len, '\0'), rq(len, '\0');
                for (size_t x = 0; x < len; ++x) {
                    rc[x] = comp_of((unsigned char)seq[s][len - 1 - x]);
                    rq[x] = qual[s][len - 1 - x];
                }
                seq[s] = rc.substr(0, rc.find('\0'));             // %s stops at the first NUL
                qual[s] = rq;
            }
            const bool un = (flag[s] & RUNMAP) != 0;
            const int ed = un ? 0 : (s ? m.ed_r2 : m.ed_r1), jc = un ? 0 : (int)m.junc_num, gm = un ? 0 : (int)(m.gm_compatible != 0);
            fprintf(w->f1, "%s\t%u\t%s\t%u\t%u\t%s\t%s\t%u\t%u\t%s\t%s\tAT:i:%d\tNM:i:%d\tJC:i:%d\tTC:i:%d\n", qname, flag[s], rname[s], pos[s], 255u, "*",
                    rnext[s], pnext[s], (unsigned)tlen[s], seq[s].c_str(), qual[s].c_str(), m.type, ed, jc, gm);
        }
    }
    return CM_OK;
}

void cm_writer_close(cm_writer *w) {
    if (!w) return;
    if (w->f1) fclose(w->f1);
    if (w->f2) fclose(w->f2);
    delete w;
}

}  // extern "C"
