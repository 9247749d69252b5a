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
    FILE *plain = nullptr;                   // set for uncompressed input: read in big blocks and tokenised on several threads
    std::vector<char> buf;
    size_t pos = 0, end = 0;
    bool eof = false;
    bool open(const char *path) {
        FILE *t = fopen(path, "rb");
        if (!t) return false;
        unsigned char magic[2] = {0, 0};
        const size_t got = fread(magic, 1, 2, t);
        if (got == 2 && magic[0] == 0x1f && magic[1] == 0x8b) {          // gzip: one inflate stream, the record-by-record path
            fclose(t);
            gz = gzopen(path, "r");
            if (!gz) return false;
            gzbuffer(gz, 1u << 20);
        } else {
            rewind(t);
            plain = t;
        }
        buf.resize(BLOCK);
        return true;
    }
    void close() {
        if (gz) gzclose(gz);
        if (plain) fclose(plain);
        gz = nullptr;
        plain = nullptr;
    }
    int read_some(char *dst, size_t cap) {
        if (plain) return (int)fread(dst, 1, cap > (1u << 30) ? (1u << 30) : cap, plain);
        return gzread(gz, dst, (unsigned)(cap > (1u << 30) ? (1u << 30) : cap));
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
            const int got = read_some(buf.data() + end, buf.size() - end);
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
    // three generations of batch storage, used in turn: the batch a call returns stays valid over the next two calls, so a
    // caller can have batch k-1 with its writer thread, batch k on the GPU and batch k+1 in the parser at the same time
    struct Gen {
        Side a, b;
        std::vector<cm_mapped_read> prior;
    } gen[3];
    int cur = 2;
    int n_threads = 0;                       // tokeniser threads of the plain-text path (0 = hardware concurrency, at most 32)
    std::vector<size_t> nl1, nl2;            // newline index of the two block buffers (plain-text path)
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

// header line "@tok0 tok1 ...": tokens separated by runs of spaces (strtok); at most FQCOMMENTCNT + 1 are kept, all are counted
int split_header(const char *p, size_t len, const char **tok, size_t *tl) {
    int nt = 0;
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
    return nt;
}
size_t name_len(int nt, const char *const *tok, const size_t *tl) {
    size_t nlen = nt ? tl[0] : 0;
    if (nlen >= 2 && tok[0][nlen - 2] == '/') nlen -= 2;          // extract_map_info :193-194
    return nlen;
}
// fill_map_info (src/fastq_parser.cpp:200-269): the MatchedRead a 23-token header carries, the fresh-read state otherwise
void state_from_header(const cm_fastq *f, int nt, const char *const *tok, const size_t *tl, cm_mapped_read &out) {
    cm_mapped_read *st = &out;
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

// one record of one stream; fills `st` (state carried in the header) when want_state
// returns 1 = record, 0 = end of input, -1 = format error
int parse_record(cm_fastq *f, Stream &s, Side &side, bool want_state, cm_mapped_read *st, bool *carried) {
    const char *p;
    size_t len;
    if (!s.line(p, len)) return 0;
    if (len == 0 && s.eof && s.pos >= s.end) return 0;
    if (len == 0 || p[0] != '@') return -1;                       // has_next asserts the '@'
    const char *tok[FQCOMMENTCNT + 1];
    size_t tl[FQCOMMENTCNT + 1];
    const int nt = split_header(p, len, tok, tl);
    size_t nlen = name_len(nt, tok, tl);
    if (nt) side.names.insert(side.names.end(), tok[0], tok[0] + nlen);
    side.names.push_back('\0');
    side.name_off.push_back(side.names.size());
    if (want_state) {
        *carried = nt == FQCOMMENTCNT;
        state_from_header(f, nt, tok, tl, *st);
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
// Append-only text buffer in front of a FILE: the writers format integers themselves (a PAM row is 20 of them), which is
// several times faster than one fprintf per record.
struct Out {
    FILE *f = nullptr;
    std::vector<char> buf;
    size_t n = 0;
    void open(FILE *fp) {
        f = fp;
        buf.resize(4u << 20);
        n = 0;
    }
    void flush() {
        if (f && n) fwrite(buf.data(), 1, n, f);
        n = 0;
    }
    void room(size_t k) {
        if (n + k > buf.size()) {
            flush();
            if (k > buf.size()) buf.resize(k);
        }
    }
    void raw(const void *p, size_t k) {
        room(k);
        memcpy(buf.data() + n, p, k);
        n += k;
    }
    void str(const char *z) { raw(z, strlen(z)); }
    void ch(char c) {
        room(1);
        buf[n++] = c;
    }
    void u64(uint64_t v) {
        char t[24];
        int k = 0;
        do {
            t[k++] = (char)('0' + v % 10);
            v /= 10;
        } while (v);
        room((size_t)k);
        while (k) buf[n++] = t[--k];
    }
    void i64(int64_t v) {
        if (v < 0) {
            ch('-');
            u64(0 - (uint64_t)v);
        } else {
            u64((uint64_t)v);
        }
    }
};

struct cm_writer {
    FILE *f1 = nullptr, *f2 = nullptr;
    Out o1, o2;
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
    f->cur = (f->cur + 1) % 3;
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
    w->o1.open(w->f1);
    if (w->f2) w->o2.open(w->f2);
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
        Out *os[2] = {&w->o1, &w->o2};
        for (int s = 0; s < 2; ++s) {
            Out &o = *os[s];
            o.ch('@');
            o.str((s ? b->names2 : b->names1) + (s ? b->name_off2 : b->name_off1)[i]);
            if (mapped_type(m.type)) {
                const char *cn = chr_name(w, m.chr_id);
                const uint32_t shift = (m.chr_id >= 0 && (size_t)m.chr_id < w->chr_shift.size()) ? w->chr_shift[(size_t)m.chr_id] : 0u;
                const uint64_t gspos = (uint64_t)(int64_t)m.contig_num * CM_CONTIG_SIZE + (uint32_t)(m.spos_r1 + shift);   // chrloc2conloc
                // " %PRId64 %d %s %u %u %d %u %u %c %d %s %u %u %d %u %u %c %d %d %d %d %d"
                o.ch(' '); o.i64((int64_t)gspos);
                o.ch(' '); o.i64(m.type);
                o.ch(' '); o.str(cn);
                o.ch(' '); o.u64(m.spos_r1);
                o.ch(' '); o.u64(m.epos_r1);
                o.ch(' '); o.i64((int)m.mlen_r1);
                o.ch(' '); o.u64(m.qspos_r1);
                o.ch(' '); o.u64(m.qepos_r1);
                o.ch(' '); o.ch(m.r1_forward ? '+' : '-');
                o.ch(' '); o.i64(m.ed_r1);
                o.ch(' '); o.str(cn);
                o.ch(' '); o.u64(m.spos_r2);
                o.ch(' '); o.u64(m.epos_r2);
                o.ch(' '); o.i64((int)m.mlen_r2);
                o.ch(' '); o.u64(m.qspos_r2);
                o.ch(' '); o.u64(m.qepos_r2);
                o.ch(' '); o.ch(m.r2_forward ? '+' : '-');
                o.ch(' '); o.i64(m.ed_r2);
                o.ch(' '); o.i64(m.tlen);
                o.ch(' '); o.i64((int)m.junc_num);
                o.ch(' '); o.i64((int)(m.gm_compatible != 0));
                o.ch(' '); o.i64(m.contig_num);
            } else {
                o.str(" * ");
                o.i64(m.type);
                o.str(" * * * * * * * * * * * * * * * * * * * *");
            }
            const uint8_t *seq = s ? b->reads.seq2 : b->reads.seq1, *q = s ? b->qual2 : b->qual1;
            const uint64_t *off = s ? b->reads.off2 : b->reads.off1;
            const size_t len = (size_t)(off[i + 1] - off[i]);
            o.ch('\n');
            o.raw(seq + off[i], len);
            o.str("\n+\n");
            o.raw(q + off[i], len);
            o.ch('\n');
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
        Out &o = w->o1;
        o.str(nm);
        if (mapped_type(m.type)) {
            const char *cn = chr_name(w, m.chr_id);
            // "%s\t%s\t%u\t%u\t%d\t%u\t%u\t%c\t%d\t%s\t%u\t%u\t%d\t%u\t%u\t%c\t%d\t%d\t%d\t%d\t%d\n"
            o.ch('\t'); o.str(cn);
            o.ch('\t'); o.u64(m.spos_r1);
            o.ch('\t'); o.u64(m.epos_r1);
            o.ch('\t'); o.i64((int)m.mlen_r1);
            o.ch('\t'); o.u64(m.qspos_r1);
            o.ch('\t'); o.u64(m.qepos_r1);
            o.ch('\t'); o.ch(m.r1_forward ? '+' : '-');
            o.ch('\t'); o.i64(m.ed_r1);
            o.ch('\t'); o.str(cn);
            o.ch('\t'); o.u64(m.spos_r2);
            o.ch('\t'); o.u64(m.epos_r2);
            o.ch('\t'); o.i64((int)m.mlen_r2);
            o.ch('\t'); o.u64(m.qspos_r2);
            o.ch('\t'); o.u64(m.qepos_r2);
            o.ch('\t'); o.ch(m.r2_forward ? '+' : '-');
            o.ch('\t'); o.i64(m.ed_r2);
            o.ch('\t'); o.i64(m.tlen);
            o.ch('\t'); o.i64((int)m.junc_num);
            o.ch('\t'); o.i64((int)(m.gm_compatible != 0));
            o.ch('\t'); o.i64(m.type);
            o.ch('\n');
        } else {
            o.str("\t*\t*\t*\t*\t*\t*\t*\t*\t*\t*\t*\t*\t*\t*\t*\t*\t*\t*\t*\t*\t*\t");
            o.i64(m.type);
            o.ch('\n');
        }
    }
    return CM_OK;
}

// SAMOutput::print_header (src/output.cpp:301-311)
int cm_write_sam_header(cm_writer *w) {
    if (!w) return CM_EINVAL;
    Out &o = w->o1;
    o.str("@HD\tVN:1.4\tSO:unsorted\n");
    for (size_t i = 0; i < w->chr_names.size(); ++i) {
        o.str("@SQ\tSN:");
        o.str(w->chr_names[i].c_str());
        o.str("\tLN:");
        o.u64(w->chr_len[i]);
        o.ch('\n');
    }
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
struct CompTable {
    char t[256];
    CompTable() {
        memset(t, 0, sizeof t);
        const char *from = "ACGTNacgtn", *to = "TGCANTGCAN";
        for (int i = 0; from[i]; ++i) t[(unsigned char)from[i]] = to[i];
    }
};
inline char comp_of(unsigned char ch) {
    static const CompTable c;
    return c.t[ch];
}
}  // namespace

// write_sam_rec_pe for the selected pairs (two records per pair)
int cm_write_sam(cm_writer *w, const cm_fastq_batch *b, const cm_mapped_read *states, const uint64_t *sel, uint64_t n_sel) {
    if (!w || !b || !states) return CM_EINVAL;
    const uint64_t n = sel ? n_sel : b->reads.n_pairs;
    std::vector<char> rc, rq;
    Out &o = w->o1;
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
            const uint8_t *sq = (s ? b->reads.seq2 : b->reads.seq1), *ql = (s ? b->qual2 : b->qual1);
            const uint64_t *off = s ? b->reads.off2 : b->reads.off1;
            const size_t len = (size_t)(off[i + 1] - off[i]);
            sq += off[i];
            ql += off[i];
            // "%s\t%u\t%s\t%u\t%u\t%s\t%s\t%u\t%u\t%s\t%s" + "\tAT:i:%d\tNM:i:%d\tJC:i:%d\tTC:i:%d" + "\n"
            o.str(qname);
            o.ch('\t'); o.u64(flag[s]);
            o.ch('\t'); o.str(rname[s]);
            o.ch('\t'); o.u64(pos[s]);
            o.str("\t255\t*\t");
            o.str(rnext[s]);
            o.ch('\t'); o.u64(pnext[s]);
            o.ch('\t'); o.u64((uint32_t)tlen[s]);                   // the reference prints the int32 through %u
            o.ch('\t');
            if (flag[s] & RREVER) {
                rc.resize(len);
                rq.resize(len);
                size_t rc_len = len;                                // %s stops at the first NUL of rcseq
                for (size_t x = 0; x < len; ++x) {
                    rc[x] = comp_of(sq[len - 1 - x]);
                    rq[x] = (char)ql[len - 1 - x];
                    if (rc[x] == '\0' && rc_len == len) rc_len = x;
                }
                o.raw(rc.data(), rc_len);
                o.ch('\t');
                o.raw(rq.data(), len);
            } else {
                o.raw(sq, len);
                o.ch('\t');
                o.raw(ql, len);
            }
            const bool un = (flag[s] & RUNMAP) != 0;
            o.str("\tAT:i:"); o.i64(m.type);
            o.str("\tNM:i:"); o.i64(un ? 0 : (s ? m.ed_r2 : m.ed_r1));
            o.str("\tJC:i:"); o.i64(un ? 0 : (int)m.junc_num);
            o.str("\tTC:i:"); o.i64(un ? 0 : (int)(m.gm_compatible != 0));
            o.ch('\n');
        }
    }
    return CM_OK;
}

void cm_writer_close(cm_writer *w) {
    if (!w) return;
    w->o1.flush();
    w->o2.flush();
    if (w->f1) fclose(w->f1);
    if (w->f2) fclose(w->f2);
    delete w;
}

}  // extern "C"
