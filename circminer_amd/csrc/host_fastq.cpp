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
#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>
#include <sys/stat.h>
#include <zlib.h>
#if defined(__x86_64__)
#include <immintrin.h>
#endif

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cerrno>
#include <cinttypes>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
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
    bool io_error = false;                   // a read failed (not: reached the end): the batch in flight is CM_EIO, not a short file
    bool open(const char *path) {
        const int fd = ::open(path, O_RDONLY | O_CLOEXEC);
        if (fd < 0) return false;
        struct stat sb;
        unsigned char magic[2] = {0, 0};
        // Only a regular file can be sniffed and read with pread(); a FIFO, a process substitution or /dev/stdin goes to zlib on
        // the same descriptor, which reads plain and gzip streams alike (what the reference's gzopen / gzread does for every
        // input, src/fastq_parser.cpp:60,85).
        const bool regular = fstat(fd, &sb) == 0 && S_ISREG(sb.st_mode);
        const bool gzip = regular && pread(fd, magic, 2, 0) == 2 && magic[0] == 0x1f && magic[1] == 0x8b;
        if (!regular || gzip) {                                           // one (inflate) stream, the record-by-record path
            gz = gzdopen(fd, "r");
            if (!gz) {
                ::close(fd);
                return false;
            }
            gzbuffer(gz, 1u << 20);
        } else {
            plain = fdopen(fd, "rb");
            if (!plain) {
                ::close(fd);
                return false;
            }
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
    uint64_t file_pos = 0;                   // plain: offset of the next unread byte
    uint64_t file_end = ~0ull;               // plain: this reader's share of the file ends here (cm_fastq_open_shard)
    int read_threads = 1;
    int read_some(char *dst, size_t cap) {
        if (cap > (1u << 30)) cap = 1u << 30;
        if (!plain) {
            const int r = gzread(gz, dst, (unsigned)cap);
            if (r < 0) io_error = true;
            return r;
        }
        // page-cache -> buffer copies are what a read() of a hot file is: several pread()s side by side
        const int fd = fileno(plain);
        if (file_pos >= file_end) return 0;
        if ((uint64_t)cap > file_end - file_pos) cap = (size_t)(file_end - file_pos);
        const int nt = (read_threads > 1 && cap >= (8u << 20)) ? read_threads : 1;
        std::vector<long> got((size_t)nt, 0);
        std::vector<char> failed((size_t)nt, 0);
        auto piece = [&](int t) {
            const size_t a = cap * (size_t)t / (size_t)nt, b = cap * (size_t)(t + 1) / (size_t)nt;
            size_t done = 0;
            while (a + done < b) {
                const ssize_t r = pread(fd, dst + a + done, b - a - done, (off_t)(file_pos + a + done));
                if (r < 0) {                       // an I/O error is not the end of the file
                    if (errno == EINTR) continue;
                    failed[(size_t)t] = 1;
                    break;
                }
                if (r == 0) break;
                done += (size_t)r;
            }
            got[(size_t)t] = (long)done;
        };
        if (nt == 1) piece(0);
        else {
            std::vector<std::thread> th;
            for (int t = 0; t < nt; ++t) th.emplace_back(piece, t);
            for (auto &x : th) x.join();
        }
        for (int t = 0; t < nt; ++t)
            if (failed[(size_t)t]) {
                io_error = true;
                return -1;
            }
        size_t total = 0;
        for (int t = 0; t < nt; ++t) {              // contiguous prefix that was actually read (a short piece means end of file)
            const size_t want = cap * (size_t)(t + 1) / (size_t)nt - cap * (size_t)t / (size_t)nt;
            total += (size_t)got[(size_t)t];
            if ((size_t)got[(size_t)t] < want) break;
        }
        file_pos += total;
        return (int)total;
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

// The part of std::vector the parser uses, without the zero-fill of resize(): the batch arrays are tens of MB and every byte
// is overwritten by the (parallel) copies right after they are sized.
// Batch arrays of hundreds of MB are filled once per batch: as transparent huge pages their first fill costs 512 times fewer
// page faults (the parser's first batches ran at a third of its steady rate).
inline void advise_huge(void *p, size_t bytes) {
#if defined(MADV_HUGEPAGE)
    if (!p || bytes < (8u << 20)) return;
    const uintptr_t a = ((uintptr_t)p + 4095) & ~(uintptr_t)4095, e = ((uintptr_t)p + bytes) & ~(uintptr_t)4095;
    if (e > a) (void)madvise((void *)a, (size_t)(e - a), MADV_HUGEPAGE);
#else
    (void)p;
    (void)bytes;
#endif
}
// Called before a block that cm_fastq_next has handed out is freed or moved (see cm_fastq_set_release_hook).
struct ReleaseHook {
    void (*fn)(void *user, const void *ptr, uint64_t bytes) = nullptr;
    void *user = nullptr;
};
std::atomic<int> g_raw_oom{0};       // a RawVec could not grow inside a worker thread (reported by the cm_fastq_* call as CM_ENOMEM)
template <class T> struct RawVec {
    T *p = nullptr;
    size_t n = 0, cap = 0;
    const ReleaseHook *hook = nullptr;      // set for the arrays a caller may have page-locked
    RawVec() = default;
    RawVec(const RawVec &) = delete;
    RawVec &operator=(const RawVec &) = delete;
    ~RawVec() { release(); }
    void release() {
        if (p && hook && hook->fn) hook->fn(hook->user, p, (uint64_t)(cap * sizeof(T)));
        free(p);
        p = nullptr;
        n = cap = 0;
    }
    T *data() { return p; }
    const T *data() const { return p; }
    size_t size() const { return n; }
    T *end() { return p + n; }
    T &operator[](size_t i) { return p[i]; }
    const T &operator[](size_t i) const { return p[i]; }
    void clear() { n = 0; }
    // A block never moves behind its owner's back: growth allocates a new block, tells the hook that the old one is going
    // (a caller that registered it with the GPU runtime unregisters it there) and only then frees it.  Out of memory: the old
    // block stays as it is and std::bad_alloc is thrown (caught at the C boundary -> CM_ENOMEM).
    void reserve(size_t c) {
        if (c <= cap) return;
        size_t nc = cap ? cap : 1024;
        while (nc < c) nc *= 2;
        T *q = (T *)malloc(nc * sizeof(T));
        if (!q) throw std::bad_alloc();
        if (n) memcpy(q, p, n * sizeof(T));
        if (p && hook && hook->fn) hook->fn(hook->user, p, (uint64_t)(cap * sizeof(T)));
        free(p);
        p = q;
        cap = nc;
        advise_huge(p, nc * sizeof(T));
    }
    void resize(size_t k) {                 // contents of new elements are unspecified
        reserve(k);
        n = k;
    }
    void push_back(const T &v) {
        reserve(n + 1);
        p[n++] = v;
    }
    template <class It> void insert(T *, It a, It b) {       // append only (the parser never inserts elsewhere)
        const size_t k = (size_t)(b - a);
        reserve(n + k);
        memcpy(p + n, &*a, k * sizeof(T));
        n += k;
    }
    void assign(size_t k, const T &v) {
        resize(k);
        for (size_t i = 0; i < k; ++i) p[i] = v;
    }
};
// body of a worker thread: an allocation failure inside it must not terminate the process
template <class F> void guarded(F &&f) {
    try {
        f();
    } catch (const std::bad_alloc &) {
        g_raw_oom = 1;
    }
}

struct Side {
    RawVec<uint8_t> seq, qual;
    RawVec<uint64_t> off;
    RawVec<char> names;
    RawVec<uint64_t> name_off;
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
    ReleaseHook hook;                        // cm_fastq_set_release_hook (declared first: the batch arrays below call it when they go)
    Stream s1, s2;
    // four generations of batch storage, used in turn: the batch a call returns stays valid over the next three calls, so a
    // caller can have batch k-1 with its writer thread, batch k on the GPU, batch k+1 staged (its H2D copy in flight) and
    // batch k+2 in the parser at the same time
    struct Gen {
        Side a, b;
        RawVec<cm_mapped_read> prior;
    } gen[4];
    int cur = 3;
    int n_threads = 0;                       // tokeniser threads of the plain-text path (0 = hardware concurrency, at most 16; CM_FASTQ_THREADS overrides, at most 32)
    std::vector<size_t> nl1, nl2;            // newline index of the two block buffers (plain-text path)
    std::vector<std::string> chr_names;
    int max_ed = 4;
    bool any_prior = false;
    std::string err;
    // cm_fastq_open_shard on input that cannot be cut at a byte offset (gzip, pipes): this reader inflates the stream from its start,
    // steps over the first skip_left records and hands out take_left (~0: to the end of the input)
    uint64_t skip_left = 0, take_left = ~0ull;
    cm_fastq() {
        for (Gen &g : gen) {                 // the arrays cm_fastq_batch::reads / prior point into
            g.a.seq.hook = g.a.off.hook = g.b.seq.hook = g.b.off.hook = &hook;
            g.prior.hook = &hook;
        }
    }
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
    if (!s.line(p, len)) return s.io_error ? -1 : 0;
    if (len == 0 || p[0] != '@') return -1;                       // has_next asserts the '@' (fastq_parser.h:64-65): blank lines at the end are malformed too
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

namespace {

// ---------------------------------------------------------------------------------------------------------------------
// Plain-text fast path of cm_fastq_next: the reference tokenises FASTQ inside one lock-protected serial section
// (src/circminer.cpp:373-379); here both files are read in large blocks and everything after the read() is data parallel:
//   1. newline index of the block (each thread scans a slice with memchr, slices concatenated in order);
//   2. four lines = one record; per record the lengths of name / sequence and the header checks (pass A);
//   3. prefix sums give every record its place in the batch arrays; bytes are copied in parallel (pass B).
// Records that do not fit the block stay in the buffer for the next call.  Same results as parse_record().
// offsets (from base) of the line feeds in base[lo, hi), appended to v.  One memchr call per line costs more than the scan itself
// on 150-byte lines (2 GB/s per thread); with AVX2 the block is compared 64 bytes at a time and the set bits are read off the mask.
#if defined(__x86_64__)
__attribute__((target("avx2"))) void scan_newlines_avx2(const char *base, size_t lo, size_t hi, RawVec<size_t> &v) {
    size_t n = v.size(), i = lo;
    v.resize(n + (hi - lo) / 32 + 1024);
    const __m256i lf = _mm256_set1_epi8('\n');
    for (; i + 64 <= hi; i += 64) {
        const __m256i a = _mm256_loadu_si256((const __m256i *)(base + i)), b = _mm256_loadu_si256((const __m256i *)(base + i + 32));
        uint64_t m = (uint64_t)(uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(a, lf)) |
                     ((uint64_t)(uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(b, lf)) << 32);
        if (n + 64 > v.size()) v.resize(v.size() * 2 + 64);
        while (m) {
            v[n++] = i + (size_t)__builtin_ctzll(m);
            m &= m - 1;
        }
    }
    for (; i < hi; ++i)
        if (base[i] == '\n') {
            if (n + 1 > v.size()) v.resize(v.size() * 2 + 64);
            v[n++] = i;
        }
    v.resize(n);
}
#endif
void scan_newlines(const char *base, size_t lo, size_t hi, RawVec<size_t> &v) {
#if defined(__x86_64__)
    static const bool avx2 = __builtin_cpu_supports("avx2") && !getenv("CM_FASTQ_NO_AVX2");
    if (avx2) {
        scan_newlines_avx2(base, lo, hi, v);
        return;
    }
#endif
    for (const char *q = base + lo, *e = base + hi; q < e;) {
        const char *z = (const char *)memchr(q, '\n', (size_t)(e - q));
        if (!z) break;
        v.push_back((size_t)(z - base));
        q = z + 1;
    }
}

template <class F> void par_for(int nt, size_t n, F f) {       // f(thread, begin, end) over [0, n) in nt contiguous pieces
    if (nt <= 1 || n < 4096) {
        f(0, (size_t)0, n);
        return;
    }
    std::vector<std::thread> th;
    for (int t = 0; t < nt; ++t)
        th.emplace_back([&f, t, n, nt]() { guarded([&]() { f(t, n * (size_t)t / (size_t)nt, n * (size_t)(t + 1) / (size_t)nt); }); });
    for (auto &x : th) x.join();
}

// makes sure s.buf[s.pos .. s.end) holds at least `want` complete records (or everything up to end of input); nl = offsets
// (relative to s.pos) of the line ends of those bytes.  Returns the number of complete records available.
size_t fill_and_index(Stream &s, std::vector<size_t> &nl, size_t want, size_t bytes_hint, int nt, size_t *tail_lines) {
    s.read_threads = nt;
    if (s.pos > 0) {                                              // drop what the previous batch consumed
        memmove(s.buf.data(), s.buf.data() + s.pos, s.end - s.pos);
        s.end -= s.pos;
        s.pos = 0;
    }
    nl.clear();
    size_t scanned = 0;
    static const bool trace = getenv("CM_FASTQ_TRACE") != nullptr;
    double t_idx = 0, t_cat = 0, t_read = 0;
    auto now = []() { return std::chrono::steady_clock::now(); };
    auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    const auto t_in = now();
    for (;;) {
        // index the bytes not scanned yet
        const size_t a = scanned, b = s.end;
        if (b > a) {
            const auto t0 = now();
            std::vector<RawVec<size_t>> part((size_t)std::max(nt, 1));
            par_for(nt, b - a, [&](int t, size_t lo, size_t hi) { scan_newlines(s.buf.data(), a + lo, a + hi, part[(size_t)t]); });
            const auto t1 = now();
            {   // the slices' lists one behind the other, copied side by side
                std::vector<size_t> at(part.size() + 1, nl.size());
                for (size_t t = 0; t < part.size(); ++t) at[t + 1] = at[t] + part[t].size();
                nl.resize(at.back());
                auto put = [&](size_t t) {
                    if (part[t].size()) memcpy(nl.data() + at[t], part[t].data(), part[t].size() * sizeof(size_t));
                };
                if (part.size() > 1 && at.back() - at[0] > (1u << 16)) {
                    std::vector<std::thread> th;
                    for (size_t t = 0; t < part.size(); ++t) th.emplace_back(put, t);
                    for (auto &x : th) x.join();
                } else
                    for (size_t t = 0; t < part.size(); ++t) put(t);
            }
            scanned = b;
            t_idx += ms(t0, t1);
            t_cat += ms(t1, now());
        }
        size_t lines = nl.size();
        if (s.eof && s.end > 0 && (nl.empty() || nl.back() != s.end - 1)) ++lines;     // last line without a newline
        if (tail_lines) *tail_lines = s.eof ? lines % 4 : 0;                           // lines behind the last whole record of the input
        if (lines / 4 >= want || s.eof) {
            if (trace) fprintf(stderr, "[fastq]   fill: move %.1f read %.1f index %.1f concat %.1f ms\n", ms(t_in, now()) - t_idx - t_cat - t_read, t_read, t_idx, t_cat);
            return lines / 4;
        }
        // more input: room for the rest of the estimate (at least one block)
        size_t need = std::max<size_t>(BLOCK, bytes_hint > s.end ? bytes_hint - s.end : BLOCK);
        if (s.end + need > s.buf.size()) {
            s.buf.resize(s.end + need);
            advise_huge(s.buf.data(), s.buf.size());
        }
        const auto t_r = now();
        while (need > 0) {
            const int got = s.read_some(s.buf.data() + s.end, need);
            if (got <= 0) {
                s.eof = true;
                break;
            }
            s.end += (size_t)got;
            need -= (size_t)got;
        }
        t_read += ms(t_r, now());
    }
}

// one side of a batch from the indexed block; returns false on a malformed record
bool build_side(cm_fastq *f, Stream &s, const std::vector<size_t> &nl, size_t n, Side &side, RawVec<cm_mapped_read> *prior, bool *any_prior,
                int nt) {
    const char *base = s.buf.data();
    auto line_of = [&](size_t k, const char *&p, size_t &len) {      // k-th line of the block
        const size_t b = k ? nl[k - 1] + 1 : 0;
        const size_t e = k < nl.size() ? nl[k] : s.end;               // the last line may have no newline
        p = base + b;
        len = e - b;
    };
    static const bool trace = getenv("CM_FASTQ_TRACE") != nullptr;
    const auto tb0 = std::chrono::steady_clock::now();
    RawVec<uint32_t> nlen, slen;
    nlen.resize(n);
    slen.resize(n);
    std::vector<uint8_t> bad((size_t)std::max(nt, 1), 0), carried((size_t)std::max(nt, 1), 0);
    if (prior) prior->resize(n);
    par_for(nt, n, [&](int t, size_t lo, size_t hi) {                 // pass A
        const char *tok[FQCOMMENTCNT + 1];
        size_t tl[FQCOMMENTCNT + 1];
        for (size_t i = lo; i < hi; ++i) {
            const char *p;
            size_t len, l2, l3, l4;
            const char *p2, *p3, *p4;
            line_of(4 * i, p, len);
            line_of(4 * i + 1, p2, l2);
            line_of(4 * i + 2, p3, l3);
            line_of(4 * i + 3, p4, l4);
            if (len == 0 || p[0] != '@' || l3 == 0 || p3[0] != '+' || l4 != l2) {
                bad[(size_t)t] = 1;
                return;
            }
            const int ntok = split_header(p, len, tok, tl);
            nlen[i] = (uint32_t)name_len(ntok, tok, tl);
            slen[i] = (uint32_t)l2;
            if (prior) {
                if (ntok == FQCOMMENTCNT) carried[(size_t)t] = 1;
                state_from_header(f, ntok, tok, tl, (*prior)[i]);
            }
        }
    });
    const auto tb1 = std::chrono::steady_clock::now();
    for (uint8_t b : bad) if (b) return false;
    if (any_prior) for (uint8_t c : carried) *any_prior = *any_prior || c;
    side.off.resize(n + 1);
    side.name_off.resize(n + 1);
    side.off[0] = 0;
    side.name_off[0] = 0;
    for (size_t i = 0; i < n; ++i) {
        side.off[i + 1] = side.off[i] + slen[i];
        side.name_off[i + 1] = side.name_off[i] + nlen[i] + 1;
    }
    side.seq.resize(side.off[n]);
    side.qual.resize(side.off[n]);
    side.names.resize(side.name_off[n]);
    const auto tb2 = std::chrono::steady_clock::now();
    par_for(nt, n, [&](int, size_t lo, size_t hi) {                   // pass B
        const char *tok[FQCOMMENTCNT + 1];
        size_t tl[FQCOMMENTCNT + 1];
        for (size_t i = lo; i < hi; ++i) {
            const char *p;
            size_t len;
            line_of(4 * i, p, len);
            const int ntok = split_header(p, len, tok, tl);
            char *nm = side.names.data() + side.name_off[i];
            if (ntok) memcpy(nm, tok[0], nlen[i]);
            nm[nlen[i]] = '\0';
            line_of(4 * i + 1, p, len);
            memcpy(side.seq.data() + side.off[i], p, len);
            line_of(4 * i + 3, p, len);
            memcpy(side.qual.data() + side.off[i], p, len);
        }
    });
    if (trace) {
        const auto tb3 = std::chrono::steady_clock::now();
        auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
        fprintf(stderr, "[fastq]   build: pass A %.1f, offsets + sizing %.1f, pass B %.1f ms\n", ms(tb0, tb1), ms(tb1, tb2), ms(tb2, tb3));
    }
    // consumed: everything up to the end of record n - 1
    const size_t last = 4 * n - 1;
    s.pos = last < nl.size() ? nl[last] + 1 : s.end;
    return true;
}

// cm_fastq_next for two plain-text files; *n_out pairs
int next_plain(cm_fastq *f, uint64_t max_pairs, cm_fastq::Gen &G, uint64_t *n_out, uint64_t *n2_out) {
    int nt = f->n_threads > 0 ? f->n_threads : (int)std::thread::hardware_concurrency();
    // more threads than this only fight over the page-cache copies and the writer's bandwidth: file to file on a 256-CPU box,
    // 32 M chr21-like pairs: 10.1 / 14.4 / 13.1 / 11.7 / 9.0 M pairs/s with 8 / 16 / 24 / 32 / 64 threads (tests/diag/e2e_reports.py)
    if (nt > 16) nt = 16;
    if (const char *e = getenv("CM_FASTQ_THREADS")) nt = atoi(e);
    nt = nt < 1 ? 1 : (nt > 32 ? 32 : nt);
    const int half = nt > 1 ? nt / 2 : 1;
    // ~ 2 x (read length + name) + 8 per record; the estimate only sizes the first read(), more is read on demand
    size_t hint = (size_t)std::min<uint64_t>(max_pairs, 1ull << 22) * 360 + (1u << 20);
    {   // never more than what is left of the larger file (a caller asking for "everything" must not cost a 1.5-GB buffer)
        struct stat sa, sb;
        if (f->s1.plain && f->s2.plain && fstat(fileno(f->s1.plain), &sa) == 0 && fstat(fileno(f->s2.plain), &sb) == 0) {
            const uint64_t left1 = (uint64_t)sa.st_size > f->s1.file_pos ? (uint64_t)sa.st_size - f->s1.file_pos : 0;
            const uint64_t left2 = (uint64_t)sb.st_size > f->s2.file_pos ? (uint64_t)sb.st_size - f->s2.file_pos : 0;
            const uint64_t left = std::max(left1, left2) + (1u << 16);
            if (left < hint) hint = (size_t)left;
        }
    }
    size_t a1 = 0, a2 = 0, tail1 = 0, tail2 = 0;
    const bool trace = getenv("CM_FASTQ_TRACE") != nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    {   // the two files are filled and indexed side by side
        std::thread t2([&]() { guarded([&]() { a2 = fill_and_index(f->s2, f->nl2, (size_t)max_pairs, hint, half, &tail2); }); });
        try {
            a1 = fill_and_index(f->s1, f->nl1, (size_t)max_pairs, hint, half, &tail1);
        } catch (const std::bad_alloc &) {
            g_raw_oom = 1;
        }
        t2.join();
    }
    if (g_raw_oom.exchange(0)) return CM_ENOMEM;
    if (f->s1.io_error || f->s2.io_error) return CM_EIO;
    if (a2 < a1 && a2 < max_pairs) return CM_EINVAL;                  // R2 ends before R1
    const size_t n = (size_t)std::min<uint64_t>(a1, max_pairs);      // R1 decides; surplus R2 records at the end of input are ignored
    // R2 records beyond R1's last (only at the end of the input) are parsed too, as the record-by-record parser does, and cut
    // off by the caller: a malformed one among them is an error there, so it is one here
    const size_t n2 = (size_t)std::min<uint64_t>(a2, max_pairs);
    *n_out = n;
    *n2_out = n2;
    // The same verdicts, in the same call, as the record-by-record parser (gzip / pipe input): what follows the last whole
    // record -- a partial record, blank lines -- is malformed input (the reference asserts the '@', fastq_parser.h:64-65), and it
    // is reported by the call that would have parsed it, i.e. unless this batch is full without it.
    if ((tail1 != 0 && a1 < max_pairs) || (tail2 != 0 && a2 < max_pairs)) return CM_EINVAL;
    if (n == 0 && n2 == 0) return CM_OK;
    bool ok1 = true, ok2 = true, any = false;
    const auto t_mid = std::chrono::steady_clock::now();
    {
        std::thread t2([&]() { guarded([&]() { ok2 = n2 == 0 || build_side(f, f->s2, f->nl2, n2, G.b, nullptr, nullptr, half); }); });
        try {
            ok1 = n == 0 || build_side(f, f->s1, f->nl1, n, G.a, &G.prior, &any, half);
        } catch (const std::bad_alloc &) {
            g_raw_oom = 1;
        }
        t2.join();
    }
    if (g_raw_oom.exchange(0)) return CM_ENOMEM;
    if (!ok1 || !ok2) return CM_EINVAL;
    f->any_prior = any;
    if (trace) fprintf(stderr, "[fastq] %zu pairs: fill+index %.1f ms, build %.1f ms\n", n, std::chrono::duration<double, std::milli>(t_mid - t0).count(),
                       std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_mid).count());
    return CM_OK;
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
    bool failed = false;                         // a write came up short: reported by the cm_write_* calls / cm_writer_flush
    void flush() {
        if (f && n && fwrite(buf.data(), 1, n, f) != n) failed = true;
        n = 0;
    }
    void room(size_t k) {
        if (n + k > buf.size()) {
            if (!f) {                                  // private buffer of a formatting thread: grow
                buf.resize(std::max(buf.size() * 2, n + k));
                return;
            }
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

// Byte offsets at which the records `rec[0..n)` (4 lines each) of a plain-text FASTQ file start: the newlines of the file are
// counted in blocks on several threads, then the block holding the wanted newline is looked at again.  A record index equal to
// the number of records maps to the end of the file.  Returns the number of records, or -1 on a read error.
static long long record_offsets(int fd, uint64_t size, const uint64_t *rec, int n, uint64_t *off, int nt) {
    constexpr uint64_t BLK = 32ull << 20;
    const uint64_t nblk = (size + BLK - 1) / BLK;
    std::vector<uint64_t> cnt(nblk + 1, 0);
    std::atomic<uint64_t> next{0};
    std::atomic<int> bad{0};
    auto read_blk = [&](uint64_t b, std::vector<char> &buf) -> size_t {
        const uint64_t a = b * BLK, e = std::min(size, a + BLK);
        buf.resize((size_t)(e - a));
        size_t done = 0;
        while (done < buf.size()) {
            const ssize_t r = pread(fd, buf.data() + done, buf.size() - done, (off_t)(a + done));
            if (r < 0 && errno == EINTR) continue;
            if (r <= 0) {
                bad = 1;
                return 0;
            }
            done += (size_t)r;
        }
        return done;
    };
    std::vector<std::thread> th;
    for (int t = 0; t < std::max(1, nt); ++t)
        th.emplace_back([&]() {
            std::vector<char> buf;
            for (uint64_t b; (b = next.fetch_add(1)) < nblk;) {
                const size_t got = read_blk(b, buf);
                cnt[b + 1] = (uint64_t)std::count(buf.data(), buf.data() + got, '\n');
            }
        });
    for (auto &x : th) x.join();
    if (bad) return -1;
    for (uint64_t b = 0; b < nblk; ++b) cnt[b + 1] += cnt[b];
    uint64_t lines = cnt[nblk];
    if (size > 0) {                                   // a last line without a newline still is a line
        char last = 0;
        if (pread(fd, &last, 1, (off_t)(size - 1)) != 1) return -1;
        if (last != '\n') ++lines;
    }
    const uint64_t n_rec = lines / 4;
    std::vector<char> buf;
    for (int i = 0; i < n; ++i) {
        const uint64_t want = rec[i] * 4;             // the record starts behind the want-th newline
        if (rec[i] >= n_rec) {
            off[i] = size;
            continue;
        }
        if (want == 0) {
            off[i] = 0;
            continue;
        }
        const uint64_t b = (uint64_t)(std::lower_bound(cnt.begin(), cnt.end(), want) - cnt.begin()) - 1;     // cnt[b] < want <= cnt[b + 1]
        const size_t got = read_blk(b, buf);
        if (bad) return -1;
        uint64_t seen = cnt[b];
        size_t k = 0;
        for (; k < got; ++k)
            if (buf[k] == '\n' && ++seen == want) break;
        off[i] = b * BLK + k + 1;
    }
    return (long long)n_rec;
}

int cm_fastq_open_shard(const char *r1_path, const char *r2_path, const cm_chr_info *chrs, uint32_t n_chr, int32_t max_ed, int32_t rank,
                        int32_t world, int n_threads, cm_fastq **out, uint64_t *first_pair, uint64_t *n_pairs) {
    if (world < 1 || rank < 0 || rank >= world) return CM_EINVAL;
    const int rc = cm_fastq_open(r1_path, r2_path, chrs, n_chr, max_ed, out);
    if (rc != CM_OK) return rc;
    cm_fastq *f = *out;
    if (first_pair) *first_pair = 0;
    if (n_pairs) *n_pairs = ~0ull;
    if (world == 1 && !n_pairs) return CM_OK;
    auto fail = [&](int e) {
        cm_fastq_close(f);
        *out = nullptr;
        return e;
    };
    // contiguous blocks of pairs need both files cut at the same RECORD: seekable plain text only (gzip members cannot be entered
    // in the middle, a pipe cannot be read twice)
    if (!f->s1.plain || !f->s2.plain) {
        if (world == 1 && !n_pairs) return CM_OK;
        // gzip members cannot be entered in the middle: the records are counted with one inflate pass over R1 (R1 decides, as
        // everywhere), every rank then inflates from the start and steps over what belongs to the ranks before it.  Slow by
        // nature (one zlib stream per file: ~ 0.4 M pairs/s) -- but the same contiguous blocks, hence the same output bytes after
        // cm_merge_parts, as plain-text input.  A pipe cannot be read twice: refused.
        struct stat sa;
        if (stat(r1_path, &sa) != 0 || !S_ISREG(sa.st_mode)) return world == 1 ? CM_OK : fail(CM_EINVAL);
        gzFile g = gzopen(r1_path, "rb");
        if (!g) return fail(CM_EIO);
        gzbuffer(g, 1u << 20);
        std::vector<char> buf(4u << 20);
        uint64_t lines = 0;
        char last = '\n';
        for (;;) {
            const int got = gzread(g, buf.data(), (unsigned)buf.size());
            if (got < 0) {
                gzclose(g);
                return fail(CM_EIO);
            }
            if (got == 0) break;
            lines += (uint64_t)std::count(buf.data(), buf.data() + got, '\n');
            last = buf[(size_t)got - 1];
        }
        gzclose(g);
        if (last != '\n') ++lines;                    // a last line without a newline still is a line
        const uint64_t n1 = lines / 4;
        const uint64_t lo = n1 * (uint64_t)rank / (uint64_t)world, hi = n1 * (uint64_t)(rank + 1) / (uint64_t)world;
        f->skip_left = lo;
        f->take_left = rank == world - 1 ? ~0ull : hi - lo;      // the last rank reads on: what follows the last whole record is seen (and refused) as in one process
        if (first_pair) *first_pair = lo;
        if (n_pairs) *n_pairs = hi - lo;
        return CM_OK;
    }
    struct stat sa, sb;
    if (fstat(fileno(f->s1.plain), &sa) != 0 || fstat(fileno(f->s2.plain), &sb) != 0) return fail(CM_EIO);
    int nt = n_threads > 0 ? n_threads : (int)std::thread::hardware_concurrency();
    nt = nt < 1 ? 1 : (nt > 32 ? 32 : nt);
    // number of records first (R1 decides, as everywhere), then this rank's [lo, hi) in both files
    uint64_t none = 0, dummy = 0;
    const long long n1 = record_offsets(fileno(f->s1.plain), (uint64_t)sa.st_size, &none, 0, &dummy, nt);
    if (n1 < 0) return fail(CM_EIO);
    const uint64_t lo = (uint64_t)n1 * (uint64_t)rank / (uint64_t)world, hi = (uint64_t)n1 * (uint64_t)(rank + 1) / (uint64_t)world;
    const uint64_t want[2] = {lo, hi};
    uint64_t o1[2], o2[2];
    if (record_offsets(fileno(f->s1.plain), (uint64_t)sa.st_size, want, 2, o1, nt) < 0) return fail(CM_EIO);
    const long long n2 = record_offsets(fileno(f->s2.plain), (uint64_t)sb.st_size, want, 2, o2, nt);
    if (n2 < 0) return fail(CM_EIO);
    if ((uint64_t)n2 < hi) return fail(CM_EINVAL);                                      // R2 ends before R1
    f->s1.file_pos = o1[0];
    f->s2.file_pos = o2[0];
    // the last rank reads to the end of both files, so that what follows the last whole record is seen (and refused) as in an
    // unsharded run; R2's surplus records, if any, stay with it too
    f->s1.file_end = rank == world - 1 ? ~0ull : o1[1];
    f->s2.file_end = rank == world - 1 ? ~0ull : o2[1];
    if (first_pair) *first_pair = lo;
    if (n_pairs) *n_pairs = hi - lo;
    return CM_OK;
}

static int fastq_next(cm_fastq *f, uint64_t max_pairs, cm_fastq_batch *out);
int cm_fastq_next(cm_fastq *f, uint64_t max_pairs, cm_fastq_batch *out) {
    if (!f || !out) return CM_EINVAL;
    try {
        return fastq_next(f, max_pairs, out);
    } catch (const std::bad_alloc &) {
        return CM_ENOMEM;
    }
}
void cm_fastq_set_release_hook(cm_fastq *f, void (*fn)(void *user, const void *ptr, uint64_t bytes), void *user) {
    if (!f) return;
    f->hook.fn = fn;
    f->hook.user = user;
}
static int fastq_next(cm_fastq *f, uint64_t max_pairs, cm_fastq_batch *out) {
    f->cur = (f->cur + 1) % 4;
    cm_fastq::Gen &G = f->gen[f->cur];
    G.a.clear();
    G.b.clear();
    G.prior.clear();
    f->any_prior = false;
    uint64_t n_fast = 0, n2_fast = 0;
    const bool fast = f->s1.plain && f->s2.plain && !getenv("CM_FASTQ_SERIAL");
    if (fast) {
        const int rc = next_plain(f, max_pairs, G, &n_fast, &n2_fast);
        if (rc != CM_OK) return rc;
    }
    // The two files are independent streams until the records are paired up: R2 is parsed (and, for .gz input, inflated)
    // on a second thread while this one does R1 and its carried state.  The reference does both inside one lock-protected
    // serial section (src/circminer.cpp:373-379), which is its ingest ceiling.
    uint64_t n = fast ? n_fast : 0, n2 = fast ? n2_fast : 0;
    int bad2 = 0;
    if (!fast && f->skip_left) {                       // a shard of gzip input: the records of the ranks before this one
        const char *lp;
        size_t ll;
        for (; f->skip_left; --f->skip_left)
            for (int k = 0; k < 4; ++k)
                if (!f->s1.line(lp, ll) || !f->s2.line(lp, ll)) return (f->s1.io_error || f->s2.io_error) ? CM_EIO : CM_EINVAL;
    }
    if (!fast && f->take_left < max_pairs) max_pairs = f->take_left;
    std::thread side_b([&]() {
        guarded([&]() {
            while (!fast && n2 < max_pairs) {
                const int r2 = parse_record(f, f->s2, G.b, false, nullptr, nullptr);
                if (r2 == 0) break;
                if (r2 < 0) {
                    bad2 = 1;
                    break;
                }
                ++n2;
            }
        });
    });
    int bad1 = 0;
    try {
        while (!fast && n < max_pairs) {
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
    } catch (const std::bad_alloc &) {
        g_raw_oom = 1;
    }
    side_b.join();
    if (g_raw_oom.exchange(0)) return CM_ENOMEM;
    if (f->s1.io_error || f->s2.io_error) return CM_EIO;
    if (bad1 || bad2 || n2 < n) return CM_EINVAL;                   // malformed record, or R2 ends before R1
    if (!fast && f->take_left != ~0ull) f->take_left -= n;
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
}  // extern "C"
// write_read_category for n pairs: pair(k) = index in the batch, state(k) = its MatchedRead
template <class PairOf, class StateOf>
static int write_remain_rows(cm_writer *w, const cm_fastq_batch *b, uint64_t n, PairOf pair, StateOf state) {
    for (uint64_t k = 0; k < n; ++k) {
        const uint64_t i = pair(k);
        if (i >= b->reads.n_pairs) return CM_EINVAL;
        const cm_mapped_read &m = state(k);
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
    return (w->o1.failed || w->o2.failed) ? CM_EIO : CM_OK;
}

extern "C" {
int cm_write_remain(cm_writer *w, const cm_fastq_batch *b, const cm_mapped_read *states, const uint64_t *sel, uint64_t n_sel) {
    if (!w || !w->f2 || !b || !states) return CM_EINVAL;
    return write_remain_rows(w, b, sel ? n_sel : b->reads.n_pairs, [&](uint64_t k) { return sel ? sel[k] : k; },
                             [&](uint64_t k) -> const cm_mapped_read & { return states[sel ? sel[k] : k]; });
}

int cm_write_remain_records(cm_writer *w, const cm_fastq_batch *b, const cm_record *recs, uint64_t n) {
    if (!w || !w->f2 || !b || (n && !recs)) return CM_EINVAL;
    return write_remain_rows(w, b, n, [&](uint64_t k) { return recs[k].pair; }, [&](uint64_t k) -> const cm_mapped_read & { return recs[k].state; });
}

// write_pam_rec_pe for the selected pairs (names of R1)
static void pam_row(const cm_writer *w, const cm_fastq_batch *b, const cm_mapped_read &m, uint64_t i, Out &o) {
    const char *nm = b->names1 + b->name_off1[i];
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

int cm_write_pam(cm_writer *w, const cm_fastq_batch *b, const cm_mapped_read *states, const uint64_t *sel, uint64_t n_sel) {
    if (!w || !b || !states) return CM_EINVAL;
    const uint64_t n = sel ? n_sel : b->reads.n_pairs;
    if (sel)
        for (uint64_t k = 0; k < n; ++k)
            if (sel[k] >= b->reads.n_pairs) return CM_EINVAL;
    // large batches: rows formatted on several threads into private buffers, written out in order
    int nt = (int)std::thread::hardware_concurrency();
    if (const char *e = getenv("CM_WRITER_THREADS")) nt = atoi(e);
    nt = nt < 1 ? 1 : (nt > 16 ? 16 : nt);
    if (n < 65536 || nt == 1) {
        for (uint64_t k = 0; k < n; ++k) {
            const uint64_t i = sel ? sel[k] : k;
            pam_row(w, b, states[i], i, w->o1);
        }
        return w->o1.failed ? CM_EIO : CM_OK;
    }
    w->o1.flush();
    std::vector<Out> part((size_t)nt);
    par_for(nt, (size_t)n, [&](int t, size_t lo, size_t hi) {
        Out &o = part[(size_t)t];
        o.buf.resize((hi - lo) * 160 + 4096);          // grows on demand (room()); never flushed: f == nullptr
        for (size_t k = lo; k < hi; ++k) {
            const uint64_t i = sel ? sel[k] : k;
            pam_row(w, b, states[i], i, o);
        }
    });
    for (Out &o : part)
        if (o.n && fwrite(o.buf.data(), 1, o.n, w->f1) != o.n) {
            w->o1.failed = true;
            return CM_EIO;
        }
    return w->o1.failed ? CM_EIO : CM_OK;
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
    return (w->o1.failed || w->o2.failed) ? CM_EIO : CM_OK;
}

int cm_writer_flush(cm_writer *w) {
    if (!w) return CM_EINVAL;
    w->o1.flush();
    w->o2.flush();
    if ((w->f1 && fflush(w->f1) != 0) || (w->f2 && fflush(w->f2) != 0)) w->o1.failed = true;
    return (w->o1.failed || w->o2.failed) ? CM_EIO : CM_OK;
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
