// On-disk genome / index formats of stock CircMiner, host side only.
//
//   <ref>.packed.fa            FASTA whose records are the packed contigs ">1", ">2", ...: chromosomes
//                              concatenated with a 50-N spacer while they fit CONTIG_SIZE
//                              (reference src/genome.cpp:96-146);
//   <ref>.packed.fa.index.info "contig \t start \t end \t chrName" per chromosome, 0-based start inside its
//                              packed contig (src/genome.cpp:127-137, read back at :147-167);
//   <ref>.packed.fa.index      mrsfast hash-table file (src/mrsfast/HashTable.c):
//       header  u8 magic (2 compact / 3 full table) | u8 WINDOW_SIZE | u8 checkSumLength | u32 maxMemSize
//               | u32 ioBufferSize | u32 CONTIG_MAX_SIZE | i32 nRecords | {i32 nameLen, name, i32 len} x nRecords
//               (:106-127; maxMemSize patched at offset 3 by finalizeSavingIHashTable :131-137)
//       per packed contig  u8 moreFollows | i16 nameLen | name | i32 offset | u32 refLen
//               | u64 packed[ceil(refLen / 21)] (3 bits per base, first base in bits 62..60)
//               | u32 nBuckets | { i32 nBytes, varbyte(hvDelta), varbyte(count14) ... } blocks
//               | (full table only) u32 memSize | GeneralIndex[memSize]                      (:197-254)
//       GeneralIndex = { u16 checksum; (2 bytes padding); i32 info } = 8 bytes.  A bucket owns count14 + 1
//       slots: [0].info = number of valid entries, then the entries sorted by (checksum, info = 1-based
//       start) (:824-839, Sort.c:116-117), then slack slots for 14-mers whose full k-mer is invalid.
//       The reference never initialises the padding bytes, the header slots' checksum and the slack
//       slots; this writer zeroes them, the reader ignores them.
//   varbyte: 7-bit little-endian groups, the last byte carries 0x80 (:74-98).
//
// The FASTA loader of the reference lives in the absent mrsfast submodule (RefGenome.c); its contract is
// fixed by the in-tree callers (HashTable.c:288-293, 618-633): one record = one contig, bases upper-cased,
// anything but A/C/G/T becomes N, offset 0.
#include <sys/mman.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cctype>
#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <chrono>
#include <memory>
#include <new>
#include <functional>
#include <vector>

#include "circminer_hot.h"

namespace {

constexpr int MIDNCNT = 50;                 // src/genome.cpp:16
constexpr uint32_t IO_BUFFER = 1u << 24;    // HashTable.c:60
constexpr uint32_t CONTIG_MAX_SIZE_DEF = 1300000000u;   // src/common.h:82

struct Fasta {
    std::vector<std::string> id, seq;
};

// GenomePacker::get_next_chr (src/genome.cpp:73-94): id = first token after '>', every following line
// contributes its first whitespace-delimited token.
bool read_fasta(const char *path, Fasta &out) {
    FILE *f = fopen(path, "rb");
    if (!f) return false;
    bool have = false;
    auto take_line = [&](const char *p, size_t n) {             // one line without its line feed
        if (n == 0) return;
        if (p[0] == '>') {
            size_t b = 1;
            while (b < n && !isspace((unsigned char)p[b])) ++b;
            out.id.emplace_back(p + 1, b - 1);
            out.seq.emplace_back();
            have = true;
        } else if (have) {
            size_t a = 0;
            while (a < n && isspace((unsigned char)p[a])) ++a;
            size_t b = a;
            while (b < n && !isspace((unsigned char)p[b])) ++b;
            out.seq.back().append(p + a, b - a);
        }
    };
    // blocks of 4 MB, lines cut out with memchr (one fgetc per base took a minute of hg38's 136 s); a line that straddles two
    // blocks is put together in `carry`
    std::vector<char> buf(4u << 20);
    std::string carry;
    size_t got;
    while ((got = fread(buf.data(), 1, buf.size(), f)) > 0) {
        const char *p = buf.data(), *e = p + got;
        while (p < e) {
            const char *z = (const char *)memchr(p, '\n', (size_t)(e - p));
            if (!z) {
                carry.append(p, (size_t)(e - p));
                break;
            }
            if (!carry.empty()) {
                carry.append(p, (size_t)(z - p));
                take_line(carry.data(), carry.size());
                carry.clear();
            } else take_line(p, (size_t)(z - p));
            p = z + 1;
        }
    }
    if (!carry.empty()) take_line(carry.data(), carry.size());
    const bool ok = !ferror(f);
    fclose(f);
    return ok;
}

inline int code3(char ch) {
    switch (ch) {
        case 'A': return 0;
        case 'C': return 1;
        case 'G': return 2;
        case 'T': return 3;
        default: return 4;
    }
}

int encode_varbyte(uint8_t *buf, uint32_t v) {          // HashTable.c:74-83
    int t = 0;
    do {
        buf[t++] = (uint8_t)(v & 127u);
        v /= 128u;
    } while (v != 0);
    buf[t - 1] |= 128u;
    return t;
}
int decode_varbyte(const uint8_t *buf, size_t avail, uint32_t *res) {     // HashTable.c:87-98
    size_t i = 0;
    uint32_t r = 0;
    uint8_t t;
    do {
        if (i >= avail || i >= 5) return -1;
        t = buf[i];
        r |= (uint32_t)(t & 127u) << (7 * i);
        ++i;
    } while ((t & 128u) == 0);
    *res = r;
    return (int)i;
}

template <class T> bool put(FILE *f, const T &v) { return fwrite(&v, sizeof(T), 1, f) == 1; }
template <class T> bool get(FILE *f, T &v) { return fread(&v, sizeof(T), 1, f) == 1; }

struct Entry {              // GeneralIndex as it is laid out in the file
    uint16_t checksum;
    uint16_t pad;
    int32_t info;
};
static_assert(sizeof(Entry) == 8, "GeneralIndex is 8 bytes");

uint32_t io_buffer_size() {           // test hook: a small buffer exercises the block splitting of saveHashTable
    const char *e = getenv("CM_INDEX_IOBUF");
    const long v = e ? atol(e) : 0;
    return v >= 64 ? (uint32_t)v : IO_BUFFER;
}

}  // namespace

// Multi-GB buffers the loader fills once per contig: 2-MB aligned and advised as transparent huge pages, so that filling them
// costs thousands of page faults instead of millions (a fresh 8.5-GB table buffer took ~1 s of faults under 32 pread()ing threads).
static void *big_alloc(size_t bytes) {
    constexpr size_t HUGE = 2u << 20;
    if (bytes < 4 * HUGE) return malloc(bytes);
    void *p = nullptr;
    if (posix_memalign(&p, HUGE, (bytes + HUGE - 1) / HUGE * HUGE) != 0) return nullptr;
#if defined(MADV_HUGEPAGE)
    (void)madvise(p, (bytes + HUGE - 1) / HUGE * HUGE, MADV_HUGEPAGE);
#endif
    return p;
}

struct cm_index_file {
    FILE *f = nullptr;
    int window = 0, checksum_len = 0, full = 0;
    uint32_t max_mem = 0, io_buf = 0, contig_max = 0;
    std::vector<std::string> names;
    std::vector<int32_t> lens;
    bool done = false;
    void *tab = nullptr;                 // the table of the contig being loaded as it is in the file; reused from contig to contig
    size_t tab_bytes = 0;
    struct RawSet {                      // cm_host_next_contig_raw: two sets take turns
        uint8_t *genome = nullptr;
        uint32_t *hv = nullptr, *cnt = nullptr;
        void *tab = nullptr;
        size_t genome_cap = 0, hdr_cap = 0, tab_cap = 0;
    } raw[2];
    int raw_turn = 0;
    ~cm_index_file() {
        free(tab);
        for (auto &r : raw) {
            free(r.genome);
            free(r.hv);
            free(r.cnt);
            free(r.tab);
        }
    }
};

extern "C" {

int cm_host_pack_genome(const char *fasta_path, const char *packed_fa_path, const char *index_info_path, uint32_t contig_size) {
    if (!fasta_path || !packed_fa_path || !index_info_path || contig_size == 0) return CM_EINVAL;
    Fasta fa;
    if (!read_fasta(fasta_path, fa)) return CM_EINVAL;
    FILE *fo = fopen(packed_fa_path, "wb"), *fi = fopen(index_info_path, "wb");
    if (!fo || !fi) {
        if (fo) fclose(fo);
        if (fi) fclose(fi);
        return CM_EINVAL;
    }
    const std::string mid(MIDNCNT, 'N');
    int contig_num = 0;
    long long cur = 0;            // the reference keeps this in an int (src/genome.cpp:104)
    for (size_t r = 0; r < fa.id.size(); ++r) {
        const long long len = (long long)fa.seq[r].size();
        if (cur == 0 || len + MIDNCNT + cur > (long long)contig_size) {
            ++contig_num;
            cur = 0;
            fprintf(fo, ">%d\n%s\n", contig_num, fa.seq[r].c_str());
            fprintf(fi, "%d\t%lld\t%lld\t%s\n", contig_num, cur, cur + len, fa.id[r].c_str());
            cur += len;
        } else {
            fprintf(fo, "%s%s\n", mid.c_str(), fa.seq[r].c_str());
            fprintf(fi, "%d\t%lld\t%lld\t%s\n", contig_num, cur + MIDNCNT, cur + MIDNCNT + len, fa.id[r].c_str());
            cur += MIDNCNT + len;
        }
    }
    fclose(fo);
    fclose(fi);
    return CM_OK;
}

int cm_host_read_index_info(const char *path, cm_chr_info **out, uint32_t *n) {
    if (!path || !out || !n) return CM_EINVAL;
    *out = nullptr;
    *n = 0;
    FILE *f = fopen(path, "rb");
    if (!f) return CM_EINVAL;
    std::vector<cm_chr_info> v;
    unsigned int contig, s, e;
    char name[4096];
    while (fscanf(f, "%u %u %u %4095s", &contig, &s, &e, name) == 4) {     // GenomePacker::load_index_info
        cm_chr_info ci;
        char *nm = (char *)malloc(strlen(name) + 1);
        strcpy(nm, name);
        ci.name = nm;
        ci.contig_id = contig;
        ci.start_pos = s;
        ci.len = e - s;
        v.push_back(ci);
    }
    fclose(f);
    cm_chr_info *arr = (cm_chr_info *)malloc((v.size() ? v.size() : 1) * sizeof(cm_chr_info));
    if (!arr) return CM_ENOMEM;
    for (size_t i = 0; i < v.size(); ++i) arr[i] = v[i];
    *out = arr;
    *n = (uint32_t)v.size();
    return CM_OK;
}

void cm_host_free_index_info(cm_chr_info *chrs, uint32_t n) {
    if (!chrs) return;
    for (uint32_t i = 0; i < n; ++i) free((void *)chrs[i].name);
    free(chrs);
}

int cm_host_write_index(const char *packed_fa_path, const char *index_path, int32_t kmer, int compact, int n_threads) {
    if (!packed_fa_path || !index_path || kmer < CM_WINDOW_SIZE || kmer > CM_WINDOW_SIZE + 8) return CM_EINVAL;
    Fasta fa;
    if (!read_fasta(packed_fa_path, fa) || fa.id.empty()) return CM_EINVAL;
    // the per-base passes below (case folding, 3-bit packing, the count of the 14-mer windows) ran on one thread: a minute of
    // hg38's 136 s; they are cut into ranges now (the window count with relaxed atomic increments: the table is shared)
    const int TP = std::max(1, std::min(n_threads, 32));
    auto par_ranges = [&](uint64_t n, uint64_t align, const std::function<void(uint64_t, uint64_t)> &fn) {
        if (TP == 1 || n < (1u << 20)) {
            fn(0, n);
            return;
        }
        std::vector<std::thread> th;
        for (int t = 0; t < TP; ++t) {
            uint64_t a = n * (uint64_t)t / TP / align * align, b = t + 1 == TP ? n : n * (uint64_t)(t + 1) / TP / align * align;
            if (b > a) th.emplace_back(fn, a, b);
        }
        for (auto &x : th) x.join();
    };
    for (auto &s : fa.seq)                                   // loadRefGenome contract: upper-case, non-ACGT -> N
        par_ranges(s.size(), 1, [&](uint64_t a, uint64_t b) {
            for (uint64_t i = a; i < b; ++i) {
                const char u = (char)toupper((unsigned char)s[i]);
                s[i] = (u == 'A' || u == 'C' || u == 'G' || u == 'T') ? u : 'N';
            }
        });
    FILE *f = fopen(index_path, "wb");
    if (!f) return CM_EINVAL;
    const uint8_t magic = compact ? 2 : 3, W = CM_WINDOW_SIZE;
    const int8_t c = (int8_t)(kmer - CM_WINDOW_SIZE);
    uint32_t max_mem = 0;
    const uint32_t io_buf = io_buffer_size(), cmax = CONTIG_MAX_SIZE_DEF;
    bool ok = put(f, magic) && put(f, W) && put(f, c) && put(f, max_mem) && put(f, io_buf) && put(f, cmax);
    const int32_t nrec = (int32_t)fa.id.size();
    ok = ok && put(f, nrec);
    for (int32_t r = 0; r < nrec && ok; ++r) {
        const int32_t nl = (int32_t)fa.id[r].size(), len = (int32_t)fa.seq[r].size();
        ok = put(f, nl) && fwrite(fa.id[r].data(), 1, (size_t)nl, f) == (size_t)nl && put(f, len);
    }
    const uint64_t nb = 1ull << (2 * CM_WINDOW_SIZE);
    std::vector<uint32_t> cnt14(nb);
    std::vector<uint8_t> buf(io_buf);
    int rc = CM_OK;
    for (int32_t r = 0; r < nrec && ok && rc == CM_OK; ++r) {
        const std::string &g = fa.seq[r];
        const uint32_t n = (uint32_t)g.size();
        const uint8_t more = (r + 1 < nrec) ? 1 : 0;
        const int16_t nl = (int16_t)fa.id[r].size();
        const int32_t off = 0;
        ok = put(f, more) && put(f, nl) && fwrite(fa.id[r].data(), 1, (size_t)nl, f) == (size_t)nl && put(f, off) && put(f, n);
        // compressSequence: 21 bases per word, 3 bits each, first base in bits 62..60, last word left-aligned
        const uint32_t nw = n / 21 + (n % 21 != 0);
        std::vector<uint64_t> packed(nw, 0);
        par_ranges(nw, 1, [&](uint64_t wa, uint64_t wb) {     // whole words per thread
            for (uint64_t w = wa; w < wb; ++w) {
                uint64_t v = 0;
                const uint64_t i0 = w * 21, i1 = std::min<uint64_t>(i0 + 21, n);
                for (uint64_t i = i0; i < i1; ++i) v |= (uint64_t)code3(g[i]) << (60 - 3 * (i - i0));
                packed[w] = v;
            }
        });
        ok = ok && (nw == 0 || fwrite(packed.data(), 8, nw, f) == nw);
        // count pass (HashTable.c:317-338): every 14-mer window without N
        std::fill(cnt14.begin(), cnt14.end(), 0u);
        uint32_t nbuckets = 0;
        {
            const uint32_t mask = (uint32_t)(nb - 1);
            std::atomic<uint32_t> nbk{0};
            // a range [a, b) counts the windows that END in it: the hash is warmed up over the 13 bases in front of a
            par_ranges(n, 1, [&](uint64_t a, uint64_t b) {
                uint32_t hv = 0, mine = 0;
                int run = 0;
                for (uint64_t i = a >= (uint64_t)(CM_WINDOW_SIZE - 1) ? a - (CM_WINDOW_SIZE - 1) : 0; i < b; ++i) {
                    const int v = code3(g[i]);
                    if (v == 4) {
                        run = 0;
                        hv = 0;
                        continue;
                    }
                    hv = ((hv << 2) | (uint32_t)v) & mask;
                    ++run;
                    if (i >= a && run >= CM_WINDOW_SIZE && __atomic_fetch_add(&cnt14[hv], 1u, __ATOMIC_RELAXED) == 0) ++mine;
                }
                nbk += mine;
            });
            nbuckets = nbk.load();
        }
        ok = ok && put(f, nbuckets);
        uint64_t mem = 0;
        {
            uint32_t k = 0, prev = 0;
            for (uint64_t h = 0; h < nb && ok; ++h) {
                if (!cnt14[h]) continue;
                mem += (uint64_t)cnt14[h] + 1;
                k += (uint32_t)encode_varbyte(buf.data() + k, (uint32_t)(h - prev));
                prev = (uint32_t)h;
                k += (uint32_t)encode_varbyte(buf.data() + k, cnt14[h]);
                if (k > io_buf - 10) {
                    const int32_t kk = (int32_t)k;
                    ok = put(f, kk) && fwrite(buf.data(), 1, k, f) == k;
                    k = 0;
                }
            }
            if (k && ok) {
                const int32_t kk = (int32_t)k;
                ok = put(f, kk) && fwrite(buf.data(), 1, k, f) == k;
            }
        }
        if (mem > 0xffffffffull) {
            rc = CM_ELIMIT;
            break;
        }
        if ((uint32_t)mem > max_mem) max_mem = (uint32_t)mem;
        if (!compact && ok) {
            cm_index_view iv{};
            rc = cm_host_build_index((const uint8_t *)g.data(), n, kmer, r, n_threads, &iv);
            if (rc != CM_OK) break;
            const uint32_t memsz = (uint32_t)mem;
            ok = put(f, memsz);
            // The table (per non-empty bucket: a header slot with the number of valid entries, the entries, zeroed slack up to
            // count14): bucket ranges are laid out side by side into one buffer -- a range's first slot is the sum of
            // (count14 + 1) over the non-empty buckets in front of it -- and written with one call.
            const int T = std::max(1, std::min(n_threads, 32));
            std::vector<uint64_t> first((size_t)T + 1, 0);
            {
                std::vector<std::thread> th;
                auto sum = [&](int t) {
                    uint64_t acc = 0;
                    for (uint64_t h = nb * (uint64_t)t / T, e = nb * (uint64_t)(t + 1) / T; h < e; ++h)
                        if (cnt14[h]) acc += (uint64_t)cnt14[h] + 1;
                    first[(size_t)t + 1] = acc;
                };
                for (int t = 1; t < T; ++t) th.emplace_back(sum, t);
                sum(0);
                for (auto &x : th) x.join();
                for (int t = 0; t < T; ++t) first[(size_t)t + 1] += first[(size_t)t];
            }
            Entry *tab = (Entry *)big_alloc(((size_t)memsz + 1) * sizeof(Entry));
            if (!tab) {
                cm_host_free_index(&iv);
                rc = CM_ENOMEM;
                break;
            }
            {
                std::vector<std::thread> th;
                auto fill = [&](int t) {
                    Entry *w = tab + first[(size_t)t];
                    for (uint64_t h = nb * (uint64_t)t / T, e = nb * (uint64_t)(t + 1) / T; h < e; ++h) {
                        if (!cnt14[h]) continue;
                        const uint32_t x0 = iv.bucket_off[h], x1 = iv.bucket_off[h + 1];
                        *w++ = Entry{0, 0, (int32_t)(x1 - x0)};
                        for (uint32_t i = x0; i < x1; ++i) *w++ = Entry{iv.checksum[i], 0, (int32_t)iv.pos[i]};
                        for (uint32_t i = x1 - x0; i < cnt14[h]; ++i) *w++ = Entry{0, 0, 0};
                    }
                };
                for (int t = 1; t < T; ++t) th.emplace_back(fill, t);
                fill(0);
                for (auto &x : th) x.join();
            }
            ok = ok && (memsz == 0 || fwrite(tab, sizeof(Entry), memsz, f) == memsz);
            free(tab);
            cm_host_free_index(&iv);
        }
    }
    if (ok && rc == CM_OK) {
        ok = fseek(f, 3, SEEK_SET) == 0 && put(f, max_mem);           // finalizeSavingIHashTable
    }
    fclose(f);
    if (rc != CM_OK) return rc;
    return ok ? CM_OK : CM_EINVAL;
}

int cm_host_open_index(const char *index_path, cm_index_file **out, int32_t *kmer, int32_t *is_full, uint32_t *n_records) {
    if (!index_path || !out) return CM_EINVAL;
    *out = nullptr;
    FILE *f = fopen(index_path, "rb");
    if (!f) return CM_EINVAL;
    cm_index_file *x = new cm_index_file();
    x->f = f;
    uint8_t magic = 0, W = 0;
    int8_t c = 0;
    int32_t nrec = 0;
    bool ok = get(f, magic) && get(f, W) && get(f, c) && get(f, x->max_mem) && get(f, x->io_buf) && get(f, x->contig_max) && get(f, nrec);
    if (!ok || (magic != 2 && magic != 3) || W != CM_WINDOW_SIZE || c < 0 || c > 8 || nrec < 0) {   // checkHashTable :485-509
        fclose(f);
        delete x;
        return CM_EINVAL;
    }
    for (int32_t r = 0; r < nrec && ok; ++r) {
        int32_t nl = 0, len = 0;
        ok = get(f, nl) && nl >= 0 && nl < 4096;
        std::string nm((size_t)(ok ? nl : 0), '\0');
        ok = ok && (nl == 0 || fread(&nm[0], 1, (size_t)nl, f) == (size_t)nl) && get(f, len);
        x->names.push_back(nm);
        x->lens.push_back(len);
    }
    if (!ok) {
        fclose(f);
        delete x;
        return CM_EINVAL;
    }
    x->window = W;
    x->checksum_len = c;
    x->full = magic == 3;
    if (kmer) *kmer = W + c;
    if (is_full) *is_full = x->full;
    if (n_records) *n_records = (uint32_t)nrec;
    *out = x;
    return CM_OK;
}

// Loads the next packed contig (loadHashTable, HashTable.c:971-1098): genome decoded to ASCII
// (pac2char_whole_contig, src/match_read.cpp:301-332) and the table in the flattened layout of
// cm_index_view.  Returns CM_OK and *loaded = 1, or *loaded = 0 after the last contig.
// The view (including its genome) is released with cm_host_free_loaded_contig.
static int next_contig(cm_index_file *x, int n_threads, cm_index_view *out, int *loaded, bool genome_only, cm_index_raw *raw_out = nullptr) {
    if (!x || (!out && !raw_out) || !loaded) return CM_EINVAL;
    if (raw_out && !x->full) return CM_EINVAL;
    cm_index_file::RawSet *RS = raw_out ? &x->raw[x->raw_turn] : nullptr;
    if (raw_out) x->raw_turn ^= 1;
    const bool trace = getenv("CM_INDEX_TRACE") != nullptr;
    auto tp = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (trace) fprintf(stderr, "[index] %s %.3f s\n", what, std::chrono::duration<double>(std::chrono::steady_clock::now() - tp).count());
        tp = std::chrono::steady_clock::now();
    };
    *loaded = 0;
    if (x->done) return CM_OK;
    FILE *f = x->f;
    uint8_t more = 0;
    if (!get(f, more)) {
        x->done = true;
        return CM_OK;
    }
    int16_t nl = 0;
    int32_t off = 0;
    uint32_t n = 0;
    char name[4096];
    if (!get(f, nl) || nl < 0 || nl >= 4096 || (nl && fread(name, 1, (size_t)nl, f) != (size_t)nl) || !get(f, off) || !get(f, n)) return CM_EINVAL;
    name[nl] = 0;
    const uint32_t nw = n / 21 + (n % 21 != 0);
    std::vector<uint64_t> packed(nw);
    if (nw && fread(packed.data(), 8, nw, f) != nw) return CM_EINVAL;
    uint8_t *g;
    if (RS) {                                      // handle-owned, reused
        if (RS->genome_cap < (size_t)n + 1) {
            free(RS->genome);
            RS->genome = (uint8_t *)big_alloc((size_t)n + 1);
            RS->genome_cap = RS->genome ? (size_t)n + 1 : 0;
        }
        g = RS->genome;
    } else g = (uint8_t *)malloc((size_t)n + 1);
    if (!g) return CM_ENOMEM;
    {   // 21 bases per 64-bit word, first base in the top bits; word ranges decoded side by side
        auto decode = [&](uint32_t w0, uint32_t w1) {
            for (uint32_t wd = w0; wd < w1; ++wd) {
                const uint64_t x = packed[wd];
                const uint32_t base = wd * 21u, cnt = n - base < 21u ? n - base : 21u;
                for (uint32_t j = 0; j < cnt; ++j) g[base + j] = (uint8_t)"ACGTNNNN"[(x >> (60 - 3 * j)) & 7u];
            }
        };
        int T = n_threads > 0 ? n_threads : 1;
        if (genome_only) T = (int)std::min<unsigned>(16u, std::max(1u, std::thread::hardware_concurrency()));
        if (nw < (1u << 20)) T = 1;
        if (T > 32) T = 32;
        std::vector<std::thread> th;
        for (int t = 1; t < T; ++t) th.emplace_back(decode, (uint32_t)((uint64_t)nw * t / T), (uint32_t)((uint64_t)nw * (t + 1) / T));
        decode(0, (uint32_t)((uint64_t)nw / T));
        for (auto &t : th) t.join();
    }
    g[n] = 0;
    lap("sequence read + decoded");
    uint32_t nbuckets = 0;
    if (!get(f, nbuckets)) {
        if (!RS) free(g);
        return CM_EINVAL;
    }
    const uint64_t nb = 1ull << (2 * CM_WINDOW_SIZE);
    // The bucket headers (hv delta, count14) come in blocks of varbyte pairs, each behind its byte length (HashTable.c:197-254).
    // The blocks are read in one sequential pass -- a block's entry count is half its number of terminator bytes -- and decoded
    // side by side: pass 1 gives every block's sum of deltas, its entries and its table slots, a prefix over the blocks gives the
    // starting hv and position of each, pass 2 writes hvs[] / cnts[].
    struct Blk { size_t off, bytes; uint64_t n, dsum, mem; };
    std::vector<Blk> blks;
    std::vector<uint8_t> hdr;
    uint64_t mem = 0;
    {
        uint64_t seen = 0;
        while (seen < nbuckets) {
            int32_t bytes = 0;
            if (!get(f, bytes) || bytes <= 0) {
                if (!RS) free(g);
                return CM_EINVAL;
            }
            const size_t at = hdr.size();
            hdr.resize(at + (size_t)bytes);
            if (fread(hdr.data() + at, 1, (size_t)bytes, f) != (size_t)bytes) {
                if (!RS) free(g);
                return CM_EINVAL;
            }
            uint64_t term = 0;                      // bytes with the top bit set end a varbyte; eight at a time
            {
                size_t k = at;
                const size_t e = at + (size_t)bytes;
                for (; k + 8 <= e; k += 8) {
                    uint64_t w;
                    memcpy(&w, hdr.data() + k, 8);
                    term += (uint64_t)__builtin_popcountll(w & 0x8080808080808080ull);
                }
                for (; k < e; ++k) term += hdr[k] >> 7;
            }
            if (term == 0 || (term & 1)) {
                if (!RS) free(g);
                return CM_EINVAL;
            }
            blks.push_back(Blk{at, (size_t)bytes, term / 2, 0, 0});
            seen += term / 2;
        }
        if (seen != nbuckets) {
            if (!RS) free(g);
            return CM_EINVAL;
        }
    }
    const int TH = std::max(1, std::min(n_threads, 32));
    auto over_blocks = [&](auto &&body) {
        std::atomic<size_t> next{0};
        std::vector<std::thread> th;
        auto run = [&]() {
            for (size_t b; (b = next.fetch_add(1)) < blks.size();) body(b);
        };
        for (int t = 1; t < TH; ++t) th.emplace_back(run);
        run();
        for (auto &t : th) t.join();
    };
    std::atomic<int> hdr_bad{0};
    if (!genome_only || !x->full)          // (stage 2 steps over the table: its size follows the blocks in the file, nothing to decode)
    over_blocks([&](size_t b) {
        Blk &B = blks[b];
        const uint8_t *p = hdr.data() + B.off;
        size_t idx = 0;
        uint64_t dsum = 0, msum = 0, cnt = 0;
        while (idx < B.bytes) {
            uint32_t d = 0, cn = 0;
            int a = decode_varbyte(p + idx, B.bytes - idx, &d);
            if (a < 0) { hdr_bad = 1; return; }
            idx += (size_t)a;
            a = decode_varbyte(p + idx, B.bytes - idx, &cn);
            if (a < 0) { hdr_bad = 1; return; }
            idx += (size_t)a;
            dsum += d;
            msum += (uint64_t)cn + 1;
            ++cnt;
        }
        if (cnt != B.n) hdr_bad = 1;
        B.dsum = dsum;
        B.mem = msum;
    });
    if (hdr_bad) {
        if (!RS) free(g);
        return CM_EINVAL;
    }
    std::unique_ptr<uint32_t[]> hvs, cnts;                 // not cleared: pass 2 fills them
    const size_t n_hdr = (size_t)nbuckets;
    {
        uint64_t hv0 = 0, at = 0;
        std::vector<uint64_t> start_hv(blks.size()), start_at(blks.size());
        for (size_t b = 0; b < blks.size(); ++b) {
            start_hv[b] = hv0;
            start_at[b] = at;
            hv0 += blks[b].dsum;
            at += blks[b].n;
            mem += blks[b].mem;
        }
        if (hv0 >= nb && nbuckets) {
            if (!RS) free(g);
            return CM_EINVAL;
        }
        if (!genome_only) {
            uint32_t *hv_w, *cn_w;
            if (RS) {
                if (RS->hdr_cap < n_hdr + 1) {
                    free(RS->hv);
                    free(RS->cnt);
                    RS->hv = (uint32_t *)big_alloc((n_hdr + 1) * sizeof(uint32_t));
                    RS->cnt = (uint32_t *)big_alloc((n_hdr + 1) * sizeof(uint32_t));
                    RS->hdr_cap = (RS->hv && RS->cnt) ? n_hdr + 1 : 0;
                }
                hv_w = RS->hv;
                cn_w = RS->cnt;
            } else {
                hvs.reset(new (std::nothrow) uint32_t[n_hdr + 1]);
                cnts.reset(new (std::nothrow) uint32_t[n_hdr + 1]);
                hv_w = hvs.get();
                cn_w = cnts.get();
            }
            if (!hv_w || !cn_w) {
                if (!RS) free(g);
                return CM_ENOMEM;
            }
            over_blocks([&](size_t b) {
                const Blk &B = blks[b];
                const uint8_t *p = hdr.data() + B.off;
                size_t idx = 0;
                uint64_t hv = start_hv[b], w = start_at[b];
                while (idx < B.bytes) {
                    uint32_t d = 0, cn = 0;
                    idx += (size_t)decode_varbyte(p + idx, B.bytes - idx, &d);
                    idx += (size_t)decode_varbyte(p + idx, B.bytes - idx, &cn);
                    hv += d;
                    hv_w[w] = (uint32_t)hv;
                    cn_w[w] = cn;
                    ++w;
                }
            });
        }
    }
    std::vector<uint8_t>().swap(hdr);
    lap("bucket headers decoded");
    const int kmer = x->window + x->checksum_len;
    const int contig_num = atoi(name) - 1;            // contigNum of the mapping loop, src/circminer.cpp:266-267
    int rc = CM_OK;
    if (genome_only) {                                   // the table is stepped over, not decoded (stage 2 needs the sequence only)
        if (x->full) {
            uint32_t memsz = 0;
            if (!get(f, memsz) || fseeko(f, (off_t)memsz * (off_t)sizeof(Entry), SEEK_CUR) != 0) {
                if (!RS) free(g);
                return CM_EINVAL;
            }
        }
        memset(out, 0, sizeof *out);
        out->contig_num = contig_num;
        out->ref_len = n;
        out->genome = g;
    } else if (x->full) {
        uint32_t memsz = 0;
        if (!get(f, memsz) || memsz != mem) {
            if (!RS) free(g);
            return CM_EINVAL;
        }
        // the table itself (8 bytes per slot, ~8.5 GB for a full-size contig): pread()s side by side into a buffer that is kept
        // for the next contig (its pages are faulted in once per file, not once per contig)
        const size_t tab_need = ((size_t)memsz + 1) * sizeof(Entry);
        void *&tab_buf = RS ? RS->tab : x->tab;
        size_t &tab_cap = RS ? RS->tab_cap : x->tab_bytes;
        if (tab_cap < tab_need) {
            free(tab_buf);
            tab_buf = big_alloc(tab_need);
            tab_cap = tab_buf ? tab_need : 0;
        }
        Entry *tab = (Entry *)tab_buf;
        if (!tab) {
            if (!RS) free(g);
            return CM_ENOMEM;
        }
        {
            const off_t at = ftello(f);
            const int fd = fileno(f);
            const size_t total_b = (size_t)memsz * sizeof(Entry);
            const int T = total_b < (64u << 20) ? 1 : std::max(1, std::min(n_threads, 32));
            std::atomic<int> bad{0};
            auto piece = [&](int t) {
                size_t a = total_b * (size_t)t / (size_t)T, b = total_b * (size_t)(t + 1) / (size_t)T;
                while (a < b) {
                    const ssize_t r = pread(fd, (char *)tab + a, std::min<size_t>(b - a, 256u << 20), at + (off_t)a);
                    if (r < 0 && errno == EINTR) continue;
                    if (r <= 0) {
                        bad = 1;
                        return;
                    }
                    a += (size_t)r;
                }
            };
            std::vector<std::thread> th;
            for (int t = 1; t < T; ++t) th.emplace_back(piece, t);
            piece(0);
            for (auto &t : th) t.join();
            if (bad || fseeko(f, at + (off_t)total_b, SEEK_SET) != 0) {
                if (!RS) free(g);
                return CM_EINVAL;
            }
        }
        lap("table read");
        if (RS) {                                      // cm_load_contig_raw flattens it on the device
            raw_out->contig_num = contig_num;
            raw_out->ref_len = n;
            raw_out->genome = g;
            raw_out->n_buckets = (uint32_t)n_hdr;
            raw_out->hv = RS->hv;
            raw_out->count14 = RS->cnt;
            raw_out->table = tab;
            raw_out->table_slots = memsz;
            if (!more) x->done = true;
            *loaded = 1;
            return CM_OK;
        }
        uint32_t *boff = (uint32_t *)calloc(nb + 1, sizeof(uint32_t));
        uint64_t total = 0, cur = 0;
        bool ok = boff != nullptr;
        // the non-empty buckets in T ranges: where each range starts in the table (one serial pass over the counts), then the
        // ranges side by side
        const int TT = std::max(1, std::min(n_threads, 32));
        std::vector<uint64_t> start((size_t)TT + 1, 0);
        {
            uint64_t run = 0;
            int t = 0;
            for (size_t b = 0; b < n_hdr; ++b) {
                while (t < TT && b == n_hdr * (size_t)t / (size_t)TT) start[(size_t)t++] = run;
                run += (uint64_t)cnts[b] + 1;
            }
            while (t <= TT) start[(size_t)t++] = run;
        }
        auto over_ranges = [&](auto &&body) {
            std::vector<std::thread> th;
            for (int t = 1; t < TT; ++t) th.emplace_back(body, t);
            body(0);
            for (auto &x : th) x.join();
        };
        if (ok) {
            std::vector<uint64_t> sum((size_t)TT, 0);
            std::vector<uint8_t> bad((size_t)TT, 0);
            over_ranges([&](int t) {
                uint64_t c0 = start[(size_t)t], s = 0;
                for (size_t b = n_hdr * (size_t)t / (size_t)TT, e = n_hdr * (size_t)(t + 1) / (size_t)TT; b < e; ++b) {
                    const int32_t c = tab[c0].info;
                    if (c < 0 || (uint32_t)c > cnts[b]) {
                        bad[(size_t)t] = 1;
                        return;
                    }
                    boff[hvs[b] + 1] = (uint32_t)c;
                    s += (uint32_t)c;
                    c0 += (uint64_t)cnts[b] + 1;
                }
                sum[(size_t)t] = s;
            });
            for (int t = 0; t < TT; ++t) {
                total += sum[(size_t)t];
                ok = ok && !bad[(size_t)t];
            }
        }
        uint16_t *cs = (uint16_t *)malloc((total ? total : 1) * sizeof(uint16_t));
        uint32_t *ps = (uint32_t *)malloc((total ? total : 1) * sizeof(uint32_t));
        if (!ok || !cs || !ps) {
            free(boff);
            free(cs);
            free(ps);
            if (!RS) free(g);
            return ok ? CM_ENOMEM : CM_EINVAL;
        }
        lap("counts placed");
        {   // inclusive prefix sum of the 4^14 bucket counts: per-thread blocks, block totals, then the offsets added back
            const int T = std::max(1, std::min(n_threads, 32));
            std::vector<uint64_t> tot((size_t)T + 1, 0);
            auto blk = [&](int t, uint64_t &lo, uint64_t &hi) { lo = 1 + nb * (uint64_t)t / (uint64_t)T; hi = 1 + nb * (uint64_t)(t + 1) / (uint64_t)T; };
            auto pass1 = [&](int t) {
                uint64_t lo, hi, run = 0;
                blk(t, lo, hi);
                for (uint64_t h = lo; h < hi; ++h) {
                    run += boff[h];
                    boff[h] = (uint32_t)run;
                }
                tot[(size_t)t + 1] = run;
            };
            auto pass2 = [&](int t) {
                uint64_t lo, hi;
                blk(t, lo, hi);
                const uint32_t add = (uint32_t)tot[(size_t)t];
                if (add)
                    for (uint64_t h = lo; h < hi; ++h) boff[h] += add;
            };
            std::vector<std::thread> th;
            for (int t = 1; t < T; ++t) th.emplace_back(pass1, t);
            pass1(0);
            for (auto &x : th) x.join();
            for (int t = 0; t < T; ++t) tot[(size_t)t + 1] += tot[(size_t)t];
            th.clear();
            for (int t = 1; t < T; ++t) th.emplace_back(pass2, t);
            for (auto &x : th) x.join();
        }
        over_ranges([&](int t) {
            uint64_t c0 = start[(size_t)t];
            for (size_t b = n_hdr * (size_t)t / (size_t)TT, e = n_hdr * (size_t)(t + 1) / (size_t)TT; b < e; ++b) {
                const uint32_t c = (uint32_t)tab[c0].info, w = boff[hvs[b]];
                for (uint32_t k = 0; k < c; ++k) {
                    cs[w + k] = tab[c0 + 1 + k].checksum;
                    ps[w + k] = (uint32_t)tab[c0 + 1 + k].info;
                }
                c0 += (uint64_t)cnts[b] + 1;
            }
        });
        (void)cur;
        lap("prefix sum + entries scattered");
        out->contig_num = contig_num;
        out->ref_len = n;
        out->genome = g;
        out->bucket_off = boff;
        out->checksum = cs;
        out->pos = ps;
        out->n_entries = total;
    } else {
        rc = cm_host_build_index(g, n, kmer, contig_num, n_threads, out);     // calculateHashTableOnFly + sortHashTable
        if (rc != CM_OK) {
            if (!RS) free(g);
            return rc;
        }
    }
    if (!more) x->done = true;
    *loaded = 1;
    return CM_OK;
}

int cm_host_next_contig(cm_index_file *x, int n_threads, cm_index_view *out, int *loaded) { return next_contig(x, n_threads, out, loaded, false); }
int cm_host_next_contig_genome(cm_index_file *x, cm_index_view *out, int *loaded) { return next_contig(x, 1, out, loaded, true); }
int cm_host_next_contig_raw(cm_index_file *x, int n_threads, cm_index_raw *out, int *loaded) {
    return next_contig(x, n_threads, nullptr, loaded, false, out);
}

void cm_host_free_loaded_contig(cm_index_view *iv) {
    if (!iv) return;
    free((void *)iv->genome);
    iv->genome = nullptr;
    cm_host_free_index(iv);
}

void cm_host_close_index(cm_index_file *x) {
    if (!x) return;
    if (x->f) fclose(x->f);
    delete x;
}

}  // extern "C"
