// cm_core.h — per-probe / per-problem / per-pair bodies of the MI355X mapping kernels.
//
// This is the product's device code.  Every function is written for one GPU lane working on flat
// HBM-resident arrays (cm_index_view / cm_annot_view with device pointers): no heap, no STL, no
// recursion, fixed-size private state.  The __global__ wrappers, launch geometry and the C-ABI
// live in cm_hot.hip.  The same bodies also compile as plain C++ so that tests/ can step through
// them on the build box, which has no GPU (tests/hostemu.cpp, never shipped or linked into
// libcmhot.so).
//
// Reference semantics reproduced here (file:line of CircMiner 0.4.5, /root/reference):
//   seeds      src/match_read.cpp:54-110,180-286 ; src/mrsfast/HashTable.c:1093-1098
//   chaining   src/chain.cpp:13-64,73-301 ; src/gene_annotation.h:123-133 ; gene_annotation.cpp:464-533
//   alignment  src/align.cpp:166-252 (edit), :254-390 (X-drop), :395-509 (banded), :556-600, :669-723
//   extension  src/extend.cpp:37-125,131-432,435-875,878-920 ; src/align.h:12-153
//   pairing    src/filter.cpp:124-395,469-551 ; src/utils.cpp:22-320,617-683,827-887
//   ordering   src/common.cpp:286-411
#pragma once
#include <stdint.h>

#include "circminer_hot.h"
#if defined(CM_STAGE2_HOST)
#include <vector>
#endif

#if defined(__HIPCC__)
#define CM_HD __host__ __device__
#define CM_NOINLINE inline
#else
#define CM_HD
#define CM_NOINLINE
#endif

// test-only work counters (defined by tests/hostemu.cpp when it wants them)
#if !defined(CM_HOOK_DP)          // host emulation only (tests/diag/dp_dup.py): every X-drop DP request, before its fast path
#define CM_HOOK_DP(kind, s, n, t, m) ((void)0)
#endif
#if defined(CM_STATS) && !defined(__HIPCC__)
extern "C" unsigned long long cm_stats[16];
#define CM_STAT(i, n) (cm_stats[i] += (unsigned long long)(n))
#else
#define CM_STAT(i, n) ((void)0)
#endif
// diagnostic builds only (-DCM_DIAG): cut the pair routine short at phase n (cm_params.reserved)
#if defined(CM_DIAG)
#define CM_DBG_STOP(n, ret) do { if (c.P.reserved == (n)) return ret; } while (0)
#else
#define CM_DBG_STOP(n, ret) ((void)0)
#endif

// Explicit address spaces for the device build.  Pointers travel through structs and private
// arrays here; without the qualifiers the compiler loses their provenance and emits flat_load /
// flat_store (no ds_*, no global_* with scalar bases), which is several times slower.
#if defined(__HIP_DEVICE_COMPILE__)
#define CM_G __attribute__((address_space(1)))
#define CM_L __attribute__((address_space(3)))
#define CM_S CM_L                                   // DP staging buffers: LDS, word-interleaved per lane
#else
#define CM_G
#define CM_L
#define CM_S
#endif

namespace cmc {

typedef const CM_G uint8_t *g_u8;
typedef const CM_G uint16_t *g_u16;
typedef const CM_G uint32_t *g_u32;
typedef const CM_G int32_t *g_i32;
typedef const CM_G uint64_t *g_u64;
typedef const CM_G cm_chain *g_chain;
typedef CM_G int *g_err;

// device mirrors of cm_index_view / cm_annot_view (same fields, global-qualified pointers)
struct IndexV {
    int32_t contig_num;
    uint32_t ref_len;
    g_u8 genome;
    g_u32 bucket_off;
    g_u16 checksum;
    g_u32 pos;
    uint64_t n_entries;
};
// Device layout of the annotation: array-of-structs records (one 32-byte load serves the five to
// seven parallel arrays of cm_annot_view that a query touches together; 12 pointers instead of 23).
// cm_load_annotation (and the host emulation) repack the caller's SoA view with build_annot_aos().
struct IvRec { uint32_t spos, epos, max_end, min_end, max_next_exon, seg_off, nseg, pad; };
struct SegRec { uint32_t start, end, next_exon_beg, gene_id, tid_off, ntid, pad0, pad1; };
struct TrRec { int32_t start_ind; uint32_t t2s_off, t2s_len, pad; };
struct GeneRec { uint32_t start, end; };
struct AnnotDev {            // plain-pointer form (kernel argument / host side)
    uint32_t n_iv, n_seg, n_trans, n_gene, n_chr, iv_bucket_shift, n_iv_bucket;
    uint32_t pair_reach;         // see pair_code(): no two chains farther apart than this share a transcript or lie in one gene span
    uint64_t n_bits;
    const IvRec *iv;
    const uint32_t *iv_seg;
    const SegRec *seg;
    const uint32_t *seg_tid;
    const TrRec *tr;
    const uint8_t *t2s;
    const GeneRec *gene;
    const uint64_t *near_border_bits, *intronic_bits;
    const uint32_t *chr_shift;
    const int32_t *chr_id;
    const uint32_t *iv_bucket;
};
struct AnnotV {              // the same with global-qualified pointers (device code)
    uint32_t n_iv, n_seg, n_trans, n_gene, n_chr, iv_bucket_shift, n_iv_bucket, pair_reach;
    uint64_t n_bits;
    const CM_G IvRec *iv;
    g_u32 iv_seg;
    const CM_G SegRec *seg;
    g_u32 seg_tid;
    const CM_G TrRec *tr;
    g_u8 t2s;
    const CM_G GeneRec *gene;
    g_u64 near_border_bits, intronic_bits;
    g_u32 chr_shift;
    g_i32 chr_id;
    g_u32 iv_bucket;
};
CM_HD inline IndexV to_dev(const cm_index_view &v) {
    IndexV d;
    d.contig_num = v.contig_num; d.ref_len = v.ref_len; d.n_entries = v.n_entries;
    d.genome = (g_u8)v.genome; d.bucket_off = (g_u32)v.bucket_off; d.checksum = (g_u16)v.checksum; d.pos = (g_u32)v.pos;
    return d;
}
CM_HD inline AnnotV to_dev(const AnnotDev &v) {
    AnnotV d;
    d.n_iv = v.n_iv; d.n_seg = v.n_seg; d.n_trans = v.n_trans; d.n_gene = v.n_gene; d.n_chr = v.n_chr;
    d.iv_bucket_shift = v.iv_bucket_shift; d.n_iv_bucket = v.n_iv_bucket; d.n_bits = v.n_bits; d.pair_reach = v.pair_reach;
    d.iv = (const CM_G IvRec *)v.iv; d.iv_seg = (g_u32)v.iv_seg; d.seg = (const CM_G SegRec *)v.seg; d.seg_tid = (g_u32)v.seg_tid;
    d.tr = (const CM_G TrRec *)v.tr; d.t2s = (g_u8)v.t2s; d.gene = (const CM_G GeneRec *)v.gene;
    d.near_border_bits = (g_u64)v.near_border_bits; d.intronic_bits = (g_u64)v.intronic_bits;
    d.chr_shift = (g_u32)v.chr_shift; d.chr_id = (g_i32)v.chr_id; d.iv_bucket = (g_u32)v.iv_bucket;
    return d;
}

constexpr int INF_I = 1000000000;          // (int)INF, src/common.h:34
constexpr uint32_t MINLB = 0u;
constexpr uint32_t MAXUB = 4294967295u;
constexpr int MAXDISCRDTLEN = 20000;       // src/common.h:40
constexpr uint32_t LARIAT2BEGTH = 1000u;   // src/common.h:53
constexpr int DPTINF = 10000000;           // src/align.cpp:12
constexpr int SC_MAT = 1, SC_MIS = -3, SC_IND = -3, SC_XD = 8;   // score_mat.init(1,-3,-3,8), src/circminer.cpp:74
constexpr int MAX_SEEDS = CM_MAX_CHAIN_FRAGS;   // seeds per read the device path supports
constexpr int MAX_BAND = 8;                 // bandWidth supported by the private DP rows
constexpr int MAX_TID = 64;                 // |common_tid| kept in registers / scratch per mate pair (larger sets: TidList)
#if defined(CM_STAGE2_HOST)
constexpr int MEMO_N = 512;                 // host (stage 2): room for every exon piece of every common transcript
#elif defined(CM_MEMO_N)
constexpr int MEMO_N = CM_MEMO_N;           // test builds (tests/test_gpu_parity.py: a tiny table sends many pairs through the re-run launch)
#else
constexpr int MEMO_N = 8;                   // memoised exon alignments per extend call
#endif

enum { ERR_POOL = 1, ERR_SEEDS = 4, ERR_BAND = 8, ERR_MEMO = 16 };

struct KCore {          // what the host passes as a kernel argument (plain pointers to device memory)
    cm_params P;
    cm_index_view X;
    AnnotDev A;
    const uint32_t *desc = nullptr;       // optional bucket descriptors (desc_pack below), null = none
};
struct Core {
    cm_params P;
    IndexV X;
    AnnotV A;
    g_u32 desc = nullptr;
};
CM_HD inline Core to_core(const KCore &k) {
    Core c;
    c.P = k.P;
    c.X = to_dev(k.X);
    c.A = to_dev(k.A);
    c.desc = (g_u32)k.desc;
    return c;
}

template <class T> CM_HD inline T cmin(T a, T b) { return a < b ? a : b; }
template <class T> CM_HD inline T cmax(T a, T b) { return a > b ? a : b; }
CM_HD inline int cabs(int a) { return a < 0 ? -a : a; }

// ------------------------------------------------------------------------------------------
// string views: reads (either orientation, any slice, optionally reversed), genome windows,
// and the all-NUL window that pac2char(start == 0) yields in the reference.
// ------------------------------------------------------------------------------------------
struct SV {
    g_u8 p;
    int32_t off;
    int32_t step;     // +1 / -1
    int32_t mode;     // 0 plain, 1 complemented, 2 all-NUL
    CM_HD inline SV sub(int a) const { return SV{p, off + a * step, step, mode}; }          // view starting at a
    CM_HD inline SV rev(int m) const { return SV{p, off + (m - 1) * step, -step, mode}; }   // first m chars reversed
};
struct Read {             // one mate in one orientation
    g_u8 p;
    int32_t len;
    int32_t rc;
    CM_HD inline SV view() const { return rc ? SV{p, len - 1, -1, 1} : SV{p, 0, 1, 0}; }
};


// ------------------------------------------------------------------------------------------
// K1 body: one k-mer probe (A1-A3)
// ------------------------------------------------------------------------------------------
struct Probe {
    uint32_t start;     // first hit in the entry arrays
    uint32_t raw;       // occurrences (0 == frags NULL)
    uint32_t touches;   // binary-search element touches (algorithmic-byte counter)
};
// The two binary searches of get_exact_locs_hash (src/match_read.cpp:54-110) over the n checksums of one bucket, element i
// (1-based in the reference) read through at(i - 1).  Counts every element looked at, like the reference's loops.
template <class AT> CM_HD inline void probe_search(const AT &at, uint32_t n, int target, uint32_t b0, Probe &r) {
    uint32_t lb = 1, ub = n, mid;
    while (lb < ub) {
        mid = (lb + ub) / 2;
        ++r.touches;
        if (target <= at(mid - 1)) ub = mid;
        else lb = mid + 1;
    }
    ++r.touches;
    if (ub < lb || target != at(lb - 1)) return;
    const uint32_t LB = lb;
    uint32_t UB = lb;
    ub = n;
    while (lb < ub) {
        mid = (lb + ub + 1) / 2;
        ++r.touches;
        if (target < at(mid - 1)) ub = mid - 1;
        else lb = mid;
    }
    ++r.touches;
    if (target == at(lb - 1)) UB = lb;
    r.start = b0 + (LB - 1);
    r.raw = UB - LB + 1;
}
// Bucket descriptors (device-side acceleration of the probe, built from the index arrays when a contig is loaded): 16 bytes per
// hash bucket = offset of its first entry, min(n, 255), and -- when they fit -- the checksums of all its entries, so that a probe
// is ONE random 16-byte read instead of a read of the offset table followed by reads of the checksum array.  A checksum has
// 2 (k - 14) bits (k = 20: 12 bits, 7 of them fit; a 1.1-Gbp contig fills a bucket with 4 k-mers on average).  Larger buckets
// keep going through the arrays.
constexpr int DESC_WORDS = 4;
CM_HD inline int desc_cbits(int kmer) { return 2 * (kmer - CM_WINDOW_SIZE); }
CM_HD inline uint32_t desc_inline(int kmer) { return desc_cbits(kmer) > 0 ? (uint32_t)(88 / desc_cbits(kmer)) : 254u; }
CM_HD inline int desc_value(uint32_t d1, uint32_t d2, uint32_t d3, uint32_t i, int cbits) {
    const uint32_t sh = 8u + i * (uint32_t)cbits, w = sh >> 5, bit = sh & 31u;
    const uint32_t lo = w == 0 ? d1 : (w == 1 ? d2 : d3), hi = w == 0 ? d2 : (w == 1 ? d3 : 0u);
    const uint64_t two = ((uint64_t)hi << 32) | lo;
    return (int)((uint32_t)(two >> bit) & ((1u << cbits) - 1u));
}
template <class CK> CM_HD inline void desc_pack(uint32_t b0, uint32_t n, const CK &cks /* cks(i) = checksum of entry b0 + i */, int kmer, uint32_t out[DESC_WORDS]) {
    const int cbits = desc_cbits(kmer);
    uint64_t lo = n < 255u ? n : 255u;      // bits 0..63 of the 96-bit field, hi = bits 64..95
    uint32_t hi = 0;
    if (n <= desc_inline(kmer) && cbits > 0)
        for (uint32_t i = 0; i < n; ++i) {
            const uint32_t sh = 8u + i * (uint32_t)cbits;
            const uint64_t v = (uint64_t)(cks(i) & ((1u << cbits) - 1u));
            if (sh < 64u) {
                lo |= v << sh;
                if (sh + (uint32_t)cbits > 64u) hi |= (uint32_t)(v >> (64u - sh));
            } else hi |= (uint32_t)(v << (sh - 64u));
        }
    out[0] = b0;
    out[1] = (uint32_t)lo;
    out[2] = (uint32_t)(lo >> 32);
    out[3] = hi;
}
CM_HD inline Probe seed_probe(const Core &c, const SV &s, int qpos) {
    Probe r{0u, 0u, 0u};
    // hashVal / checkSumVal without branches (a 4-way switch per base made k_seed scalar-ALU bound): a base is
    // valid iff it is upper-case A/C/G/T *after* the orientation's transform -- the forward strand is taken as
    // is, the reverse strand went through FASTQParser::set_comp, which also upper-cases (src/fastq_parser.cpp:141-153)
    int hv = 0, cv = 0;
    uint32_t bad = 0;
    const uint32_t rc = s.mode == 1 ? 1u : 0u;
    for (int i = 0; i < c.P.kmer; ++i) {
        uint32_t u = s.mode == 2 ? 0u : (uint32_t)s.p[s.off + (qpos + i) * s.step];
        u = rc ? (u & 0xDFu) : u;
        bad |= (uint32_t)!((u == 'A') | (u == 'C') | (u == 'G') | (u == 'T'));
        uint32_t code = ((u >> 1) ^ (u >> 2)) & 3u;                  // A 0, C 1, G 2, T 3
        code = rc ? 3u - code : code;
        if (i < CM_WINDOW_SIZE) hv = (hv << 2) | (int)code;
        else cv = (cv << 2) | (int)code;
    }
    if (bad) return r;
    const int target = (int)(int16_t)cv;     // int16 quirk, src/match_read.cpp:77
    if (c.desc) {
        const g_u32 d = c.desc + (uint64_t)(uint32_t)hv * DESC_WORDS;
        uint32_t w[DESC_WORDS];
        __builtin_memcpy(w, (const CM_G uint8_t *)d, sizeof(w));     // one 16-byte load
        const uint32_t n8 = w[1] & 0xffu;
        if (n8 == 0) return r;
        if (n8 <= desc_inline(c.P.kmer)) {
            const int cbits = desc_cbits(c.P.kmer);
            const uint32_t d1 = w[1], d2 = w[2], d3 = w[3];
            probe_search([&](uint32_t i) { return cbits > 0 ? desc_value(d1, d2, d3, i, cbits) : 0; }, n8, target, w[0], r);
            return r;
        }
    }
    const uint32_t b0 = c.X.bucket_off[hv], b1 = c.X.bucket_off[hv + 1];
    if (b1 == b0) return r;
    const g_u16 it = c.X.checksum + b0;
    probe_search([&](uint32_t i) { return (int)it[i]; }, b1 - b0, target, b0, r);
    return r;
}

// ------------------------------------------------------------------------------------------
// annotation queries (A7)
// ------------------------------------------------------------------------------------------
CM_HD inline bool bit_at(g_u64 b, uint64_t n, uint64_t p) { return p < n && ((b[p >> 6] >> (p & 63)) & 1ull); }
CM_HD inline uint32_t iv_nseg(const AnnotV &A, int iv) { return A.iv[iv].nseg; }
CM_HD inline uint32_t iv_segid(const AnnotV &A, int iv, uint32_t i) { return A.iv_seg[A.iv[iv].seg_off + i]; }

CM_HD inline int iv_find_ind(const AnnotV &A, uint32_t pos, int &ind) {   // interval_tree_impl.h:136-175
    CM_STAT(10, 1);
    ind = -1;
    if (pos < A.iv[0].spos) return -1;
    // FlatIntervalTree::search returns the number of intervals whose spos <= pos; narrow the range
    // with the bucket table when the caller supplied one (same result, ~2 probes instead of log2 n)
    int beg = 0, end = (int)A.n_iv;
    const uint32_t b = pos >> A.iv_bucket_shift;
    if (A.iv_bucket && b + 1 < A.n_iv_bucket) {
        beg = (int)A.iv_bucket[b];
        end = (int)A.iv_bucket[b + 1];
        if (beg > 0) --beg;                     // keep the invariant spos[beg] <= pos (spos[0] <= pos was checked)
        if (end < (int)A.n_iv) ++end;
        if (end <= beg) end = beg + 1;
    }
    while (end - beg > 1) {
        const int mid = (beg + end) / 2;
        if (pos < A.iv[mid].spos) end = mid;
        else beg = mid;
    }
    ind = end - 1;
    if (ind < 0 || A.iv[ind].epos < pos) return -1;
    return ind;
}
CM_HD inline int overlap_ind(const Core &c, uint32_t loc, int &ind) {            // gene_annotation.cpp:555-568
    int r = iv_find_ind(c.A, loc, ind);
    if (r < 0 || iv_nseg(c.A, r) == 0) return -1;
    return r;
}
CM_HD inline int overlap(const Core &c, uint32_t loc) { int ind; return overlap_ind(c, loc, ind); }

CM_HD inline uint32_t upper_bound_lookup(const Core &c, uint32_t spos, uint32_t mlen, uint32_t rlen, uint32_t &max_end, int &ol) {
    const AnnotV &A = c.A;
    max_end = 0;
    int it_ind = -1;
    int ov = iv_find_ind(A, spos, it_ind);
    const uint32_t epos = spos + mlen - 1;
    if (ov < 0 || iv_nseg(A, ov) == 0) {
        ol = -1;
        int nx = it_ind + 1;
        max_end = ((nx < 0 || nx >= (int)A.n_iv) ? 0u : A.iv[nx].spos) - 1u;
        if (max_end < epos) return 0;
        return cmin(spos + rlen + (uint32_t)c.P.max_ed, max_end - mlen + 1);
    }
    ol = -1;
    uint32_t min_end = 1000000000u, max_next = 0;
    if (epos > A.iv[ov].epos) {
        const uint32_t n = iv_nseg(A, ov);
        for (uint32_t i = 0; i < n; ++i) {
            uint32_t s = iv_segid(A, ov, i);
            uint32_t e = A.seg[s].end;
            if (e >= epos) {
                max_end = cmax(max_end, e);
                min_end = cmin(min_end, e);
                max_next = cmax(max_next, A.seg[s].next_exon_beg);
            }
        }
    } else {
        max_end = A.iv[ov].max_end;
        min_end = A.iv[ov].min_end;
        max_next = A.iv[ov].max_next_exon;
    }
    if (max_end > 0 && max_end >= epos) {
        ol = ov;
        if (min_end < rlen + epos && max_next != 0) return max_next + mlen - 1;
        return max_end - mlen + 1;
    }
    max_end = 0;
    ol = -1;
    return 0;
}
CM_HD inline uint32_t upper_bound(const Core &c, uint32_t spos, uint32_t mlen, uint32_t rlen, uint32_t &max_end, int &ol) {
    if (bit_at(c.A.near_border_bits, c.A.n_bits, spos)) return upper_bound_lookup(c, spos, mlen, rlen, max_end, ol);
    max_end = 0;
    ol = -1;
    return spos + rlen + (uint32_t)c.P.max_ed;
}
CM_HD inline int chr_row(const Core &c, uint32_t loc) {      // GTFParser::get_shift
    uint32_t i;
    for (i = 1; i < c.A.n_chr; ++i)
        if (loc < c.A.chr_shift[i]) return (int)i - 1;
    return (int)i - 1;
}

// ------------------------------------------------------------------------------------------
// K2 body: k-best chaining of one (read, orientation) (A6)
// ------------------------------------------------------------------------------------------
struct Event { double score; uint32_t cell; uint32_t pad; };
struct ChainWork {
    CM_G double *dp_score;            // this problem's cells
    CM_G int32_t *dp_prev;            // (list << 16 | ind) or -1
    CM_G uint8_t *pool;               // event pool shared by the launch
    unsigned long long pool_bytes;
    CM_G unsigned long long *pool_cursor;
    g_err err;
};

#if defined(__HIP_DEVICE_COMPILE__)
CM_HD inline unsigned long long pool_take(CM_G unsigned long long *cur, unsigned long long n) { return atomicAdd((unsigned long long *)cur, n); }
CM_HD inline void flag_err(g_err e, int bits) { atomicOr((int *)e, bits); }
#else
inline unsigned long long pool_take(unsigned long long *cur, unsigned long long n) { return __sync_fetch_and_add(cur, n); }
inline void flag_err(int *e, int bits) { __sync_fetch_and_or(e, bits); }
#endif

CM_HD inline bool check_junction(const Core &c, uint32_t s1, uint32_t s2, int ol, int kmer, int read_dist, int &trans_dist) {
    trans_dist = INF_I;
    if (ol < 0) return false;
    const uint32_t e1 = s1 + kmer - 1;
    if (s2 <= e1) return false;
    int td2intron = -1;
    const AnnotV &A = c.A;
    const uint32_t n = iv_nseg(A, ol);
    for (uint32_t i = 0; i < n; ++i) {
        const uint32_t s = iv_segid(A, ol, i);
        const int e12end = (int)(A.seg[s].end - e1);
        const int beg2s2 = (int)(s2 - A.seg[s].next_exon_beg);
        if (e12end >= 0 && e12end < read_dist && beg2s2 + kmer < 0) td2intron = (int)(s2 - e1 - 1);
        if (e12end < 0 || beg2s2 < 0) continue;
        trans_dist = e12end + beg2s2;
        if (cabs(trans_dist - read_dist) <= c.P.max_ed) return true;
    }
    if (td2intron != -1) {
        trans_dist = td2intron;
        return true;
    }
    trans_dist = INF_I;
    return false;
}

// Storage policies of the chaining DP.
//  * ChainStoreGlobal: cells (score, back pointer) in the launch's HBM workspace, hit positions read from
//    the index, improvement log in the shared pool (grows by x4, flags ERR_POOL when the pool is exhausted);
//  * ChainStoreSmall: everything in lane-private arrays — for problems with at most SMALL_W (hit, later
//    hit) pairs, i.e. at most SMALL_W improvements and SMALL_W + 1 cells: >95 % of all problems, and no
//    global atomics, no scattered HBM cells for them.
constexpr int SMALL_W = 32;
struct ChainStoreGlobal {
    ChainWork &w;
    g_u32 POS;
    const uint32_t *start;
    const uint32_t *cnt_;
    const uint32_t *base_;
    uint32_t lb_[MAX_SEEDS];
    CM_G Event *ev;
    uint32_t n_ev, cap_ev;
    bool lost;
    CM_HD ChainStoreGlobal(ChainWork &ww, g_u32 pos, const uint32_t *st, const uint32_t *cn, const uint32_t *bs)
        : w(ww), POS(pos), start(st), cnt_(cn), base_(bs), ev(nullptr), n_ev(0), cap_ev(0), lost(false) {}
    CM_HD inline uint32_t cnt(int s) const { return cnt_[s]; }
    CM_HD inline uint32_t base(int s) const { return base_[s]; }
    CM_HD inline uint32_t total(int kc) const { return base_[kc]; }
    CM_HD inline void lb_reset(int kc) { for (int k = 0; k < kc; ++k) lb_[k] = 0; }
    CM_HD inline uint32_t lb(int s) const { return lb_[s]; }
    CM_HD inline void set_lb(int s, uint32_t v) { lb_[s] = v; }
    CM_HD inline void init(uint32_t n_cells, double v) {
        for (uint32_t x = 0; x < n_cells; ++x) {
            w.dp_score[x] = v;
            w.dp_prev[x] = -1;
        }
    }
    CM_HD inline uint32_t pos(int s, uint32_t i) const { return POS[start[s] + i]; }
    CM_HD inline double score(uint32_t x) const { return w.dp_score[x]; }
    CM_HD inline int32_t prev(uint32_t x) const { return w.dp_prev[x]; }
    CM_HD inline void set(uint32_t x, double sc, int32_t pv) {
        w.dp_score[x] = sc;
        w.dp_prev[x] = pv;
    }
    CM_HD inline void push(double sc, uint32_t cell) {
        if (n_ev == cap_ev && !lost) {
            const uint32_t ncap = cap_ev ? cap_ev * 4u : 32u;
            const unsigned long long bytes = (unsigned long long)ncap * sizeof(Event);
            const unsigned long long off = pool_take(w.pool_cursor, bytes);
            if (off + bytes > w.pool_bytes) {
                lost = true;
                flag_err(w.err, ERR_POOL);
            } else {
                CM_G Event *ne = (CM_G Event *)(w.pool + off);
                for (uint32_t q = 0; q < n_ev; ++q) ne[q] = ev[q];
                ev = ne;
                cap_ev = ncap;
            }
        }
        if (n_ev < cap_ev) {
            ev[n_ev].score = sc;
            ev[n_ev].cell = cell;
            ++n_ev;
        }
    }
    CM_HD inline uint32_t n_events() const { return n_ev; }
    CM_HD inline double ev_score(uint32_t q) const { return ev[q].score; }
    CM_HD inline uint32_t ev_cell(uint32_t q) const { return ev[q].cell; }
};
struct ChainStoreSmall {
    double sc[SMALL_W + 2];
    int32_t pv[SMALL_W + 2];
    uint32_t hp[SMALL_W + 2];
    double es[SMALL_W];
    uint32_t ec[SMALL_W];
    const uint32_t *cnt_;
    const uint32_t *base_;
    uint32_t lb_[MAX_SEEDS];
    uint32_t n_ev;
    CM_HD ChainStoreSmall(g_u32 POS, const uint32_t *start, const uint32_t *cn, const uint32_t *bs, int kc) : cnt_(cn), base_(bs), n_ev(0) {
        for (int s = 0; s < kc; ++s)
            for (uint32_t i = 0; i < cn[s]; ++i) hp[bs[s] + i] = POS[start[s] + i];
    }
    CM_HD inline uint32_t cnt(int s) const { return cnt_[s]; }
    CM_HD inline uint32_t base(int s) const { return base_[s]; }
    CM_HD inline uint32_t total(int kc) const { return base_[kc]; }
    CM_HD inline void lb_reset(int kc) { for (int k = 0; k < kc; ++k) lb_[k] = 0; }
    CM_HD inline uint32_t lb(int s) const { return lb_[s]; }
    CM_HD inline void set_lb(int s, uint32_t v) { lb_[s] = v; }
    CM_HD inline void init(uint32_t n_cells, double v) {
        for (uint32_t x = 0; x < n_cells; ++x) {
            sc[x] = v;
            pv[x] = -1;
        }
    }
    CM_HD inline uint32_t pos(int s, uint32_t i) const { return hp[base_[s] + i]; }
    CM_HD inline double score(uint32_t x) const { return sc[x]; }
    CM_HD inline int32_t prev(uint32_t x) const { return pv[x]; }
    CM_HD inline void set(uint32_t x, double v, int32_t p) {
        sc[x] = v;
        pv[x] = p;
    }
    CM_HD inline void push(double v, uint32_t cell) {
        if (n_ev < (uint32_t)SMALL_W) {       // cannot overflow: improvements <= (hit, later hit) pairs <= SMALL_W
            es[n_ev] = v;
            ec[n_ev] = cell;
            ++n_ev;
        }
    }
    CM_HD inline uint32_t n_events() const { return n_ev; }
    CM_HD inline double ev_score(uint32_t q) const { return es[q]; }
    CM_HD inline uint32_t ev_cell(uint32_t q) const { return ec[q]; }
};

// seeds: ordinal s has qpos s*kmer, cnt[s] hits (positions through the store).  Returns #chains.
template <class ST>
CM_HD inline int chain_kbest_t(const Core &c, int seq_len, int kc, ST &S, CM_G cm_chain *out) {
    const int kmer = c.P.kmer;
    const uint32_t max_best = (uint32_t)c.P.max_chain_len;
    S.init(S.total(kc), (double)kmer);
    uint32_t max_exon_end = 0;
    int ol = -1;

    for (int ii = kc - 2; ii >= 0; --ii) {
        const uint32_t read_remain = (uint32_t)(seq_len - ii * kmer - kmer);
        S.lb_reset(kc);
        const uint32_t cnt_ii = S.cnt(ii), base_ii = S.base(ii);
        for (uint32_t i = 0; i < cnt_ii; ++i) {
            const int32_t cur_info = (int32_t)S.pos(ii, i);
            const uint32_t seg_start = (uint32_t)cur_info, seg_end = (uint32_t)cur_info + kmer - 1;
            uint32_t max_lpos_lim = MAXUB;
            double my_score = S.score(base_ii + i);
            for (int jj = ii + 1; jj < kc; ++jj) {
                const uint32_t pcn = S.cnt(jj);
                uint32_t lb = S.lb(jj);
                if (pcn == 0 || lb >= pcn) continue;
                uint32_t pinfo = S.pos(jj, lb);
                if (cur_info + c.P.max_intron < (int32_t)pinfo) continue;
                while ((int32_t)pinfo <= cur_info) {
                    if (++lb >= pcn) break;
                    pinfo = S.pos(jj, lb);
                }
                S.set_lb(jj, lb);
                if (lb >= pcn) continue;
                if (max_lpos_lim == MAXUB) max_lpos_lim = upper_bound(c, seg_start, (uint32_t)kmer, read_remain, max_exon_end, ol);
                const int distr = (jj - ii) * kmer - kmer;
                const uint32_t base_jj = S.base(jj);
                uint32_t j = lb;
                for (; j < pcn; ++j) {
                    if (j != lb) pinfo = S.pos(jj, j);
                    if (pinfo > max_lpos_lim) break;
                    int genome_dist, distt, trans_dist;
                    if (max_exon_end == 0 || (pinfo + kmer - 1) <= max_exon_end) genome_dist = (int)(pinfo - seg_end - 1);
                    else genome_dist = INF_I;
                    if (cabs(genome_dist - distr) <= c.P.max_ed) {
                        distt = genome_dist;
                    } else if (check_junction(c, seg_start, pinfo, ol, kmer, distr, trans_dist)) {
                        distt = trans_dist;
                    } else {
                        continue;
                    }
                    const int maxd = distr < distt ? distt : distr, mind = distr < distt ? distr : distt;
                    const double beta = 0.1 * (double)(maxd - mind);
                    const double alpha = 2e4 * (double)kmer;
                    const double t1 = S.score(base_jj + j) + alpha;     // (prev + alpha) - beta, no contraction
                    const double temp_score = t1 - beta;
                    if (temp_score > my_score) {
                        my_score = temp_score;
                        S.set(base_ii + i, temp_score, (int32_t)(((uint32_t)jj << 16) | j));
                        S.push(temp_score, ((uint32_t)ii << 16) | i);
                    }
                }
            }
        }
    }

    // back-tracking, src/chain.cpp:242-281: score groups descending, insertion order inside a group,
    // at most max_best cells per group (the cap applied at insertion time in the reference).
    uint32_t best_count = 0;
    const uint32_t n_ev = S.n_events();
    if (n_ev > 0) {
        double best_score = S.ev_score(0);
        for (uint32_t q = 1; q < n_ev; ++q) best_score = S.ev_score(q) > best_score ? S.ev_score(q) : best_score;
        double cur = best_score;
        bool have = true;
        while (have && best_count < max_best) {
            uint32_t in_group = 0;
            for (uint32_t q = 0; q < n_ev && best_count < max_best; ++q) {
                if (S.ev_score(q) != cur) continue;
                if (in_group >= max_best) break;
                ++in_group;
                const uint32_t cell = S.ev_cell(q);
                int bl = (int)(cell >> 16);
                uint32_t bi = cell & 0xffffu;
                const uint32_t spos = S.pos(bl, bi);
                if (cur < best_score) {
                    bool rep = false;     // repeats: rpos of any non-first fragment already emitted
                    for (uint32_t a = 0; a < best_count && !rep; ++a)
                        for (uint32_t b = 1; b < out[a].chain_len; ++b)
                            if (out[a].rpos[b] == spos) { rep = true; break; }
                    if (rep) continue;
                }
                CM_G cm_chain &ch = out[best_count++];
                uint32_t n = 0;
                while (true) {
                    ch.rpos[n] = S.pos(bl, bi);
                    ch.qpos[n] = bl * kmer;
                    ++n;
                    const int32_t pv = S.prev(S.base(bl) + bi);
                    if (pv < 0) break;
                    bl = (int)((uint32_t)pv >> 16);
                    bi = (uint32_t)pv & 0xffffu;
                }
                ch.score = (float)cur;
                ch.chain_len = n;
            }
            // next lower score
            have = false;
            double nxt = 0;
            for (uint32_t q = 0; q < n_ev; ++q)
                if (S.ev_score(q) < cur && (!have || S.ev_score(q) > nxt)) {
                    nxt = S.ev_score(q);
                    have = true;
                }
            cur = nxt;
        }
    }
    if (best_count == 0) {      // singletons, src/chain.cpp:283-298
        for (int ii = kc - 1; ii >= 0; --ii)
            for (uint32_t i = 0; i < S.cnt(ii); ++i) {
                if (best_count >= max_best) break;
                CM_G cm_chain &ch = out[best_count++];
                ch.rpos[0] = S.pos(ii, i);
                ch.qpos[0] = ii * kmer;
                ch.score = (float)S.score(S.base(ii) + i);
                ch.chain_len = 1;
            }
    }
    return (int)best_count;
}

CM_HD inline int chain_kbest(const Core &c, int seq_len, int n_seeds, const uint32_t *start, const uint32_t *cnt, ChainWork &w, CM_G cm_chain *out) {
    int kc = n_seeds;
    while (kc >= 1 && cnt[kc - 1] == 0) --kc;
    if (kc <= 0) return 0;
    uint32_t base[MAX_SEEDS + 1];
    base[0] = 0;
    unsigned long long pairs = 0;
    for (int s = 0; s < kc; ++s) {
        pairs += (unsigned long long)cnt[s] * base[s];      // hits of s x hits of all earlier seeds
        base[s + 1] = base[s] + cnt[s];
    }
    if (pairs <= (unsigned long long)SMALL_W && base[kc] <= (uint32_t)SMALL_W + 1u) {
        ChainStoreSmall S(c.X.pos, start, cnt, base, kc);
        return chain_kbest_t(c, seq_len, kc, S, out);
    }
    ChainStoreGlobal S(w, c.X.pos, start, cnt, base);
    return chain_kbest_t(c, seq_len, kc, S, out);
}

// ------------------------------------------------------------------------------------------
// alignment (A14, A15)
// ------------------------------------------------------------------------------------------
struct AlignRes { uint32_t pos; int ed, sclen, indel, qcovlen, rcovlen, score; };
struct MemoKey { uint32_t rspos, rlen, qspos, qlen; };
struct MemoSpill { MemoKey k; AlignRes v; };     // one overflow entry of the extension memo, in global memory (see Memo)
typedef CM_G MemoSpill *g_spill;
CM_HD inline AlignRes ar_init(uint32_t p) { return AlignRes{p, 0, 0, 0, 0, 0, -INF_I}; }
CM_HD inline void ar_set(AlignRes &a, uint32_t p, int e, int s, int i, int qc, int scr) {
    a.pos = p; a.ed = e; a.sclen = s; a.indel = i; a.qcovlen = qc; a.rcovlen = qc - i; a.score = scr;
}
CM_HD inline void ar_update(AlignRes &a, int e, int s, uint32_t np, int i, int qc, int scr) {
    a.pos = np; a.ed += e; a.sclen = s; a.indel += i; a.qcovlen += qc; a.rcovlen += qc - i; a.score = scr;
}
CM_HD inline bool ar_by_score(AlignRes &b, const AlignRes &r, bool right) {
    const bool pos_better = right ? (r.pos < b.pos) : (r.pos > b.pos);
    if (b.score < r.score || (b.score == r.score && pos_better)) {
        ar_set(b, r.pos, r.ed, r.sclen, r.indel, r.qcovlen, r.score);
        return true;
    }
    return false;
}
CM_HD inline void ar_side(const Core &c, AlignRes &b, const AlignRes &r, bool right) {
    bool take = false;
    if (r.qcovlen > b.qcovlen) {
        take = r.ed <= c.P.max_ed && r.sclen <= c.P.max_sc && 2 * (r.ed - b.ed) < (r.qcovlen - b.qcovlen);
    } else if (r.qcovlen < b.qcovlen) {
        take = r.ed <= c.P.max_ed && r.sclen <= c.P.max_sc && 2 * (b.ed - r.ed) >= (b.qcovlen - r.qcovlen);
    } else {
        const bool pos_better = right ? (r.pos < b.pos) : (r.pos > b.pos);
        take = (r.ed < b.ed) || (r.ed == b.ed && r.sclen < b.sclen) || (r.ed == b.ed && r.sclen == b.sclen && pos_better);
    }
    if (take) ar_set(b, r.pos, r.ed, r.sclen, r.indel, r.qcovlen, r.score);
}
struct Cand { int ed, sclen, indel, score; };
CM_HD inline bool cand_less(const Cand &a, const Cand &b) {     // AlignCandid::operator<, src/align.h:132-139
    if (a.score != b.score) return a.score > b.score;
    if (a.ed != b.ed) return a.ed < b.ed;
    return cabs(a.indel) < cabs(b.indel);
}

// ---- per-lane staged strings -------------------------------------------------------------
// Every DP first stages its two strings as base codes (A0 C1 G2 T3 case-insensitive, anything
// else 4 in the first string and 5 in the second, so "other" never matches anything — N vs N
// included, src/align.cpp:745-759) into a private buffer, already in the order the DP walks them
// (the left-hand variants stage reversed views).  On the GPU the buffer is LDS, word-interleaved
// across the 64 lanes of the wave (code i of lane l lives in word (i/8)*64 + l), so concurrent
// per-lane accesses fall in distinct banks; on the host build it is a plain array.
#if defined(__HIP_DEVICE_COMPILE__)
constexpr int LSTRIDE = 64;
#else
constexpr int LSTRIDE = 1;
#endif
// Codes are < 8, so the buffer keeps them as nibbles, eight per 32-bit word (half the LDS of a byte per code:
// the staging buffers are what limits the pair kernels to two waves per SIMD otherwise).  `cap` counts
// characters; the storage is cap / 8 + 1 words (window() may touch one word past the last code).
struct LBuf {
    CM_S uint8_t *b;
    int cap;
    CM_HD inline uint32_t word(int w) const {                 // codes 8w .. 8w+7, code 8w in the low nibble
#if defined(__HIP_DEVICE_COMPILE__)
        return ((CM_S const uint32_t *)b)[w * LSTRIDE];
#else
        uint32_t x;
        __builtin_memcpy(&x, b + 4 * w, 4);
        return x;
#endif
    }
    CM_HD inline uint8_t get(int i) const { return (uint8_t)((word(i >> 3) >> (4 * (i & 7))) & 0xFu); }
    // 8 consecutive codes starting at index `at` (0 <= at <= cap - 8), one per byte: code at+r in bits 8r..8r+7
    CM_HD inline uint64_t window(int at) const {
        const int w = at >> 3;
        const uint64_t two = ((uint64_t)word(w + 1) << 32) | (uint64_t)word(w);
        uint64_t x = (uint32_t)(two >> (4 * (at & 7)));      // 8 nibbles
        x = (x | (x << 16)) & 0x0000FFFF0000FFFFull;
        x = (x | (x << 8)) & 0x00FF00FF00FF00FFull;
        x = (x | (x << 4)) & 0x0F0F0F0F0F0F0F0Full;
        return x;
    }
    CM_HD inline void put(int i, uint8_t v) const {           // host staging only (the device stores whole words)
        uint32_t x;
        __builtin_memcpy(&x, (const uint8_t *)b + 4 * (i >> 3) * LSTRIDE, 4);
        x = (x & ~(0xFu << (4 * (i & 7)))) | ((uint32_t)(v & 0xFu) << (4 * (i & 7)));
        __builtin_memcpy((uint8_t *)b + 4 * (i >> 3) * LSTRIDE, &x, 4);
    }
};
// four one-byte codes -> four nibbles (16 bits)
CM_HD inline uint32_t pack_nibbles(uint32_t x) { return (x & 0xFu) | ((x >> 4) & 0xF0u) | ((x >> 8) & 0xF00u) | ((x >> 12) & 0xF000u); }
// 4 ASCII bases -> 4 one-byte codes (SWAR).  A/a 0, C/c 1, T/t 2, G/g 3 (= (ch >> 1) & 3 of the
// upper-cased byte; only equality of codes matters to the DPs), complement = code ^ 2, any other
// byte -> `other`.
CM_HD inline uint32_t swar_zero_bytes(uint32_t t) { return ~(((t & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | t | 0x7F7F7F7Fu); }   // 0x80 where byte == 0
CM_HD inline uint32_t code4(uint32_t w, bool comp, uint8_t other) {
    const uint32_t x = w & 0xDFDFDFDFu;
    const uint32_t valid = swar_zero_bytes(x ^ 0x41414141u) | swar_zero_bytes(x ^ 0x43434343u) | swar_zero_bytes(x ^ 0x47474747u) |
                           swar_zero_bytes(x ^ 0x54545454u);
    const uint32_t m = (valid >> 7) * 0xFFu;                  // 0xFF in valid bytes
    uint32_t code = (x >> 1) & 0x03030303u;
    if (comp) code ^= 0x02020202u;
    return (code & m) | ((0x01010101u * other) & ~m);
}
CM_HD inline uint8_t code1(uint8_t ch, bool comp, uint8_t other) { return (uint8_t)(code4(ch, comp, other) & 0xFFu); }

// Copy the first n characters of view v into the staging buffer as codes.
// Device build: 16-byte global loads (the caller guarantees CM_STAGE_PAD readable bytes on both
// sides of every global string, see cm_hot.hip) + whole-word LDS stores.
constexpr int CM_STAGE_PAD = 64;
#if defined(__HIP_DEVICE_COMPILE__)
// codes of characters 4*w0 .. 4*w0+15 of view v, four per word
CM_HD inline void load_codes16(const SV &v, int w0, uint8_t other, uint32_t q[4]) {
    if (v.mode == 2) {
#pragma unroll
        for (int k = 0; k < 4; ++k) q[k] = 0x01010101u * other;
        return;
    }
    uint32_t r[4];
    if (v.step > 0) {
        __builtin_memcpy(r, (const CM_G uint8_t *)(v.p + v.off + 4 * w0), 16);
#pragma unroll
        for (int k = 0; k < 4; ++k) q[k] = code4(r[k], v.mode == 1, other);
    } else {                                                   // chars 4*w0 .. 4*w0+15 live at p[off - i]
        __builtin_memcpy(r, (const CM_G uint8_t *)(v.p + v.off - 4 * w0 - 15), 16);
#pragma unroll
        for (int k = 0; k < 4; ++k) q[k] = code4(__builtin_bswap32(r[3 - k]), v.mode == 1, other);
    }
}
#endif
CM_HD inline void stage(const SV &v, int n, const LBuf &d, uint8_t other) {
    CM_STAT(8, 1);
#if defined(CM_ABL_NOSTAGE)     // ablation study only: no staging either
    return;
#endif
#if defined(__HIP_DEVICE_COMPILE__)
    CM_S uint32_t *dw = (CM_S uint32_t *)d.b;
    const int nw = (n + 7) >> 3;                   // words of eight codes
    for (int w0 = 0; w0 < nw; w0 += 8) {           // four 16-character loads in flight per trip to memory (they used to go one by one)
        uint32_t q[4][4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (w0 + 2 * u < nw) load_codes16(v, 2 * (w0 + 2 * u), other, q[u]);         // characters 8*w .. 8*w+15, w = w0 + 2u
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int w = w0 + 2 * u;
            if (w < nw) dw[w * LSTRIDE] = pack_nibbles(q[u][0]) | (pack_nibbles(q[u][1]) << 16);
            if (w + 1 < nw) dw[(w + 1) * LSTRIDE] = pack_nibbles(q[u][2]) | (pack_nibbles(q[u][3]) << 16);
        }
    }
#else
    for (int i = 0; i < n; ++i) d.put(i, v.mode == 2 ? other : code1(v.p[v.off + i * v.step], v.mode == 1, other));
#endif
}
// number of positions i < len where a[i] and b[i] are not the same base (N never matches)
CM_HD inline int prefix_mismatches(const SV &a, const SV &b, int len) {
    int mm = 0;
#if defined(__HIP_DEVICE_COMPILE__)
    for (int w0 = 0; 4 * w0 < len; w0 += 4) {
        uint32_t qa[4], qb[4];
        load_codes16(a, w0, 4, qa);
        load_codes16(b, w0, 5, qb);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int base = 4 * (w0 + k);
            if (base >= len) continue;
            uint32_t x = qa[k] ^ qb[k];
            x = (x | (x >> 1) | (x >> 2)) & 0x01010101u;                   // 1 per differing byte (codes < 8)
            if (len - base < 4) x &= (1u << (8 * (len - base))) - 1u;
            mm += __builtin_popcount(x);
        }
    }
#else
    for (int i = 0; i < len; ++i) {
        const uint8_t x = a.mode == 2 ? 4 : code1(a.p[a.off + i * a.step], a.mode == 1, 4);
        const uint8_t y = b.mode == 2 ? 5 : code1(b.p[b.off + i * b.step], b.mode == 1, 5);
        mm += x != y;
    }
#endif
    return mm;
}
// prefix_mismatches(a, b, len) == 0, leaving at the first 16 bases that differ
CM_HD inline bool prefix_equal(const SV &a, const SV &b, int len) {
#if defined(__HIP_DEVICE_COMPILE__)
    for (int w0 = 0; 4 * w0 < len; w0 += 4) {
        uint32_t qa[4], qb[4];
        load_codes16(a, w0, 4, qa);
        load_codes16(b, w0, 5, qb);
        uint32_t any = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int base = 4 * (w0 + k);
            uint32_t x = qa[k] ^ qb[k];
            if (base >= len) x = 0;
            else if (len - base < 4) x &= (1u << (8 * (len - base))) - 1u;
            any |= x;
        }
        if (any) return false;
    }
    return true;
#else
    return prefix_mismatches(a, b, len) == 0;
#endif
}
CM_HD inline int ldiff(uint8_t a, uint8_t b) { return a == b ? 0 : 1; }
CM_HD inline int lscore(uint8_t a, uint8_t b) { return a == b ? SC_MAT : SC_MIS; }

// full Levenshtein for the tiny fallbacks (n <= 2w or m <= w, or n <= w for the one-sided form)
constexpr int TINY = 4 * MAX_BAND + 4;
CM_HD inline void tiny_full_dp(const LBuf &s, int n, const LBuf &t, int m, int *col /* n+1 */) {
    for (int i = 0; i <= n; ++i) col[i] = i;
    for (int j = 1; j <= m; ++j) {
        int diag = col[0];
        col[0] = j;
        const uint8_t tj = t.get(j - 1);
        for (int i = 1; i <= n; ++i) {
            const int v = cmin(cmin(diag + ldiff(s.get(i - 1), tj), col[i - 1] + 1), col[i] + 1);
            diag = col[i];
            col[i] = v;
        }
    }
}

// WT > 0: band known at compile time (rows live in registers, loops fully unrolled);
// WT == 0: run-time band up to MAX_BAND.
template <int WT> struct BandC { static constexpr int WC = WT > 0 ? WT : MAX_BAND; };

// Alignment::global_one_side_banded_alignment, src/align.cpp:219-252 (m == n + w in every call)
template <int WT>
CM_HD CM_NOINLINE int one_side_banded_impl(const LBuf &s, int n, const LBuf &t, int m, int w) {
    constexpr int WC = BandC<WT>::WC;
    if (w < 0 || n <= w) {
        int col[TINY];
        if (n >= TINY) n = TINY - 1;
        tiny_full_dp(s, n, t, m, col);
        return col[n];
    }
    int prev[WC + 3], cur[WC + 3];     // index c+1, c = j - i in [0,w]; everything outside is DPTINF
#pragma unroll
    for (int x = 0; x < WC + 3; ++x) prev[x] = (x >= 1 && x <= w + 1) ? x - 1 : DPTINF;   // dp[0][j] = j
    for (int i = 1; i <= n; ++i) {
        cur[0] = DPTINF;                            // dp[i][i-1]
        const uint8_t si = s.get(i - 1);
#pragma unroll
        for (int c = 0; c <= WC; ++c) {
            int v = DPTINF;
            if (c <= w) v = cmin(cmin(prev[c + 1] + ldiff(si, t.get(i + c - 1)), prev[c + 2] + 1), cur[c] + 1);
            cur[c + 1] = v;
        }
        cur[WC + 2] = DPTINF;
#pragma unroll
        for (int x = 0; x < WC + 3; ++x) prev[x] = cur[x];
    }
    int r = DPTINF;
#pragma unroll
    for (int x = 1; x <= WC + 1; ++x)
        if (x == (m - n) + 1) r = prev[x];
    return r;
}

// Alignment::global_banded_alignment[_reverse] + local_alignment_right/left, src/align.cpp:395-600.
// Forward form only: the reverse variant is the same DP on reversed strings (the caller stages them so).
template <int WT>
CM_HD CM_NOINLINE int local_alignment_side_impl(const Core &c, const LBuf &s, int n, const LBuf &t, int m, int &indel, int &align_score) {
    constexpr int WC = BandC<WT>::WC;
    const int w = WT > 0 ? WT : c.P.band;
    const int max_edit = c.P.max_ed;
    Cand best{max_edit + 1, c.P.max_sc + 1, w + 1, -1 * (c.P.max_sc + 1) - 2 * (max_edit + 1)};
    if (w < 0 || n <= 2 * w || m <= w) {
        int col[TINY];
        const int nn = n < TINY ? n : TINY - 1;
        tiny_full_dp(s, nn, t, m, col);
        for (int i = cmax(0, m - w); i <= cmin(m + w, nn); ++i)
            if (col[i] <= max_edit) {
                Cand x{col[i], 0, m - i, -2 * col[i]};
                if (cand_less(x, best)) best = x;
            }
    } else {
        // column j holds rows i in [j-w, j+w]; slot k = i - j + w + 1 (pads at 0 and 2w+2)
        int prev[2 * WC + 3], cur[2 * WC + 3];
#pragma unroll
        for (int k = 0; k < 2 * WC + 3; ++k) prev[k] = (k >= w + 1 && k <= 2 * w + 1) ? k - w - 1 : DPTINF;   // column 0
        CM_STAT(5, m * (2 * w + 1));
        for (int j = 1; j <= m; ++j) {
            const uint8_t tj = t.get(j - 1);
            cur[0] = DPTINF;
#pragma unroll
            for (int k = 1; k <= 2 * WC + 1; ++k) {
                const int i = j + k - w - 1;
                int v = DPTINF;
                if (k <= 2 * w + 1) {
                    if (i == 0) v = (j <= w) ? j : DPTINF;                  // dp[0][j]
                    else if (i > 0 && i <= n) v = cmin(cmin(prev[k] + ldiff(s.get(i - 1), tj), cur[k - 1] + 1), prev[k + 1] + 1);
                }
                cur[k] = v;
            }
            cur[2 * WC + 2] = DPTINF;
#pragma unroll
            for (int k = 0; k < 2 * WC + 3; ++k) prev[k] = cur[k];
        }
        for (int i = cmax(0, m - w); i <= cmin(m + w, n); ++i) {
            const int slot = i - m + w + 1;
            int v = DPTINF;
#pragma unroll
            for (int k = 1; k <= 2 * WC + 1; ++k)
                if (k == slot) v = prev[k];
            if (v <= max_edit) {
                Cand x{v, 0, m - i, -2 * v};
                if (cand_less(x, best)) best = x;
            }
        }
    }
    align_score = -1 * best.ed;
    indel = best.indel;
    return best.ed;
}

// Alignment::global_banded_alignment_drop + DropAlignment::local_alignment_{right,left}_sc,
// src/align.cpp:254-390, 669-723.  s = reference window (n), t = read residual (m); the left variant
// is the same DP on reversed strings.  Anti-diagonal k is evaluated slot by slot (slot q <-> i - j =
// q - w - 1) in ascending i, the reference's order, so the ">=" tie rule picks the same best cell.
template <int WT>
CM_HD CM_NOINLINE int local_alignment_sc_impl(const Core &c, const LBuf &s, int n, const LBuf &t, int m, int &sc_len, int &indel, int &align_score) {
    constexpr int WC = BandC<WT>::WC;
    constexpr int W2 = 2 * WC + 3;                  // slot = (i - j) + w + 1, pads at 0 and 2w+2
    const int w = WT > 0 ? WT : c.P.band;
    int d0[W2], d1[W2], d2[W2];                     // anti-diagonals k, k-1, k-2
    int on_s = 0, on_t = 0, best_score = 0;
    if (m > 0 && n > 0) {
#pragma unroll
        for (int q = 0; q < W2; ++q) {
            d2[q] = (q == w + 1) ? 0 : -DPTINF;                                    // anti-diagonal 0: (0,0)
            d1[q] = (w >= 1 && (q == w + 2 || q == w)) ? SC_IND : -DPTINF;         // anti-diagonal 1: (1,0), (0,1)
        }
        int pre_optimum = 0, cur_optimum = 0;
        int lb = 1, ub = 1, pre_ub = 0;
        for (int k = 2; k <= m + n; ++k) {
            int new_ub = -1;
            CM_STAT(4, ub - lb + 1);
#pragma unroll
            for (int q = 0; q < W2; ++q) {
                int v = -DPTINF;
                if (k <= w && (q == k + w + 1 || q == w + 1 - k)) v = k * SC_IND;   // boundary cells (k,0), (0,k)
                const int twice_i = k + q - w - 1;
                if (q >= 1 && q <= 2 * w + 1 && !(twice_i & 1)) {
                    const int i = twice_i >> 1;
                    if (i >= lb && i <= ub) {
                        const int j = k - i;
                        v = cmax(cmax(d2[q] + lscore(s.get(i - 1), t.get(j - 1)), d1[q - 1] + SC_IND), d1[q + 1] + SC_IND);
                        cur_optimum = cmax(cur_optimum, v);
                        if (v >= cur_optimum) {
                            cur_optimum = v;
                            on_s = i;
                            on_t = j;
                            best_score = v;
                        }
                        if (v + SC_XD < pre_optimum) v = -DPTINF;
                        if (v > -DPTINF) new_ub = i;
                    }
                }
                d0[q] = v;
            }
            const int lb_t = k - lb;
            if (lb_t == m || (k > w && ((k - w) % 2 == 0))) ++lb;
            if (ub < n && (k <= w || (k > w && ((k - w) % 2 == 1)))) ++ub;
            if ((pre_ub == -1 && new_ub == -1) || lb > ub) break;
            pre_ub = new_ub;
            pre_optimum = cmax(pre_optimum, cur_optimum);
#pragma unroll
            for (int q = 0; q < W2; ++q) {
                d2[q] = d1[q];
                d1[q] = d0[q];
            }
        }
    }
    const int score = best_score;                   // dpx[on_s][on_t]; 0 when the best cell is (0,0)
    const uint32_t ed = (uint32_t)((SC_MAT * cmax(on_s, on_t) - score) / (SC_MAT - SC_MIS));
    Cand best{c.P.max_ed + 1, cmax(c.P.max_sc, m) + 1, w + 1, 0};
    if (ed <= (uint32_t)c.P.max_ed) {
        // right: best.update(cand) ; left: best.set(cand).  The X-drop optimum is >= 0 == best.score,
        // and on a tie (score 0, ed 0) cand wins update() on ed, so both forms select cand.
        Cand x{(int)ed, m - on_t, on_t - on_s, score};
        best = x;
    }
    align_score = score;
    sc_len = best.sclen;
    indel = best.indel;
    return best.ed;
}

struct XdropW3 {
    int d[9];
    int pre_optimum, cur_optimum, lb, ub, pre_ub, on_s, on_k;
};
// eight codes starting at code `at` (0 <= at < cap), one per nibble, code `at` lowest
CM_HD inline uint32_t nib8(const LBuf &b, int at) {
    const int w = at >> 3;
    const uint64_t two = ((uint64_t)b.word(w + 1) << 32) | (uint64_t)b.word(w);
    return (uint32_t)(two >> (4 * (at & 7)));
}
// one anti-diagonal k with k & 1 == PAR (so every slot index below is a compile-time constant); EARLY: k <= W is possible
// (boundary cells).  `tr` holds the read residual REVERSED (tr[x] = t[m - 1 - x]): the cells of an anti-diagonal pair
// s[i0 - 1 + r] with t[k - i0 - 1 - r] = tr[m - k + i0 + r], r = 0..3, so one 8-nibble window of each string, XORed,
// gives the match / mismatch bits of all its cells at once.  Returns false when the reference's loop breaks.
template <int PAR, bool EARLY>
CM_HD inline bool xdrop_w3_step(XdropW3 &x, const LBuf &s, int n, const LBuf &tr, int m, int k, int top) {
    constexpr int W = 3;
    constexpr int q0 = PAR ? 1 : 2;
    int new_ub = -1;
    CM_STAT(4, x.ub - x.lb + 1);
    const int i0 = (k + q0 - W - 1) >> 1;        // row of the first slot (k + q0 - W - 1 is even and >= 0)
    const int a = i0 - 1, b = m - k + i0;        // first code of the two windows; a >= -1, b >= -3 while a cell is valid
    uint32_t S = nib8(s, a < 0 ? 0 : (a > top ? top : a));
    S = a < 0 ? S << 4 : S;
    uint32_t T = nib8(tr, b < 0 ? 0 : (b > top ? top : b));
    T = b < 0 ? T << (4 * (-b & 7)) : T;
    uint32_t ne = S ^ T;                         // codes are < 8: three bits tell two of them apart
    ne = (ne | (ne >> 1) | (ne >> 2)) & 0x11111111u;
    const int lb = x.lb, ub = x.ub, pre = x.pre_optimum;
    bool took = false;
#pragma unroll
    for (int r = 0; r <= W; ++r) {
        constexpr int QMAX = 2 * W + 1;
        const int q = q0 + 2 * r;
        if (q > QMAX) continue;                  // the even class has only W slots
        const int i = i0 + r;
        const bool valid = (i >= lb) && (i <= ub);
        const int sub = ((ne >> (4 * r)) & 1u) ? SC_MIS : SC_MAT;
        const int nb = x.d[q - 1] > x.d[q + 1] ? x.d[q - 1] : x.d[q + 1];
        int v = x.d[q] + sub;
        v = v > nb + SC_IND ? v : nb + SC_IND;
        const bool take = valid && (v >= x.cur_optimum);     // ascending i: a later cell wins a tie, as in the reference
        x.cur_optimum = take ? v : x.cur_optimum;
        x.on_s = take ? i : x.on_s;
        took = took || take;
        v = (v + SC_XD < pre) ? -DPTINF : v;
        new_ub = (valid && v > -DPTINF) ? i : new_ub;
        if (EARLY) {
            const bool bnd = (k <= W) && (q == k + W + 1 || q == W + 1 - k);     // boundary cells (k,0), (0,k)
            x.d[q] = valid ? v : (bnd ? k * SC_IND : -DPTINF);
        } else {
            x.d[q] = valid ? v : -DPTINF;
        }
    }
    x.on_k = took ? k : x.on_k;                  // on_t = on_k - on_s
    if (EARLY) {
        if (k - x.lb == m || (k > W && ((k - W) % 2 == 0))) ++x.lb;
        if (x.ub < n && (k <= W || (k > W && ((k - W) % 2 == 1)))) ++x.ub;
    } else if (PAR) {                            // k > W, k - W even
        ++x.lb;
    } else {                                     // k > W, k - W odd
        if (k - x.lb == m) ++x.lb;
        if (x.ub < n) ++x.ub;
    }
    if ((x.pre_ub == -1 && new_ub == -1) || x.lb > x.ub) return false;
    x.pre_ub = new_ub;
    x.pre_optimum = cmax(x.pre_optimum, x.cur_optimum);
    return true;
}
// The anti-diagonals k > W of the band-3 DP, in a RELATIVE score domain: slot q holds R = score - (k' >> 1), k' = the anti-diagonal
// the slot's value belongs to (a dead cell holds -DPTINF itself).  With that offset a match adds nothing (R - 0), a mismatch costs 4,
// a gap from anti-diagonal k - 1 costs 4 for even k and 3 for odd k -- kept per slot as Rm = R - (what its consumers subtract),
// so that a cell is one bit-field extract, one subtract and one three-way max; the running optimum and the X-drop threshold are
// shifted into the same domain once per anti-diagonal.  Exactly the values, ties and pruning of xdrop_w3_step (score = R + (k >> 1)
// for every live cell, by induction: (k - 2 >> 1) + 1 = k >> 1; (k - 1 >> 1) = (k >> 1) - (k even)); the "last live row" new_ub is
// only ever compared with -1, so a flag `some cell alive` replaces it.  (VALU instructions per two anti-diagonals on gfx950:
// 164 -> see DESIGN; the anti-diagonal counter stays scalar because the caller's loop is wave-uniform.)
struct XdropW3R {
    int R[9], Rm[9];
    int pre_optimum, cur_optimum, lb, ub, on_s, on_k;
    bool alive_prev;
};
template <int PAR>
CM_HD inline bool xdrop_w3_rel(XdropW3R &x, const LBuf &s, int n, const LBuf &tr, int m, int k, int top) {
    constexpr int W = 3;
    constexpr int q0 = PAR ? 1 : 2;
    constexpr int G_OUT = PAR ? 4 : 3;           // what the consumers of this parity's slots subtract for a gap (odd slots feed even k: 4)
    CM_STAT(4, x.ub - x.lb + 1);
    const int ck = k >> 1;
    const int i0 = (k + q0 - W - 1) >> 1;        // row of the first slot
    const int a = i0 - 1, b = m - k + i0;        // first code of the two windows; a >= 0 here, b >= -3 while a cell is valid
    const uint32_t S = nib8(s, a > top ? top : a);
    uint32_t T = nib8(tr, b < 0 ? 0 : (b > top ? top : b));
    T = b < 0 ? T << (4 * (-b & 7)) : T;
    uint32_t ne = S ^ T;                         // codes are < 8: three bits tell two of them apart
    ne = ((ne | (ne >> 1) | (ne >> 2)) & 0x11111111u) << 2;      // 4 where the codes differ, per nibble
    const int lb = x.lb, ub = x.ub;
    const int thr = x.pre_optimum - SC_XD - ck;  // a cell below this is dropped
    int cur = x.cur_optimum - ck;
    bool took = false;
    int live = -DPTINF;
#pragma unroll
    for (int r = 0; r <= W; ++r) {
        constexpr int QMAX = 2 * W + 1;
        const int q = q0 + 2 * r;
        if (q > QMAX) continue;                  // the even class has only W slots
        const int i = i0 + r;
        const bool valid = (i >= lb) && (i <= ub);
        const int diag = x.R[q] - (int)((ne >> (4 * r)) & 4u);
        const int nb = x.Rm[q - 1] > x.Rm[q + 1] ? x.Rm[q - 1] : x.Rm[q + 1];
        const int v = diag > nb ? diag : nb;
        const bool take = valid && (v >= cur);   // ascending i: a later cell wins a tie, as in the reference
        cur = take ? v : cur;
        x.on_s = take ? i : x.on_s;
        took = took || take;
        const int nv = (valid && v >= thr) ? v : -DPTINF;
        x.R[q] = nv;
        x.Rm[q] = nv - G_OUT;
        live = live > nv ? live : nv;
    }
    x.on_k = took ? k : x.on_k;                  // on_t = on_k - on_s
    x.cur_optimum = cur + ck;
    if (PAR) {                                   // k > W, k - W even
        ++x.lb;
    } else {                                     // k > W, k - W odd
        if (k - x.lb == m) ++x.lb;
        if (x.ub < n) ++x.ub;
    }
    const bool alive = live > -DPTINF;
    if ((!x.alive_prev && !alive) || x.lb > x.ub) return false;
    x.alive_prev = alive;
    x.pre_optimum = cmax(x.pre_optimum, x.cur_optimum);
    return true;
}
#if defined(__HIP_DEVICE_COMPILE__)
#define CM_ANY_LANE(x) (__builtin_amdgcn_ballot_w64(x) != 0ull)
#else
#define CM_ANY_LANE(x) (x)
#endif
// Band-3 specialisation of the X-drop DP (the default bandWidth, >85 % of all DP cells).
// Same recurrence, evaluation order and pruning as local_alignment_sc_impl, restructured for a GPU lane:
//  * one register per slot, updated in place: anti-diagonal k only owns the slots q == k (mod 2), so
//    d[q] still holds (k-2, q) when (k, q) is computed and d[q +- 1] hold anti-diagonal k-1;
//  * the loop is unrolled by two anti-diagonals so the slot class of each half is static, and it runs for as long as ANY lane
//    of the wave has anti-diagonals left (a lane that is done is masked): the counter k is then a scalar, and with it the
//    window addresses of the reference string;
//  * branch-free cells (selects), at most 4 per anti-diagonal instead of a 9-slot loop;
//  * the substitution scores of an anti-diagonal come from one XOR of two nibble windows (the read residual is staged
//    reversed for this: `tr`); the best cell is kept as (row, anti-diagonal), its score is cur_optimum;
//  * beyond the first two anti-diagonals (boundary cells) the cells live in a relative score domain (xdrop_w3_rel).
// One band-3 DP in flight, resumable two anti-diagonals at a time: local_alignment_sc_w3 runs one to its end; the DP engine of
// k_pair_heavy keeps one per lane and hands a lane the next queued request as soon as its DP has ended.
struct XdropLane {
    XdropW3R y;
    int k, kmax, n, m;
    bool go;                 // anti-diagonals left
};
CM_HD inline void xdrop_w3_begin(XdropLane &L, const LBuf &s, int n, const LBuf &tr, int m, int top) {
    constexpr int W = 3;
    XdropW3 x;
    x.on_s = x.on_k = x.cur_optimum = 0;
    x.pre_optimum = 0;
    x.lb = x.ub = 1;
    x.pre_ub = 0;
    L.go = false;
    L.n = n;
    L.m = m;
    L.kmax = m + n;
    L.k = 4;
    if (m > 0 && n > 0) {
#pragma unroll
        for (int q = 0; q < 2 * W + 3; ++q) x.d[q] = (q == W + 1) ? 0 : ((q == W + 2 || q == W) ? SC_IND : -DPTINF);
        L.go = xdrop_w3_step<0, true>(x, s, n, tr, m, 2, top) && L.kmax >= 3 && xdrop_w3_step<1, true>(x, s, n, tr, m, 3, top);
    }
    XdropW3R &y = L.y;
#pragma unroll
    for (int q = 0; q < 2 * W + 3; ++q) {        // into the relative domain: every slot holds anti-diagonal 2 or 3 (offset 1)
        const int d = (m > 0 && n > 0) ? x.d[q] : -DPTINF;
        y.R[q] = (q >= 1 && q <= 2 * W + 1 && d > -DPTINF) ? d - 1 : -DPTINF;
        y.Rm[q] = y.R[q] - ((q & 1) ? 4 : 3);
    }
    y.pre_optimum = x.pre_optimum;
    y.cur_optimum = x.cur_optimum;
    y.lb = x.lb;
    y.ub = x.ub;
    y.on_s = x.on_s;
    y.on_k = x.on_k;
    y.alive_prev = x.pre_ub != -1;
    L.go = L.go && L.k <= L.kmax;
}
// anti-diagonals k, k + 1 of a DP that has some left (L.go)
CM_HD inline void xdrop_w3_advance(XdropLane &L, const LBuf &s, const LBuf &tr, int top) {
    const int k = L.k;
    if (!xdrop_w3_rel<0>(L.y, s, L.n, tr, L.m, k, top)) L.go = false;
    else if (k + 1 > L.kmax) L.go = false;
    else if (!xdrop_w3_rel<1>(L.y, s, L.n, tr, L.m, k + 1, top)) L.go = false;
    L.k = k + 2;
    L.go = L.go && L.k <= L.kmax;
}
CM_HD inline int xdrop_w3_end(const Core &c, const XdropLane &L, int &sc_len, int &indel, int &align_score) {
    constexpr int W = 3;
    const int score = L.y.cur_optimum, on_s = L.y.on_s, on_t = L.y.on_k - L.y.on_s;
    const uint32_t ed = (uint32_t)((SC_MAT * cmax(on_s, on_t) - score) / (SC_MAT - SC_MIS));
    Cand best{c.P.max_ed + 1, cmax(c.P.max_sc, L.m) + 1, W + 1, 0};
    if (ed <= (uint32_t)c.P.max_ed) {
        Cand z{(int)ed, L.m - on_t, on_t - on_s, score};
        best = z;
    }
    align_score = score;
    sc_len = best.sclen;
    indel = best.indel;
    return best.ed;
}
CM_HD CM_NOINLINE int local_alignment_sc_w3(const Core &c, const LBuf &s, int n, const LBuf &tr, int m, int &sc_len, int &indel, int &align_score) {
    const int top = (s.cap < tr.cap ? s.cap : tr.cap) - 1;
    XdropLane L;
    xdrop_w3_begin(L, s, n, tr, m, top);
#if defined(CM_ABL_NODP)        // ablation study only (tests/diag/ablate.sh): what the kernels issue without the DP's main loop (results wrong)
    L.go = false;
#endif
    // (every lane of the wave starts at k = 4 here, so the counter is wave-uniform -- a scalar -- for as long as any lane runs)
    for (int k = 4; CM_ANY_LANE(L.go); k += 2) {
        if (L.go) {
            L.k = k;
            xdrop_w3_advance(L, s, tr, top);
        }
    }
    return xdrop_w3_end(c, L, sc_len, indel, align_score);
}

// A soft-clip X-drop DP computed ahead of the code that asks for it (k_pair_heavy: the DPs of a whole wave's tasks are collected,
// sorted by length and run 64 at a time, instead of every lane running its own whenever its control flow arrives there).
// The DP is a pure function of its two strings; an entry carries the identity of the request it answers -- reference window
// (offset, direction, length), read residual (offset inside the read, which of the pair's two read views, length) -- and
// local_alignment_sc() takes an entry only if every field matches the call, so a request nobody predicted, or predicted
// differently, is simply computed in place: results never depend on what the table holds.
struct PreDP {
    uint32_t s_off;          // SV::off of the reference window (as passed: the far end for a reversed window)
    uint32_t key;            // n | m << 10 | t.off << 20 | (s.step < 0) << 30 | (t.mode == 1) << 31
    uint32_t res;            // ed | sclen << 8 | (indel + 64) << 20 | 1 << 31 (valid)
    int32_t score;
};
typedef CM_G const PreDP *g_pre;          // the table lives in the wave's slice of a global scratch area (L2-resident)
CM_HD inline bool pre_keyable(const SV &s, int n, const SV &t, int m) {
    return n >= 0 && n < 1024 && m >= 0 && m < 1024 && t.off >= 0 && t.off < 1024 && s.mode == 0 && t.mode != 2;
}
CM_HD inline uint32_t pre_key(const SV &s, int n, const SV &t, int m) {
    return (uint32_t)n | ((uint32_t)m << 10) | ((uint32_t)t.off << 20) | (s.step < 0 ? 1u << 30 : 0u) | (t.mode == 1 ? 1u << 31 : 0u);
}
CM_HD inline uint32_t pre_pack(int ed, int sclen, int indel) { return (uint32_t)ed | ((uint32_t)sclen << 8) | ((uint32_t)(indel + 64) << 20) | (1u << 31); }
// Staging + dispatch on the (wave-uniform) band.  `sm` = the lane's two staging buffers.
#if !(defined(CM_DIAG) && defined(CM_DIAG_CNT) && defined(__HIPCC__) && defined(__HIP_DEVICE_COMPILE__))
#define CM_CNT(sm_, k) ((void)0)      // event counts of a diagnostic build (-DCM_DIAG -DCM_DIAG_CNT; contended atomics: not together with the timers)
#endif
#if defined(CM_DIAG) && defined(__HIPCC__)
// Diagnostic section timers.  acc[]: per-lane time between ticks (includes waiting for other lanes).
// w (LDS, one per wave): w[0] = time of the wave's last tick, w[1+id] += wave time attributed to section id,
// w[33+id] += that time x lanes arriving at the tick together.
struct Tick { unsigned long long last; unsigned long long acc[32]; CM_L unsigned long long *w; int wave_on; unsigned long long *ctr; };
#if defined(__HIP_DEVICE_COMPILE__)
__device__ inline void cm_tick(Tick *tk, int id) {
    const unsigned long long n_ = wall_clock64();
    tk->acc[id] += n_ - tk->last;
    tk->last = n_;
    if (tk->wave_on) {
        const unsigned long long m = __ballot(1);
        const unsigned int me = __lane_id();
        if (me == (unsigned int)(__ffsll((long long)m) - 1)) {
            const unsigned long long dt = n_ - tk->w[0];
            tk->w[1 + id] += dt;
            tk->w[33 + id] += dt * (unsigned long long)__popcll(m);
            tk->w[0] = n_;
        }
    }
}
#define CM_TICK(sm_, id) cm_tick((sm_).tk, id)
#if defined(CM_DIAG_CNT)
#define CM_CNT(sm_, k) do { if ((sm_).tk->ctr) atomicAdd(&(sm_).tk->ctr[k], 1ull); } while (0)      // event counts (cm_debug_counters)
#endif
#else
#define CM_TICK(sm_, id) ((void)0)
#endif
struct DpMem { LBuf a, b; g_err err; Tick *tk; g_spill spill; int spill_cap; g_pre pre; int n_pre; };
#elif defined(CM_STAGE2_HOST)
#define CM_TICK(sm_, id) ((void)0)
struct DpMem { LBuf a, b; g_err err; bool edit; g_spill spill; int spill_cap; g_pre pre; int n_pre; };      // edit: EditDistAlignment instead of DropAlignment (ProcessCirc, src/process_circ.cpp:25)
#else
#define CM_TICK(sm_, id) ((void)0)
struct DpMem { LBuf a, b; g_err err; g_spill spill; int spill_cap; g_pre pre; int n_pre; };   // spill: see Memo (null in the first pass of a pair); pre: see PreDP
#endif
CM_HD inline bool dp_fits(const DpMem &sm, int n, int m) {
    if (n <= sm.a.cap && m <= sm.b.cap && n >= 0 && m >= 0) return true;
    flag_err(sm.err, ERR_BAND);
    return false;
}
// Closed forms used below (each provably equal to the DP it replaces; NOTES.md note 3):
//  * one-sided DP with w == 0 is the Hamming distance; with w > 0 and s == t[0..n) it is w (= |m - n|);
//  * banded edit DP with s[0..m) == t: dp[m][m] = 0 beats every other end row (score 0 vs <= -2);
//  * X-drop DP with s[0..m) == t (all m bases valid): the diagonal is never pruned, cell (m, m) scores m
//    and no other cell can (score <= matches - 3*gaps), so on_s = on_t = m, ed = 0, no clip, no indel.
CM_HD inline int one_side_banded(const Core &c, const DpMem &sm, const SV &s, int n, const SV &t, int m, int w) {
    if (w == 0 && m == n && n >= 0) return prefix_mismatches(s, t, n);
    if (w > 0 && n > w && m == n + w && prefix_mismatches(s, t, n) == 0) return w;
    if (!dp_fits(sm, n, m)) return c.P.max_ed + 1;
    stage(s, n, sm.a, 4);
    stage(t, m, sm.b, 5);
    return c.P.band == 3 ? one_side_banded_impl<3>(sm.a, n, sm.b, m, w) : one_side_banded_impl<0>(sm.a, n, sm.b, m, w);
}
CM_HD inline int local_alignment_side(const Core &c, const DpMem &sm, const SV &s, int n, const SV &t, int m, bool rev, int &indel, int &align_score) {
    if (m >= 1 && n >= m && prefix_mismatches(rev ? s.rev(n) : s, rev ? t.rev(m) : t, m) == 0) {
        indel = 0;
        align_score = 0;
        return 0;
    }
    if (!dp_fits(sm, n, m)) { indel = c.P.band + 1; align_score = -(c.P.max_ed + 1); return c.P.max_ed + 1; }
    stage(rev ? s.rev(n) : s, n, sm.a, 4);
    stage(rev ? t.rev(m) : t, m, sm.b, 5);
    return c.P.band == 3 ? local_alignment_side_impl<3>(c, sm.a, n, sm.b, m, indel, align_score)
                         : local_alignment_side_impl<0>(c, sm.a, n, sm.b, m, indel, align_score);
}
#if defined(CM_STAGE2_HOST)
// EditDistAlignment::local_alignment_right_sc / _left_sc (src/align.cpp:602-660), host only (stage 2): the banded edit DP of
// global_banded_alignment (:395-450; full DP when n <= 2w or m <= w) over the staged strings, then the end cell (i, j), j = m down
// to m - min(maxSc, m), |i - j| <= w, with dp <= maxEd that is best under AlignCandid's order (first seen wins ties):
// soft clip m - j, indel j - i; a query of at most maxEd bases may also be taken as all mismatches.  Score = m - clip - 2 ed.
inline int local_alignment_sc_edit(const Core &c, const LBuf &s, int n, const LBuf &t, int m, int &sc_len, int &indel, int &align_score) {
    const int w = c.P.band, BIG = DPTINF;
    const bool banded = !(w < 0 || n <= 2 * w || m <= w);
    // Banded: only the cells with |i - j| <= w are ever finite (everything else is BIG in the reference's matrix), so only they are
    // kept: row i holds columns i - w .. i + w at slots 0 .. 2w.  Not banded (a few rows or columns): the whole matrix.
    const int W = banded ? 2 * w + 1 : m + 1;
    static thread_local std::vector<int> cells;
    cells.resize((size_t)(n + 1) * (size_t)W);
    int *dp = cells.data();
    auto at = [&](int i, int j) -> int {                       // dp[i][j]; BIG outside the band
        if (!banded) return dp[(size_t)i * W + j];
        const int k = j - i + w;
        return (k < 0 || k >= W) ? BIG : dp[(size_t)i * W + k];
    };
    for (int i = 0; i <= n; ++i) {
        const int j0 = banded ? cmax(0, i - w) : 0, j1 = banded ? cmin(m, i + w) : m;
        for (int j = j0; j <= j1; ++j) {
            int v;
            if (i == 0) v = j;
            else if (j == 0) v = i;
            else {
                const int d = at(i - 1, j - 1) + ldiff(s.get(i - 1), t.get(j - 1));
                const int u = at(i - 1, j) + 1, l = at(i, j - 1) + 1;
                v = cmin(d, cmin(u, l));
                if (v > BIG) v = BIG;
            }
            dp[(size_t)i * W + (banded ? j - i + w : j)] = v;
        }
    }
    const int max_sclen = cmin(c.P.max_sc, m);
    Cand best{c.P.max_ed + 1, c.P.max_sc + 1, w + 1, -(c.P.max_sc + 1) - 2 * (c.P.max_ed + 1)};
    for (int j = m; j >= m - max_sclen; --j)
        for (int i = cmax(0, j - w); i <= cmin(j + w, n); ++i) {
            const int v = at(i, j);
            if (v <= c.P.max_ed) {
                const Cand x{v, m - j, j - i, -(m - j) - 2 * v};
                if (cand_less(x, best)) best = x;
            }
        }
    if (m <= c.P.max_ed) {
        const Cand x{m, 0, 0, -2 * m};
        if (cand_less(x, best)) best = x;
    }
    align_score = m - best.sclen - 2 * best.ed;
    sc_len = best.sclen;
    indel = best.indel;
    return best.ed;
}
#endif
// The two halves of local_alignment_sc (below), also called one after the other on other lanes by the DP pool of k_pair_heavy.
// sc_closed_form: the read residual equals the start of the reference window base for base (closed form, see above).
CM_HD inline bool sc_closed_form(const SV &s, int n, const SV &t, int m) { return m >= 1 && n >= m && prefix_equal(s, t, m); }
// sc_run_dp: staging + the DP proper (the strings fit the staging buffers: dp_fits)
CM_HD inline int sc_run_dp(const Core &c, const DpMem &sm, const SV &s, int n, const SV &t, int m, int &sc_len, int &indel, int &align_score) {
    const bool w3 = c.P.band == 3;
    CM_HOOK_DP(1, s, n, t, m);
    stage(s, n, sm.a, 4);
    stage(w3 ? t.rev(m) : t, m, sm.b, 5);                     // the band-3 DP walks the read residual from its far end
    CM_STAT(11 + (m > 16) + (m > 32) + (m > 64), 1);          // X-drop DPs by read-residual length (test-only counters)
    CM_TICK(sm, 27);
    const int r = w3 ? local_alignment_sc_w3(c, sm.a, n, sm.b, m, sc_len, indel, align_score)
                     : local_alignment_sc_impl<0>(c, sm.a, n, sm.b, m, sc_len, indel, align_score);
    CM_TICK(sm, 28);
    return r;
}
// the caller passes already-reversed views for the left variant
CM_HD inline int local_alignment_sc(const Core &c, const DpMem &sm, const SV &s, int n, const SV &t, int m, int &sc_len, int &indel, int &align_score) {
    CM_TICK(sm, 24);
    CM_HOOK_DP(0, s, n, t, m);
#if defined(CM_STAGE2_HOST)
    if (sm.edit) {
        if (!dp_fits(sm, n, m)) { sc_len = c.P.max_sc + 1; indel = c.P.band + 1; align_score = 0; return c.P.max_ed + 1; }
        stage(s, n, sm.a, 4);
        stage(t, m, sm.b, 5);
        return local_alignment_sc_edit(c, sm.a, n, sm.b, m, sc_len, indel, align_score);
    }
#endif
    if (sm.pre && pre_keyable(s, n, t, m)) {                  // computed ahead (PreDP)?
        // the entries of an item sit in a fixed order -- [forward read: left end, right end][backward read: left, right] (n_pre = 4)
        // or [left end, right end] (n_pre = 2) -- so the call's own views say which one could be its answer: one load
        const uint32_t key = pre_key(s, n, t, m);
        const int at = (s.step < 0 ? 0 : 1) + (sm.n_pre == 4 && t.mode == 1 ? 2 : 0);
        const PreDP e = sm.pre[at];
        if ((e.res >> 31) && e.key == key && e.s_off == (uint32_t)s.off) {
            CM_CNT(sm, 10);
            sc_len = (int)((e.res >> 8) & 0xFFFu);
            indel = (int)((e.res >> 20) & 0x7Fu) - 64;
            align_score = e.score;
            return (int)(e.res & 0xFFu);
        }
    }
    if (sc_closed_form(s, n, t, m)) {
        CM_CNT(sm, 13);
        CM_STAT(9, 1);
        sc_len = 0;
        indel = 0;
        align_score = m * SC_MAT;
        CM_TICK(sm, 25);
        return 0;
    }
    CM_TICK(sm, 26);
    if (!dp_fits(sm, n, m)) { sc_len = cmax(c.P.max_sc, m) + 1; indel = c.P.band + 1; align_score = 0; return c.P.max_ed + 1; }
    CM_CNT(sm, 11);
    return sc_run_dp(c, sm, s, n, t, m, sc_len, indel, align_score);
}


// GenomeSeeder::pac2char, src/match_read.cpp:288-299
CM_HD inline bool pac2char(const Core &c, uint32_t start, int len, SV &out) {
    const int ref_len = (int)c.X.ref_len;
    if ((int)start < 0 || (int)start + len - 1 > ref_len) return false;
    CM_STAT(15, len > 0 ? len : 0);          // reference bytes the algorithm asks for (host emulation only: tests/diag/window_bytes.py)
    if (start == 0) out = SV{c.X.genome, 0, 1, 2};
    else out = SV{c.X.genome, (int32_t)(start - 1), 1, 0};
    return true;
}

// ------------------------------------------------------------------------------------------
// mates, chains, ordering (A15-A18)
// ------------------------------------------------------------------------------------------
struct MM {            // MatchedMate, src/common.h:260-307
    uint32_t spos, epos, qspos, qepos, matched_len;
    int right_ed, left_ed, middle_ed, sclen_right, sclen_left, dir, type;
    int exon_ind_spos, exon_ind_epos, exons_spos, exons_epos;
    uint16_t junc_num;
    bool is_concord, left_ok, right_ok, looked_up_spos, looked_up_epos;
};
CM_HD inline MM mm_init(const Core &c) {
    MM m;
    m.spos = m.epos = m.qspos = m.qepos = m.matched_len = 0;
    m.right_ed = m.left_ed = m.middle_ed = c.P.max_ed + 1;
    m.sclen_right = m.sclen_left = 0;
    m.dir = 0;
    m.type = CM_ORPHAN;
    m.exon_ind_spos = m.exon_ind_epos = -1;
    m.exons_spos = m.exons_epos = -1;
    m.junc_num = 0;
    m.is_concord = m.left_ok = m.right_ok = m.looked_up_spos = m.looked_up_epos = false;
    return m;
}
CM_HD inline int mm_ed(const MM &m) { return m.left_ed + m.middle_ed + m.right_ed; }

// One stored chain copied into private memory when a task starts: the pair logic reads fragment
// fields dozens of times, and per-lane private arrays cost one coalesced access per wave where a
// global field read costs 64 separate cache lines.
struct CH {
    uint32_t r[MAX_SEEDS];
    int32_t q[MAX_SEEDS];
    uint32_t n;
    int kmer;
    CM_HD CH(g_chain p, int k) : n(p->chain_len), kmer(k) {
        for (uint32_t i = 0; i < (uint32_t)MAX_SEEDS; ++i) {
            r[i] = i < n ? p->rpos[i] : 0u;
            q[i] = i < n ? p->qpos[i] : 0;
        }
    }
    CM_HD inline uint32_t len() const { return n; }
    CM_HD inline uint32_t rpos(uint32_t i) const { return r[i]; }
    CM_HD inline int32_t qpos(uint32_t i) const { return q[i]; }
    CM_HD inline uint32_t rend_excl() const { return r[n - 1] + (uint32_t)kmer; }
    CM_HD inline int32_t qend_excl() const { return q[n - 1] + kmer; }
};
struct CHEnds {         // just the reference span of a chain (pairing predicate)
    uint32_t r0, rend;
    CM_HD CHEnds(g_chain p, int k) : r0(p->rpos[0]), rend(p->rpos[p->chain_len - 1] + (uint32_t)k) {}
    CM_HD CHEnds(uint32_t a, uint32_t b) : r0(a), rend(b) {}
};

CM_HD inline void default_mr(const Core &c, cm_mapped_read &m) {
    m.spos_r1 = m.spos_r2 = m.epos_r1 = m.epos_r2 = 0;
    m.qspos_r1 = m.qspos_r2 = m.qepos_r1 = m.qepos_r2 = 0;
    m.mlen_r1 = m.mlen_r2 = 0;
    m.ed_r1 = m.ed_r2 = c.P.max_ed + 1;
    m.type = CM_NOPROC_NOMATCH;
    m.tlen = INF_I;
    m.contig_num = 0;
    m.chr_id = -1;
    m.junc_num = 0;
    m.r1_forward = 1;
    m.r2_forward = 1;
    m.gm_compatible = 0;
    m.pad[0] = m.pad[1] = m.pad[2] = 0;
}
CM_HD inline bool mapped_type(int t) {
    return t == CM_CONCRD || t == CM_DISCRD || t == CM_CHIORF || t == CM_CHIBSJ || t == CM_CHI2BSJ || t == CM_CONGNM || t == CM_CONGEN;
}
CM_HD inline bool go_for_update(const cm_mapped_read &t, const MM &r1, const MM &r2, int32_t tlen, bool gm, int type) {
    if (type < t.type) return true;
    if (type > t.type) return false;
    if (gm && !t.gm_compatible) return true;
    if (!gm && t.gm_compatible) return false;
    const int ed = mm_ed(r1) + mm_ed(r2);
    const uint32_t ml = r1.matched_len + r2.matched_len;
    if (type < CM_CHIBSJ) {
        if ((t.ed_r1 + t.ed_r2) > ed) return true;
        if ((t.ed_r1 + t.ed_r2) < ed) return false;
        if (t.tlen > tlen) return true;
        if (t.tlen < tlen) return false;
        if ((t.mlen_r1 + t.mlen_r2) < ml) return true;
        if ((t.mlen_r1 + t.mlen_r2) > ml) return false;
    } else {
        if ((t.mlen_r1 + t.mlen_r2) < ml) return true;
        if ((t.mlen_r1 + t.mlen_r2) > ml) return false;
        if ((t.ed_r1 + t.ed_r2) > ed) return true;
        if ((t.ed_r1 + t.ed_r2) < ed) return false;
    }
    return false;
}
CM_HD inline bool mr_update(const Core &c, cm_mapped_read &t, const MM &r1, const MM &r2, int row, int32_t tlen, int jun_between,
                            bool gm, int type, bool r1_first) {
    if (!go_for_update(t, r1, r2, tlen, gm, type)) return false;
    const uint32_t shift = c.A.chr_shift[row];
    const MM &a = r1_first ? r1 : r2;
    const MM &b = r1_first ? r2 : r1;
    t.type = type;
    t.chr_id = c.A.chr_id[row];
    t.spos_r1 = a.spos - shift; t.epos_r1 = a.epos - shift; t.qspos_r1 = a.qspos; t.qepos_r1 = a.qepos;
    t.mlen_r1 = a.matched_len; t.ed_r1 = mm_ed(a);
    t.spos_r2 = b.spos - shift; t.epos_r2 = b.epos - shift; t.qspos_r2 = b.qspos; t.qepos_r2 = b.qepos;
    t.mlen_r2 = b.matched_len; t.ed_r2 = mm_ed(b);
    t.r1_forward = a.dir > 0;
    t.r2_forward = b.dir > 0;
    t.tlen = tlen;
    t.junc_num = (uint16_t)(jun_between + r1.junc_num + r2.junc_num);
    t.gm_compatible = gm;
    t.contig_num = c.X.contig_num;
    return true;
}
CM_HD inline void mr_update_type(cm_mapped_read &t, int type) { if (type < t.type) t.type = type; }

CM_HD inline void update_match_mate_info(const Core &c, bool lok, bool rok, int err, MM &mm) {
    mm.left_ok = lok && (mm.sclen_left <= c.P.max_sc);
    mm.right_ok = rok && (mm.sclen_right <= c.P.max_sc);
    if (lok && rok && (err <= c.P.max_ed) && (mm.sclen_right <= c.P.max_sc) && (mm.sclen_left <= c.P.max_sc)) {
        mm.is_concord = true;
        mm.type = CM_CONCRD;
    } else if (lok || rok) mm.type = CM_CANDID;
    else mm.type = CM_ORPHAN;
}
CM_HD inline int estimate_middle_error(const Core &c, const CH &ch) {
    int mid = 0;
    for (uint32_t i = 0; i + 1 < ch.len(); ++i)
        if (ch.qpos(i + 1) > ch.qpos(i) + ch.kmer) {
            const int diff = (int)(ch.rpos(i + 1) - ch.rpos(i)) - (ch.qpos(i + 1) - ch.qpos(i));
            if (diff == 0) ++mid;
            else if (diff > 0 && diff <= c.P.band) mid += diff;
            else if (diff < 0 && diff >= -c.P.band) mid -= diff;
        }
    return mid;
}
template <class CHT> CM_HD inline bool is_concord_impl(const CHT &a, uint32_t seq_len, MM &mr, bool v2) {
    if (a.len() < 2) {
        mr.is_concord = false;
    } else if ((uint32_t)(a.qend_excl() - a.qpos(0)) >= seq_len) {
        mr.is_concord = true;
        mr.type = CM_CONCRD;
        mr.spos = a.rpos(0);
        mr.epos = a.rend_excl() - 1;
        mr.matched_len = (uint32_t)(a.qend_excl() - a.qpos(0));
        mr.qspos = (uint32_t)a.qpos(0);
        mr.qepos = (uint32_t)(a.qend_excl() - 1);
    } else {
        mr.is_concord = false;
        if (v2 && (a.qpos(0) == 0 || (uint32_t)a.qend_excl() == seq_len)) mr.type = CM_CANDID;
    }
    return mr.is_concord;
}
CM_HD inline void overlap_to_epos(const Core &c, MM &m) {
    if (m.looked_up_epos || m.exons_epos >= 0) return;
    m.exons_epos = overlap_ind(c, m.epos, m.exon_ind_epos);
    m.looked_up_epos = true;
}
CM_HD inline void overlap_to_spos(const Core &c, MM &m) {
    if (m.looked_up_spos || m.exons_spos >= 0) return;
    m.exons_spos = overlap_ind(c, m.spos, m.exon_ind_spos);
    m.looked_up_spos = true;
}
CM_HD inline int calc_tlen(const Core &c, const MM &sm, const MM &lm, int &intron_num) {
    const AnnotV &A = c.A;
    int min_tlen = INF_I;
    const uint32_t ns = iv_nseg(A, sm.exons_epos);
    for (uint32_t i = 0; i < ns; ++i) {
        const uint32_t sg = iv_segid(A, sm.exons_epos, i);
        for (uint32_t j = A.seg[sg].tid_off; j < A.seg[sg].tid_off + A.seg[sg].ntid; ++j) {
            const uint32_t tid = A.seg_tid[j];
            const int start_ind = A.tr[tid].start_ind;
            const uint32_t sti = (uint32_t)(sm.exon_ind_epos - start_ind);
            const uint32_t eti = (uint32_t)(lm.exon_ind_spos - start_ind);
            const uint32_t tsz = A.tr[tid].t2s_len;
            const g_u8 t2s = A.t2s + A.tr[tid].t2s_off;
            if (lm.exon_ind_spos < start_ind || eti >= tsz || t2s[eti] == 0) continue;
            int in = 0, tlen;
            if (sti == eti) {
                tlen = (int)(lm.spos - sm.epos + 1);
            } else {
                bool pre_zero = false;
                tlen = (int)(A.iv[sm.exons_epos].epos - sm.epos + 1);
                int it = sm.exon_ind_epos;
                for (uint32_t k = sti + 1; k < eti; ++k) {
                    ++it;
                    if (t2s[k] != 0) {
                        tlen += (int)(A.iv[it].epos - A.iv[it].spos + 1);
                        pre_zero = false;
                    } else {
                        if (!pre_zero) ++in;
                        pre_zero = true;
                    }
                }
                tlen += (int)(lm.spos - A.iv[lm.exons_spos].spos + 1);
            }
            if (tlen < min_tlen) {
                intron_num = in;
                min_tlen = tlen;
            }
        }
    }
    return (min_tlen == INF_I) ? -1 : (int)(min_tlen + sm.matched_len - 1 + lm.matched_len - 1);
}
CM_HD inline bool same_gene_span(const Core &c, int iv, uint32_t s, uint32_t e) {      // utils.cpp:617-639
    const AnnotV &A = c.A;
    const uint32_t n = iv_nseg(A, iv);
    for (uint32_t i = 0; i < n; ++i) {
        const uint32_t g = A.seg[iv_segid(A, iv, i)].gene_id;
        if (A.gene[g].start <= s && e <= A.gene[g].end) return true;
    }
    return false;
}
CM_HD inline bool share_gene(const Core &c, int a, int b) {
    const AnnotV &A = c.A;
    const uint32_t na = iv_nseg(A, a), nb = iv_nseg(A, b);
    for (uint32_t i = 0; i < na; ++i)
        for (uint32_t j = 0; j < nb; ++j)
            if (A.seg[iv_segid(A, a, i)].gene_id == A.seg[iv_segid(A, b, j)].gene_id) return true;
    return false;
}
// same_transcript + intersect_trans, utils.cpp:322-354: the transcripts of interval s's segments (in that order) that also
// occur in interval r.  f(tid) is called for the elements with index >= skip; returns false as soon as f does.
template <class F>
CM_HD inline bool common_tids_from(const Core &c, int s, int r, int skip, F &&f) {
    if (s < 0 || r < 0) return true;
    const AnnotV &A = c.A;
    int n = 0;
    const uint32_t ns = iv_nseg(A, s), nr = iv_nseg(A, r);
    for (uint32_t i = 0; i < ns; ++i) {
        const uint32_t g = iv_segid(A, s, i);
        for (uint32_t k = A.seg[g].tid_off; k < A.seg[g].tid_off + A.seg[g].ntid; ++k) {
            const uint32_t t1 = A.seg_tid[k];
            bool found = false;
            for (uint32_t j = 0; j < nr && !found; ++j) {
                const uint32_t h = iv_segid(A, r, j);
                for (uint32_t l = A.seg[h].tid_off; l < A.seg[h].tid_off + A.seg[h].ntid; ++l)
                    if (A.seg_tid[l] == t1) { found = true; break; }
            }
            if (found && n++ >= skip && !f(t1)) return false;
        }
    }
    return true;
}
CM_HD inline bool any_common_tid(const Core &c, int s, int r) {
    return !common_tids_from(c, s, r, 0, [](uint32_t) { return false; });
}
// The common transcripts of one mate pair as the extension consumes them: the first MAX_TID are kept in `t`; when the
// intersection is larger (`more`), the rest is derived again from the two intervals, in the same order, each time it is
// walked -- genes with hundreds of isoforms cost time, not an error.
struct TidList {
    const uint32_t *t;
    int n, s, r;
    bool more;
};
CM_HD inline TidList common_tids(const Core &c, int s, int r, uint32_t *out) {
    TidList tl{out, 0, s, r, false};
    common_tids_from(c, s, r, 0, [&](uint32_t t1) {
        if (tl.n < MAX_TID) {
            out[tl.n++] = t1;
            return true;
        }
        tl.more = true;
        return false;
    });
    return tl;
}

CM_HD inline bool concordant_explanation(const Core &c, const MM &sm, const MM &lm, cm_mapped_read &mr, int row, bool r1_sm, int pair_type) {
    if (sm.spos > lm.spos) return false;
    const AnnotV &A = c.A;
    int32_t tlen;
    const bool on_cdna = sm.exons_spos >= 0 && sm.exons_epos >= 0 && lm.exons_spos >= 0 && lm.exons_epos >= 0;
    const int good = (pair_type == 0) ? CM_CONCRD : CM_CONGEN;
    if (sm.exons_spos < 0 || lm.exons_spos < 0) {
        tlen = (int32_t)(lm.spos - sm.epos - 1 + lm.matched_len + sm.matched_len);
        if (tlen <= c.P.max_tlen || tlen <= MAXDISCRDTLEN) mr_update(c, mr, sm, lm, row, tlen, 0, false, CM_CONGNM, r1_sm);
    } else {
        const uint32_t na = iv_nseg(A, sm.exons_spos), nb = iv_nseg(A, lm.exons_spos);
        for (uint32_t i = 0; i < na; ++i)
            for (uint32_t j = 0; j < nb; ++j) {
                const uint32_t x = iv_segid(A, sm.exons_spos, i), y = iv_segid(A, lm.exons_spos, j);
                if (A.seg[x].start == A.seg[y].start && A.seg[x].end == A.seg[y].end) {
                    tlen = (int32_t)(lm.spos + lm.matched_len - sm.spos);
                    mr_update(c, mr, sm, lm, row, tlen, 0, on_cdna, tlen <= c.P.max_tlen ? good : CM_DISCRD, r1_sm);
                }
            }
    }
    if (sm.exons_epos < 0 || lm.exons_spos < 0) {
        tlen = (int32_t)(lm.spos - sm.epos - 1 + sm.matched_len + lm.matched_len);
        if (tlen <= c.P.max_tlen || tlen <= MAXDISCRDTLEN) mr_update(c, mr, sm, lm, row, tlen, 0, false, CM_CONGNM, r1_sm);
    } else {
        int intron_num = 0;
        tlen = calc_tlen(c, sm, lm, intron_num);
        if (tlen >= 0 && tlen <= c.P.max_tlen) {
            mr_update(c, mr, sm, lm, row, tlen, intron_num, on_cdna, good, r1_sm);
        } else {
            if (tlen < 0) {
                tlen = (int32_t)(lm.spos - sm.epos - 1 + sm.matched_len + lm.matched_len);
                intron_num = 0;
            }
            mr_update(c, mr, sm, lm, row, tlen, intron_num, on_cdna, CM_DISCRD, r1_sm);
        }
    }
    return mr.type == CM_CONCRD;
}
CM_HD inline void check_chimeric(const Core &c, const MM &sm, const MM &lm, cm_mapped_read &mr, int row, bool r1_sm) {
    if (mr.type == CM_CONCRD) return;
    if (sm.exons_spos < 0 || lm.exons_spos < 0) return;
    if (sm.spos < lm.spos && share_gene(c, sm.exons_spos, lm.exons_spos))
        mr_update(c, mr, sm, lm, row, (int32_t)(lm.epos - sm.spos + 1), 0, false, CM_CHIORF, r1_sm);
}
CM_HD inline void bsj_tail(const Core &c, const MM &sm, const MM &lm, cm_mapped_read &mr, int row, bool r1_sm, int type) {
    const AnnotV &A = c.A;
    const int32_t tl = (int32_t)(lm.epos - sm.spos + 1);
    if (sm.exons_spos < 0 || lm.exons_spos < 0) {
        if ((sm.exons_spos >= 0 && same_gene_span(c, sm.exons_spos, lm.spos, lm.epos)) ||
            (lm.exons_spos >= 0 && same_gene_span(c, lm.exons_spos, sm.spos, sm.epos))) {
            mr_update(c, mr, sm, lm, row, tl, 0, false, type, r1_sm);
            return;
        }
        if (bit_at(A.intronic_bits, A.n_bits, sm.spos) && bit_at(A.intronic_bits, A.n_bits, lm.spos) && sm.exon_ind_spos >= 0 &&
            lm.exon_ind_epos >= 0 && sm.exon_ind_spos == lm.exon_ind_epos &&
            (uint32_t)(sm.spos - A.iv[sm.exon_ind_spos].epos) <= LARIAT2BEGTH)
            mr_update(c, mr, sm, lm, row, tl, 0, false, type, r1_sm);
        return;
    }
    if (share_gene(c, sm.exons_spos, lm.exons_spos)) mr_update(c, mr, sm, lm, row, tl, 0, false, type, r1_sm);
}
CM_HD inline void check_bsj(const Core &c, const MM &sm, const MM &lm, cm_mapped_read &mr, int row, bool r1_sm) {
    if (mr.type == CM_CONCRD || mr.type == CM_DISCRD) return;
    if (!sm.right_ok || !lm.left_ok) return;
    bsj_tail(c, sm, lm, mr, row, r1_sm, CM_CHIBSJ);
}
CM_HD inline void check_2bsj(const Core &c, const MM &sm, const MM &lm, cm_mapped_read &mr, int row, bool r1_sm) {
    if (mr.type < CM_CHI2BSJ) return;
    if (sm.spos > lm.spos) return;
    if (sm.right_ok && lm.right_ok && sm.spos != lm.spos) return;
    if (sm.left_ok && lm.left_ok && sm.epos != lm.epos) return;
    if (sm.left_ok && lm.right_ok) return;
    bsj_tail(c, sm, lm, mr, row, r1_sm, CM_CHI2BSJ);
}
CM_HD inline bool is_left_chain(const CH &a, const CH &b, int read_length) {
    const uint32_t a_beg = a.rpos(0), b_beg = b.rpos(0);
    const uint32_t a_end = a.rend_excl() - 1, b_end = b.rend_excl() - 1;
    if ((b_beg > a_end) || (a_beg > b_end)) return a_beg < b_beg;
    uint32_t i = 0, j = 0;
    int best_d = INF_I, bi = -1, bj = -1;
    while (i < a.len() && j < b.len()) {
        const uint32_t bj_beg = b.rpos(j), ai_end = a.rpos(i) + a.kmer - 1;
        if (ai_end < bj_beg) {
            const int d = (int)(bj_beg - ai_end);
            if (d < best_d) { best_d = d; bi = (int)i; bj = (int)j; }
            ++i;
            continue;
        }
        const uint32_t ai_beg = a.rpos(i), bj_end = b.rpos(j) + b.kmer - 1;
        if (bj_end < ai_beg) {
            const int d = (int)(ai_beg - bj_end);
            if (d < best_d) { best_d = d; bi = (int)i; bj = (int)j; }
            ++j;
            continue;
        }
        bi = (int)i;
        bj = (int)j;
        break;
    }
    const uint32_t common_bp = cmax(a.rpos(bi), b.rpos(bj));
    const int32_t a_ov = a.qpos(bi) + (int32_t)(common_bp - a.rpos(bi));
    const int32_t b_ov = b.qpos(bj) + (int32_t)(common_bp - b.rpos(bj));
    if (a_ov < read_length && b_ov < read_length) return a_ov >= b_ov;
    return a_beg < b_beg;
}

// ------------------------------------------------------------------------------------------
// extension (A11-A13, A19)
// ------------------------------------------------------------------------------------------
struct Memo {
    MemoKey k[MEMO_N];
    AlignRes v[MEMO_N];
    int n;
    int flags;          // bit 0: an insert was dropped (table full); bit 1: an end piece shorter than its query was seen
    g_spill sp;         // entries MEMO_N .. MEMO_N + sp_cap - 1 (the exact re-run of a pair whose memo overflowed); null otherwise
    int sp_cap;
};
CM_HD inline bool memo_key_eq(const MemoKey &a, const MemoKey &k) { return a.rspos == k.rspos && a.rlen == k.rlen && a.qspos == k.qspos && a.qlen == k.qlen; }
CM_HD inline int memo_find(const Memo &m, const MemoKey &k) {
    const int np = m.n < MEMO_N ? m.n : MEMO_N;
    for (int i = 0; i < np; ++i)
        if (memo_key_eq(m.k[i], k)) return i;
    for (int i = MEMO_N; i < m.n; ++i) {
        const MemoKey a = m.sp[i - MEMO_N].k;
        if (memo_key_eq(a, k)) return i;
    }
    return -1;
}
CM_HD inline AlignRes memo_get(const Memo &m, int i) { return i < MEMO_N ? m.v[i] : m.sp[i - MEMO_N].v; }
// std::map::insert semantics (no overwrite).  The reference's memo (std::map<AllCoord, AlignRes>, src/extend.cpp:299,375) is a
// pure cache -- same key, same computation -- EXCEPT when a "middle" piece and an "end" piece share a key, where it serves the
// first-inserted kind to both.  A middle key has rlen < qlen; an end key has rlen = min(remaining window, exon) >= qlen unless
// the indels of the pieces before it sum to less than -band (rlen = qlen + band + sum(indel)).  So a full table that stops
// memoising (recomputing instead) is exact unless BOTH happen in one extend call: an insert was dropped and an end piece with
// rlen < qlen was seen.  extend_side flags that combination (ERR_MEMO) instead of returning a result that may differ from the
// reference's; the pair kernels then leave the pair untouched and map it again with a spill area behind the table (m.sp:
// entries MEMO_N .. in global memory, cm_hot.hip RetryArgs), so no pair fails a batch for the size of this table.
CM_HD inline void memo_put(Memo &m, const MemoKey &k, const AlignRes &v) {      // callers have just looked k up and missed
    if (m.n < MEMO_N) {
        m.k[m.n] = k;
        m.v[m.n] = v;
        ++m.n;
    } else if (m.n - MEMO_N < m.sp_cap) {
        m.sp[m.n - MEMO_N].k = k;
        m.sp[m.n - MEMO_N].v = v;
        ++m.n;
    } else m.flags |= 1;
}

struct Ext {
    const Core &c;
    const DpMem &sm;
    CM_HD Ext(const Core &cc, const DpMem &mem) : c(cc), sm(mem) {}

    CM_HD bool extend_middle(uint32_t pos, uint32_t exon_len, const SV &q, uint32_t qlen, int ed_th, AlignRes &best, AlignRes &curr,
                             AlignRes &exon_res, bool right) const {
        SV ref;
        if (!pac2char(c, right ? pos + 1 : pos - exon_len, (int)exon_len, ref)) return false;
        int indel, sc;
        const uint32_t seq_remain = cmin<uint32_t>(exon_len + (uint32_t)c.P.band, qlen);
        const int ed = local_alignment_side(c, sm, q, (int)seq_remain, ref, (int)exon_len, !right, indel, sc);
        const uint32_t np = right ? pos + exon_len : pos - exon_len;
        ar_set(exon_res, np, ed, 0, -indel, (int)exon_len - indel, sc);
        if (curr.ed + ed <= ed_th) {
            ar_update(curr, ed, 0, np, -indel, (int)exon_len - indel, sc);
            ar_side(c, best, curr, right);
            return true;
        }
        return false;
    }
    CM_HD void extend_end(uint32_t pos, uint32_t ref_len, const SV &q, int qlen, int ed_th, AlignRes &best, AlignRes &curr,
                          AlignRes &exon_res, bool right) const {
        SV ref;
        if (!pac2char(c, right ? pos + 1 : pos - ref_len, (int)ref_len, ref)) return;
        int sclen, indel, sc, ed;
        if (right) ed = local_alignment_sc(c, sm, ref, (int)ref_len, q, qlen, sclen, indel, sc);
        else ed = local_alignment_sc(c, sm, ref.rev((int)ref_len), (int)ref_len, q.rev(qlen), qlen, sclen, indel, sc);
        const uint32_t np = right ? pos + qlen - indel : pos - qlen + indel;
        ar_set(exon_res, np, ed, sclen, indel, qlen, sc);
        if ((curr.ed + ed <= ed_th) && (sclen <= c.P.max_sc) && (qlen - sclen >= sclen)) {
            ar_update(curr, ed, sclen, np, indel, qlen, sc);
            ar_by_score(best, curr, right);
        }
    }
    CM_HD bool middle_step(Memo &memo, const MemoKey &key, uint32_t pos, uint32_t exon_len, const SV &q, uint32_t qlen, int ed_th,
                           AlignRes &best, AlignRes &curr, AlignRes &exon_res, bool right, int &indel) const {
        CM_TICK(sm, 20);
        const int f = memo_find(memo, key);
        if (f >= 0) {
            const AlignRes r = memo_get(memo, f);
            if (curr.ed + r.ed > ed_th) return false;
            ar_update(curr, r.ed, r.sclen, r.pos, r.indel, r.qcovlen, r.score);
            ar_side(c, best, curr, right);
            indel = r.indel;
            return true;
        }
        const bool ok = extend_middle(pos, exon_len, q, qlen, ed_th, best, curr, exon_res, right);
        CM_TICK(sm, 21);
        memo_put(memo, key, exon_res);
        if (!ok) return false;
        indel = exon_res.indel;
        return true;
    }
    CM_HD void end_step(Memo &memo, const MemoKey &key, uint32_t pos, uint32_t ref_len, const SV &q, int qlen, int ed_th, AlignRes &best,
                        AlignRes &curr, AlignRes &exon_res, bool right) const {
        CM_TICK(sm, 16);
        if (key.rlen < key.qlen) memo.flags |= 2;
        const int f = memo_find(memo, key);
        if (f >= 0) {
            const AlignRes r = memo_get(memo, f);
            if ((curr.ed + r.ed > ed_th) || (r.sclen > c.P.max_sc) || (r.qcovlen - r.sclen < r.sclen)) return;
            ar_update(curr, r.ed, r.sclen, r.pos, r.indel, r.qcovlen, r.score);
            ar_by_score(best, curr, right);
            CM_TICK(sm, 17);
        } else {
            extend_end(pos, ref_len, q, qlen, ed_th, best, curr, exon_res, right);
            CM_TICK(sm, 18);
            memo_put(memo, key, exon_res);
            CM_TICK(sm, 19);
        }
    }

    // it_seg / it_ind: get_location_overlap_ind(pos), looked up once per extend call (same pos for every tid)
    CM_HD void right_trans(uint32_t tid, uint32_t pos, int it_seg, int it_ind, int ref_len, const SV &q, int qlen, int ed_th, uint32_t ub,
                           AlignRes &best, bool &consecutive, Memo &memo) const {
        const AnnotV &A = c.A;
        consecutive = false;
        CM_STAT(7, 1);
        AlignRes curr = ar_init(ub), exon_res = ar_init(ub);
        if (it_seg < 0) return;
        int covered = 0;
        const int it_start = A.tr[tid].start_ind;
        const int rel_ind = it_ind - it_start;
        const uint32_t tsz = A.tr[tid].t2s_len;
        const g_u8 t2s = A.t2s + A.tr[tid].t2s_off;
        uint32_t rspos = pos;
        int exon_len = (int)(A.iv[it_seg].epos - pos);
        int remain_ref_len = ref_len;
        int indel = 0;
        for (unsigned int i = (unsigned int)(rel_ind + 1); i < tsz; ++i) {
            if (exon_len >= qlen - covered) break;
            const uint8_t st = t2s[i];
            if (st == 1) {
                indel = 0;
                if (exon_len > 0) {
                    if (rspos + exon_len > ub) return;
                    const uint32_t rq = (uint32_t)cmin(exon_len + c.P.band, qlen - covered);
                    const MemoKey key{rspos, (uint32_t)exon_len, (uint32_t)covered, rq};
                    if (!middle_step(memo, key, rspos, (uint32_t)exon_len, q.sub(covered), rq, ed_th, best, curr, exon_res, true, indel)) return;
                }
                remain_ref_len -= exon_len;
                covered += exon_len + indel;
                exon_len = 0;
                rspos = A.iv[(int)i + it_start].spos - 1;
            }
            if (st != 0) {
                const int iv = (int)i + it_start;
                exon_len += (int)(A.iv[iv].epos - A.iv[iv].spos + 1);
            }
        }
        if ((exon_len > 0) && (exon_len < qlen - covered) && (rspos + exon_len <= ub)) {
            const uint32_t rq = (uint32_t)cmin(exon_len + c.P.band, qlen - covered);
            const MemoKey key{rspos, (uint32_t)exon_len, (uint32_t)covered, rq};
            middle_step(memo, key, rspos, (uint32_t)exon_len, q.sub(covered), rq, ed_th, best, curr, exon_res, true, indel);
            return;
        }
        if (covered >= qlen || (rspos + qlen - covered > ub) || (exon_len < qlen - covered)) return;
        consecutive = (rspos == pos);
        remain_ref_len = cmin(remain_ref_len, exon_len);
        const MemoKey key{rspos, (uint32_t)remain_ref_len, (uint32_t)covered, (uint32_t)(qlen - covered)};
        end_step(memo, key, rspos, (uint32_t)remain_ref_len, q.sub(covered), qlen - covered, ed_th, best, curr, exon_res, true);
    }

    CM_HD void left_trans(uint32_t tid, uint32_t pos, int it_seg, int it_ind, int ref_len, const SV &q, int qlen, int ed_th, uint32_t lb,
                          AlignRes &best, bool &consecutive, Memo &memo) const {
        const AnnotV &A = c.A;
        consecutive = false;
        AlignRes curr = ar_init(lb), exon_res = ar_init(lb);
        if (it_seg < 0) return;
        int covered = 0;
        const int it_start = A.tr[tid].start_ind;
        const int rel_ind = it_ind - it_start;
        const uint32_t tsz = A.tr[tid].t2s_len;
        const g_u8 t2s = A.t2s + A.tr[tid].t2s_off;
        uint32_t lepos = pos;
        int exon_len = 0;
        int remain_ref_len = ref_len;
        int indel = 0;
        bool first_seg = true;
        for (int i = rel_ind; i >= 0; --i) {
            const uint8_t st = ((uint32_t)i < tsz) ? t2s[i] : 0;
            if (st != 0) {
                const int iv = i + it_start;
                if (first_seg) {
                    exon_len = (int)(pos - A.iv[iv].spos);
                    first_seg = false;
                } else {
                    if (exon_len == 0) lepos = A.iv[iv].epos + 1;
                    exon_len += (int)(A.iv[iv].epos - A.iv[iv].spos + 1);
                }
            }
            if (exon_len >= qlen - covered) break;
            if (st == 1) {
                indel = 0;
                if (exon_len > 0) {
                    if (lepos < lb + exon_len) return;
                    const uint32_t rq = (uint32_t)cmin(exon_len + c.P.band, qlen - covered);
                    const MemoKey key{lepos, (uint32_t)exon_len, (uint32_t)covered, rq};
                    if (!middle_step(memo, key, lepos, (uint32_t)exon_len, q.sub(qlen - covered - (int)rq), rq, ed_th, best, curr, exon_res, false, indel)) return;
                }
                remain_ref_len -= exon_len;
                covered += exon_len + indel;
                exon_len = 0;
            }
        }
        if ((exon_len > 0) && (exon_len < qlen - covered) && (lepos >= lb + exon_len)) {
            const uint32_t rq = (uint32_t)cmin(exon_len + c.P.band, qlen - covered);
            const MemoKey key{lepos, (uint32_t)exon_len, (uint32_t)covered, rq};
            middle_step(memo, key, lepos, (uint32_t)exon_len, q.sub(qlen - covered - (int)rq), rq, ed_th, best, curr, exon_res, false, indel);
            return;
        }
        if (covered >= qlen || (lepos < lb + qlen - covered) || (exon_len < qlen - covered)) return;
        consecutive = (lepos == pos);
        remain_ref_len = cmin(remain_ref_len, exon_len);
        const MemoKey key{lepos, (uint32_t)remain_ref_len, (uint32_t)covered, (uint32_t)(qlen - covered)};
        end_step(memo, key, lepos, (uint32_t)remain_ref_len, q, qlen - covered, ed_th, best, curr, exon_res, false);
    }

    // The genome fall-back of extend_right / extend_left (src/extend.cpp:334-343, 410-419): the window of len + band bases next to
    // orig_pos against the whole residual q (len chars), as local_alignment_sc takes them.  Shared with the DP pool of k_pair_heavy,
    // which must predict exactly these views.
    CM_HD bool fallback_views(uint32_t orig_pos, int len, const SV &q, bool right, SV &s, int &n, SV &t, int &m) const {
        const int ref_len = len + c.P.band;
        SV ref;
        if (!pac2char(c, right ? orig_pos + 1 : orig_pos - ref_len, ref_len, ref)) return false;
        s = right ? ref : ref.rev(ref_len);
        t = right ? q : q.rev(len);
        n = ref_len;
        m = len;
        return true;
    }
    // extend_right / extend_left, src/extend.cpp:285-432; q = the residual (len chars)
    CM_HD bool extend_side(const TidList &tl, const SV &q, uint32_t &pos, int len, int ed_th, uint32_t bound, AlignRes &best,
                           bool right) const {
        CM_STAT(3, 1);
        const int seq_len = len, ref_len = len + c.P.band;
        const uint32_t orig_pos = pos;
        bool consecutive = false;
        ar_set(best, pos, ed_th + 1, len + 1, c.P.band + 1, 0, 0);
        Memo memo;
        memo.n = 0;
        memo.flags = 0;
        memo.sp = sm.spill;
        memo.sp_cap = sm.spill ? sm.spill_cap : 0;
        int it_ind = -1, it_seg = -1;
        CM_TICK(sm, 22);
        if (tl.n > 0) it_seg = overlap_ind(c, pos, it_ind);
        CM_TICK(sm, 23);
        auto along = [&](uint32_t tid) {
            if (right) right_trans(tid, pos, it_seg, it_ind, ref_len, q, seq_len, ed_th, bound, best, consecutive, memo);
            else left_trans(tid, pos, it_seg, it_ind, ref_len, q, seq_len, ed_th, bound, best, consecutive, memo);
            return true;
        };
        for (int i = 0; i < tl.n; ++i) along(tl.t[i]);
        if (tl.more) common_tids_from(c, tl.s, tl.r, MAX_TID, along);
#if defined(CM_MEMO_STRICT)          // test builds: every dropped insert sends the pair to the re-run launch
        if (memo.flags & 1) flag_err(sm.err, ERR_MEMO);
#else
        if (memo.flags == 3) flag_err(sm.err, ERR_MEMO);
#endif
        int min_ed = best.ed, sclen_best = best.sclen;
        CM_TICK(sm, 11);
        if (min_ed <= ed_th) {
            pos = right ? best.pos - sclen_best : best.pos + sclen_best;
            if (best.qcovlen >= seq_len && sclen_best <= c.P.max_sc) return true;
        }
        SV fs, ft;
        int fn, fm;
        if (!consecutive && fallback_views(orig_pos, len, q, right, fs, fn, ft, fm)) {
            int indel, sc;
            min_ed = local_alignment_sc(c, sm, fs, fn, ft, fm, sclen_best, indel, sc);
            CM_TICK(sm, 12);
            if (min_ed <= ed_th && sclen_best <= c.P.max_sc) {
                const uint32_t np = right ? orig_pos + seq_len - indel : orig_pos - seq_len + indel;
                AlignRes curr = ar_init(bound);
                ar_set(curr, np, min_ed, sclen_best, indel, seq_len, sc);
                if (ar_by_score(best, curr, right)) {
                    pos = right ? np - sclen_best : np + sclen_best;
                    return true;
                }
            }
        }
        if (best.qcovlen <= 0) {
            pos = orig_pos;
            ar_set(best, pos, 0, 0, 0, 0, -INF_I);
        }
        const int qremain = seq_len - best.qcovlen;
        if (qremain + best.sclen <= c.P.max_sc) {
            ar_set(best, pos, best.ed, best.sclen + qremain, best.indel, seq_len, best.score);
            return true;
        }
        return (best.qcovlen >= seq_len && best.ed <= ed_th);
    }

    template <class CHT> CM_HD bool chain_right(const TidList &tl, const CHT &ch, const SV &seq, int seq_len, uint32_t ub, MM &mr, int &err) const {
        uint32_t rm_pos = ch.rend_excl() - 1;
        int remain_end = seq_len - ch.qend_excl();
        bool right_ok = (remain_end <= 0);
        AlignRes best = ar_init(ub);
        if (remain_end > 0) right_ok = extend_side(tl, seq.sub(seq_len - remain_end), rm_pos, remain_end, c.P.max_ed - err, ub, best, true);
        const int sclen_right = best.sclen, err_right = best.ed;
        remain_end -= best.qcovlen;
        mr.epos = rm_pos;
        mr.matched_len -= (uint32_t)(right_ok ? sclen_right : remain_end);
        mr.qepos -= (uint32_t)(right_ok ? sclen_right : remain_end);
        mr.sclen_right = sclen_right;
        mr.right_ed = best.ed;
        err += err_right;
        return right_ok;
    }
    template <class CHT> CM_HD bool chain_left(const TidList &tl, const CHT &ch, const SV &seq, int32_t qspos, uint32_t lb, MM &mr, int &err) const {
        uint32_t lm_pos = ch.rpos(0);
        int remain_beg = ch.qpos(0) - qspos;
        bool left_ok = (remain_beg <= 0);
        AlignRes best = ar_init(lb);
        if (remain_beg > 0) left_ok = extend_side(tl, seq, lm_pos, remain_beg, c.P.max_ed - err, lb, best, false);
        const int sclen_left = best.sclen, err_left = best.ed;
        remain_beg -= best.qcovlen;
        mr.spos = lm_pos;
        mr.matched_len -= (uint32_t)(left_ok ? sclen_left : remain_beg);
        mr.qspos += (uint32_t)(left_ok ? sclen_left : remain_beg);
        mr.sclen_left = sclen_left;
        mr.left_ed = best.ed;
        err += err_left;
        return left_ok;
    }
    template <class CHT> CM_HD int calc_middle_ed(const CHT &ch, int edth, const SV &q) const {
        int mid = 0;
        if (ch.len() == 0) return 0;
        for (uint32_t i = 0; i + 1 < ch.len(); ++i) {
            if (ch.qpos(i + 1) > ch.qpos(i) + ch.kmer) {
                const int diff = (int)(ch.rpos(i + 1) - ch.rpos(i)) - (ch.qpos(i + 1) - ch.qpos(i));
                const int32_t qspos = ch.qpos(i) + ch.kmer;
                const int qlen = ch.qpos(i + 1) - qspos;
                const uint32_t rspos = ch.rpos(i) + (uint32_t)ch.kmer;
                int rlen = qlen + diff;
                if (rlen < 0) rlen = 0;
                if (diff >= -c.P.band && diff <= c.P.band) {
                    SV ref;
                    if (!pac2char(c, rspos, rlen, ref)) ref = SV{c.X.genome, 0, 1, 2};    // defined as an all-NUL window
                    const SV qs = q.sub(qspos);
                    const bool qf = diff >= 0;                      // one call site, arguments selected
                    mid += one_side_banded(c, sm, qf ? qs : ref, qf ? qlen : rlen, qf ? ref : qs, qf ? rlen : qlen, qf ? diff : -diff);
                }
                if (mid > edth) return edth + 1;
            }
        }
        return mid;
    }
    CM_HD bool both_mates(const CH &lch, const CH &rch, const TidList &tl, const Read &lr, const Read &rr, MM &lmm, MM &rmm) const {
        const int maxEd = c.P.max_ed;
        const SV lseq = lr.view(), rseq = rr.view();
        CM_TICK(sm, 3);
        lmm.middle_ed = calc_middle_ed(lch, maxEd, lseq);
        rmm.middle_ed = calc_middle_ed(rch, maxEd, rseq);
        if (lmm.middle_ed <= maxEd) is_concord_impl(lch, (uint32_t)lr.len, lmm, true);
        if (rmm.middle_ed <= maxEd) is_concord_impl(rch, (uint32_t)rr.len, rmm, true);
        if (lmm.middle_ed > maxEd || rmm.middle_ed > maxEd) return false;
        CM_DBG_STOP(2, false);
        CM_TICK(sm, 4);
        lmm.is_concord = false;
        rmm.is_concord = false;
        int lerr = lmm.middle_ed, rerr = rmm.middle_ed;
        // chain_len is never 0 for a stored chain, so both mates extend (src/extend.cpp:60-74)
        lmm.matched_len = (uint32_t)lr.len;
        lmm.qspos = 1;
        lmm.qepos = (uint32_t)lr.len;
        const bool llok = chain_left(tl, lch, lseq, 0, MINLB, lmm, lerr);
        CM_DBG_STOP(3, false);
        CM_TICK(sm, 5);
        rmm.matched_len = (uint32_t)rr.len;
        rmm.qspos = 1;
        rmm.qepos = (uint32_t)rr.len;
        const bool rlok = chain_left(tl, rch, rseq, 0, lmm.spos, rmm, rerr);
        CM_DBG_STOP(4, false);
        CM_TICK(sm, 6);
        const bool rrok = chain_right(tl, rch, rseq, rr.len, MAXUB, rmm, rerr);
        CM_DBG_STOP(5, false);
        CM_TICK(sm, 7);
        const bool lrok = chain_right(tl, lch, lseq, lr.len, rmm.epos, lmm, lerr);
        CM_DBG_STOP(6, false);
        CM_TICK(sm, 8);
        update_match_mate_info(c, llok, lrok, lerr, lmm);
        update_match_mate_info(c, rlok, rrok, rerr, rmm);
        return true;
    }
    CM_HD int chain_both_sides(const CH &ch, const Read &rd, MM &mr, int dir) const {
        CM_STAT(2, 1);
        const int maxEd = c.P.max_ed;
        const SV seq = rd.view();
        const int seq_len = rd.len;
        mr.is_concord = false;
        mr.middle_ed = estimate_middle_error(c, ch);
        if (is_concord_impl(ch, (uint32_t)seq_len, mr, false)) {
            mr.dir = dir;
            return mr.type;
        }
        uint32_t lm_pos = ch.rpos(0);
        int remain_beg = ch.qpos(0);
        bool left_ok = (remain_beg <= 0);
        AlignRes bl = ar_init(MINLB);
        if (remain_beg > 0) left_ok = extend_side(TidList{nullptr, 0, -1, -1, false}, seq, lm_pos, remain_beg, maxEd - mr.middle_ed, MINLB, bl, false);
        const int err_left = bl.ed, sclen_left = bl.sclen;
        remain_beg -= bl.qcovlen;
        uint32_t rm_pos = ch.rend_excl() - 1;
        int remain_end = seq_len - ch.qend_excl();
        bool right_ok = (remain_end <= 0);
        AlignRes br = ar_init(MAXUB);
        if (remain_end > 0) right_ok = extend_side(TidList{nullptr, 0, -1, -1, false}, seq.sub(seq_len - remain_end), rm_pos, remain_end, maxEd - mr.middle_ed - err_left, MAXUB, br, true);
        const int err_right = br.ed, sclen_right = br.sclen;
        remain_end -= br.qcovlen;
        mr.spos = lm_pos;
        mr.epos = rm_pos;
        mr.matched_len = (uint32_t)seq_len;
        mr.matched_len -= (uint32_t)(left_ok ? sclen_left : remain_beg);
        mr.matched_len -= (uint32_t)(right_ok ? sclen_right : remain_end);
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

// ------------------------------------------------------------------------------------------
// K3 body: pair chains, extend, classify (A8-A10, A20)
// ------------------------------------------------------------------------------------------
struct ChainSet {          // chains of one (mate, orientation)
    g_chain ch;
    int n;
};

// One mate pair of process_mates' loop (filter.cpp:261-342), split in two so that the expensive,
// order-independent half can run as parallel tasks:
//   extend_task : decide the left mate, extend both mates, look up the exon intervals of the ends;
//   fold_task   : apply the outcome to the pair's MatchedRead — must be applied in (i, j) order;
//                 returns true where the reference returns CONCRD from inside the loop.
CM_HD inline void extend_task(const Core &c, const Ext &ext, const CH &F, const CH &R, const TidList &tl, const Read &frd,
                              const Read &brd, MM &r1, MM &r2, bool &is_left, bool &ok, int &row) {
    r1 = mm_init(c);
    r2 = mm_init(c);
    r1.dir = 1;
    r2.dir = -1;
    row = 0;
    is_left = is_left_chain(F, R, frd.len);
    // one call site (see process_read): left mate first, whichever read it is
    ok = ext.both_mates(is_left ? F : R, is_left ? R : F, tl, is_left ? frd : brd, is_left ? brd : frd, is_left ? r1 : r2, is_left ? r2 : r1);
    if (ok) {
        row = chr_row(c, is_left ? r1.spos : r2.spos);
        overlap_to_epos(c, r1); overlap_to_spos(c, r1);
        overlap_to_epos(c, r2); overlap_to_spos(c, r2);
    }
}
CM_HD inline bool fold_task(const Core &c, const MM &r1, const MM &r2, bool is_left, bool ok, int row, int pair_type, bool r1_forward,
                            cm_mapped_read &mr) {
    if (!ok) return false;
    const bool both_c = r1.type == CM_CONCRD && r2.type == CM_CONCRD;
    const bool one_c = (r1.type == CM_CANDID && r2.type == CM_CONCRD) || (r1.type == CM_CONCRD && r2.type == CM_CANDID);
    const bool both_x = r1.type == CM_CANDID && r2.type == CM_CANDID;
    if (is_left) {
        if (both_c) {
            if (concordant_explanation(c, r1, r2, mr, row, r1_forward, pair_type) && c.P.scan_level == 0) return true;
        } else if (one_c) check_bsj(c, r1, r2, mr, row, r1_forward);
        else if (both_x) check_2bsj(c, r1, r2, mr, row, r1_forward);
    } else {
        if (both_c) check_chimeric(c, r2, r1, mr, row, !r1_forward);
        else if (one_c) check_bsj(c, r2, r1, mr, row, !r1_forward);
        else if (both_x) check_2bsj(c, r2, r1, mr, row, !r1_forward);
    }
    return false;
}
// the pairing predicate of pair_chains for one (i, j): 0 = not paired, else pair type + 1
// Quick exact reject (the 30 x 30 chain pairs of a read from a repeat family are mostly copies megabases apart): let d = the
// distance of the two chain starts.  d > MAXDISCRDTLEN makes tlen > MAXDISCRDTLEN (tlen spans both starts).  same_tr needs a
// transcript T with a segment in both starts' intervals, so d <= the extent of T's intervals; same_gen needs a gene whose span
// contains one chain while a segment of it lies in the other start's interval, so d <= the hull of that interval and that gene
// span.  pair_reach = the largest such extent / hull of the contig's annotation (cm_aos.h); beyond it none of the three holds.
CM_HD inline uint32_t pair_code(const Core &c, const CHEnds &F, const CHEnds &R, int fe_i, int re_j, int saved_type) {
    const uint32_t fs = F.r0, rs = R.r0, fe_ = F.rend, re_ = R.rend;
    {
        const uint32_t d = fs > rs ? fs - rs : rs - fs;
        if (d > (uint32_t)MAXDISCRDTLEN && d > c.A.pair_reach) return 0u;
    }
    const int tlen = (int)((fs < rs) ? (re_ - fs) : (fe_ - rs));
    bool same_tr = false, same_gen = false;
    if (fe_i >= 0 && re_j >= 0) same_tr = any_common_tid(c, fe_i, re_j);
    if (!same_tr && fe_i >= 0 && ((c.P.scan_level == 0 && saved_type > CM_CONGEN) || (c.P.scan_level > 0 && saved_type >= CM_CONGEN)))
        same_gen = same_gene_span(c, fe_i, rs, re_);
    if (!same_gen && re_j >= 0 && saved_type >= CM_CONGEN) same_gen = same_gene_span(c, re_j, fs, fe_);
    if (same_tr || same_gen || ((tlen <= MAXDISCRDTLEN) && (saved_type >= CM_CONGNM))) return same_tr ? 1u : (same_gen ? 2u : 3u);
    return 0u;
}
// the category process_mates derives from the unpaired-chain extensions (filter.cpp:387-393)
CM_HD inline int leftover_type(int min_ret1, int min_ret2, bool r1_genic, bool r2_genic) {
    return (((min_ret1 == CM_ORPHAN) && (min_ret2 == CM_CONCRD)) || ((min_ret1 == CM_CONCRD) && (min_ret2 == CM_ORPHAN))) ? CM_OEANCH
         : ((min_ret1 == CM_ORPHAN) || (min_ret2 == CM_ORPHAN)) ? CM_ORPHAN
         : ((min_ret1 == CM_CONCRD) && (min_ret2 == CM_CONCRD) && (r1_genic && r2_genic)) ? CM_CHIFUS
         : ((min_ret1 == CM_CONCRD) && (min_ret2 == CM_CONCRD)) ? CM_OEA2 : CM_CANDID;
}

// The unpaired-chain extensions of process_mates (filter.cpp:356-393) feed nothing but leftover_type(), whose
// value is applied with mr_update_type, i.e. only if it is lower than the type T the pair already has.
// leftover_type >= CHIFUS; its values below OEANCH need both sides to end without ORPHAN, those below
// CANDID (CHIFUS, OEA2) need both sides to end CONCRD.  can1 / can2: side still has chains to extend.
// Returns false when no outcome of the remaining extensions can change mr.type: they are dead work
// (full-length DPs of spurious chains, typically) and are skipped; results are identical.
CM_HD inline bool leftovers_matter(int T, int min_ret1, bool can1, int min_ret2, bool can2) {
    if (T <= CM_CHIFUS) return false;
    if (T <= CM_CANDID && ((min_ret1 != CM_CONCRD && !can1) || (min_ret2 != CM_CONCRD && !can2))) return false;
    if (T <= CM_OEANCH && ((min_ret1 == CM_ORPHAN && !can1) || (min_ret2 == CM_ORPHAN && !can2))) return false;
    return true;
}

// FilterRead::process_mates (filter.cpp:244-395) with pair_chains (filter.cpp:484-551) fused in:
// pass 1 evaluates the pairing predicate for every (i, j) (needed up-front for the *_paired
// flags), pass 2 walks the accepted pairs in i-major order.
CM_HD inline int process_mates(const Core &c, const DpMem &sm, const ChainSet &fwd, const Read &frd, const ChainSet &bwd, const Read &brd,
                               cm_mapped_read &mr, bool r1_forward, g_err err) {
    const Ext ext(c, sm);
    const int kmer = c.P.kmer;
    const int saved_type = mr.type;
    int fe[CM_BESTCHAINLIM], re[CM_BESTCHAINLIM];
    CM_TICK(sm, 0);
    for (int i = 0; i < fwd.n; ++i) fe[i] = overlap(c, fwd.ch[i].rpos[0]);
    for (int j = 0; j < bwd.n; ++j) re[j] = overlap(c, bwd.ch[j].rpos[0]);
    uint32_t ptype[(CM_BESTCHAINLIM * CM_BESTCHAINLIM * 2 + 31) / 32];   // 2 bits per (i,j): 0 none, 1..3 = type+1
    const int n_ptype = fwd.n > 0 ? (((fwd.n - 1) * CM_BESTCHAINLIM + bwd.n) >> 4) + 1 : 0;      // words the loops below touch
    for (int x = 0; x < n_ptype; ++x) ptype[x] = 0;
    uint32_t fpaired = 0, bpaired = 0;
    uint32_t tids[MAX_TID];
    CM_STAT(1, fwd.n * bwd.n);
    CM_STAT(6, 1);
    for (int i = 0; i < fwd.n; ++i)
        for (int j = 0; j < bwd.n; ++j) {
            const CHEnds F{fwd.ch + i, kmer}, R{bwd.ch + j, kmer};
            const uint32_t code = pair_code(c, F, R, fe[i], re[j], saved_type);
            if (code) {
                const int idx = i * CM_BESTCHAINLIM + j;
                ptype[idx >> 4] |= code << ((idx & 15) * 2);
                fpaired |= 1u << i;
                bpaired |= 1u << j;
            }
        }
    CM_DBG_STOP(1, mr.type);
    CM_TICK(sm, 1);
    int min_ret1 = CM_ORPHAN, min_ret2 = CM_ORPHAN;
    bool r1_genic = false, r2_genic = false;
    for (int i = 0; i < fwd.n; ++i)
        for (int j = 0; j < bwd.n; ++j) {
            const int idx = i * CM_BESTCHAINLIM + j;
            const uint32_t code = (ptype[idx >> 4] >> ((idx & 15) * 2)) & 3u;
            if (code == 0) continue;
            const int pair_type = (int)code - 1;
            const TidList tl = (code == 1) ? common_tids(c, fe[i], re[j], tids) : TidList{tids, 0, -1, -1, false};
            // (when same_tr is false the reference's common_tid is empty: same_transcript clears it)
            CM_STAT(0, 1);
            MM r1, r2;
            bool is_left, ok;
            int row;
            const CH F{fwd.ch + i, kmer}, R{bwd.ch + j, kmer};
            CM_TICK(sm, 2);
            extend_task(c, ext, F, R, tl, frd, brd, r1, r2, is_left, ok, row);
            CM_TICK(sm, 9);
            if (fold_task(c, r1, r2, is_left, ok, row, pair_type, r1_forward, mr)) return CM_CONCRD;
            CM_TICK(sm, 10);
            min_ret1 = cmin(r1.type, min_ret1);
            min_ret2 = cmin(r2.type, min_ret2);
            r1_genic = (r1.exons_spos >= 0) || (r1.exons_epos >= 0);
            r2_genic = (r2.exons_spos >= 0) || (r2.exons_epos >= 0);
        }
    if (mr.type == CM_CONCRD || mr.type == CM_DISCRD || mr.type == CM_CHIORF || mr.type == CM_CHIBSJ || mr.type == CM_CHI2BSJ) return mr.type;
    const uint32_t fun = ~fpaired & (fwd.n >= 32 ? 0xffffffffu : ((1u << fwd.n) - 1u));
    const uint32_t bun = ~bpaired & (bwd.n >= 32 ? 0xffffffffu : ((1u << bwd.n) - 1u));
    CM_TICK(sm, 10);
    if (!leftovers_matter(mr.type, min_ret1, min_ret1 != CM_CONCRD && fun != 0, min_ret2, min_ret2 != CM_CONCRD && bun != 0)) return mr.type;
    MM mm1 = mm_init(c);      // deliberately not reset between chains (stale looked_up_* caches, filter.cpp:356-370)
    if (min_ret1 != CM_CONCRD)
        for (int i = 0; i < fwd.n; ++i)
            if (!((fpaired >> i) & 1u)) {
                const CH F{fwd.ch + i, kmer};
                const int ex = ext.chain_both_sides(F, frd, mm1, 1);
                min_ret1 = cmin(ex, min_ret1);
                overlap_to_spos(c, mm1);
                overlap_to_epos(c, mm1);
                r1_genic = (mm1.exons_spos >= 0) || (mm1.exons_epos >= 0);
            }
    if (!leftovers_matter(mr.type, min_ret1, false, min_ret2, min_ret2 != CM_CONCRD && bun != 0)) return mr.type;
    MM mm2 = mm_init(c);
    if (min_ret2 != CM_CONCRD)
        for (int j = 0; j < bwd.n; ++j)
            if (!((bpaired >> j) & 1u)) {
                const CH R{bwd.ch + j, kmer};
                const int ex = ext.chain_both_sides(R, brd, mm2, -1);
                min_ret2 = cmin(ex, min_ret2);
                overlap_to_spos(c, mm2);
                overlap_to_epos(c, mm2);
                r2_genic = (mm2.exons_spos >= 0) || (mm2.exons_epos >= 0);
            }
    const int new_type = leftover_type(min_ret1, min_ret2, r1_genic, r2_genic);
    mr_update_type(mr, new_type);
    CM_TICK(sm, 14);
    return mr.type;
}

// FilterRead::process_read (PE), filter.cpp:124-241, after seeding + chaining.
// sets[0..3] = chains of (R1 fwd, R1 rc, R2 fwd, R2 rc); high[] = high_hits of each.
CM_HD inline int process_read(const Core &c, const DpMem &sm, g_u8 s1, int len1, g_u8 s2, int len2, const ChainSet *sets,
                              const int *high, cm_mapped_read &mr, g_err err) {
    const int n1 = sets[0].n + sets[1].n, n2 = sets[2].n + sets[3].n;
    if (n1 + n2 <= 0) {
        if ((high[0] + high[1] > 0) && (high[2] + high[3] > 0)) {
            mr_update_type(mr, CM_NOPROC_MANYHIT);
            return CM_NOPROC_MANYHIT;
        }
        mr_update_type(mr, CM_NOPROC_NOMATCH);
        return CM_NOPROC_NOMATCH;
    }
    if (n1 <= 0 || n2 <= 0) {
        mr_update_type(mr, CM_OEANCH);
        return CM_OEANCH;
    }
    const float fc1 = sets[0].n > 0 ? sets[0].ch[0].score : 0.f, bc1 = sets[1].n > 0 ? sets[1].ch[0].score : 0.f;
    const float fc2 = sets[2].n > 0 ? sets[2].ch[0].score : 0.f, bc2 = sets[3].n > 0 ? sets[3].ch[0].score : 0.f;
    const float lhs = fc1 + bc2, rhs = fc2 + bc1;
    const Read r1f{s1, len1, 0}, r1b{s1, len1, 1}, r2f{s2, len2, 0}, r2b{s2, len2, 1};
    const bool first = lhs >= rhs;
    for (int attempt = 0; attempt < 2; ++attempt) {
        // one call site for both orientations (arguments selected, not branched on): the lanes of a wave whose pairs
        // have opposite orientations stay converged inside process_mates
        const bool r1_fwd = (attempt == 0) == first;                   // forward R1 / backward R2, else forward R2 / backward R1
        const int a = process_mates(c, sm, r1_fwd ? sets[0] : sets[2], r1_fwd ? r1f : r2f, r1_fwd ? sets[3] : sets[1], r1_fwd ? r2b : r1b, mr,
                                    r1_fwd, err);
        if (c.P.scan_level == 0 && a == CM_CONCRD) return CM_CONCRD;
    }
    return mr.type;
}

// the skip / requeue rule of map_reads (src/circminer.cpp:386-397) plus what the next round's
// fill_map_info would read back from the remain FASTQ header (fastq_parser.cpp:214-267)
CM_HD inline void finish_round(const Core &c, int state, int is_last, int len1, int len2, cm_mapped_read &mr, uint8_t &active) {
    const bool skip = (c.P.scan_level == 0 && state == CM_CONCRD) ||
                      (c.P.scan_level == 1 && state == CM_CONCRD && mr.gm_compatible && (mr.ed_r1 + mr.ed_r2 == 0) &&
                       (mr.mlen_r1 + mr.mlen_r2 == (uint32_t)(len1 + len2)));
    const bool requeue = (!is_last && !skip) || (is_last && (mr.type == CM_CHIBSJ || mr.type == CM_CHI2BSJ));
    active = requeue ? 1 : 0;
    if (requeue && !is_last && !mapped_type(mr.type)) {
        const int t = mr.type;
        default_mr(c, mr);
        mr.type = t;
    }
}

}  // namespace cmc
