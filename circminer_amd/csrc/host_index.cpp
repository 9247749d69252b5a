// Host-side k-mer index builder (stays on host: BASELINE.json north_star "index build ... on host").
//
// In-memory equivalent of generateHashTableOnDisk for one packed contig
// (reference src/mrsfast/HashTable.c:257-380 count pass, :769-821 scatter pass,
// :824-839 + src/mrsfast/Sort.c:116-117 per-bucket order), emitted in the flattened layout of
// cm_index_view instead of the reference's pointer table + (count14+1)-stride arena:
//
//   * a k-mer (k = 14 + c) is indexed iff all k bases are upper-case A/C/G/T
//     (HashTable.c:274-279, 799-806); its bucket is the 2-bit value of the first 14 bases,
//     its checksum the 2-bit value of the remaining c bases, its position the 1-based start;
//   * inside a bucket entries are ordered by (checksum, position).
//
// Not built here (not needed by the probe side): the slack slots the reference allocates for
// 14-mers whose full k-mer is invalid (HashTable.c:1065-1090) and the [0].info count header
// (the count is bucket_off[hv+1]-bucket_off[hv]).
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "circminer_hot.h"

namespace {

struct BaseLut {
    uint8_t v[256];
    BaseLut() {
        memset(v, 4, sizeof v);
        v[(uint8_t)'A'] = 0;
        v[(uint8_t)'C'] = 1;
        v[(uint8_t)'G'] = 2;
        v[(uint8_t)'T'] = 3;
    }
};
const BaseLut BASE_LUT;
inline int base_code(uint8_t ch) { return BASE_LUT.v[ch]; }

// Calls f(bucket, checksum, start1) for every indexable k-mer, in ascending start order.
template <class F>
void for_each_kmer(const uint8_t *g, uint32_t n, int k, int c, F f) {
    const uint64_t kmask = (k == 32) ? ~0ull : ((1ull << (2 * k)) - 1);
    const uint64_t cmask = c ? ((1ull << (2 * c)) - 1) : 0;
    uint64_t v = 0;
    int run = 0;
    for (uint32_t i = 0; i < n; ++i) {
        int b = base_code(g[i]);
        if (b == 4) {
            run = 0;
            v = 0;
            continue;
        }
        v = ((v << 2) | (uint64_t)b) & kmask;
        if (++run >= k) f((uint32_t)(v >> (2 * c)), (uint16_t)(v & cmask), i + 2 - (uint32_t)k);
    }
}

}  // namespace

// Parallel build without a shared random-access phase.  The 4^14 buckets are cut into 4096 ranges (top 12 bits of the bucket
// number); the contig is cut into slices.
//   pass 1  (parallel over slices)  count the indexable k-mers of a slice per range;
//   pass 2  (parallel over slices)  write each k-mer as one 64-bit word (low 16 bucket bits | checksum | position) into the
//                                   temp array at [range][slice] order: 4096 sequential write streams per thread;
//   pass 3  (parallel over ranges)  a range's words are contiguous in temp and already in ascending position; count its 65536
//                                   buckets, prefix-sum, scatter into the final (checksum, pos) arrays and order each bucket
//                                   by checksum -- all inside a few hundred KB, i.e. in cache.
// A range's span in temp IS its span in the final arrays, so bucket offsets are range start + local prefix.  Every bucket
// receives its entries in ascending position order, like the serial scatter of the reference (HashTable.c:769-821), so the
// result is identical for any thread count.  1.06 Gbp on 16 threads: a few seconds (was 40-60 s with a random scatter).
template <class F>
static void run_threads(int nt, F f) {
    if (nt <= 1) {
        f(0);
        return;
    }
    std::vector<std::thread> th;
    for (int t = 0; t < nt; ++t) th.emplace_back(f, t);
    for (auto &t : th) t.join();
}

extern "C" int cm_host_build_index(const uint8_t *genome, uint32_t ref_len, int32_t kmer, int32_t contig_num,
                                   int n_threads, cm_index_view *out) {
    if (!genome || !out || kmer < CM_WINDOW_SIZE || kmer > CM_WINDOW_SIZE + 8) return CM_EINVAL;
    const int c = kmer - CM_WINDOW_SIZE;
    const uint64_t nb = 1ull << (2 * CM_WINDOW_SIZE);
    constexpr int RBITS = 12, NR = 1 << RBITS, LOWBITS = 2 * CM_WINDOW_SIZE - RBITS;      // 4096 ranges of 65536 buckets
    const int nt = std::max(1, std::min(n_threads, 64));
    const uint32_t n_slices = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)nt * 4, ((uint64_t)ref_len >> 16) + 1));
    auto slice_lo = [&](uint32_t s) { return (uint32_t)((uint64_t)ref_len * s / n_slices); };
    // k-mers whose START lies in [lo, hi): scan [lo, min(hi + k - 1, n))
    auto scan_slice = [&](uint32_t s, auto &&f) {
        const uint32_t lo = slice_lo(s), hi = slice_lo(s + 1);
        const uint32_t end = (uint32_t)std::min<uint64_t>((uint64_t)hi + (uint32_t)kmer - 1, ref_len);
        const uint64_t kmask = (1ull << (2 * kmer)) - 1, cmask = c ? ((1ull << (2 * c)) - 1) : 0;
        uint64_t v = 0;
        int run = 0;
        for (uint32_t i = lo; i < end; ++i) {
            const int b = base_code(genome[i]);
            if (b == 4) {
                run = 0;
                v = 0;
                continue;
            }
            v = ((v << 2) | (uint64_t)b) & kmask;
            if (++run >= kmer) f((uint32_t)(v >> (2 * c)), (uint32_t)(v & cmask), i + 2 - (uint32_t)kmer);
        }
    };
    // pass 1
    std::vector<uint64_t> cnt((size_t)n_slices * NR, 0);
    std::atomic<uint32_t> next{0};
    run_threads(nt, [&](int) {
        for (uint32_t s; (s = next.fetch_add(1)) < n_slices;) {
            uint64_t *cs = cnt.data() + (size_t)s * NR;
            scan_slice(s, [&](uint32_t h, uint32_t, uint32_t) { ++cs[h >> LOWBITS]; });
        }
    });
    // temp offsets in [range][slice] order
    std::vector<uint64_t> base((size_t)n_slices * NR), range_start(NR + 1);
    uint64_t total = 0;
    for (int r = 0; r < NR; ++r) {
        range_start[r] = total;
        for (uint32_t s = 0; s < n_slices; ++s) {
            base[(size_t)s * NR + r] = total;
            total += cnt[(size_t)s * NR + r];
        }
    }
    range_start[NR] = total;
    if (total > 0xffffffffull) return CM_ELIMIT;
    uint32_t *off = (uint32_t *)malloc((nb + 1) * sizeof(uint32_t));
    uint16_t *cs = (uint16_t *)malloc((total ? total : 1) * sizeof(uint16_t));
    uint32_t *ps = (uint32_t *)malloc((total ? total : 1) * sizeof(uint32_t));
    uint64_t *tmp = (uint64_t *)malloc((total ? total : 1) * sizeof(uint64_t));
    if (!off || !cs || !ps || !tmp) {
        free(off);
        free(cs);
        free(ps);
        free(tmp);
        return CM_ENOMEM;
    }
    // pass 2
    next = 0;
    run_threads(nt, [&](int) {
        std::vector<uint64_t> cur(NR);
        for (uint32_t s; (s = next.fetch_add(1)) < n_slices;) {
            for (int r = 0; r < NR; ++r) cur[r] = base[(size_t)s * NR + r];
            scan_slice(s, [&](uint32_t h, uint32_t ck, uint32_t p) {
                tmp[cur[h >> LOWBITS]++] = ((uint64_t)(h & ((1u << LOWBITS) - 1)) << 48) | ((uint64_t)ck << 32) | p;
            });
        }
    });
    // pass 3
    next = 0;
    run_threads(nt, [&](int) {
        std::vector<uint32_t> lc((1u << LOWBITS) + 1);
        std::vector<std::pair<uint16_t, uint32_t>> srt;
        for (uint32_t r; (r = next.fetch_add(1)) < (uint32_t)NR;) {
            const uint64_t a = range_start[r], b = range_start[r + 1];
            std::fill(lc.begin(), lc.end(), 0u);
            for (uint64_t i = a; i < b; ++i) ++lc[(tmp[i] >> 48) + 1];
            for (uint32_t h = 0; h < (1u << LOWBITS); ++h) lc[h + 1] += lc[h];
            uint32_t *o = off + ((uint64_t)r << LOWBITS);
            for (uint32_t h = 0; h < (1u << LOWBITS); ++h) o[h] = (uint32_t)a + lc[h];
            for (uint64_t i = a; i < b; ++i) {
                const uint64_t w = tmp[i];
                const uint32_t d = (uint32_t)a + lc[w >> 48]++;
                cs[d] = (uint16_t)(w >> 32);
                ps[d] = (uint32_t)w;
            }
            if (c > 0) {          // per-bucket order (checksum, pos): positions are ascending already, a stable sort on the checksum is enough
                for (uint32_t h = 0; h < (1u << LOWBITS); ++h) {
                    const uint32_t x = o[h], y = (h + 1 < (1u << LOWBITS)) ? o[h + 1] : (uint32_t)b;
                    if (y - x < 2) continue;
                    bool sorted = true;
                    for (uint32_t i = x + 1; i < y && sorted; ++i) sorted = cs[i - 1] <= cs[i];
                    if (sorted) continue;
                    srt.resize(y - x);
                    for (uint32_t i = x; i < y; ++i) srt[i - x] = {cs[i], ps[i]};
                    std::stable_sort(srt.begin(), srt.end(),
                                     [](const std::pair<uint16_t, uint32_t> &p, const std::pair<uint16_t, uint32_t> &q) { return p.first < q.first; });
                    for (uint32_t i = x; i < y; ++i) {
                        cs[i] = srt[i - x].first;
                        ps[i] = srt[i - x].second;
                    }
                }
            }
        }
    });
    off[nb] = (uint32_t)total;
    free(tmp);
    out->contig_num = contig_num;
    out->ref_len = ref_len;
    out->genome = genome;  // borrowed
    out->bucket_off = off;
    out->checksum = cs;
    out->pos = ps;
    out->n_entries = total;
    return CM_OK;
}

extern "C" void cm_host_free_index(cm_index_view *iv) {
    if (!iv) return;
    free((void *)iv->bucket_off);
    free((void *)iv->checksum);
    free((void *)iv->pos);
    iv->bucket_off = nullptr;
    iv->checksum = nullptr;
    iv->pos = nullptr;
    iv->n_entries = 0;
}

extern "C" int cm_host_index_stats(const cm_index_view *iv, int32_t seed_lim, int n_threads, uint64_t out[4]) {
    if (!iv || !out || !iv->bucket_off || (iv->n_entries && !iv->checksum)) return CM_EINVAL;
    const uint64_t nb = 1ull << (2 * CM_WINDOW_SIZE);
    const int nt = std::max(1, std::min(n_threads, 64));
    std::vector<uint64_t> part((size_t)nt * 4, 0);
    const uint32_t *off = iv->bucket_off;
    const uint16_t *cs = iv->checksum;
    run_threads(nt, [&](int t) {
        uint64_t *p = part.data() + (size_t)t * 4;
        const uint64_t h0 = nb * (uint64_t)t / nt, h1 = nb * (uint64_t)(t + 1) / nt;
        for (uint64_t h = h0; h < h1; ++h) {
            for (uint32_t i = off[h], e = off[h + 1]; i < e;) {           // runs of equal checksum inside a bucket = one k-mer
                uint32_t j = i + 1;
                while (j < e && cs[j] == cs[i]) ++j;
                const uint64_t m = j - i;
                p[0] += m;
                if (m > 1) p[1] += m;
                if (m > (uint64_t)std::max(seed_lim, 0)) p[2] += m;
                ++p[3];
                i = j;
            }
        }
    });
    for (int k = 0; k < 4; ++k) {
        out[k] = 0;
        for (int t = 0; t < nt; ++t) out[k] += part[(size_t)t * 4 + k];
    }
    return CM_OK;
}
