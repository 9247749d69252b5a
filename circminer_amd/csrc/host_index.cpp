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
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "circminer_hot.h"

namespace {

inline int base_code(uint8_t ch) {
    switch (ch) {
        case 'A': return 0;
        case 'C': return 1;
        case 'G': return 2;
        case 'T': return 3;
        default: return 4;
    }
}

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

// Both passes run on n_threads threads that each scan the whole contig and keep only the k-mers of their own
// bucket range: bucket ranges are disjoint, so the counters / cursors need no atomics and every bucket still
// receives its entries in ascending position order (pass 2), exactly like the serial scatter of the reference.
// Pass 1 splits the bucket space evenly, pass 2 by entry count (boundaries taken from the prefix sum).  A rolling
// 2-bit code costs ~1 ns per base, so T scans in parallel are cheap next to the random scatter they divide by T:
// a 1.06 Gbp contig builds in ~6 s on 16 threads instead of ~40 s on one.
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
    int nt = std::max(1, std::min(n_threads, 64));
    if (ref_len < (1u << 22)) nt = 1;        // small contigs: the scan is the whole cost
    // off[h+2] counts bucket h during pass 1; after the prefix sum off[h+1] is bucket h's
    // write cursor, and once pass 2 is done off[h] is bucket h's start.
    uint32_t *off = (uint32_t *)calloc(nb + 2, sizeof(uint32_t));
    if (!off) return CM_ENOMEM;
    std::vector<uint64_t> part(nt, 0);
    run_threads(nt, [&](int t) {
        const uint32_t lo = (uint32_t)(nb * (uint64_t)t / nt), span = (uint32_t)(nb * (uint64_t)(t + 1) / nt) - lo;
        uint64_t cnt = 0;
        for_each_kmer(genome, ref_len, kmer, c, [&](uint32_t h, uint16_t, uint32_t) {
            if (h - lo < span) {
                ++off[h + 2];
                ++cnt;
            }
        });
        part[t] = cnt;
    });
    uint64_t total = 0;
    for (int t = 0; t < nt; ++t) total += part[t];
    if (total > 0xffffffffull) {
        free(off);
        return CM_ELIMIT;
    }
    for (uint64_t h = 2; h < nb + 2; ++h) off[h] += off[h - 1];
    uint16_t *cs = (uint16_t *)malloc((total ? total : 1) * sizeof(uint16_t));
    uint32_t *ps = (uint32_t *)malloc((total ? total : 1) * sizeof(uint32_t));
    if (!cs || !ps) {
        free(off);
        free(cs);
        free(ps);
        return CM_ENOMEM;
    }
    // bucket boundaries that give every thread about the same number of entries (off[h+1] = start of bucket h now)
    std::vector<uint64_t> cut(nt + 1, 0);
    cut[nt] = nb;
    for (int t = 1; t < nt; ++t) {
        const uint32_t want = (uint32_t)(total * (uint64_t)t / nt);
        cut[t] = (uint64_t)(std::lower_bound(off + 1, off + 1 + nb, want) - (off + 1));
        if (cut[t] < cut[t - 1]) cut[t] = cut[t - 1];
    }
    run_threads(nt, [&](int t) {
        const uint32_t lo = (uint32_t)cut[t], span = (uint32_t)(cut[t + 1] - cut[t]);
        if (!span) return;
        for_each_kmer(genome, ref_len, kmer, c, [&](uint32_t h, uint16_t ck, uint32_t p) {
            if (h - lo < span) {
                const uint32_t w = off[h + 1]++;
                cs[w] = ck;
                ps[w] = p;
            }
        });
    });
    // Per-bucket order (checksum, pos).  Pass 2 wrote ascending pos, so a stable sort on the
    // checksum is enough.
    if (c > 0) {
        run_threads(nt, [&](int t) {
            std::vector<std::pair<uint16_t, uint32_t>> tmp;
            for (uint64_t h = cut[t]; h < cut[t + 1]; ++h) {
                uint32_t a = off[h], b = off[h + 1];
                if (b - a < 2) continue;
                bool sorted = true;
                for (uint32_t i = a + 1; i < b && sorted; ++i) sorted = cs[i - 1] <= cs[i];
                if (sorted) continue;
                tmp.resize(b - a);
                for (uint32_t i = a; i < b; ++i) tmp[i - a] = {cs[i], ps[i]};
                std::stable_sort(tmp.begin(), tmp.end(),
                                 [](const std::pair<uint16_t, uint32_t> &x, const std::pair<uint16_t, uint32_t> &y) {
                                     return x.first < y.first;
                                 });
                for (uint32_t i = a; i < b; ++i) {
                    cs[i] = tmp[i - a].first;
                    ps[i] = tmp[i - a].second;
                }
            }
        });
    }
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
