// hostemu.cpp — TEST-ONLY harness that runs the device bodies of circminer_amd/csrc/cm_core.h
// lane-by-lane on the host CPU, so the kernel logic can be stepped through and compared with
// the oracle on the build box (which has no GPU).  It mirrors the launch structure of
// cm_hot.hip (seed -> cells/scan -> chain -> pair) but is never linked into libcmhot.so, never
// shipped, and is not a fallback: the product C-ABI has no CPU path.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <vector>

#include "circminer_hot.h"
#define CM_STATS 1
// diagnostic (tests/diag/dp_dup.py): every X-drop DP request of a pair (kind 0) and every one that runs the recurrence (kind 1)
static void emu_hook_dp(int kind, const void *sp, int soff, int sstep, int smode, int n, const void *tp, int toff, int tstep, int tmode, int m);
#define CM_HOOK_DP(kind, s, n, t, m) emu_hook_dp(kind, (const void *)(s).p, (s).off, (s).step, (s).mode, n, (const void *)(t).p, (t).off, (t).step, (t).mode, m)
#include "cm_core.h"
#include "cm_aos.h"
extern "C" { unsigned long long cm_stats[16]; }

using cmc::Core;
#include <array>
#include <set>
static std::set<std::array<long long, 10>> g_dp_seen[2];
static uint32_t g_dp_n[2];
static bool g_dp_on = false;
static void emu_hook_dp(int kind, const void *sp, int soff, int sstep, int smode, int n, const void *tp, int toff, int tstep, int tmode, int m) {
    if (!g_dp_on) return;
    ++g_dp_n[kind];
    g_dp_seen[kind].insert({(long long)(size_t)sp, soff, sstep, smode, n, (long long)(size_t)tp, toff, tstep, tmode, m});
}

namespace {
struct Emu {
    Core core;
    const cm_reads *R;
    int S;
    std::vector<uint32_t> sstart, scnt, sraw;
    std::vector<cm_chain> chains;
    std::vector<int32_t> nchain, high;
    std::vector<uint8_t> pool;
    cmc::AnnotAosHost aos;
    unsigned long long cursor = 0;
    int err = 0;
};

int n_seeds_of(const cm_params *P, const cm_reads *R) {
    uint64_t mx = 0;
    for (uint64_t i = 0; i < R->n_pairs; ++i) {
        mx = std::max<uint64_t>(mx, R->off1[i + 1] - R->off1[i]);
        mx = std::max<uint64_t>(mx, R->off2[i + 1] - R->off2[i]);
    }
    return (int)(mx / (uint64_t)P->kmer);
}

void seed_all(Emu &e, const uint8_t *active) {
    const uint64_t n = e.R->n_pairs;
    const int S = e.S;
    e.sstart.assign(n * 4 * S + 1, 0);
    e.scnt.assign(n * 4 * S + 1, 0);
    e.sraw.assign(n * 4 * S + 1, 0);
    for (uint64_t q = 0; q < n * 4 * (uint64_t)S; ++q) {
        const uint32_t s = (uint32_t)(q % S);
        const uint64_t r = q / S;
        const int orient = r & 1, mate = (r >> 1) & 1;
        const uint64_t p = r >> 2;
        if (active && !active[p]) continue;
        const uint64_t o0 = mate ? e.R->off2[p] : e.R->off1[p], o1 = mate ? e.R->off2[p + 1] : e.R->off1[p + 1];
        const int len = (int)(o1 - o0), k = e.core.P.kmer;
        if ((int)(s + 1) * k > len) continue;
        const cmc::Read rd{(mate ? e.R->seq2 : e.R->seq1) + o0, len, orient};
        const cmc::Probe pr = cmc::seed_probe(e.core, rd.view(), (int)s * k);
        e.sstart[q] = pr.start;
        e.sraw[q] = pr.raw;
        e.scnt[q] = pr.raw > (uint32_t)e.core.P.seed_lim ? 0u : pr.raw;
    }
}

void chain_all(Emu &e, const uint8_t *active) {
    const uint64_t n = e.R->n_pairs;
    const int S = e.S;
    e.chains.assign(n * 4 * CM_BESTCHAINLIM + 1, cm_chain{});
    e.nchain.assign(n * 4 + 1, 0);
    e.high.assign(n * 4 + 1, 0);
    e.pool.resize(64u << 20);
    std::vector<double> dps;
    std::vector<int32_t> dpp;
    for (uint64_t r = 0; r < n * 4; ++r) {
        const uint64_t p = r >> 2;
        if (active && !active[p]) continue;
        const int mate = (r >> 1) & 1;
        const int len = (int)(mate ? e.R->off2[p + 1] - e.R->off2[p] : e.R->off1[p + 1] - e.R->off1[p]);
        uint32_t st[cmc::MAX_SEEDS], cn[cmc::MAX_SEEDS];
        size_t cells = 0;
        int hh = 0;
        for (int s = 0; s < S; ++s) {
            st[s] = e.sstart[r * S + s];
            cn[s] = e.scnt[r * S + s];
            cells += cn[s];
            if (e.sraw[r * S + s] > 0 && cn[s] == 0) ++hh;
        }
        dps.assign(cells + 1, 0);
        dpp.assign(cells + 1, 0);
        e.cursor = 0;
        cmc::ChainWork w{dps.data(), dpp.data(), e.pool.data(), (unsigned long long)e.pool.size(), &e.cursor, &e.err};
        e.nchain[r] = cmc::chain_kbest(e.core, len, S, st, cn, w, e.chains.data() + r * CM_BESTCHAINLIM);
        e.high[r] = hh;
    }
}
}  // namespace

extern "C" {

// bucket descriptors (cmc::desc_pack), built on request for ONE index view; every entry point below uses them for that view
static std::vector<uint32_t> g_desc;
static const void *g_desc_of = nullptr;
int emu_build_desc(const cm_params *P, const cm_index_view *X) {
    const uint64_t nb = (uint64_t)1 << (2 * CM_WINDOW_SIZE);
    g_desc.assign((size_t)nb * cmc::DESC_WORDS, 0u);
    for (uint64_t hv = 0; hv < nb; ++hv) {
        const uint32_t b0 = X->bucket_off[hv], n = X->bucket_off[hv + 1] - b0;
        cmc::desc_pack(b0, n, [&](uint32_t i) { return (uint32_t)X->checksum[b0 + i]; }, P->kmer, g_desc.data() + hv * cmc::DESC_WORDS);
    }
    g_desc_of = X->bucket_off;
    return 0;
}
void emu_free_desc() {
    std::vector<uint32_t>().swap(g_desc);
    g_desc_of = nullptr;
}
static const uint32_t *desc_for(const cm_index_view *X) { return (g_desc_of && g_desc_of == X->bucket_off) ? g_desc.data() : nullptr; }

int emu_seed_batch(const cm_params *P, const cm_index_view *X, const cm_reads *R, uint32_t n_slots, uint32_t *out_start, uint32_t *out_cnt,
                   uint32_t *out_raw) {
    Emu e;
    e.core.P = *P;
    e.core.X = cmc::to_dev(*X);
    e.core.desc = desc_for(X);
    memset(&e.core.A, 0, sizeof e.core.A);
    e.R = R;
    e.S = n_seeds_of(P, R);
    if ((uint32_t)e.S != n_slots) return -100;
    seed_all(e, nullptr);
    const size_t k = (size_t)R->n_pairs * 4 * e.S;
    memcpy(out_start, e.sstart.data(), k * 4);
    memcpy(out_cnt, e.scnt.data(), k * 4);
    memcpy(out_raw, e.sraw.data(), k * 4);
    return 0;
}

int emu_chain_batch(const cm_params *P, const cm_index_view *X, const cm_annot_view *A, const cm_reads *R, cm_chain *out_chains,
                    int32_t *out_nchain, int32_t *out_high) {
    Emu e;
    e.core.P = *P;
    e.core.X = cmc::to_dev(*X);
    e.core.desc = desc_for(X);
    cmc::build_annot_aos(*A, e.aos);
    e.core.A = cmc::to_dev(cmc::annot_dev_host(*A, e.aos));
    e.R = R;
    e.S = n_seeds_of(P, R);
    seed_all(e, nullptr);
    chain_all(e, nullptr);
    const size_t np = (size_t)R->n_pairs * 4;
    memcpy(out_chains, e.chains.data(), np * CM_BESTCHAINLIM * sizeof(cm_chain));
    memcpy(out_nchain, e.nchain.data(), np * 4);
    memcpy(out_high, e.high.data(), np * 4);
    return e.err;
}

// diagnostic (tests/diag/lane_cost.py): when set, emu_map_round writes the CPU time of each pair's pair stage (ns) here
static unsigned long long *g_pair_stats = nullptr;      // per pair: the 16 cm_stats counters of its pair stage
void emu_set_stats_out(unsigned long long *out) { g_pair_stats = out; }
static uint32_t *g_dp_dup = nullptr;        // per pair: requests, distinct requests, recurrences run, distinct recurrences
void emu_set_dp_dup_out(uint32_t *out) { g_dp_dup = out; g_dp_on = out != nullptr; }
static double *g_pair_ns = nullptr;
static uint32_t *g_pair_dps = nullptr;      // number of real DPs (cmc::stage calls) of each pair
void emu_set_cost_out(double *out) { g_pair_ns = out; }
void emu_set_dp_out(uint32_t *out) { g_pair_dps = out; }

// spill_cap > 0: the extension memo gets that many overflow entries behind its MEMO_N (what the device's re-run launch of
// k_pair does for a pair whose first pass flagged ERR_MEMO, cm_hot.hip RetryArgs)
int emu_map_round_spill(const cm_params *P, const cm_index_view *X, const cm_annot_view *A, const cm_reads *R, int is_last, cm_mapped_read *state,
                        uint8_t *active, int32_t *category, int spill_cap);
int emu_map_round(const cm_params *P, const cm_index_view *X, const cm_annot_view *A, const cm_reads *R, int is_last, cm_mapped_read *state,
                  uint8_t *active, int32_t *category) {
    return emu_map_round_spill(P, X, A, R, is_last, state, active, category, 0);
}
int emu_map_round_spill(const cm_params *P, const cm_index_view *X, const cm_annot_view *A, const cm_reads *R, int is_last, cm_mapped_read *state,
                        uint8_t *active, int32_t *category, int spill_cap) {
    Emu e;
    std::vector<cmc::MemoSpill> spill((size_t)(spill_cap > 0 ? spill_cap : 0));
    e.core.P = *P;
    e.core.X = cmc::to_dev(*X);
    e.core.desc = desc_for(X);
    cmc::build_annot_aos(*A, e.aos);
    e.core.A = cmc::to_dev(cmc::annot_dev_host(*A, e.aos));
    e.R = R;
    e.S = n_seeds_of(P, R);
    seed_all(e, active);
    chain_all(e, active);
    for (uint64_t p = 0; p < R->n_pairs; ++p) {
        if (!active[p]) {
            category[p] = -1;
            continue;
        }
        cmc::ChainSet sets[4];
        int hh[4];
        for (int x = 0; x < 4; ++x) {
            sets[x].ch = e.chains.data() + (p * 4 + x) * CM_BESTCHAINLIM;
            sets[x].n = e.nchain[p * 4 + x];
            hh[x] = e.high[p * 4 + x];
        }
        const int l1 = (int)(R->off1[p + 1] - R->off1[p]), l2 = (int)(R->off2[p + 1] - R->off2[p]);
        uint8_t bufa[1024], bufb[1024];
        cmc::DpMem sm{cmc::LBuf{bufa, 1024}, cmc::LBuf{bufb, 1024}, &e.err};
        if (spill_cap > 0) {
            sm.spill = spill.data();
            sm.spill_cap = spill_cap;
        }
        if (g_dp_dup) { g_dp_seen[0].clear(); g_dp_seen[1].clear(); g_dp_n[0] = g_dp_n[1] = 0; }
        unsigned long long st0[16];
        memcpy(st0, cm_stats, sizeof st0);
        const auto t0 = std::chrono::steady_clock::now();
        const unsigned long long dp0 = cm_stats[8];
        const int st = cmc::process_read(e.core, sm, R->seq1 + R->off1[p], l1, R->seq2 + R->off2[p], l2, sets, hh, state[p], &e.err);
        if (g_pair_ns) g_pair_ns[p] = std::chrono::duration<double, std::nano>(std::chrono::steady_clock::now() - t0).count();
        if (g_pair_dps) g_pair_dps[p] = (uint32_t)(cm_stats[8] - dp0);
        if (g_pair_stats)
            for (int x = 0; x < 16; ++x) g_pair_stats[p * 16 + x] = cm_stats[x] - st0[x];
        if (g_dp_dup) {
            g_dp_dup[p * 4 + 0] = g_dp_n[0]; g_dp_dup[p * 4 + 1] = (uint32_t)g_dp_seen[0].size();
            g_dp_dup[p * 4 + 2] = g_dp_n[1]; g_dp_dup[p * 4 + 3] = (uint32_t)g_dp_seen[1].size();
        }
        cmc::finish_round(e.core, st, is_last, l1, l2, state[p], active[p]);
        category[p] = st;
    }
    return e.err;
}

// stand-alone DP bodies for the property tests of A14
int emu_edit_side(const cm_params *P, const uint8_t *s, int n, const uint8_t *t, int m, int left, int *indel, int *score) {
    Core c{};
    c.P = *P;
    uint8_t bufa[2048], bufb[2048];
    int err = 0;
    const cmc::DpMem sm{cmc::LBuf{bufa, 2048}, cmc::LBuf{bufb, 2048}, &err};
    return cmc::local_alignment_side(c, sm, cmc::SV{s, 0, 1, 0}, n, cmc::SV{t, 0, 1, 0}, m, left != 0, *indel, *score);
}
int emu_drop_sc(const cm_params *P, const uint8_t *s, int n, const uint8_t *t, int m, int left, int *sclen, int *indel, int *score) {
    Core c{};
    c.P = *P;
    cmc::SV a{s, 0, 1, 0}, b{t, 0, 1, 0};
    uint8_t bufa[2048], bufb[2048];
    int err = 0;
    const cmc::DpMem sm{cmc::LBuf{bufa, 2048}, cmc::LBuf{bufb, 2048}, &err};
    if (left) return cmc::local_alignment_sc(c, sm, a.rev(n), n, b.rev(m), m, *sclen, *indel, *score);
    return cmc::local_alignment_sc(c, sm, a, n, b, m, *sclen, *indel, *score);
}
int emu_leftovers_matter(int T, int min_ret1, int can1, int min_ret2, int can2) { return cmc::leftovers_matter(T, min_ret1, can1 != 0, min_ret2, can2 != 0) ? 1 : 0; }
int emu_leftover_type(int min_ret1, int min_ret2, int g1, int g2) { return cmc::leftover_type(min_ret1, min_ret2, g1 != 0, g2 != 0); }
// one k-mer probe of a forward-strand string: occurrences, first entry and the search-touch count (test_seed_touch_count)
uint32_t emu_probe(const cm_params *P, const cm_index_view *X, const uint8_t *seq, int qpos, uint32_t *start, uint32_t *touches) {
    Core c{};
    c.P = *P;
    c.X = cmc::to_dev(*X);
    c.desc = desc_for(X);
    const cmc::Probe pr = cmc::seed_probe(c, cmc::SV{seq, 0, 1, 0}, qpos);
    *start = pr.start;
    *touches = pr.touches;
    return pr.raw;
}
int emu_one_side(const cm_params *P, const uint8_t *s, int n, const uint8_t *t, int m, int w) {
    Core c{};
    c.P = *P;
    uint8_t bufa[2048], bufb[2048];
    int err = 0;
    const cmc::DpMem sm{cmc::LBuf{bufa, 2048}, cmc::LBuf{bufb, 2048}, &err};
    return cmc::one_side_banded(c, sm, cmc::SV{s, 0, 1, 0}, n, cmc::SV{t, 0, 1, 0}, m, w);
}
}
