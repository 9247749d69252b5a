// cm_hot.hip — gfx950 kernels + the C-ABI of include/circminer_hot.h.
//
// Launch structure of one mapping round (cm_map_round) over the resident read batch, per tile of
// pairs:  k_seed (one lane per k-mer probe)  ->  k_cells + k_scan (cells per chaining problem,
// exclusive offsets)  ->  k_chain (one lane per (mate, orientation) problem, DP cells and the
// improvement log in HBM workspace)  ->  k_pair (one lane per pair: pairing, extension,
// classification, round bookkeeping).  Index, genome, annotation, reads and per-pair state stay
// in HBM between rounds; nothing is copied back until cm_reads_download.
//
// There is no CPU path in this file: without a HIP device cm_create fails with CM_ENODEV.
#include <chrono>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <unordered_map>
#include <functional>
#include <thread>
#include <vector>

#include "circminer_hot.h"
#include "cm_core.h"
#include "cm_aos.h"

using cmc::Core;
using cmc::KCore;

namespace {

constexpr int MAX_SLOTS = 16;
constexpr uint32_t TILE_PAIRS = 1u << 20;          // pairs per launch group (workspace sizing: ~46 KB of HBM per pair) ...
constexpr uint32_t TILE_PAIRS_MAX = 1u << 21;      // ... up to this for batches of more than two such groups (prepare_resident)
constexpr int BLK = 256;
constexpr int BLK_CHAIN = 64;
constexpr int BLK_PAIR = 64;
// bytes one staged string of `cap` characters takes per lane (cm_core.h LBuf: eight codes per word + one spare word)
__host__ __device__ constexpr int lbuf_bytes(int cap) { return (cap / 8 + 1) * 4; }

struct ReadsDev {
    const uint8_t *seq1, *seq2;
    const uint64_t *off1, *off2;
};

// ------------------------------------------------------------------ kernels
__global__ void __launch_bounds__(BLK) k_seed(KCore kc, ReadsDev rd, const uint8_t *active, uint64_t pair0, uint32_t n_tile, int S,
                                              uint32_t *sstart, uint32_t *scnt, uint32_t *sraw, unsigned long long *counters) {
    const Core c = cmc::to_core(kc);
    __shared__ unsigned int sh[3];
    if (threadIdx.x < 3) sh[threadIdx.x] = 0;
    __syncthreads();
    const uint64_t q = (uint64_t)blockIdx.x * BLK + threadIdx.x;
    const uint64_t total = (uint64_t)n_tile * 4u * (uint64_t)S;
    if (q < total) {
        const uint32_t s = (uint32_t)(q % (uint64_t)S);
        const uint64_t r = q / (uint64_t)S;
        const int orient = (int)(r & 1u), mate = (int)((r >> 1) & 1u);
        const uint64_t p = pair0 + (r >> 2);
        uint32_t st = 0, cn = 0, rw = 0;
        if (active[p]) {
            const uint64_t o0 = mate ? rd.off2[p] : rd.off1[p], o1 = mate ? rd.off2[p + 1] : rd.off1[p + 1];
            const int len = (int)(o1 - o0);
            const int k = c.P.kmer;
            if ((int)(s + 1) * k <= len) {
                const cmc::Read R{(cmc::g_u8)((mate ? rd.seq2 : rd.seq1) + o0), len, orient};
                const cmc::Probe pr = cmc::seed_probe(c, R.view(), (int)s * k);
                st = pr.start;
                rw = pr.raw;
                cn = (pr.raw > (uint32_t)c.P.seed_lim) ? 0u : pr.raw;
                atomicAdd(&sh[0], 1u);
                atomicAdd(&sh[1], pr.touches);
                atomicAdd(&sh[2], cn);
            }
        }
        sstart[q] = st;
        scnt[q] = cn;
        sraw[q] = rw;
    }
    __syncthreads();
    if (threadIdx.x < 3 && sh[threadIdx.x]) atomicAdd(&counters[threadIdx.x], (unsigned long long)sh[threadIdx.x]);
}

// cells per chaining problem + exclusive scan over the problems of a tile (3 phases):
// k_scan_a: per block of SCAN_ELEMS problems, cells[r] = sum of its seed counts, block-local
// exclusive offsets and the block total; k_scan_b: one workgroup scans the block totals;
// k_scan_c: adds the block base.  out[n] = grand total.
constexpr int SCAN_T = 256;
constexpr int SCAN_ELEMS = 1024;       // 4 per thread
__global__ void __launch_bounds__(SCAN_T) k_scan_a(const uint32_t *scnt, int S, uint32_t n, unsigned long long *out, unsigned long long *bsum,
                                                   unsigned int *bmax) {
    // element k * SCAN_T + t of the block belongs to thread t: consecutive lanes read consecutive problems
    // (28-byte stride, the S loads of a lane reuse its sectors) instead of four problems per lane
    __shared__ unsigned long long sh[SCAN_T];
    __shared__ unsigned int shmax;
    const uint32_t base = blockIdx.x * SCAN_ELEMS;
    unsigned long long run = 0;
    unsigned int my_max = 0;                     // the largest problem of the block (cells = retained hits): sizes k_chain_heavy's LDS
    if (threadIdx.x == 0) shmax = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ELEMS / SCAN_T; ++k) {
        const uint32_t r = base + k * SCAN_T + threadIdx.x;
        uint32_t c = 0;
        if (r < n)
            for (int s = 0; s < S; ++s) c += scnt[(uint64_t)r * S + s];
        my_max = c > my_max ? c : my_max;
        sh[threadIdx.x] = c;
        __syncthreads();
        for (int d = 1; d < SCAN_T; d <<= 1) {
            const unsigned long long x = (threadIdx.x >= (unsigned)d) ? sh[threadIdx.x - d] : 0ull;
            __syncthreads();
            sh[threadIdx.x] += x;
            __syncthreads();
        }
        if (r < n) out[r] = run + sh[threadIdx.x] - c;
        run += sh[SCAN_T - 1];
        __syncthreads();
    }
    atomicMax(&shmax, my_max);
    __syncthreads();
    if (threadIdx.x == 0) {
        bsum[blockIdx.x] = run;
        bmax[blockIdx.x] = shmax;
    }
}
__global__ void __launch_bounds__(1024) k_scan_b(unsigned long long *bsum, uint32_t nb, unsigned long long *total, const unsigned int *bmax) {
    __shared__ unsigned long long part[1024];
    __shared__ unsigned int gmax;
    const uint32_t t = threadIdx.x;
    const uint32_t chunk = (nb + 1023u) / 1024u;
    const uint32_t a = t * chunk, b = (a + chunk < nb) ? a + chunk : nb;
    unsigned long long s = 0;
    unsigned int m = 0;
    if (t == 0) gmax = 0;
    __syncthreads();
    for (uint32_t i = a; i < b; ++i) {
        s += bsum[i];
        m = bmax[i] > m ? bmax[i] : m;
    }
    atomicMax(&gmax, m);
    part[t] = s;
    __syncthreads();
    for (uint32_t d = 1; d < 1024; d <<= 1) {
        const unsigned long long v = (t >= d) ? part[t - d] : 0ull;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    unsigned long long run = t ? part[t - 1] : 0ull;
    for (uint32_t i = a; i < b; ++i) {
        const unsigned long long x = bsum[i];
        bsum[i] = run;
        run += x;
    }
    if (t == 1023) {
        total[0] = part[1023];
        total[1] = gmax;                          // (every atomicMax above is followed by a barrier of the scan)
    }
}
__global__ void __launch_bounds__(SCAN_T) k_scan_c(unsigned long long *out, const unsigned long long *bsum, uint32_t n) {
    const uint32_t i = blockIdx.x * SCAN_T + threadIdx.x;
    if (i < n) out[i] += bsum[i / SCAN_ELEMS];
}

// bases of a problem's best chain left to extend on the left / right (work proxies of the pair-stage ordering):
// left in the low byte, right in the high byte, each capped at 255
__device__ inline int pack_resid(int left, int right) {
    left = left < 0 ? 0 : (left > 255 ? 255 : left);
    right = right < 0 ? 0 : (right > 255 ? 255 : right);
    return left | (right << 8);
}
#ifndef CM_CHAIN_WAVES
#define CM_CHAIN_WAVES 5
#endif
__global__ void __launch_bounds__(BLK_CHAIN, CM_CHAIN_WAVES) k_chain(KCore kc, ReadsDev rd, const uint8_t *active, uint64_t pair0, uint32_t r0, uint32_t r1, int S,
                                                     const uint32_t *sstart, const uint32_t *scnt, const uint32_t *sraw,
                                                     const unsigned long long *celloff, unsigned long long cellbase, double *dp_score,
                                                     int32_t *dp_prev, uint8_t *pool, unsigned long long pool_bytes,
                                                     unsigned long long *pool_cursor, cm_chain *chains, int32_t *nchain, int32_t *high, int *err,
                                                     uint16_t *resid, const uint32_t *perm, const unsigned int *perm_lo, const unsigned int *perm_hi) {
    // perm == null: problems r0..r1 in index order;
    // perm != null: the light problems perm[*perm_lo .. *perm_hi), grouped by work class so that the lanes of a wave carry similar DPs
    uint32_t r = r0 + blockIdx.x * BLK_CHAIN + threadIdx.x;
    if (perm) {
        r += *perm_lo;
        if (r >= *perm_hi) return;
        r = perm[r];
    } else if (r >= r1) return;
    const Core c = cmc::to_core(kc);
    const uint64_t p = pair0 + (r >> 2);
    int n = 0, hh = 0, rs = 0;
    if (active[p]) {
        const int mate = (int)((r >> 1) & 1u);
        const uint64_t o0 = mate ? rd.off2[p] : rd.off1[p], o1 = mate ? rd.off2[p + 1] : rd.off1[p + 1];
        const int len = (int)(o1 - o0);
        uint32_t st[cmc::MAX_SEEDS], cn[cmc::MAX_SEEDS];
        for (int s = 0; s < S; ++s) {
            st[s] = sstart[(uint64_t)r * S + s];
            cn[s] = scnt[(uint64_t)r * S + s];
            if (sraw[(uint64_t)r * S + s] > 0 && cn[s] == 0) ++hh;       // get_best_chains high_hits
        }
        cmc::ChainWork w;
        w.dp_score = (CM_G double *)(dp_score + (celloff[r] - cellbase));
        w.dp_prev = (CM_G int32_t *)(dp_prev + (celloff[r] - cellbase));
        w.pool = (CM_G uint8_t *)pool;
        w.pool_bytes = pool_bytes;
        w.pool_cursor = (CM_G unsigned long long *)pool_cursor;
        w.err = (cmc::g_err)err;
        n = cmc::chain_kbest(c, len, S, st, cn, w, (CM_G cm_chain *)(chains + (uint64_t)r * CM_BESTCHAINLIM));
        if (n > 0) {          // bases of the best chain left to extend (work proxy used by the pair-stage ordering)
            const cm_chain &b = chains[(uint64_t)r * CM_BESTCHAINLIM];
            rs = pack_resid(b.qpos[0], len - (b.qpos[b.chain_len - 1] + c.P.kmer));
        }
    }
    nchain[r] = n;
    high[r] = hh;
    resid[r] = (uint16_t)rs;
}

// The reference has no per-pair capacity limits inside maxReadLength (its extension memo is an unbounded std::map,
// src/extend.cpp:299,375); the device keeps 8 memo entries per extend call in registers / scratch.  A pair that would need more
// (and could observe the difference, cm_core.h memo_put) is not allowed to fail the batch: both pair kernels leave it untouched,
// append it to `list`, and a second launch of k_pair over that list -- queued behind them unconditionally, it reads `count` on
// the device -- maps it with a spill area of `spill_cap` more entries per lane in global memory and longer DP staging buffers.
struct RetryArgs {
    uint32_t *pair_err;          // one word per pair of the tile, all zero between launches
    uint32_t *list;              // pairs (tile-relative) to re-run
    unsigned int *count;
    cmc::MemoSpill *spill;       // re-run only
    int spill_cap;
    int first;                   // 1: first pass (record, skip, queue)   0: the re-run
};

#ifndef CM_PAIR_WAVES
#define CM_PAIR_WAVES 4       // waves per SIMD the pair kernels are compiled for (128 VGPRs; LDS: 2 x lbuf_bytes x 64 per wave)
#endif
// The pair stage's light kernel and its re-run (RetryArgs) are the same code: FIRST = the first pass over the tile (a pair that
// hits a device capacity is recorded, skipped and queued), !FIRST = the re-run of the queued pairs (k_pair_rerun: spill memo,
// longer staging buffers, limits fail the call).  Two kernel symbols, so that profiles tell the two launches apart.
template <bool FIRST>
__device__ __forceinline__ void pair_kernel(const KCore &kc, const ReadsDev &rd, uint64_t pair0, uint32_t n_tile, const cm_chain *chains, const int32_t *nchain,
                                            const int32_t *high, cm_mapped_read *state, uint8_t *active, int32_t *cat, int is_last,
                                            int *err, unsigned long long *counters, int str_cap, unsigned long long *lane_clk,
                                            const uint32_t *perm, const unsigned int *n_light, unsigned int *next_chunk,
                                            const RetryArgs &ra) {
    const unsigned long long clk0 = lane_clk ? wall_clock64() : 0ull;
    // per-lane staging buffers for the two DP strings, word-interleaved across the wave (cm_core.h LBuf)
    extern __shared__ uint32_t lds_words[];
    CM_S uint8_t *lane_base = (CM_S uint8_t *)lds_words + 4 * threadIdx.x;
    const int str_stride = lbuf_bytes(str_cap) * BLK_PAIR;
#if defined(CM_DIAG)
    __shared__ unsigned long long tick_w[65];
    cmc::Tick tick;
    for (int i = 0; i < 32; ++i) tick.acc[i] = 0;
    tick.last = wall_clock64();
    tick.w = (CM_L unsigned long long *)tick_w;
    tick.wave_on = 1;
    for (int i = threadIdx.x; i < 65; i += BLK_PAIR) tick_w[i] = i == 0 ? tick.last : 0ull;
    __syncthreads();
    cmc::DpMem sm{cmc::LBuf{lane_base, str_cap}, cmc::LBuf{lane_base + str_stride, str_cap}, (cmc::g_err)err, &tick};
#else
    cmc::DpMem sm{cmc::LBuf{lane_base, str_cap}, cmc::LBuf{lane_base + str_stride, str_cap}, (cmc::g_err)err};
#endif
    if (ra.spill) {          // the exact re-run of the pairs the first pass gave up on: a memo that holds every exon piece
        sm.spill = (cmc::g_spill)(ra.spill + ((size_t)blockIdx.x * BLK_PAIR + threadIdx.x) * (size_t)ra.spill_cap);
        sm.spill_cap = ra.spill_cap;
    }
    const Core c = cmc::to_core(kc);
    // persistent grid: the launch places every workgroup at once (gridDim <= resident capacity), so the dispatcher is free
    // for the kernels of the next round that other streams run at the same time.  A wave takes the next 64 pairs of the list
    // (light pairs only, in bucket order: most expensive classes first) from a shared cursor whenever it is done with its
    // last: waves that drew cheap pairs take more of them, and all of them finish within one chunk of each other.
    const uint32_t n_l = *n_light;
    auto take = [&]() { return (uint32_t)__shfl((int)(threadIdx.x == 0 ? atomicAdd(next_chunk, 1u) : 0u), 0); };
    for (uint32_t chunk = take(); (uint64_t)chunk * BLK_PAIR < n_l; chunk = take()) {
    const uint32_t slot = chunk * BLK_PAIR + threadIdx.x;
    if (slot >= n_l) continue;
    const unsigned long long it0 = lane_clk ? wall_clock64() : 0ull;
    const uint32_t t = perm[slot];
    const uint64_t p = pair0 + t;
    const uint64_t a0 = rd.off1[p], a1 = rd.off1[p + 1], b0 = rd.off2[p], b1 = rd.off2[p + 1];
    cmc::ChainSet sets[4];
    int hh[4];
    for (int x = 0; x < 4; ++x) {
        const uint64_t r = (uint64_t)t * 4 + x;
        sets[x].ch = (cmc::g_chain)(chains + r * CM_BESTCHAINLIM);
        sets[x].n = nchain[r];
        hh[x] = high[r];
    }
    cm_mapped_read mr = state[p];
    // First pass: a device-capacity limit hit by this pair (cmc::ERR_MEMO / ERR_BAND) is recorded in the pair's own word; such
    // a pair keeps its inputs (nothing is written) and is queued for the re-run.  Re-run (ra.first == 0): limits go to the
    // launch-wide word and fail the call -- with the spill memo and the longer staging buffers none is known to be reachable.
#if defined(CM_AB_GLOBAL_ERR)          // A/B experiment only: what the per-pair error word costs (results wrong for pairs over a capacity)
    int *perr = err;
#else
    int *perr = FIRST ? (int *)(ra.pair_err + t) : err;
    sm.err = (cmc::g_err)perr;
#endif
    const int st = cmc::process_read(c, sm, (cmc::g_u8)(rd.seq1 + a0), (int)(a1 - a0), (cmc::g_u8)(rd.seq2 + b0), (int)(b1 - b0), sets, hh, mr, (cmc::g_err)perr);
    bool keep = true;
    if (FIRST) {
        if (__hip_atomic_load(ra.pair_err + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
            ra.list[atomicAdd(ra.count, 1u)] = t;
            active[pair0 + t] = 1;          // until the re-run writes its flag: counted as active (the next item's seeds may be computed before that)
            keep = false;
        }
    } else ra.pair_err[t] = 0u;
    if (keep) {
        uint8_t act = 1;
        cmc::finish_round(c, st, is_last, (int)(a1 - a0), (int)(b1 - b0), mr, act);
        state[p] = mr;
        active[p] = act;
        cat[p] = st;
    }
    {   // pair-rounds counter: one atomic per wave (a pair left to the re-run is counted there)
        const unsigned long long m = __ballot(1), mk = __ballot(keep);
        if ((threadIdx.x & 63) == (unsigned)__ffsll((long long)m) - 1 && mk) {
            atomicAdd(&counters[3], (unsigned long long)__popcll(mk));
            if (!FIRST) atomicAdd(&counters[4], (unsigned long long)__popcll(mk));       // pairs mapped by the re-run launch
        }
    }
#if defined(CM_DIAG)
    CM_TICK(sm, 13);
    if (lane_clk) {                                           // section timing study: 16 words per pair
        for (int i = 0; i < 15; ++i) lane_clk[p * 16 + i] = tick.acc[i];
        lane_clk[p * 16 + 15] = wall_clock64() - clk0;
        // wave-level rows behind the per-pair rows (second half of the buffer): [wave][0] = wave time, [wave][1] = lane-weighted
        const unsigned long long m = __ballot(1);
        if ((threadIdx.x & 63) == (unsigned)__ffsll((long long)m) - 1) {
            unsigned long long *wr = lane_clk + (unsigned long long)n_tile * 16 + (unsigned long long)blockIdx.x * 64;
            for (int i = 0; i < 64; ++i) wr[i] = tick_w[1 + i];
        }
    }
#else
    // diagnostic only (CM_LANE_CLK=1), in processing order (64 consecutive slots = one wave iteration): low word = the
    // wave's time for this iteration (100 MHz ticks; the clock is read after the lanes have reconverged), high word = the pair
    if (lane_clk) lane_clk[slot] = ((wall_clock64() - it0) & 0xFFFFFFFFull) | ((unsigned long long)t << 32);
#endif
    }
}


__global__ void __launch_bounds__(BLK_PAIR, CM_PAIR_WAVES) k_pair(KCore kc, ReadsDev rd, uint64_t pair0, uint32_t n_tile, const cm_chain *chains, const int32_t *nchain,
                                                   const int32_t *high, cm_mapped_read *state, uint8_t *active, int32_t *cat, int is_last,
                                                   int *err, unsigned long long *counters, int str_cap, unsigned long long *lane_clk,
                                                   const uint32_t *perm, const unsigned int *n_light, unsigned int *next_chunk,
                                                   RetryArgs ra) {
    pair_kernel<true>(kc, rd, pair0, n_tile, chains, nchain, high, state, active, cat, is_last, err, counters, str_cap, lane_clk, perm, n_light, next_chunk, ra);
}
__global__ void __launch_bounds__(BLK_PAIR, CM_PAIR_WAVES) k_pair_rerun(KCore kc, ReadsDev rd, uint64_t pair0, uint32_t n_tile, const cm_chain *chains,
                                                         const int32_t *nchain, const int32_t *high, cm_mapped_read *state, uint8_t *active, int32_t *cat,
                                                         int is_last, int *err, unsigned long long *counters, int str_cap,
                                                         unsigned long long *lane_clk, const uint32_t *perm, const unsigned int *n_light,
                                                         unsigned int *next_chunk, RetryArgs ra) {
    pair_kernel<false>(kc, rd, pair0, n_tile, chains, nchain, high, state, active, cat, is_last, err, counters, str_cap, lane_clk, perm, n_light, next_chunk, ra);
}

// ---- wave-cooperative chaining of heavy problems ----------------------------------------------
// chain_seeds_sorted_kbest (src/chain.cpp:73-301) for ONE problem per wave.  For slot ii the cells i are
// independent given the slots > ii, except for the reference's shared cursor lb_ind[jj]
// (src/chain.cpp:133-160).  That cursor only differs from upper_bound(hits of jj, hit i) when an earlier
// cell skipped the list with the maxIntronLen test, and the difference is observable only if a later
// cell's window (max_lpos_lim) reaches a hit more than maxIntronLen away, i.e. only if the annotation
// has an exon or an exon->next-exon hop longer than maxIntronLen.  cm_load_annotation checks that
// (Slot::chain_parallel_ok); if it ever fails, heavy problems stay on the sequential kernel.
// Order of the improvement log = reference insertion order (ii desc, i asc, then (jj, j) asc): each batch
// of 64 cells is evaluated twice, first to count the improvements per cell, then (after a wave scan) to
// store them at their final offsets.
template <class T> __device__ inline T wave_excl_scan(T v, int lane, T &total) {
    T x = v;
    for (int o = 1; o < 64; o <<= 1) {
        const T y = __shfl_up(x, o);
        if (lane >= o) x += y;
    }
    total = __shfl(x, 63);
    return x - v;
}
struct HeavyChainCtx {
#if defined(CM_CHAIN_DIAG)
    unsigned long long *tk;      // per-lane ticks: [0] binary searches, [1] upper_bound, [2] window loops, [3] cells, [4] pair evaluations
#endif
    const Core *c;
    CM_L const uint32_t *LP;     // hit positions of every slot, concatenated (LDS)
    CM_L const uint32_t *NB;     // near-border bit of every cell of the slots that are evaluated (LDS bitmask, see k_chain_heavy)
    const uint32_t *base, *cnt;  // per slot (uniform)
    int kc, seq_len;
    CM_G double *dps;
    CM_G int32_t *dpp;
};
// evaluates cell (ii, i); returns the number of strict improvements; stores them to ev[] when ev != null
__device__ inline uint32_t heavy_cell(const HeavyChainCtx &h, int ii, uint32_t i, CM_G cmc::Event *ev, double &out_score, int32_t &out_prev,
                                     double &e0, double &e1) {
    const Core &c = *h.c;
    const int kmer = c.P.kmer;
    const uint32_t read_remain = (uint32_t)(h.seq_len - ii * kmer - kmer);
    const int32_t cur_info = (int32_t)h.LP[h.base[ii] + i];
    const uint32_t seg_start = (uint32_t)cur_info, seg_end = (uint32_t)cur_info + kmer - 1;
    uint32_t max_lpos_lim = cmc::MAXUB, max_exon_end = 0;
    int ol = -1;
    double my_score = (double)kmer;
    int32_t my_prev = -1;
    uint32_t n = 0;
    // cmc::upper_bound of a hit near an exon border was resolved before the DP (k_chain_heavy) and parked in the cell's own score /
    // back-pointer slots, which nothing reads before this cell has been evaluated; asked for now, needed after the first binary search
    const uint32_t cell = h.base[ii] + i;
    const bool near = (h.NB[cell >> 5] >> (cell & 31)) & 1u;
    uint2 parked = make_uint2(0u, 0u);
    int parked_ol = -1;
    if (near) {
        parked = *(CM_G const uint2 *)(h.dps + cell);
        parked_ol = h.dpp[cell];
    }
#if defined(CM_CHAIN_DIAG)
    if (!ev) h.tk[3] += 1;
#define CD_T0 const unsigned long long cd_t = wall_clock64()
#define CD_ADD(k) h.tk[k] += wall_clock64() - cd_t
#else
#define CD_T0
#define CD_ADD(k)
#endif
    for (int jj = ii + 1; jj < h.kc; ++jj) {
        const uint32_t pcn = h.cnt[jj];
        if (pcn == 0) continue;
        CM_L const uint32_t *pp = h.LP + h.base[jj];
        uint32_t lo = 0, hi = pcn;                   // first hit of jj strictly right of this hit
        {
            CD_T0;
            while (lo < hi) {
                const uint32_t mid = (lo + hi) >> 1;
                if ((int32_t)pp[mid] <= cur_info) lo = mid + 1;
                else hi = mid;
            }
            CD_ADD(0);
        }
        if (lo >= pcn) continue;
        if (cur_info + c.P.max_intron < (int32_t)pp[lo]) continue;      // nothing within maxIntronLen
        if (max_lpos_lim == cmc::MAXUB) {           // cmc::upper_bound with its bit test hoisted (see `near`)
            CD_T0;
            if (near) {
                max_lpos_lim = parked.x;
                max_exon_end = parked.y;
                ol = parked_ol;
            } else {
                max_exon_end = 0;
                ol = -1;
                max_lpos_lim = seg_start + read_remain + (uint32_t)c.P.max_ed;
            }
            CD_ADD(1);
        }
        const int distr = (jj - ii) * kmer - kmer;
        CD_T0;
        for (uint32_t j = lo; j < pcn && pp[j] <= max_lpos_lim; ++j) {
#if defined(CM_CHAIN_DIAG)
            if (!ev) h.tk[4] += 1;
#endif
            const uint32_t pinfo = pp[j];
            int genome_dist, distt, trans_dist;
            if (max_exon_end == 0 || (pinfo + kmer - 1) <= max_exon_end) genome_dist = (int)(pinfo - seg_end - 1);
            else genome_dist = cmc::INF_I;
            if (cmc::cabs(genome_dist - distr) <= c.P.max_ed) distt = genome_dist;
            else if (cmc::check_junction(c, seg_start, pinfo, ol, kmer, distr, trans_dist)) distt = trans_dist;
            else continue;
            const int maxd = distr < distt ? distt : distr, mind = distr < distt ? distr : distt;
            const double beta = 0.1 * (double)(maxd - mind);
            const double alpha = 2e4 * (double)kmer;
            const double t1 = h.dps[h.base[jj] + j] + alpha;
            const double temp_score = t1 - beta;
            if (temp_score > my_score) {
                my_score = temp_score;
                my_prev = (int32_t)(((uint32_t)jj << 16) | j);
                if (ev) {
                    ev[n].score = temp_score;
                    ev[n].cell = ((uint32_t)ii << 16) | i;
                }
                if (n == 0) e0 = temp_score;           // the first two improvements are handed back: most cells have no more,
                else if (n == 1) e1 = temp_score;      // and the caller then skips the second (storing) evaluation
                ++n;
            }
        }
        CD_ADD(2);
    }
    out_score = my_score;
    out_prev = my_prev;
    return n;
}

// back-tracking state of one heavy problem (LDS): the candidates of the current score, their chains as (cell, slot) lists, the
// non-first fragments of every chain emitted so far (the reference's `repeats` set, src/chain.cpp:248,274-276)
struct HeavyBackTrack {
    uint32_t base[cmc::MAX_SEEDS + 1];
    uint32_t cand[CM_BESTCHAINLIM + 2];
    uint16_t cidx[CM_BESTCHAINLIM][CM_MAX_CHAIN_FRAGS];
    uint8_t cbl[CM_BESTCHAINLIM][CM_MAX_CHAIN_FRAGS];
    uint8_t clen[CM_BESTCHAINLIM + 2];
    uint8_t emit[CM_BESTCHAINLIM + 2];
    uint32_t rep[CM_BESTCHAINLIM * (CM_MAX_CHAIN_FRAGS - 1)];
};

#ifndef CM_CHEAVY_WAVES
#define CM_CHEAVY_WAVES 3      // waves per SIMD k_chain_heavy is compiled for (158 VGPRs as it stands)
#endif
__global__ void __launch_bounds__(64, CM_CHEAVY_WAVES) k_chain_heavy(KCore kc_, ReadsDev rd, uint64_t pair0, int S, const uint32_t *sstart, const uint32_t *scnt,
                                                    const unsigned long long *celloff, double *dp_score, int32_t *dp_prev, uint8_t *pool,
                                                    unsigned long long pool_bytes, unsigned long long *pool_cursor, cm_chain *chains, int32_t *nchain,
                                                    int *err, uint16_t *resid, const uint32_t *perm, const unsigned int *n_perm, unsigned int *next_problem,
                                                    unsigned long long *counters) {
    extern __shared__ uint32_t lds_words[];
    CM_L uint32_t *LP = (CM_L uint32_t *)lds_words;
    __shared__ HeavyBackTrack BT;
    const int lane = threadIdx.x;
    const Core c = cmc::to_core(kc_);
    const int kmer = c.P.kmer;
    const uint32_t max_best = (uint32_t)c.P.max_chain_len;
    const unsigned int n_heavy = *n_perm;
    // the list starts with the heaviest class; a wave takes the next problem from a shared cursor when it is done with its last
    auto take = [&]() { return (unsigned int)__shfl((int)(lane == 0 ? atomicAdd(next_problem, 1u) : 0u), 0); };
    for (unsigned int hidx = take(); hidx < n_heavy; hidx = take()) {
        const uint32_t r = perm[hidx];
        const uint64_t p = pair0 + (r >> 2);
        const int mate = (int)((r >> 1) & 1u);
        const int len = (int)(mate ? rd.off2[p + 1] - rd.off2[p] : rd.off1[p + 1] - rd.off1[p]);
#if defined(CM_CHAIN_DIAG)      // wave time per phase (100 MHz ticks) into counters[5..7]: load + init, DP, back-tracking
        const unsigned long long dg0 = wall_clock64();
#endif
        uint32_t st[cmc::MAX_SEEDS], cn[cmc::MAX_SEEDS], base[cmc::MAX_SEEDS + 1];
        int kc = S;
        for (int s = 0; s < S; ++s) {
            st[s] = sstart[(uint64_t)r * S + s];
            cn[s] = scnt[(uint64_t)r * S + s];
        }
        while (kc >= 1 && cn[kc - 1] == 0) --kc;
        base[0] = 0;
        for (int s = 0; s < kc; ++s) base[s + 1] = base[s] + cn[s];
        const uint32_t ncell = base[kc];
        CM_G double *dps = (CM_G double *)(dp_score + celloff[r]);
        CM_G int32_t *dpp = (CM_G int32_t *)(dp_prev + celloff[r]);
        // hit positions -> LDS, cells initialised
        for (int s = 0; s < kc; ++s)
            for (uint32_t i = lane; i < cn[s]; i += 64) LP[base[s] + i] = c.X.pos[st[s] + i];
        for (uint32_t x = lane; x < ncell; x += 64) {
            dps[x] = (double)kmer;
            dpp[x] = -1;
        }
        __threadfence_block();
        __syncthreads();
        // cmc::upper_bound of every hit the DP will evaluate (slots 0 .. kc - 2), ahead of the DP.  One hit in seven lies near an exon
        // border (60 k genes) and needs the annotation look-up, four or five dependent loads; inside the DP every batch of 64 cells
        // waited for its few such lanes (51 % of the DP's lane time, DESIGN note 30).  Here the near-border bits of all hits are read
        // first (one load each, every lane busy; they stay in LDS as a bitmask), the near hits are queued, and the look-ups run 64 to
        // a batch; the results are parked in the cells' own score / back-pointer slots (see heavy_cell).
        CM_L uint32_t *NB = LP + ncell;
        CM_L uint16_t *Q = (CM_L uint16_t *)(NB + 2 * ((ncell + 63) >> 6));
        {
            const uint32_t n_pre = kc >= 1 ? base[kc - 1] : 0u;
            const unsigned long long lt = (1ull << lane) - 1ull;
            uint32_t qn = 0;
            auto look_up = [&](uint32_t cnt) {
                if ((uint32_t)lane < cnt) {
                    const uint32_t x = Q[lane];
                    int ii = 0;
                    for (int s = 1; s < kc; ++s) ii += x >= base[s] ? 1 : 0;
                    const uint32_t read_remain = (uint32_t)(len - ii * kmer - kmer);
                    uint32_t mx = 0;
                    int ol = -1;
                    const uint32_t lim = cmc::upper_bound_lookup(c, LP[x], (uint32_t)kmer, read_remain, mx, ol);
                    *(CM_G uint2 *)(dps + x) = make_uint2(lim, mx);
                    dpp[x] = ol;
                }
            };
            for (uint32_t x0 = 0; x0 < n_pre; x0 += 256) {               // four bit reads in flight per lane
                bool nr[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const uint32_t x = x0 + 64 * u + lane;
                    nr[u] = x < n_pre && cmc::bit_at(c.A.near_border_bits, c.A.n_bits, LP[x]);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const uint32_t xb = x0 + 64 * u;
                    if (xb >= n_pre) break;
                    const unsigned long long m = __ballot(nr[u]);
                    if (lane == 0) NB[xb >> 5] = (uint32_t)m;
                    if (lane == 32) NB[(xb >> 5) + 1] = (uint32_t)(m >> 32);
                    if (nr[u]) Q[qn + __popcll(m & lt)] = (uint16_t)(xb + lane);
                    qn += (uint32_t)__popcll(m);
                }
                __syncthreads();
                while (qn >= 64) {                                       // (at most 63 + 256 entries queued)
                    look_up(64);
                    uint16_t keep[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) keep[u] = 64u * u + (uint32_t)lane < qn - 64 ? Q[64 + 64 * u + lane] : (uint16_t)0;
                    __syncthreads();
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if (64u * u + (uint32_t)lane < qn - 64) Q[64 * u + lane] = keep[u];
                    qn -= 64;
                    __syncthreads();
                }
            }
            if (qn) look_up(qn);
        }
        __threadfence_block();
        __syncthreads();
#if defined(CM_CHAIN_DIAG)
        const unsigned long long dg1 = wall_clock64();
#endif
#if defined(CM_CHAIN_DIAG)
        unsigned long long tk[5] = {0, 0, 0, 0, 0};
        unsigned long long wv[4] = {0, 0, 0, 0};       // wave time: first evaluation, scan + log growth, store / second evaluation, barrier
        HeavyChainCtx H{tk, &c, LP, NB, base, cn, kc, len, dps, dpp};
#else
        HeavyChainCtx H{&c, LP, NB, base, cn, kc, len, dps, dpp};
#endif
        CM_G cmc::Event *ev = nullptr;
        uint32_t n_ev = 0, cap_ev = 0;
        bool lost = false;
        for (int ii = kc - 2; ii >= 0; --ii) {
            for (uint32_t i0 = 0; i0 < cn[ii]; i0 += 64) {
                const uint32_t i = i0 + lane;
                const bool on = i < cn[ii];
                double sc = 0, e0 = 0, e1 = 0;
                int32_t pv = -1;
#if defined(CM_CHAIN_DIAG)
                const unsigned long long w0 = wall_clock64();
#endif
                const uint32_t mine = on ? heavy_cell(H, ii, i, (CM_G cmc::Event *)nullptr, sc, pv, e0, e1) : 0u;
#if defined(CM_CHAIN_DIAG)
                const unsigned long long w1 = wall_clock64();
#endif
                uint32_t total;
                const uint32_t off = wave_excl_scan(mine, lane, total);
                if (n_ev + total > cap_ev && !lost) {                 // grow the log (uniform decision)
                    uint32_t ncap = cap_ev ? cap_ev : 256u;
                    while (ncap < n_ev + total) ncap *= 4u;
                    const unsigned long long bytes = (unsigned long long)ncap * sizeof(cmc::Event);
                    unsigned long long o = 0;
                    if (lane == 0) o = atomicAdd(pool_cursor, bytes);
                    o = ((unsigned long long)__shfl((unsigned int)(o >> 32), 0) << 32) | (unsigned long long)__shfl((unsigned int)o, 0);
                    if (o + bytes > pool_bytes) {
                        lost = true;
                        if (lane == 0) atomicOr(err, cmc::ERR_POOL);
                    } else {
                        CM_G cmc::Event *ne = (CM_G cmc::Event *)((CM_G uint8_t *)pool + o);
                        for (uint32_t q = lane; q < n_ev; q += 64) ne[q] = ev[q];
                        ev = ne;
                        cap_ev = ncap;
                    }
                }
                if (on) {
                    if (mine && !lost) {
                        CM_G cmc::Event *dst = ev + n_ev + off;
                        if (mine <= 2) {
                            dst[0].score = e0;
                            dst[0].cell = ((uint32_t)ii << 16) | i;
                            if (mine == 2) {
                                dst[1].score = e1;
                                dst[1].cell = ((uint32_t)ii << 16) | i;
                            }
                        } else heavy_cell(H, ii, i, dst, sc, pv, e0, e1);
                    }
                    dps[base[ii] + i] = sc;
                    dpp[base[ii] + i] = pv;
                }
                if (!lost) n_ev += total;
#if defined(CM_CHAIN_DIAG)
                const unsigned long long w3 = wall_clock64();
                wv[0] += w1 - w0;
                wv[2] += w3 - w1;
#endif
            }
#if defined(CM_CHAIN_DIAG)
            const unsigned long long w4 = wall_clock64();
#endif
            __threadfence_block();
            __syncthreads();
#if defined(CM_CHAIN_DIAG)
            wv[3] += wall_clock64() - w4;
#endif
        }
#if defined(CM_CHAIN_DIAG)
        for (int k = 0; k < 5; ++k) atomicAdd(&counters[8 + k], tk[k]);          // lane sums
        if (lane == 0)
            for (int k = 0; k < 4; ++k) atomicAdd(&counters[16 + k], wv[k]);      // wave times
#endif
        // ---- back-tracking (src/chain.cpp:242-298), wave-parallel.  The reference walks the scores downwards; per score it takes
        // the first <= maxChainLen logged cells in insertion order, skips one whose start is a non-first fragment of a chain
        // already emitted (only below the best score), and emits the others until maxChainLen chains are out.  Here, per score:
        // (1) the candidates are picked out of the log with ballots, (2) every candidate's chain is walked by its own lane
        // (dependent loads of the back pointers: 30 walks overlap instead of queueing on lane 0), (3) the skip rule runs over
        // the candidates in order against the emitted fragments kept in LDS, (4) the emitted chains are written by all lanes.
        CM_G cm_chain *out = (CM_G cm_chain *)(chains + (uint64_t)r * CM_BESTCHAINLIM);
        uint32_t best_count = 0;
        int first_q0 = 0, first_qlast = 0;            // of chain 0 (for the residual key below)
        if (lane <= kc) BT.base[lane] = base[lane];   // base[] by a lane-varying slot: from LDS
        __threadfence_block();
        __syncthreads();
#if defined(CM_CHAIN_DIAG)
        const unsigned long long dg2 = wall_clock64();
#endif
        if (n_ev > 0) {
            double best_score = -1.0;
            for (uint32_t q = lane; q < n_ev; q += 64) best_score = ev[q].score > best_score ? ev[q].score : best_score;
            for (int o = 32; o >= 1; o >>= 1) {
                const double u = __shfl_xor(best_score, o);
                best_score = u > best_score ? u : best_score;
            }
            double cur = best_score;
            bool have = true;
            uint32_t n_rep = 0;
#if defined(CM_CHAIN_DIAG)
            unsigned long long dg_levels = 0;
#endif
            while (have && best_count < max_best) {
#if defined(CM_CHAIN_DIAG)
                ++dg_levels;
#endif
                // (1) + next lower score, one pass over the log
                uint32_t n_c = 0;
                double nxt = -1.0;
                bool hv = false;
                for (uint32_t q0 = 0; q0 < n_ev; q0 += 64) {
                    const uint32_t q = q0 + lane;
                    const double v = q < n_ev ? ev[q].score : 0.0;
                    const bool hit = q < n_ev && v == cur;
                    if (q < n_ev && v < cur && (!hv || v > nxt)) {
                        nxt = v;
                        hv = true;
                    }
                    const unsigned long long m = __ballot(hit);
                    if (m && n_c < max_best) {
                        const uint32_t rank = n_c + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
                        if (hit && rank < max_best) BT.cand[rank] = ev[q].cell;
                        n_c += (uint32_t)__popcll(m);
                        if (n_c > max_best) n_c = max_best;
                    }
                }
                for (int o = 32; o >= 1; o >>= 1) {
                    const double u = __shfl_xor(nxt, o);
                    const int uh = __shfl_xor((int)hv, o);
                    if (uh && (!hv || u > nxt)) {
                        nxt = u;
                        hv = true;
                    }
                }
                __syncthreads();
                // (2) one lane per candidate walks its chain
                if ((uint32_t)lane < n_c) {
                    const uint32_t cell = BT.cand[lane];
                    uint32_t bl = cell >> 16, bi = cell & 0xffffu, n = 0;
                    while (true) {
                        const uint32_t x = BT.base[bl] + bi;
                        BT.cidx[lane][n] = (uint16_t)x;
                        BT.cbl[lane][n] = (uint8_t)bl;
                        ++n;
                        const int32_t pv = dpp[x];
                        if (pv < 0 || n >= (uint32_t)CM_MAX_CHAIN_FRAGS) break;
                        bl = (uint32_t)pv >> 16;
                        bi = (uint32_t)pv & 0xffffu;
                    }
                    BT.clen[lane] = (uint8_t)n;
                }
                __syncthreads();
                // (3) the skip rule, candidates in order
                const uint32_t first = best_count;
                uint32_t n_emit = 0;
                for (uint32_t a = 0; a < n_c && best_count < max_best; ++a) {
                    if (cur < best_score) {
                        const uint32_t spos = LP[BT.cidx[a][0]];
                        bool mine = false;
                        for (uint32_t x = lane; x < n_rep; x += 64) mine = mine || BT.rep[x] == spos;
                        if (__ballot(mine) != 0ull) continue;
                    }
                    const uint32_t cl = BT.clen[a];
                    if (lane == 0) BT.emit[n_emit] = (uint8_t)a;
                    if ((uint32_t)lane >= 1u && (uint32_t)lane < cl) BT.rep[n_rep + lane - 1] = LP[BT.cidx[a][lane]];
                    if (best_count == 0) {
                        first_q0 = (int)BT.cbl[a][0] * kmer;
                        first_qlast = (int)BT.cbl[a][cl - 1] * kmer;
                    }
                    n_rep += cl - 1;
                    ++n_emit;
                    ++best_count;
                    __syncthreads();
                }
                __syncthreads();
                // (4) write the chains emitted for this score
                for (uint32_t x = lane; x < n_emit * (uint32_t)CM_MAX_CHAIN_FRAGS; x += 64) {
                    const uint32_t k = x / (uint32_t)CM_MAX_CHAIN_FRAGS, f = x % (uint32_t)CM_MAX_CHAIN_FRAGS;
                    const uint32_t a = BT.emit[k];
                    if (f < BT.clen[a]) {
                        CM_G cm_chain &ch = out[first + k];
                        ch.rpos[f] = LP[BT.cidx[a][f]];
                        ch.qpos[f] = (int32_t)BT.cbl[a][f] * kmer;
                    }
                }
                if ((uint32_t)lane < n_emit) {
                    CM_G cm_chain &ch = out[first + lane];
                    ch.score = (float)cur;
                    ch.chain_len = BT.clen[BT.emit[lane]];
                }
                __syncthreads();
                have = hv;
                cur = nxt;
            }
#if defined(CM_CHAIN_DIAG)      // shape of the back-tracking: events, score levels walked, passes over the log (64 events each)
            if (lane == 0) {
                atomicAdd(&counters[20], (unsigned long long)n_ev);
                atomicAdd(&counters[21], dg_levels);
                atomicAdd(&counters[22], 1ull);
                atomicAdd(&counters[23], dg_levels * (unsigned long long)((n_ev + 63) / 64));
                atomicAdd(&counters[24], (unsigned long long)ncell);
            }
#endif
        }
        if (best_count == 0) {          // singletons (lane 0 emits; every lane keeps the count)
            for (int ii = kc - 1; ii >= 0; --ii)
                for (uint32_t i = 0; i < cn[ii]; ++i) {
                    if (best_count >= max_best) break;
                    if (best_count == 0) first_q0 = first_qlast = ii * kmer;
                    if (lane == 0) {
                        CM_G cm_chain &ch = out[best_count];
                        ch.rpos[0] = LP[base[ii] + i];
                        ch.qpos[0] = ii * kmer;
                        ch.score = (float)dps[base[ii] + i];
                        ch.chain_len = 1;
                    }
                    ++best_count;
                }
        }
        __threadfence_block();
        __syncthreads();
        if (lane == 0) {
            nchain[r] = (int32_t)best_count;
            int rs = 0;
            if (best_count > 0) rs = pack_resid(first_q0, len - (first_qlast + kmer));
            resid[r] = (uint16_t)rs;
#if defined(CM_CHAIN_DIAG)
            const unsigned long long dg3 = wall_clock64();
            atomicAdd(&counters[5], dg1 - dg0);
            atomicAdd(&counters[6], dg2 - dg1);
            atomicAdd(&counters[7], dg3 - dg2);
#endif
        }
        __syncthreads();
    }
}

// ---- light / heavy split of the pair stage -------------------------------------------------
// A pair whose chain lists can produce many mate pairs and unpaired-chain extensions (reads from
// repeats: up to 30 x 30 pairs plus 60 full-length extensions) costs 100x the median pair; as one lane
// it would hold its whole wave for milliseconds.  Such pairs are listed by k_classify and mapped a
// few per *wave* by k_pair_heavy (the 64 lanes share the pairing predicates, the mate-pair extensions and
// the unpaired-chain extensions of HG pairs; an owner lane per pair folds the outcomes in the reference's
// order).  Everything else stays one pair per lane in k_pair.
constexpr int HEAVY_COST = 6;             // hg38-like bench, ms per step at 3 / 4 / 5 / 6 / 8 / 12 / 16: 32.1 / 31.3 / 26.5 / 26.6 / 28.0 / 30.1 / 31.7
constexpr int N_BUCKETS = 7;             // residual-length buckets
constexpr int HEAVY_CLS = 15;            // class of the pairs mapped by k_pair_heavy
// class of a pair for the pair stage: -2 inactive, HEAVY_CLS heavy (k_pair_heavy), else
// (genic ? 7 : 0) + bucket of the total residual length of its best chains (bases left to extend).
// genic: some best chain starts inside an annotated exon, i.e. the pair will walk transcripts during extension
// while the others only extend on the genome.  k_pair walks the light pairs class by class so the lanes of a
// wave carry similar work; the classes are a heuristic, results do not depend on them.
// (Measured: a "some residual is inexact" flag as a further key costs as much in k_pair_cls as it saves in k_pair.)
__device__ inline int pair_class(const Core &c, const cm_chain *chains, const uint16_t *resid4, const int32_t *nchain,
                                 const uint8_t *active, uint64_t pair0, uint32_t t, int heavy_cost, int *sub) {
    const uint64_t p = pair0 + t;
    if (sub) *sub = 0;
    if (!active[p]) return -2;
    const int32_t *nc = nchain + 4 * (uint64_t)t;
    const int a = nc[0], b = nc[1], cc = nc[2], d = nc[3];
    // a pair with chains on one mate only is settled without any extension (OEANCH, src/filter.cpp:196-203): never heavy
    if ((a + b) > 0 && (cc + d) > 0 && (a * d + cc * b + a + b + cc + d) > heavy_cost) {
        if (sub) {          // heavy pairs: cost level as the first radix key, so that k_pair_heavy starts with the longest ones
            const int cost = a * d + cc * b + a + b + cc + d;
            const int lv = cost <= 12 ? 0 : cost <= 16 ? 1 : cost <= 24 ? 2 : cost <= 32 ? 3 : cost <= 48 ? 4 : cost <= 64 ? 5 : cost <= 96 ? 6 : cost <= 128 ? 7
                           : cost <= 192 ? 8 : cost <= 256 ? 9 : cost <= 384 ? 10 : cost <= 512 ? 11 : cost <= 768 ? 12 : 13;
            *sub = lv << 4;
        }
        return HEAVY_CLS;
    }
    const uint16_t *q = resid4 + 4 * (uint64_t)t;
    int resid = 0;
    for (int x = 0; x < 4; ++x) resid += (q[x] & 0xff) + (q[x] >> 8);
    if (sub) {          // which of the four extensions of the main orientation have bases to extend (LL, LR, RL, RR of both_mates)
        const bool main0 = a > 0 && d > 0;                 // forward R1 / backward R2 carries chains, else the other orientation
        const uint16_t f = q[main0 ? 0 : 2], bk = q[main0 ? 3 : 1];
        *sub = ((f & 0xff) ? 1 : 0) | ((f >> 8) ? 2 : 0) | ((bk & 0xff) ? 4 : 0) | ((bk >> 8) ? 8 : 0);
        int mx = f & 0xff;                                  // longest single residual of the main orientation, 16 levels
        mx = (f >> 8) > mx ? (f >> 8) : mx;
        mx = (bk & 0xff) > mx ? (bk & 0xff) : mx;
        mx = (bk >> 8) > mx ? (bk >> 8) : mx;
        const int lv = mx < 5 ? 0 : mx < 10 ? 1 : mx < 15 ? 2 : mx < 20 ? 3 : mx < 25 ? 4 : mx < 30 ? 5 : mx < 40 ? 6 : mx < 50 ? 7 : mx < 60 ? 8 : mx < 70 ? 9
                       : mx < 80 ? 10 : mx < 90 ? 11 : mx < 100 ? 12 : mx < 115 ? 13 : mx < 130 ? 14 : 15;
        *sub |= lv << 4;
    }
    const int bucket = resid < 25 ? 0 : resid < 50 ? 1 : resid < 100 ? 2 : resid < 150 ? 3 : resid < 200 ? 4 : resid < 300 ? 5 : 6;
    bool genic = false;
    for (int x = 0; x < 4 && !genic; ++x) {
        if (nc[x] <= 0) continue;
        genic = cmc::overlap(c, chains[((uint64_t)t * 4 + x) * CM_BESTCHAINLIM].rpos[0]) >= 0;
    }
    return (genic ? N_BUCKETS : 0) + bucket;
}
// Atomic-free counting sort of the tile's pairs by class (bucket 0..7, heavy = class 8):
// k_cls_count: class per pair + per-block class histogram;  k_cls_scan: one workgroup turns the
// [class][block] histogram into exclusive bases (heaviest light bucket first) and totals;
// k_cls_place: writes perm[] (light pairs) / hlist[] (heavy pairs) at base + rank inside the block.
constexpr int CHAIN_LIGHT_CLS = 12;     // chaining classes below this are light
constexpr int CLS_T = 1024;            // elements per block of the counting sorts
constexpr int CLS_W = 256;             // threads per block: every wave takes CLS_T / CLS_W stretches of 64 elements in turn (blocks of 1024 threads waited up
                                       // to 1.4 ms for sixteen free wave slots on one CU next to the seeding kernel)
constexpr int CLS_REP = CLS_T / CLS_W;
constexpr int N_CLS = 16;                // classes a sort can use (pairs: 0..13 light + 15 heavy; chaining: 0..11 light + 12..15 heavy)
constexpr int CTR_SUM = 16, CTR_BASE = 32, CTR_WORDS = 64;
constexpr int CTR_NEXT = 48;             // spare words of the pair stage's class counters: work cursors of k_pair / k_pair_heavy, then the
constexpr int CTR_RETRY = 50;            // re-run list's length and the re-run launch's cursor (RetryArgs)
constexpr unsigned HEAVY_GRID_MAX = 4096;   // d_hres holds the task outcomes + scratch of this many k_pair_heavy blocks
constexpr int RETRY_GRID = 8;            // blocks of the re-run launch of k_pair (pairs beyond the first pass's capacity: a handful per run, if any)
constexpr int RETRY_SPILL = 2040;        // + MEMO_N in registers: 2048 memoised exon pieces per extend call
__device__ inline void block_class_ranks(int k, unsigned int (*wcnt)[N_CLS], unsigned int &rank_in_wave, int lane, int wave) {
    // wcnt[w][c] = number of lanes of wave w with class c; rank_in_wave = rank of this lane among its class in its wave
    rank_in_wave = 0;
#pragma unroll
    for (int c = 0; c < N_CLS; ++c) {
        const unsigned long long m = __ballot(k == c);
        if (lane == 0) wcnt[wave][c] = (unsigned int)__popcll(m);
        if (k == c) rank_in_wave = (unsigned int)__popcll(m & ((1ull << lane) - 1ull));
    }
}
// Where the light / heavy line of the pair stage goes depends on how much heavy work a tile holds.  The two pair kernels run side
// by side and the stage ends with the later one.  With little heavy work (chr21, the round-2 genome: a few percent of the pairs
// come from repeats) the heavy kernel is done long before the light one, and every multi-chain pair is best taken off the light
// kernel's waves (threshold 6: 95 vs 57 M pairs/s on chr21 against 48).  On the section-8(d) genome a sixth of the pairs are heavy,
// most of them with 30 x 30 chain pairs; the heavy kernel is the long pole and the light kernel has slack.  With one pair per wave
// in the heavy kernel the mid-cost pairs (7 .. 48) were best left to the light kernel (threshold 48: 16.7 vs 15.0 M pairs/s);
// with several pairs per wave (HG) the heavy kernel is the more efficient place for a pair with a dozen chain pairs, and the
// wide threshold is 16 (92.1 ms per step against 95.5 at 48 and 92.7 at 6).
// k_pair_cost adds up the cost beyond HEAVY_COST over the tile; k_pair_cls takes the wide threshold when that sum exceeds
// HEAVY_LOAD per pair of the tile.  Results never depend on the split.
constexpr int HEAVY_COST_WIDE = 16;
constexpr unsigned long long HEAVY_LOAD = 16;
__global__ void __launch_bounds__(BLK) k_pair_cost(const int32_t *nchain, const uint8_t *active, uint64_t pair0, uint32_t n_tile, int base_cost,
                                                  unsigned long long *sum) {
    __shared__ unsigned long long sh[BLK / 64];
    const uint32_t t = blockIdx.x * BLK + threadIdx.x;
    unsigned long long v = 0;
    if (t < n_tile && active[pair0 + t]) {
        const int32_t *nc = nchain + 4 * (uint64_t)t;
        const int a = nc[0], b = nc[1], cc = nc[2], d = nc[3];
        const int cost = a * d + cc * b + a + b + cc + d;
        if ((a + b) > 0 && (cc + d) > 0 && cost > base_cost) v = (unsigned long long)cost;
    }
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long s2 = 0;
        for (int w = 0; w < BLK / 64; ++w) s2 += sh[w];
        if (s2) atomicAdd(sum, s2);
    }
}
__global__ void __launch_bounds__(BLK) k_pair_cls(KCore kc, const cm_chain *chains, const uint16_t *resid, const int32_t *nchain,
                                                 const uint8_t *active, uint64_t pair0, uint32_t n_tile, int8_t *cls, int32_t *cat, int heavy_cost,
                                                 int8_t *cls_sub, int8_t *cls_sub2, uint8_t *act_out, const unsigned long long *heavy_load) {
    const uint32_t t = blockIdx.x * BLK + threadIdx.x;
    if (t >= n_tile) return;
    const Core c = cmc::to_core(kc);
    if (heavy_load && *heavy_load > HEAVY_LOAD * (unsigned long long)n_tile) heavy_cost = HEAVY_COST_WIDE;
    int sub = 0;
    const int k = pair_class(c, chains, resid, nchain, active, pair0, t, heavy_cost, &sub);
    cls[t] = (int8_t)k;
    cls_sub[t] = (int8_t)(k < 0 ? -2 : (k == HEAVY_CLS ? 0 : (sub & 15)));    // secondary key
    cls_sub2[t] = (int8_t)(k < 0 ? -2 : (sub >> 4));                           // tertiary key (first pass of the radix sort)
    if (k == -2) {                                     // retired in an earlier round: not mapped, stays retired
        cat[pair0 + t] = -1;
        act_out[pair0 + t] = 0;
    }
}
// work class of one chaining problem: number of (hit, later hit) pairs the DP may have to examine
__global__ void __launch_bounds__(BLK) k_chain_cls(const uint32_t *scnt, const uint32_t *sraw, int S, uint32_t n_prob, int8_t *cls, int32_t *high,
                                                  unsigned long long light_w, unsigned int light_cells, int32_t *nchain, uint16_t *resid,
                                                  const uint8_t *active, uint64_t pair0) {
    const uint32_t r = blockIdx.x * BLK + threadIdx.x;
    if (r >= n_prob) return;
    // nchain == null: the chain records are still being read by an earlier pair stage; `high` is a scratch array of the seed set and
    // k_chain_apply carries the three fields over when the records are free (the class -2 stands for "no chain, no kernel visits it")
    if (!active[pair0 + (r >> 2)]) {    // seeded under older flags (a superset), retired since: no chains wanted
        high[r] = 0;
        if (nchain) {
            nchain[r] = 0;
            resid[r] = 0;
        }
        cls[r] = -2;
        return;
    }
    unsigned long long w = 0, suffix = 0;
    int hh = 0;
    for (int s = S - 1; s >= 0; --s) {
        const unsigned long long c = scnt[(uint64_t)r * S + s];
        w += c * suffix;
        suffix += c;
        if (sraw[(uint64_t)r * S + s] > 0 && c == 0) ++hh;       // get_best_chains high_hits (also for problems k_chain skips)
    }
    high[r] = hh;
    if (suffix == 0 && nchain) {        // no retained hit: no chain, and no kernel visits this problem
        nchain[r] = 0;
        resid[r] = 0;
    }
    // classes 0..11 = light (one lane each, k_chain; grouped so the lanes of a wave carry similar DPs),
    // 12..15 = heavy (one wave each, k_chain_heavy, heaviest class first)
    const bool light = w <= light_w && suffix <= light_cells;
    const unsigned int n = (unsigned int)suffix;
    cls[r] = (int8_t)(suffix == 0 ? -2
                      : light ? (n <= 3 ? 0 : n <= 6 ? 1 : n <= 7 ? 2 : n <= 9 ? 3 : n <= 12 ? 4 : n <= 16 ? 5 : n <= 24 ? 6 : n <= 32 ? 7 : n <= 48 ? 8 : n <= 64 ? 9
                                 : n <= 96 ? 10 : 11)
                      : w <= 4096 ? 12 : w <= 65536 ? 13 : w <= 1048576 ? 14 : 15);
}
__global__ void __launch_bounds__(BLK) k_chain_apply(uint32_t n_prob, const int8_t *cls, const int32_t *thigh, int32_t *high, int32_t *nchain, uint16_t *resid) {
    const uint32_t r = blockIdx.x * BLK + threadIdx.x;
    if (r >= n_prob) return;
    high[r] = thigh[r];
    if (cls[r] == -2) {
        nchain[r] = 0;
        resid[r] = 0;
    }
}
// `order` (optional): visit the elements in this order (second pass of an LSD radix sort: order = the permutation of the
// first pass, *n_order entries); the element at position i is order[i]
__global__ void __launch_bounds__(CLS_W) k_cls_hist(const int8_t *cls, uint32_t n, unsigned int *blk_cnt, uint32_t nb, const uint32_t *order,
                                                    const unsigned int *n_order) {
    __shared__ unsigned int wcnt[CLS_T / 64][N_CLS];
    const uint32_t lim = n_order ? (*n_order < n || order ? *n_order : n) : n;      // (without an order: a device-side count, at most n)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // a workgroup takes the stretches of CLS_T elements blockIdx.x, blockIdx.x + gridDim.x, ... (one each when the grid has nb
    // workgroups; a smaller grid when the count is only known on the device and most of nb would have nothing to do)
    for (uint32_t vb = blockIdx.x; vb < nb; vb += gridDim.x) {
        if (!order && n_order && vb * (uint32_t)CLS_T >= lim) break;                 // (k_cls_scan is told the same count: nothing to report)
#pragma unroll
        for (int j = 0; j < CLS_REP; ++j) {
            const int vw = wave * CLS_REP + j;                 // stretch of 64 elements within the block
            const uint32_t i = vb * CLS_T + (uint32_t)vw * 64u + (uint32_t)lane;
            const int k = i < lim ? (int)cls[order ? order[i] : i] : -2;
            unsigned int r;
            block_class_ranks(k, wcnt, r, lane, vw);
        }
        __syncthreads();
        if (threadIdx.x < N_CLS) {
            unsigned int tot = 0;
            for (int w = 0; w < CLS_T / 64; ++w) tot += wcnt[w][threadIdx.x];
            blk_cnt[(size_t)threadIdx.x * nb + vb] = tot;
        }
        __syncthreads();
    }
}
// ctr[c] = total of class c, ctr[CTR_SUM] = entries placed in perm[], ctr[CTR_BASE + c] = base offset of class c in perm[]
// (highest class first); the classes in the bit mask `separate` (none: -1) go to their own lists instead
// n_dev (optional): the elements there are, on the device; only the blocks that hold some are scanned (rows keep their stride nb).
// One workgroup of four waves, four class rows each (a 1024-thread workgroup waited 0.1 - 3 ms for sixteen free wave slots on one CU).
constexpr int SCAN_CLS_T = 256;
__global__ void __launch_bounds__(SCAN_CLS_T) k_cls_scan(unsigned int *blk_cnt, uint32_t nb, unsigned int *ctr, int separate, int n_used,
                                                         const unsigned int *n_dev = nullptr) {
    // one wave per class row at a time: 64 block counts per step, exclusive scan inside the wave, running total carried on
    __shared__ unsigned int tot[N_CLS];
    const uint32_t t = threadIdx.x;
    const int lane = (int)(t & 63u), wave = (int)(t >> 6);
    if (t < N_CLS) tot[t] = 0;
    __syncthreads();
    uint32_t nb_eff = nb;
    if (n_dev) {
        const uint32_t used = (uint32_t)(((unsigned long long)*n_dev + CLS_T - 1) / CLS_T);
        nb_eff = used < nb ? used : nb;
    }
    for (int c = wave; c < N_CLS; c += SCAN_CLS_T / 64) {
        if (c >= n_used) break;                          // classes >= n_used are not produced by this caller
        unsigned int *row = blk_cnt + (size_t)c * nb;
        unsigned int run = 0;
        for (uint32_t i0 = 0; i0 < nb_eff; i0 += 64) {
            const uint32_t i = i0 + (uint32_t)lane;
            const unsigned int x = i < nb_eff ? row[i] : 0u;
            unsigned int incl = x;
            for (int o = 1; o < 64; o <<= 1) {
                const unsigned int y = __shfl_up(incl, o);
                if (lane >= o) incl += y;
            }
            if (i < nb_eff) row[i] = run + incl - x;
            run += __shfl(incl, 63);
        }
        if (lane == 0) tot[c] = run;
    }
    __syncthreads();
    if (t == 0) {
        unsigned int off = 0;
        for (int c = N_CLS - 1; c >= 0; --c) {          // heaviest class first
            if (separate >= 0 && ((separate >> c) & 1)) {      // classes with their own list
                ctr[CTR_BASE + c] = 0;
                continue;
            }
            ctr[CTR_BASE + c] = off;
            off += tot[c];
        }
        ctr[CTR_SUM] = off;
        for (int c = 0; c < N_CLS; ++c) ctr[c] = tot[c];
    }
}
__global__ void __launch_bounds__(CLS_W) k_cls_place(const int8_t *cls, uint32_t n_tile, const unsigned int *blk_base, uint32_t nb,
                                                     const unsigned int *ctr, uint32_t *perm, uint32_t *hlist, const uint32_t *order,
                                                     const unsigned int *n_order) {
    __shared__ unsigned int wcnt[CLS_T / 64][N_CLS];
    const uint32_t lim = n_order ? (*n_order < n_tile || order ? *n_order : n_tile) : n_tile;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (uint32_t vb = blockIdx.x; vb < nb; vb += gridDim.x) {                       // (see k_cls_hist)
        if (!order && n_order && vb * (uint32_t)CLS_T >= lim) break;
        uint32_t t[CLS_REP];
        int k[CLS_REP];
        unsigned int r[CLS_REP];
#pragma unroll
        for (int j = 0; j < CLS_REP; ++j) {
            const int vw = wave * CLS_REP + j;
            const uint32_t i = vb * CLS_T + (uint32_t)vw * 64u + (uint32_t)lane;
            t[j] = i < lim ? (order ? order[i] : i) : 0u;
            k[j] = i < lim ? (int)cls[t[j]] : -2;
            block_class_ranks(k[j], wcnt, r[j], lane, vw);
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < CLS_REP; ++j) {
            if (k[j] < 0) continue;
            const int vw = wave * CLS_REP + j;
            unsigned int before = 0;
            for (int w = 0; w < vw; ++w) before += wcnt[w][k[j]];
            const unsigned int pos = blk_base[(size_t)k[j] * nb + vb] + before + r[j];
            if (k[j] == HEAVY_CLS && hlist) hlist[pos] = t[j];
            else perm[ctr[CTR_BASE + k[j]] + pos] = t[j];
        }
        __syncthreads();
    }
}

struct HRes {            // outcome of one mate-pair task, handed from the computing lane to the pair's owner lane
    cmc::MM r1, r2;
    int32_t row;
    int32_t pair_type;
    uint8_t ok, is_left, pad[2];
};
// k_pair_heavy maps HG heavy pairs per wave side by side.  A heavy pair has hundreds of pairing-predicate evaluations but few
// accepted mate pairs (section-8(d) genome: 5.5 per process_mates call, tests/diag/heavy_shape.py): one pair per wave left 55 of
// the 64 lanes idle through the extensions, which are four fifths of the kernel (7.9 lanes per VALU instruction, VALU issue
// saturated).  Here every phase of process_mates runs over ONE work list of the whole wave -- chain ends, predicate evaluations,
// accepted mate pairs, unpaired chains of all HG pairs, one item per lane -- and lane g < HG (the "owner" of slot g) keeps pair g's
// MatchedRead and folds its outcomes in the reference's order.  What a lane needs of a pair it reads from the pair's slot in LDS.
#ifndef CM_HEAVY_G
#define CM_HEAVY_G 6            // 4 / 6 / 8 / 16: 89.3 / 85.9 / 89.4 / 96 ms per step on the hg38-like bench (33 tasks per 64-lane batch at 6)
#endif
constexpr int HG = CM_HEAVY_G;
static_assert(HG >= 1 && HG <= 16, "the task list packs the slot in 4 bits");
constexpr int HEAVY_LIST = HG * CM_BESTCHAINLIM * CM_BESTCHAINLIM;      // accepted (i, j) of all slots, worst case
constexpr int HEAVY_SCRATCH = HEAVY_LIST * 2;     // bytes of per-block global scratch behind HRes[64]: the task list (u16)
// One pair in flight (LDS).  Written by its owner lane unless noted.  Packed: LDS is allocated in steps of 1 280 bytes on this part
// and this kernel's request (two staging buffers + six slots: 14 552 bytes = 12 steps) must not grow into the next one -- a 13th
// step costs a resident wave per CU and 6 ms per bench step (NOTES 44).
struct HSlot {
    cmc::g_u8 fseq, bseq;                // the two reads (forward read: as stored; backward read: reverse complement)
    uint32_t fch_i, bch_i;               // chain lists of this attempt (record index into the tile's chain records): forward read's, backward read's
    uint32_t t;                          // the pair (tile-relative): its capacity-limit word is pair_err[t] (RetryArgs)
    unsigned int inv_nb;                 // ceil(65536 / nb): idx / nb = idx * inv_nb >> 16 for idx < 900
    unsigned int fp, bp;                 // chains that found a mate (atomicOr by the predicate lanes)
    int ntask;                           // accepted mate pairs (atomicAdd by the predicate lanes)
    unsigned int fun, bun;               // unpaired chains to extend
    int exf, exb;                        // outcome of the unpaired-chain extensions (atomicMin)
    int16_t flen, blen;
    int16_t nf, nb;                      // chains; 0 / 0 while the slot sits an attempt out
    int16_t nfu, nbu;
    int8_t saved_type;                   // MatchedRead.type at the start of the attempt (pairing predicate)
    int8_t done;                         // the fold returned CONCRD: the slot's remaining tasks are dead
    int8_t gf, gb;                       // genic flag of each side's first unpaired chain (written by the k == 0 lane)
    int fe[CM_BESTCHAINLIM], re[CM_BESTCHAINLIM];                        // exon interval of each chain's first fragment (written by the chain-end lanes)
    uint32_t r0[2 * CM_BESTCHAINLIM], rend[2 * CM_BESTCHAINLIM];         // reference span of the chains: forward, then (from CM_BESTCHAINLIM) backward
};
__device__ inline cmc::g_chain slot_fch(CM_L const HSlot &s, const cm_chain *base) { return (cmc::g_chain)(base + s.fch_i); }
__device__ inline cmc::g_chain slot_bch(CM_L const HSlot &s, const cm_chain *base) { return (cmc::g_chain)(base + s.bch_i); }
__device__ inline cmc::g_err slot_perr(CM_L const HSlot &s, uint32_t *pair_err) { return (cmc::g_err)(pair_err + s.t); }
static_assert(sizeof(HSlot) <= 808, "six slots + two staging buffers of 72-byte rows fit 11 LDS allocation steps");

__device__ inline int nth_set_bit(uint32_t m, int k) {
    for (int x = 0; x < k; ++x) m &= m - 1;
    return __ffs((int)m) - 1;
}
// item x of a work list that concatenates the slots' items: pre[g] = items before slot g's, pre[HG] = all
__device__ inline void locate(const int (&pre)[HG + 1], int x, int &g, int &k) {
    g = 0;
    int b = 0;
#pragma unroll
    for (int s = 1; s < HG; ++s) {
        const bool ge = x >= pre[s];
        g += ge ? 1 : 0;
        b = ge ? pre[s] : b;
    }
    k = x - b;
}

#ifndef CM_HEAVY_WAVES
#define CM_HEAVY_WAVES CM_PAIR_WAVES
#endif
// process_read (src/filter.cpp:124-241) + process_mates (src/filter.cpp:244-395) for HG pairs per wave iteration.
__global__ void __launch_bounds__(BLK_PAIR, CM_HEAVY_WAVES) k_pair_heavy(KCore kc, ReadsDev rd, uint64_t pair0, const uint32_t *hlist, const unsigned int *hcount,
                                                         const cm_chain *chains, const int32_t *nchain, const int32_t *high, cm_mapped_read *state,
                                                         uint8_t *active, int32_t *cat, int is_last, int *err, unsigned long long *counters,
                                                         int str_cap, unsigned long long *dbg_rows, HRes *hres, unsigned int *next_pair,
                                                         RetryArgs ra) {
    extern __shared__ uint32_t lds_words[];
    const int lane = threadIdx.x;
    CM_L uint8_t *base = (CM_L uint8_t *)lds_words;
    CM_S uint8_t *lane_base = base + 4 * lane;
    const int lds_stage_bytes = 2 * lbuf_bytes(str_cap) * BLK_PAIR;
    const int str_stride = lbuf_bytes(str_cap) * BLK_PAIR;
#if defined(CM_DIAG)
    __shared__ unsigned long long tick_w[65];
    cmc::Tick tick{};
    tick.w = (CM_L unsigned long long *)tick_w;
    tick.wave_on = 1;
    tick.last = wall_clock64();
    for (int i = threadIdx.x; i < 65; i += BLK_PAIR) tick_w[i] = i == 0 ? tick.last : 0ull;
    __syncthreads();
    cmc::DpMem sm{cmc::LBuf{lane_base, str_cap}, cmc::LBuf{lane_base + str_stride, str_cap}, (cmc::g_err)err, &tick};
#else
    cmc::DpMem sm{cmc::LBuf{lane_base, str_cap}, cmc::LBuf{lane_base + str_stride, str_cap}, (cmc::g_err)err};
#endif
    CM_L HSlot *S = (CM_L HSlot *)(base + lds_stage_bytes);
    CM_G HRes *res = (CM_G HRes *)(hres + (size_t)blockIdx.x * 64);
    CM_G uint16_t *list = (CM_G uint16_t *)((CM_G uint8_t *)(hres + (size_t)gridDim.x * 64) + (size_t)blockIdx.x * HEAVY_SCRATCH);
    const Core c = cmc::to_core(kc);
    const cmc::Ext ext(c, sm);
    const int kmer = c.P.kmer;
    const unsigned int n_heavy = *hcount;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    uint32_t tids[cmc::MAX_TID];
    // HG pairs at a time from a shared cursor (the list starts with the most expensive pairs; neighbours cost about the same)
    auto take = [&]() { return (unsigned int)__shfl((int)(lane == 0 ? atomicAdd(next_pair, (unsigned int)HG) : 0u), 0); };
    for (unsigned int h0 = take(); h0 < n_heavy; h0 = take()) {
        // ---- owners: lane g holds pair h0 + g -------------------------------------------------------------------------
        const bool owner = lane < HG && h0 + (unsigned int)lane < n_heavy;
        uint32_t t = 0;
        uint64_t p = 0;
        int len1 = 0, len2 = 0, st = -1;
        cmc::ChainSet sets[4];
        cmc::g_u8 s1 = nullptr, s2 = nullptr;
        cm_mapped_read mr{};
        bool first = true;
        for (int x = 0; x < 4; ++x) { sets[x].ch = nullptr; sets[x].n = 0; }
        if (owner) {
            t = hlist[h0 + lane];
            p = pair0 + t;
            const uint64_t a0 = rd.off1[p], a1 = rd.off1[p + 1], b0 = rd.off2[p], b1 = rd.off2[p + 1];
            len1 = (int)(a1 - a0);
            len2 = (int)(b1 - b0);
            s1 = (cmc::g_u8)(rd.seq1 + a0);
            s2 = (cmc::g_u8)(rd.seq2 + b0);
            int hh[4];
            for (int x = 0; x < 4; ++x) {
                const uint64_t r = (uint64_t)t * 4 + x;
                sets[x].ch = (cmc::g_chain)(chains + r * CM_BESTCHAINLIM);
                sets[x].n = nchain[r];
                hh[x] = high[r];
            }
            mr = state[p];
            const int n1 = sets[0].n + sets[1].n, n2 = sets[2].n + sets[3].n;
            if (n1 + n2 <= 0) {             // unreachable for a pair classified heavy; kept for completeness
                st = ((hh[0] + hh[1] > 0) && (hh[2] + hh[3] > 0)) ? CM_NOPROC_MANYHIT : CM_NOPROC_NOMATCH;
                cmc::mr_update_type(mr, st);
            } else if (n1 <= 0 || n2 <= 0) {
                st = CM_OEANCH;
                cmc::mr_update_type(mr, st);
            } else {
                const float fc1 = sets[0].n > 0 ? sets[0].ch[0].score : 0.f, bc1 = sets[1].n > 0 ? sets[1].ch[0].score : 0.f;
                const float fc2 = sets[2].n > 0 ? sets[2].ch[0].score : 0.f, bc2 = sets[3].n > 0 ? sets[3].ch[0].score : 0.f;
                first = (fc1 + bc2) >= (fc2 + bc1);
            }
        }
        for (int attempt = 0; attempt < 2; ++attempt) {
            // ---- owners publish the attempt of their pair (process_read's loop: forward R1 / backward R2, or the other way) ----
            const bool on = owner && st < 0;
            const bool r1_fwd = (attempt == 0) == first;
            if (lane < HG) {
                CM_L HSlot &s = S[lane];
                const cmc::ChainSet &F = r1_fwd ? sets[0] : sets[2], &B = r1_fwd ? sets[3] : sets[1];
                s.fch_i = (uint32_t)(F.ch ? F.ch - (cmc::g_chain)chains : 0);
                s.bch_i = (uint32_t)(B.ch ? B.ch - (cmc::g_chain)chains : 0);
                s.fseq = r1_fwd ? s1 : s2;
                s.bseq = r1_fwd ? s2 : s1;
                s.flen = (int16_t)(r1_fwd ? len1 : len2);
                s.blen = (int16_t)(r1_fwd ? len2 : len1);
                s.t = t;
                s.nf = (int16_t)(on ? F.n : 0);
                s.nb = (int16_t)(on ? B.n : 0);
                s.inv_nb = (on && B.n > 0) ? (65536u + (unsigned int)B.n - 1u) / (unsigned int)B.n : 0u;
                s.saved_type = (int8_t)mr.type;
                s.fp = 0u;
                s.bp = 0u;
                s.ntask = 0;
                s.done = 0;
                s.nfu = 0;
                s.nbu = 0;
            }
            if (__ballot(on) == 0ull) break;                   // uniform: every pair of this iteration is settled
            __syncthreads();
            int pre[HG + 1];
            // ---- chain ends: reference span + exon interval of the first fragment, one chain per lane -------------------------
            pre[0] = 0;
#pragma unroll
            for (int g = 0; g < HG; ++g) pre[g + 1] = pre[g] + S[g].nf + S[g].nb;
            for (int x = lane; x < pre[HG]; x += 64) {
                int g, k;
                locate(pre, x, g, k);
                CM_L HSlot &s = S[g];
                const int nf = s.nf;
                const bool back = k >= nf;
                const int ci = back ? k - nf : k;
                const cmc::CHEnds e{(back ? slot_bch(s, chains) : slot_fch(s, chains)) + ci, kmer};
                s.r0[(back ? CM_BESTCHAINLIM : 0) + ci] = e.r0;
                s.rend[(back ? CM_BESTCHAINLIM : 0) + ci] = e.rend;
                (back ? s.re : s.fe)[ci] = cmc::overlap(c, e.r0);
            }
            __syncthreads();
            // ---- pairing predicate of every (i, j) of every slot; the accepted ones go to the task list in (slot, i, j) order ----
            pre[0] = 0;
#pragma unroll
            for (int g = 0; g < HG; ++g) pre[g + 1] = pre[g] + S[g].nf * S[g].nb;
            int n_task = 0;
            for (int b0 = 0; b0 < pre[HG]; b0 += 64) {
                const int x = b0 + lane;
                uint32_t code = 0;
                int g = 0, idx = 0;
                if (x < pre[HG]) {
                    locate(pre, x, g, idx);
                    CM_L HSlot &s = S[g];
                    const int i = (int)(((unsigned int)idx * s.inv_nb) >> 16), j = idx - i * s.nb;
                    const cmc::CHEnds F{s.r0[i], s.rend[i]}, R{s.r0[CM_BESTCHAINLIM + j], s.rend[CM_BESTCHAINLIM + j]};
                    code = cmc::pair_code(c, F, R, s.fe[i], s.re[j], s.saved_type);
                    if (code) {
                        atomicOr((unsigned int *)&s.fp, 1u << i);
                        atomicOr((unsigned int *)&s.bp, 1u << j);
                        atomicAdd((int *)&s.ntask, 1);
                    }
                }
                const unsigned long long m = __ballot(code != 0);
                if (code) list[n_task + __popcll(m & lt_mask)] = (uint16_t)((unsigned int)idx | (code << 10) | ((unsigned int)g << 12));
                n_task += __popcll(m);
            }
            __syncthreads();
            CM_TICK(sm, 29);
            // ---- mate-pair tasks: one per lane, 64 at a time; each owner folds its slot's outcomes in order --------------------
            int t_lo = 0, t_hi = 0;                                     // the owner's slice of the task list
            if (lane < HG) {
                for (int g = 0; g < HG; ++g) t_lo += g < lane ? S[g].ntask : 0;
                t_hi = t_lo + S[lane].ntask;
            }
            int min_ret1 = CM_ORPHAN, min_ret2 = CM_ORPHAN, g1 = 0, g2 = 0;          // meaningful on the owners
            bool early = false;
            for (int b0 = 0; b0 < n_task; b0 += 64) {
                const int x = b0 + lane;
                if (x < n_task) {
                    const unsigned int e = list[x];
                    CM_L HSlot &s = S[e >> 12];
                    if (!s.done) {
                        const uint32_t code = (e >> 10) & 3u;
                        const int idx = (int)(e & 1023u);
                        const int i = (int)(((unsigned int)idx * s.inv_nb) >> 16), j = idx - i * s.nb;
                        sm.err = slot_perr(s, ra.pair_err);
                        const cmc::TidList tl = (code == 1) ? cmc::common_tids(c, s.fe[i], s.re[j], tids) : cmc::TidList{tids, 0, -1, -1, false};
                        const cmc::CH F{slot_fch(s, chains) + i, kmer}, R{slot_bch(s, chains) + j, kmer};
                        const cmc::Read frd{s.fseq, s.flen, 0}, brd{s.bseq, s.blen, 1};
                        cmc::MM r1, r2;
                        bool il, ok;
                        int row;
                        cmc::extend_task(c, ext, F, R, tl, frd, brd, r1, r2, il, ok, row);
                        res[lane].r1 = r1;
                        res[lane].r2 = r2;
                        res[lane].row = row;
                        res[lane].pair_type = (int)code - 1;
                        res[lane].ok = ok;
                        res[lane].is_left = il;
                    }
                }
                __syncthreads();
                CM_TICK(sm, 30);
                if (on && !early) {
                    const int lo = t_lo > b0 ? t_lo : b0, hi = t_hi < b0 + 64 ? t_hi : b0 + 64;
                    for (int y = lo; y < hi; ++y) {
                        const cmc::MM r1 = res[y - b0].r1, r2 = res[y - b0].r2;
                        if (cmc::fold_task(c, r1, r2, res[y - b0].is_left != 0, res[y - b0].ok != 0, res[y - b0].row, res[y - b0].pair_type, r1_fwd, mr)) {
                            early = true;
                            S[lane].done = 1;
                            break;
                        }
                        min_ret1 = r1.type < min_ret1 ? r1.type : min_ret1;
                        min_ret2 = r2.type < min_ret2 ? r2.type : min_ret2;
                        g1 = (r1.exons_spos >= 0) || (r1.exons_epos >= 0);
                        g2 = (r2.exons_spos >= 0) || (r2.exons_epos >= 0);
                    }
                }
                __syncthreads();
                CM_TICK(sm, 31);
            }
            // ---- owners: does the pair go on to the unpaired-chain extensions? (src/filter.cpp:344-393) ------------------------
            int a = -1;                                                 // what process_mates returns
            bool do_f = false, do_b = false;
            if (on) {
                CM_L HSlot &s = S[lane];
                if (early) a = CM_CONCRD;
                else if (mr.type == CM_CONCRD || mr.type == CM_DISCRD || mr.type == CM_CHIORF || mr.type == CM_CHIBSJ || mr.type == CM_CHI2BSJ) a = mr.type;
                else {
                    const uint32_t fun = ~s.fp & (s.nf >= 32 ? 0xffffffffu : ((1u << s.nf) - 1u));
                    const uint32_t bun = ~s.bp & (s.nb >= 32 ? 0xffffffffu : ((1u << s.nb) - 1u));
                    do_f = min_ret1 != CM_CONCRD && fun != 0;
                    do_b = min_ret2 != CM_CONCRD && bun != 0;
                    if (!cmc::leftovers_matter(mr.type, min_ret1, do_f, min_ret2, do_b)) a = mr.type;
                    else {
                        s.fun = fun;
                        s.bun = bun;
                        s.nfu = (int16_t)(do_f ? __popc(fun) : 0);
                        s.nbu = (int16_t)(do_b ? __popc(bun) : 0);
                        s.exf = 99;
                        s.exb = 99;
                        s.gf = 0;
                        s.gb = 0;
                    }
                }
            }
            __syncthreads();
            // ---- unpaired chains: one full-length extension per lane.  The reference reuses one MatchedMate for all chains of a
            // side, so only the first chain's exon lookups ever happen (stale looked_up_* flags, filter.cpp:356-385): the lane
            // with k == 0 computes the genic flag of its side.
            pre[0] = 0;
#pragma unroll
            for (int g = 0; g < HG; ++g) pre[g + 1] = pre[g] + S[g].nfu + S[g].nbu;
            for (int x = lane; x < pre[HG]; x += 64) {
                int g, u;
                locate(pre, x, g, u);
                CM_L HSlot &s = S[g];
                const bool back = u >= s.nfu;
                const int k = back ? u - s.nfu : u;
                const int ci = nth_set_bit(back ? s.bun : s.fun, k);
                const cmc::CH ch{(back ? slot_bch(s, chains) : slot_fch(s, chains)) + ci, kmer};
                const cmc::Read frd{s.fseq, s.flen, 0}, brd{s.bseq, s.blen, 1};
                cmc::MM m = cmc::mm_init(c);
                sm.err = slot_perr(s, ra.pair_err);
                const int ex = ext.chain_both_sides(ch, back ? brd : frd, m, back ? -1 : 1);
                atomicMin((int *)(back ? &s.exb : &s.exf), ex);
                if (k == 0) {
                    cmc::overlap_to_spos(c, m);
                    cmc::overlap_to_epos(c, m);
                    (back ? s.gb : s.gf) = (int8_t)((m.exons_spos >= 0) || (m.exons_epos >= 0));
                }
            }
            __syncthreads();
            CM_TICK(sm, 15);
            if (on) {
                if (a < 0) {
                    CM_L HSlot &s = S[lane];
                    if (do_f) {
                        min_ret1 = s.exf < min_ret1 ? s.exf : min_ret1;
                        g1 = s.gf;
                    }
                    if (do_b) {
                        min_ret2 = s.exb < min_ret2 ? s.exb : min_ret2;
                        g2 = s.gb;
                    }
                    cmc::mr_update_type(mr, cmc::leftover_type(min_ret1, min_ret2, g1 != 0, g2 != 0));
                    a = mr.type;
                }
                if (c.P.scan_level == 0 && a == CM_CONCRD) st = CM_CONCRD;
            }
            __syncthreads();                                           // the slots are rewritten at the top of the next attempt
        }
        // the wave's own atomicOr's on the pairs' words have reached the L2 after this (a workgroup-scope fence = s_waitcnt for a
        // one-wave block; an agent-scope __threadfence() here invalidated the CU's vector L1 once per heavy pair and cost every
        // kernel on the chip 10 - 15 %)
        __threadfence_block();
        if (owner) {
            if (st < 0) st = mr.type;
            if (__hip_atomic_load(ra.pair_err + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
                ra.list[atomicAdd(ra.count, 1u)] = t;          // left as it was; the re-run launch of k_pair maps it (one lane, exact)
                active[p] = 1;                                 // (its flag: active until then)
            } else {
                uint8_t act = 1;
                cmc::finish_round(c, st, is_last, len1, len2, mr, act);
                state[p] = mr;
                active[p] = act;
                cat[p] = st;
                atomicAdd(&counters[3], 1ull);
            }
        }
        __syncthreads();
        CM_TICK(sm, 13);
    }
#if defined(CM_DIAG)
    if (dbg_rows && lane == 0)
        for (int i = 0; i < 64; ++i) dbg_rows[(size_t)blockIdx.x * 64 + i] = tick_w[1 + i];
#endif
}

#include "cm_heavy_pipe.h"

__global__ void __launch_bounds__(BLK) k_active_cls(const uint8_t *active, uint64_t n, int8_t *cls) {
    const uint64_t i = (uint64_t)blockIdx.x * BLK + threadIdx.x;
    if (i < n) cls[i] = active[i] ? 0 : -2;
}
__global__ void __launch_bounds__(BLK) k_gather_active(const uint32_t *perm, const unsigned int *count, unsigned long long cap, const cm_mapped_read *state,
                                                       unsigned long long *out_idx, cm_mapped_read *out_state) {
    const uint64_t i = (uint64_t)blockIdx.x * BLK + threadIdx.x;
    if (i >= *count || i >= cap) return;
    const uint32_t p = perm[i];
    out_idx[i] = p;
    out_state[i] = state[p];
}

__global__ void __launch_bounds__(BLK) k_gather_records(const uint32_t *perm, const unsigned int *count, unsigned long long cap, const cm_mapped_read *state,
                                                        unsigned long long index_base, cm_record *out) {
    const uint64_t i = (uint64_t)blockIdx.x * BLK + threadIdx.x;
    if (i >= *count || i >= cap) return;
    const uint32_t p = perm[i];
    out[i].pair = index_base + p;
    out[i].state = state[p];
}

// final MatchedRead::type histogram of a batch (cm_type_histogram): block-private counts in LDS, 14 atomics per block
__global__ void __launch_bounds__(BLK) k_type_hist(const cm_mapped_read *state, uint64_t n, unsigned long long *hist) {
    __shared__ unsigned int h[16];
    if (threadIdx.x < 16) h[threadIdx.x] = 0;
    __syncthreads();
    for (uint64_t i = (uint64_t)blockIdx.x * BLK + threadIdx.x; i < n; i += (uint64_t)gridDim.x * BLK) {
        const int t = state[i].type;
        if (t >= 0 && t < 14) atomicAdd(&h[t], 1u);
    }
    __syncthreads();
    if (threadIdx.x < 14 && h[threadIdx.x]) atomicAdd(&hist[threadIdx.x], (unsigned long long)h[threadIdx.x]);
}
__global__ void k_err_clear(int *err, int mask) {
    if (threadIdx.x == 0 && blockIdx.x == 0) atomicAnd(err, ~mask);
}
// bucket descriptors of a loaded contig (cmc::desc_pack): one thread per hash bucket
__global__ void __launch_bounds__(BLK) k_build_desc(const uint32_t *bucket_off, const uint16_t *checksum, uint64_t n_buckets, int kmer, uint32_t *desc) {
    const uint64_t hv = (uint64_t)blockIdx.x * BLK + threadIdx.x;
    if (hv >= n_buckets) return;
    const uint32_t b0 = bucket_off[hv], n = bucket_off[hv + 1] - b0;
    uint32_t out[cmc::DESC_WORDS];
    cmc::desc_pack(b0, n, [&](uint32_t i) { return (uint32_t)checksum[b0 + i]; }, kmer, out);
    uint4 v;
    v.x = out[0]; v.y = out[1]; v.z = out[2]; v.w = out[3];
    ((uint4 *)desc)[hv] = v;
}
// ---- cm_load_contig_raw: the index table flattened on the device --------------------------------------------------
// Multi-block scan of uint32 (totals stay below 2^32 here: table slots / entries of one contig): k_scan32_a scans blocks of
// S32_B items (8 consecutive items per thread) and leaves the block totals, k_scan32_b scans those in one workgroup,
// k_scan32_c adds the block bases.  add = 1: the items are in[i] + 1; inclusive: out[i] includes item i.
constexpr int S32_T = 1024, S32_E = 8, S32_B = S32_T * S32_E;
__global__ void __launch_bounds__(S32_T) k_scan32_a(const uint32_t *in, uint64_t n, uint32_t *out, uint32_t *bsum, uint32_t add, int inclusive) {
    __shared__ uint32_t wsum[S32_T / 64];
    const uint64_t base = (uint64_t)blockIdx.x * S32_B + (uint64_t)threadIdx.x * S32_E;
    uint32_t v[S32_E], run = 0;
#pragma unroll
    for (int k = 0; k < S32_E; ++k) {
        const uint64_t i = base + k;
        v[k] = i < n ? in[i] + add : 0u;
        run += v[k];
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t incl = run;
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t y = __shfl_up(incl, o);
        if (lane >= o) incl += y;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    uint32_t wbase = 0;
    for (int w = 0; w < wave; ++w) wbase += wsum[w];
    uint32_t pre = wbase + incl - run;               // items of this block in front of this thread's
#pragma unroll
    for (int k = 0; k < S32_E; ++k) {
        const uint64_t i = base + k;
        if (i < n) out[i] = inclusive ? pre + v[k] : pre;
        pre += v[k];
    }
    if (threadIdx.x == S32_T - 1) bsum[blockIdx.x] = pre;
}
__global__ void __launch_bounds__(1024) k_scan32_b(uint32_t *bsum, uint32_t nb) {
    __shared__ uint32_t part[1024];
    const uint32_t t = threadIdx.x, chunk = (nb + 1023u) / 1024u;
    const uint32_t a = t * chunk, b = (a + chunk < nb) ? a + chunk : nb;
    uint32_t sum = 0;
    for (uint32_t i = a; i < b; ++i) sum += bsum[i];
    part[t] = sum;
    __syncthreads();
    for (uint32_t d = 1; d < 1024; d <<= 1) {
        const uint32_t v = (t >= d) ? part[t - d] : 0u;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    uint32_t run = t ? part[t - 1] : 0u;
    for (uint32_t i = a; i < b; ++i) {
        const uint32_t x = bsum[i];
        bsum[i] = run;
        run += x;
    }
}
__global__ void __launch_bounds__(S32_T) k_scan32_c(uint32_t *out, uint64_t n, const uint32_t *bsum) {
    const uint32_t add = bsum[blockIdx.x];
    const uint64_t base = (uint64_t)blockIdx.x * S32_B + (uint64_t)threadIdx.x * S32_E;
    if (add)
#pragma unroll
        for (int k = 0; k < S32_E; ++k)
            if (base + k < n) out[base + k] += add;
}
struct RawEntry { uint16_t checksum, pad; int32_t info; };        // GeneralIndex as the index file holds it (8 bytes)
// bucket i of the file: header slot at start[i] (info = number of valid entries c <= count14[i]), then its slots.
// counts[hv + 1] = c (the array is zero elsewhere: its inclusive scan is bucket_off)
__global__ void __launch_bounds__(BLK) k_raw_counts(const RawEntry *tab, const uint32_t *start, const uint32_t *hv, const uint32_t *cnt14, uint32_t n_b,
                                                   uint64_t n_buckets_all, uint32_t *counts, int *bad) {
    const uint32_t i = blockIdx.x * BLK + threadIdx.x;
    if (i >= n_b) return;
    const int32_t c = tab[start[i]].info;
    if (c < 0 || (uint32_t)c > cnt14[i] || (uint64_t)hv[i] >= n_buckets_all || (i && hv[i] <= hv[i - 1])) {
        atomicOr(bad, 1);
        return;
    }
    counts[hv[i] + 1] = (uint32_t)c;
}
__global__ void __launch_bounds__(BLK) k_raw_scatter(const RawEntry *tab, const uint32_t *start, const uint32_t *hv, uint32_t n_b, const uint32_t *bucket_off,
                                                    uint16_t *checksum, uint32_t *pos) {
    const uint32_t i = blockIdx.x * BLK + threadIdx.x;
    if (i >= n_b) return;
    const uint32_t h = hv[i], w = bucket_off[h], c = bucket_off[h + 1] - w;
    const RawEntry *e = tab + start[i] + 1;
    for (uint32_t k = 0; k < c; ++k) {
        const RawEntry x = e[k];
        checksum[w + k] = x.checksum;
        pos[w + k] = (uint32_t)x.info;
    }
}

__global__ void k_init_state(KCore kc, cm_mapped_read *state, uint8_t *active, int32_t *cat, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * BLK + threadIdx.x;
    if (i >= n) return;
    Core c;
    c.P = kc.P;
    cmc::default_mr(c, state[i]);
    active[i] = 1;
    cat[i] = -1;
}

// ------------------------------------------------------------------ host side
struct Slot {
    bool loaded = false, has_annot = false;
    uint64_t gen = 0;                    // bumped by every (un)load: work prepared against an older content of the slot is stale
    uint32_t *d_desc = nullptr;          // bucket descriptors (cmc::desc_pack), one of idx_allocs; null when switched off
    bool chain_parallel_ok = false;      // see k_chain_heavy: no annotated hop longer than maxIntronLen
    cm_index_view X{};
    cmc::AnnotDev A{};
    std::vector<void *> idx_allocs, ann_allocs;
};

struct ProfRec { hipEvent_t a, b; int cls; };

}  // namespace

struct cm_ctx {
    cm_params P{};
    unsigned id = 0;                          // serial number of the context in this process: names it in teardown diagnostics
    int teardown_errors = 0;
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr;            // heavy work of a stage, concurrent with the light kernel on `stream`
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    // The pair stage of round r runs on its own pair of streams while `stream` / `stream2` already seed and chain round r + 1
    // (cm_map_rounds): seeds and chains are functions of (read, contig) only, the carried state enters in the pair stage.
    hipStream_t stream_p = nullptr, stream_p2 = nullptr, stream_p3 = nullptr;      // p3: the re-run launch of the pair stage (RetryArgs)
    hipStream_t stream_o = nullptr;           // work classes + ordered lists of a pair stage, computed under the pair stage of the item before
    hipEvent_t ev_order[2] = {nullptr, nullptr};      // ... of set b are complete
    // The re-run launch of a pair stage (RetryArgs), decided late: see settle_pair.
    struct Rerun {
        bool deferred = false;
        cmc::KCore core;
        ReadsDev rd;
        uint64_t p0 = 0;
        uint32_t nt = 0;
        cm_chain *chains = nullptr;
        int32_t *nchain = nullptr, *high = nullptr;
        uint8_t *act_out = nullptr;
        int is_last = 0, cap2 = 0;
        size_t lds2 = 0;
        uint32_t *pair_err = nullptr, *retry_list = nullptr;
        unsigned int *retry_ctr = nullptr;
        // the fall-back launch of the heavy-pair pipeline (k_pair_heavy over the pairs that did not fit its arrays), decided late as well
        bool fall = false;
        const uint32_t *fall_list = nullptr;
        unsigned int *fall_ctr = nullptr, *fall_cursor = nullptr;
        size_t lds_heavy = 0;
        int str_cap = 0;
        unsigned fall_grid = 0;
    } rerun[2];
    hipStream_t stream_s = nullptr;           // seeding of the NEXT item, issued while the chain stage of this one runs (map_rounds_issue)
    hipEvent_t ev_seed[2] = {nullptr, nullptr};       // seeds + cell offsets of a seed set are complete
    hipEvent_t ev_first[2] = {nullptr, nullptr};      // an item's two pair kernels are done (set b): its re-run may start
    hipEvent_t ev_join_p = nullptr, ev_prep[2] = {nullptr, nullptr}, ev_pair[2] = {nullptr, nullptr}, ev_tail = nullptr;
    hipEvent_t ev_flags = nullptr;            // main stream: the pair stage two items back is complete (its flags may be read: early seeding)
    bool pair_pending[2] = {false, false};
    // cross-batch prefetch (cm_map_rounds): the staged batch's first round seeded and chained under this batch's last pair stage
    int item_base = 0;                        // items (tile x round) mapped so far: item i uses chain-record set (item_base + i) & 1
    bool pre_launched = false;                // done for the staged batch ...
    bool pre_ready = false;                   // ... which cm_reads_swap has made the current one
    int pre_slot = -1, pre_b = 0;
    uint64_t pre_gen = 0, pre_n = 0;
    uint32_t pre_nt = 0;                      // pairs of the prefetched item (the staged batch's first tile)
    uint8_t *d_ones = nullptr;                // all-active flags of a fresh batch
    uint64_t ones_cap = 0;
    unsigned long long *h_pin = nullptr;          // page-locked landing zone of the scalar read-backs (cell total, error flags, counts)
    uint32_t h_pin_nt = 0;                        // pairs of the tile whose heavy load was last sent to h_pin[4] (0: none yet)
    std::string err = "";
    Slot slots[MAX_SLOTS];
    // reads
    uint64_t n_pairs = 0;
    uint8_t *d_seq1 = nullptr, *d_seq2 = nullptr, *d_seq1_base = nullptr, *d_seq2_base = nullptr;
    uint64_t *d_off1 = nullptr, *d_off2 = nullptr;
    cm_mapped_read *d_state = nullptr;
    uint8_t *d_active = nullptr;              // current flags (valid after the last completed round)
    uint8_t *d_active_b = nullptr;            // the other parity: a round reads one array and writes the other
    int32_t *d_cat = nullptr;
    int n_seeds = 0, max_len = 0;
    // staged batch (cm_reads_stage): a second set of read buffers filled on the copy stream while the resident batch is mapped
    hipStream_t stream_copy = nullptr;
    hipEvent_t ev_staged = nullptr, ev_retired = nullptr;
    bool staged = false;
    uint64_t st_n_pairs = 0;
    uint8_t *st_seq1_base = nullptr, *st_seq2_base = nullptr;
    uint64_t *st_off1 = nullptr, *st_off2 = nullptr;
    cm_mapped_read *st_prior = nullptr;
    bool st_has_prior = false;
    int st_max_len = 0;
    // workspace
    uint32_t tile = 0;
    uint32_t *d_sstart = nullptr, *d_scnt = nullptr, *d_sraw = nullptr, *d_cells = nullptr;
    unsigned long long *d_celloff = nullptr, *d_bsum = nullptr;
    unsigned int *d_bmax = nullptr;
    // second seed set (SeedBufs): item i + 1 is seeded into it while the chain stage of item i reads the first, and vice versa
    uint32_t *d_sstart_b = nullptr, *d_scnt_b = nullptr, *d_sraw_b = nullptr;
    unsigned long long *d_celloff_b = nullptr, *d_bsum_b = nullptr;
    unsigned int *d_bmax_b = nullptr;
    unsigned int *d_cctr_b = nullptr, *d_cblk_b = nullptr;
    int8_t *d_cls4_b = nullptr;
    int32_t *d_thigh = nullptr, *d_thigh_b = nullptr;
    bool cls_deferred[2] = {false, false};      // seed set s: classes made before the chain records were free, k_chain_apply still to run
    uint32_t *d_perm4_b = nullptr;
    double *d_dpscore = nullptr;
    int32_t *d_dpprev = nullptr;
    unsigned long long cells_cap = 0;
    cm_chain *d_chains = nullptr;             // chains of a round: two sets, the pair stage of round r reads set r & 1 while
    int32_t *d_nchain = nullptr, *d_high = nullptr;       // the chaining of round r + 1 fills the other
    cm_chain *d_chains_b = nullptr;
    int32_t *d_nchain_b = nullptr, *d_high_b = nullptr;
    uint16_t *d_resid_b = nullptr;
    unsigned int *d_cctr = nullptr, *d_cblk = nullptr;    // class counters / block histograms of the CHAIN stage's sort (own copies: the
                                                          // pair stage of the previous round uses d_cls_ctr / d_blk_cnt at the same time)
    unsigned long long *d_lane_clk = nullptr;     // diagnostic build of the timing study only
    int8_t *d_cls = nullptr, *d_cls4 = nullptr, *d_cls_sub = nullptr, *d_cls_sub2 = nullptr;
    uint32_t *d_perm1 = nullptr, *d_perm0 = nullptr;
    unsigned int *d_cls_ctr2 = nullptr, *d_cls_ctr3 = nullptr;
    uint32_t *d_perm4 = nullptr;
    uint16_t *d_resid = nullptr;
    uint32_t *d_perm = nullptr;
    unsigned int *d_cls_ctr = nullptr, *d_blk_cnt = nullptr;
    unsigned long long *d_collect_idx = nullptr;
    cm_mapped_read *d_collect_st = nullptr;
    cm_record *d_collect_rec = nullptr;
    uint64_t collect_rec_cap = 0;
    std::unordered_map<const void *, size_t> caps;   // bytes behind each grow-only per-batch buffer, keyed by the pointer field (ensure())
    uint64_t collect_cap = 0;
    int8_t *d_col_cls = nullptr;
    uint32_t *d_col_perm = nullptr;
    unsigned int *d_col_blk = nullptr, *d_col_ctr = nullptr;
    uint32_t *d_hlist = nullptr;
    HRes *d_hres = nullptr;          // task outcomes of k_pair_heavy: 64 per resident block
    // the heavy pairs as a pipeline of kernels (cm_heavy_pipe.h)
    HPair *d_hp = nullptr;
    uint32_t *d_hp_list2 = nullptr, *d_hp_fall = nullptr, *d_hp_q = nullptr, *d_hp_q2 = nullptr;
    HTask *d_hp_T = nullptr;
    int8_t *d_hp_tcls = nullptr;               // work class of every task, the order k_hp_tasks walks them in, the counting sort's scratch
    uint32_t *d_hp_tperm = nullptr;
    unsigned int *d_hp_tblk = nullptr, *d_hp_tctr = nullptr;
    HUnp *d_hp_U = nullptr;
    cmc::PreDP *d_hp_pre = nullptr, *d_hp_pre2 = nullptr;
    HRes *d_hp_res = nullptr;
    uint16_t *d_hp_lists = nullptr;
    unsigned int *d_hp_ctr = nullptr;
    uint32_t hp_tasks_cap = 0, hp_unp_cap = 0;
    unsigned int *d_hp_fallctr = nullptr;      // [set]
    unsigned long long *d_type_hist = nullptr;
    unsigned int *d_retry_ctr = nullptr;                          // [set][count, cursor]
    unsigned long long *d_heavy_load = nullptr;                   // cost beyond HEAVY_COST summed over the tile in the pair stage (k_pair_cost)
    uint32_t *d_pair_err = nullptr, *d_retry_list = nullptr;      // per-pair capacity flags of a tile (zero between launches), pairs to re-run (RetryArgs)
    cmc::MemoSpill *d_spill = nullptr;                            // RETRY_GRID x 64 lanes x RETRY_SPILL overflow entries of the extension memo
    uint8_t *d_pool = nullptr;
    unsigned long long pool_bytes = 0;
    unsigned long long *d_pool_cursor = nullptr;
    int *d_err = nullptr;
    unsigned long long *d_counters = nullptr;
    std::vector<unsigned long long> h_celloff;
    // profiling
    bool prof = false;
    std::vector<ProfRec> recs;
    std::vector<hipEvent_t> ev_free;           // timing events are recycled: creating them is slow and comes in bursts
    double ms[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    uint64_t launches[8] = {0, 0, 0, 0, 0, 0, 0, 0};
};

namespace {

int fail(cm_ctx *ctx, int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (ctx) ctx->err = buf;
    return code;
}
#define HIPCHK(ctx, call)                                                                             \
    do {                                                                                              \
        hipError_t e_ = (call);                                                                       \
        if (e_ != hipSuccess) return fail(ctx, (e_ == hipErrorOutOfMemory) ? CM_ENOMEM : CM_EHIP, "%s: %s", #call, hipGetErrorString(e_)); \
    } while (0)

// A HIP status nobody can return to the caller (teardown, frees inside grow-only buffers).  An asynchronous error -- a fault of a
// kernel this context launched -- is sticky and would otherwise first show as an abort inside whatever context runs next;
// reported here it carries the id of the context that raised it.  The first few per context go to stderr.
void report_hip(cm_ctx *ctx, const char *what, hipError_t e) {
    if (e == hipSuccess) return;
    const int k = ctx ? ctx->teardown_errors++ : 0;
    if (k < 8) fprintf(stderr, "[cmhot] context #%u: %s failed: %s (%d)\n", ctx ? ctx->id : 0u, what, hipGetErrorString(e), (int)e);
}

template <class T>
int up(cm_ctx *ctx, std::vector<void *> &allocs, const T *host, size_t n, const T **dev) {
    void *d = nullptr;
    size_t bytes = (n ? n : 1) * sizeof(T);
    HIPCHK(ctx, hipMalloc(&d, bytes));
    allocs.push_back(d);
    if (n) HIPCHK(ctx, hipMemcpyAsync(d, host, n * sizeof(T), hipMemcpyHostToDevice, ctx->stream));
    *dev = (const T *)d;
    return CM_OK;
}
void free_all(cm_ctx *ctx, std::vector<void *> &v) {
    for (void *p : v) report_hip(ctx, "hipFree", hipFree(p));
    v.clear();
}
template <class T>
void dfree(cm_ctx *ctx, T *&p) {
    if (p) report_hip(ctx, "hipFree", hipFree((void *)p));
    p = nullptr;
}

// Grow-only per-batch buffers: a batch re-uses the previous batch's allocation when it is large enough.  hipFree + hipMalloc
// of the multi-GB workspaces cost ~0.85 s per 1 M-pair batch on MI355X, 85x the mapping itself (tests/diag/upload_rate.py).
template <class T>
hipError_t ensure(cm_ctx *c, T *&p, size_t bytes) {
    size_t &cap = c->caps[(const void *)&p];
    if (p && cap >= bytes) return hipSuccess;
    dfree(c, p);
    cap = 0;
    const hipError_t e = hipMalloc((void **)&p, bytes ? bytes : 1);
    if (e == hipSuccess) cap = bytes;
    return e;
}

void free_reads(cm_ctx *c) {
    dfree(c, c->d_seq1_base); dfree(c, c->d_seq2_base); c->d_seq1 = c->d_seq2 = nullptr; dfree(c, c->d_off1); dfree(c, c->d_off2);
    dfree(c, c->d_state); dfree(c, c->d_active); dfree(c, c->d_active_b); dfree(c, c->d_cat);
    dfree(c, c->d_chains_b); dfree(c, c->d_nchain_b); dfree(c, c->d_high_b); dfree(c, c->d_resid_b); dfree(c, c->d_cctr); dfree(c, c->d_cblk);
    dfree(c, c->d_sstart); dfree(c, c->d_scnt); dfree(c, c->d_sraw); dfree(c, c->d_cells); dfree(c, c->d_celloff); dfree(c, c->d_bsum); dfree(c, c->d_bmax);
    dfree(c, c->d_sstart_b); dfree(c, c->d_scnt_b); dfree(c, c->d_sraw_b); dfree(c, c->d_celloff_b); dfree(c, c->d_bsum_b); dfree(c, c->d_bmax_b);
    dfree(c, c->d_cctr_b); dfree(c, c->d_cblk_b); dfree(c, c->d_cls4_b); dfree(c, c->d_perm4_b); dfree(c, c->d_thigh); dfree(c, c->d_thigh_b);
    dfree(c, c->d_dpscore); dfree(c, c->d_dpprev); dfree(c, c->d_chains); dfree(c, c->d_nchain); dfree(c, c->d_high);
    dfree(c, c->d_pool); dfree(c, c->d_lane_clk); dfree(c, c->d_cls); dfree(c, c->d_cls4); dfree(c, c->d_perm4); dfree(c, c->d_resid); dfree(c, c->d_perm); dfree(c, c->d_cls_ctr); dfree(c, c->d_cls_ctr2); dfree(c, c->d_cls_sub); dfree(c, c->d_perm1); dfree(c, c->d_cls_ctr3); dfree(c, c->d_cls_sub2); dfree(c, c->d_perm0); dfree(c, c->d_blk_cnt); dfree(c, c->d_hlist); dfree(c, c->d_hres);
    dfree(c, c->d_hp); dfree(c, c->d_hp_list2); dfree(c, c->d_hp_fall); dfree(c, c->d_hp_fallctr); dfree(c, c->d_hp_q); dfree(c, c->d_hp_q2); dfree(c, c->d_hp_T); dfree(c, c->d_hp_tcls); dfree(c, c->d_hp_tperm); dfree(c, c->d_hp_tblk); dfree(c, c->d_hp_tctr); dfree(c, c->d_hp_U);
    dfree(c, c->d_hp_pre); dfree(c, c->d_hp_pre2); dfree(c, c->d_hp_res); dfree(c, c->d_hp_lists); dfree(c, c->d_hp_ctr);
    dfree(c, c->d_pair_err); dfree(c, c->d_retry_list); dfree(c, c->d_spill); dfree(c, c->d_type_hist); dfree(c, c->d_retry_ctr); dfree(c, c->d_heavy_load);
    dfree(c, c->d_col_cls); dfree(c, c->d_col_perm); dfree(c, c->d_col_blk); dfree(c, c->d_col_ctr);
    if (c->stream_copy) report_hip(c, "hipStreamSynchronize(copy stream)", hipStreamSynchronize(c->stream_copy));
    dfree(c, c->st_seq1_base); dfree(c, c->st_seq2_base); dfree(c, c->st_off1); dfree(c, c->st_off2); dfree(c, c->st_prior);
    dfree(c, c->d_ones);
    c->ones_cap = 0;
    c->pre_launched = c->pre_ready = false;
    c->staged = false;
    c->st_n_pairs = 0;
    c->n_pairs = 0;
    c->tile = 0;
}

static hipEvent_t take_event(cm_ctx *c) {
    hipEvent_t e = nullptr;
    if (!c->ev_free.empty()) {
        e = c->ev_free.back();
        c->ev_free.pop_back();
    } else (void)hipEventCreate(&e);
    return e;
}
struct Timer {
    cm_ctx *c;
    int cls;
    ProfRec r{};
    bool on;
    hipStream_t st;
    Timer(cm_ctx *ctx, int k, hipStream_t stream = nullptr) : c(ctx), cls(k), on(ctx->prof), st(stream ? stream : ctx->stream) {
        if (on) {
            r.a = take_event(c);
            r.b = take_event(c);
            r.cls = cls;
            (void)hipEventRecord(r.a, st);
        }
    }
    ~Timer() {
        if (on) {
            (void)hipEventRecord(r.b, st);
            c->recs.push_back(r);
        }
    }
};

KCore make_core(const cm_ctx *c, const Slot &s) {
    KCore k;
    k.P = c->P;
    k.X = s.X;
    k.A = s.A;
    k.desc = s.d_desc;
    return k;
}

int check_slot(cm_ctx *ctx, int slot, bool need_annot) {
    if (slot < 0 || slot >= MAX_SLOTS) return fail(ctx, CM_EINVAL, "slot %d out of range", slot);
    if (!ctx->slots[slot].loaded) return fail(ctx, CM_ESTATE, "contig slot %d not loaded", slot);
    if (need_annot && !ctx->slots[slot].has_annot) return fail(ctx, CM_ESTATE, "annotation of slot %d not loaded", slot);
    return CM_OK;
}

// seeds (+ optionally chains) of one tile; leaves results in the workspace
struct RoundBufs { cm_chain *chains; int32_t *nchain, *high; uint16_t *resid; };
RoundBufs round_bufs(cm_ctx *c, int b) {
    return b ? RoundBufs{c->d_chains_b, c->d_nchain_b, c->d_high_b, c->d_resid_b} : RoundBufs{c->d_chains, c->d_nchain, c->d_high, c->d_resid};
}

ReadsDev current_reads(const cm_ctx *ctx) { return ReadsDev{ctx->d_seq1, ctx->d_seq2, ctx->d_off1, ctx->d_off2}; }

// What the seeding of a tile leaves for its chain stage: the seed ranges of every probe, the DP-cell offsets of every chaining
// problem (+ total and largest problem, also copied to h_pin[8 + 2 s ..]).  Two sets: see map_rounds_issue.
// The work classes of the tile's chaining problems and their sorted list (k_chain_cls + counting sort), the zeroed cursors of the
// chain kernels (improvement-log pool, heavy work list) -- everything the chain kernels need but the chain records themselves.
struct SeedBufs {
    uint32_t *sstart, *scnt, *sraw;
    unsigned long long *celloff, *bsum;
    unsigned int *bmax;
    int8_t *cls4;
    unsigned int *cblk, *cctr;
    uint32_t *perm4;
    unsigned long long *pool_cursor;
    int32_t *thigh;              // high_hits per problem while the chain records are not free yet (k_chain_apply)
};
SeedBufs seed_bufs(cm_ctx *c, int s) {
    return s ? SeedBufs{c->d_sstart_b, c->d_scnt_b, c->d_sraw_b, c->d_celloff_b, c->d_bsum_b, c->d_bmax_b, c->d_cls4_b, c->d_cblk_b, c->d_cctr_b, c->d_perm4_b,
                        c->d_pool_cursor + 1, c->d_thigh_b}
             : SeedBufs{c->d_sstart, c->d_scnt, c->d_sraw, c->d_celloff, c->d_bsum, c->d_bmax, c->d_cls4, c->d_cblk, c->d_cctr, c->d_perm4, c->d_pool_cursor, c->d_thigh};
}
constexpr unsigned HP_PLAN_GRID = 2048;      // workgroups of k_hp_plan (each with its own task-list scratch)
// CM_HEAVY_PIPELINE=0: the heavy pairs of a tile through k_pair_heavy alone (the round-3 path)
static bool heavy_pipeline() {
    static const bool v = !(getenv("CM_HEAVY_PIPELINE") && getenv("CM_HEAVY_PIPELINE")[0] == '0');
    return v;
}
static unsigned long long chain_light_w() {
    static const unsigned long long v = getenv("CM_CHAIN_LIGHT_W") ? strtoull(getenv("CM_CHAIN_LIGHT_W"), nullptr, 10) : 256ull;
    return v;
}
static unsigned int chain_light_cells() {
    static const unsigned int v = getenv("CM_CHAIN_LIGHT_CELLS") ? (unsigned)atoi(getenv("CM_CHAIN_LIGHT_CELLS")) : 96u;
    return v;
}

// seeds of one tile into seed set s, on stream st: k_seed, the scan of the problems' DP cells, the two totals to the host; with
// rb (the chain records the tile's chain stage will fill: their per-problem counters are initialised here) also the work
// classes + sorted lists of the chaining problems and the zeroed cursors.  ev_seed[s] is recorded behind them.
// The second half of run_seed_tile: the work classes + sorted lists of the chaining problems of seed set s, the per-problem counters of
// the chain records rb and the zeroed cursors.  (On its own when the seeds were computed before the chain records were free: the
// cross-batch prefetch.)
static int seed_classes(cm_ctx *ctx, uint64_t pair0, uint32_t n_tile, const uint8_t *act, int s, hipStream_t st, const RoundBufs *rb, bool defer = false) {
    const int S = ctx->n_seeds;
    const uint32_t n_prob = n_tile * 4u;
    if ((uint64_t)n_prob * (uint64_t)S == 0) return CM_OK;
    const SeedBufs sb = seed_bufs(ctx, s);
    // Light problems: one lane each, index order.  Heavy problems (many hits): one wave each (k_chain_heavy), heaviest class
    // first.  (Used by run_chain_tile when it takes the split path; computed here because these five small launches, queued
    // behind the persistent pair kernels of the previous item, took 6 ms of the chain stage's critical path.)
    Timer t(ctx, 5, st);
    const uint32_t nbk = (n_prob + CLS_T - 1) / CLS_T;
    // defer: the chain records rb are still read by an earlier pair stage -- nothing is written into them here, run_chain_tile does that
    // (k_chain_apply) when it is ordered behind that stage
    hipLaunchKernelGGL(k_chain_cls, dim3((n_prob + BLK - 1) / BLK), dim3(BLK), 0, st, sb.scnt, sb.sraw, S, n_prob, sb.cls4, defer ? sb.thigh : rb->high,
                       chain_light_w(), chain_light_cells(), defer ? (int32_t *)nullptr : rb->nchain, defer ? (uint16_t *)nullptr : rb->resid, act, pair0);
    ctx->cls_deferred[s] = defer;
    hipLaunchKernelGGL(k_cls_hist, dim3(nbk), dim3(CLS_W), 0, st, sb.cls4, n_prob, sb.cblk, nbk, (const uint32_t *)nullptr,
                       (const unsigned int *)nullptr);
    hipLaunchKernelGGL(k_cls_scan, dim3(1), dim3(SCAN_CLS_T), 0, st, sb.cblk, nbk, sb.cctr, -1, N_CLS);
    hipLaunchKernelGGL(k_cls_place, dim3(nbk), dim3(CLS_W), 0, st, sb.cls4, n_prob, sb.cblk, nbk, sb.cctr, sb.perm4,
                       (uint32_t *)nullptr, (const uint32_t *)nullptr, (const unsigned int *)nullptr);
    ctx->launches[5] += 4;
    HIPCHK(ctx, hipMemsetAsync(sb.pool_cursor, 0, sizeof(unsigned long long), st));
    HIPCHK(ctx, hipMemsetAsync(sb.cctr + 48, 0, sizeof(unsigned int), st));       // spare word of the class counters: work cursor of k_chain_heavy
    HIPCHK(ctx, hipGetLastError());
    return CM_OK;
}
int run_seed_tile(cm_ctx *ctx, const KCore &core, const ReadsDev &rd, uint64_t pair0, uint32_t n_tile, const uint8_t *act, int s, hipStream_t st,
                  const RoundBufs *rb, bool defer_rb = false) {
    const int S = ctx->n_seeds;
    const uint64_t total = (uint64_t)n_tile * 4u * (uint64_t)S;
    if (total == 0) return CM_OK;
    const SeedBufs sb = seed_bufs(ctx, s);
    {
        Timer t(ctx, 0, st);
        hipLaunchKernelGGL(k_seed, dim3((unsigned)((total + BLK - 1) / BLK)), dim3(BLK), 0, st, core, rd, act, pair0, n_tile, S,
                           sb.sstart, sb.scnt, sb.sraw, ctx->d_counters);
        ++ctx->launches[0];
    }
    const uint32_t n_prob = n_tile * 4u;
    {
        Timer t(ctx, 3, st);
        const uint32_t nb = (n_prob + SCAN_ELEMS - 1) / SCAN_ELEMS;
        hipLaunchKernelGGL(k_scan_a, dim3(nb), dim3(SCAN_T), 0, st, sb.scnt, S, n_prob, sb.celloff, sb.bsum, sb.bmax);
        hipLaunchKernelGGL(k_scan_b, dim3(1), dim3(1024), 0, st, sb.bsum, nb, sb.celloff + n_prob, (const unsigned int *)sb.bmax);
        hipLaunchKernelGGL(k_scan_c, dim3((n_prob + SCAN_T - 1) / SCAN_T), dim3(SCAN_T), 0, st, sb.celloff, sb.bsum, n_prob);
        ctx->launches[3] += 3;
    }
    HIPCHK(ctx, hipMemcpyAsync(ctx->h_pin + 8 + 2 * s, sb.celloff + n_prob, 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
    if (rb) {
        const int rc = seed_classes(ctx, pair0, n_tile, act, s, st, rb, defer_rb);
        if (rc) return rc;
    }
    HIPCHK(ctx, hipEventRecord(ctx->ev_seed[s], st));
    HIPCHK(ctx, hipGetLastError());
    return CM_OK;
}

// chains of one tile from seed set s (run_seed_tile) into the chain records rb.  `after_launch` (may be empty) is called once, when
// the chain kernels of the tile have been launched and before the host waits for them.
int run_chain_tile(cm_ctx *ctx, const KCore &core, const ReadsDev &rd, uint64_t pair0, uint32_t n_tile, bool parallel_ok, const uint8_t *act,
                   const RoundBufs &rb, int s, const std::function<int()> &after_launch = {}) {
    const int S = ctx->n_seeds;
    const uint32_t n_prob = n_tile * 4u;
    if (n_prob == 0 || S == 0) return after_launch ? after_launch() : CM_OK;
    const SeedBufs sb = seed_bufs(ctx, s);
    HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_seed[s], 0));
    HIPCHK(ctx, hipEventSynchronize(ctx->ev_seed[s]));
    if (ctx->cls_deferred[s]) {           // classes made while the chain records were in use (the caller has ordered this stream behind that)
        hipLaunchKernelGGL(k_chain_apply, dim3((n_prob + BLK - 1) / BLK), dim3(BLK), 0, ctx->stream, n_prob, (const int8_t *)sb.cls4, (const int32_t *)sb.thigh,
                           rb.high, rb.nchain, rb.resid);
        ctx->cls_deferred[s] = false;
    }
    const unsigned long long total = ctx->h_pin[8 + 2 * s];
    const unsigned long long max_cells = ctx->h_pin[9 + 2 * s];       // of one problem
    // problem ranges whose DP cells fit the workspace
    std::vector<std::pair<uint32_t, uint32_t>> ranges;
    if (total <= ctx->cells_cap) {
        ranges.push_back({0u, n_prob});
    } else {
        ctx->h_celloff.resize((size_t)n_prob + 1);
        HIPCHK(ctx, hipMemcpyAsync(ctx->h_celloff.data(), sb.celloff, ((size_t)n_prob + 1) * sizeof(unsigned long long), hipMemcpyDeviceToHost,
                                   ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        uint32_t a = 0;
        while (a < n_prob) {
            uint32_t b = a;
            while (b < n_prob && ctx->h_celloff[b + 1] - ctx->h_celloff[a] <= ctx->cells_cap) ++b;
            if (b == a) return fail(ctx, CM_ELIMIT, "one chaining problem needs %llu DP cells (> workspace %llu)",
                                    ctx->h_celloff[a + 1] - ctx->h_celloff[a], ctx->cells_cap);
            ranges.push_back({a, b});
            a = b;
        }
    }
    // CM_CHAIN_SPLIT=0 keeps everything on the sequential kernel (the class lists of run_seed_tile go unused).
    static const char *split_env = getenv("CM_CHAIN_SPLIT");
    // k_chain_heavy keeps a problem's hit positions in LDS: sized for the largest problem of this tile (a multiple of 2 KB, so
    // that launches of similar tiles share a configuration), not for the n_seeds x seed_lim a problem could have in theory --
    // the kernel waits on memory most of the time and the LDS request decides how many waves a CU holds.
    // (+ behind the hits: one bit per cell and a queue of 320 hits, see the kernel's upper_bound pass)
    const size_t lds_extra = ((size_t)max_cells / 64 + 2) * 8 + 640;
    const size_t heavy_lds = std::min<size_t>((size_t)S * (size_t)ctx->P.seed_lim * sizeof(uint32_t),
                                              ((size_t)max_cells * sizeof(uint32_t) + 2047) / 2048 * 2048 + 2048) + (lds_extra + 255) / 256 * 256;
    const bool split = !(split_env && split_env[0] == '0') && ranges.size() == 1 && parallel_ok && heavy_lds <= 152u * 1024u;
    bool fresh = true;            // the cursors are still as run_seed_tile zeroed them
    // one launch group over problems [a, b) whose DP cells start at `base`: the improvement log of every problem comes out of the
    // shared pool, whose cursor starts at 0 for every group
    auto launch_group = [&](uint32_t a, uint32_t b, unsigned long long base, bool use_split) -> int {
        const uint32_t n = b - a;
        if (!fresh) {
            HIPCHK(ctx, hipMemsetAsync(sb.pool_cursor, 0, sizeof(unsigned long long), ctx->stream));
            HIPCHK(ctx, hipMemsetAsync(sb.cctr + 48, 0, sizeof(unsigned int), ctx->stream));       // spare word of the class counters: work cursor
        }
        fresh = false;
        if (use_split) {          // the few long problems run on the second stream, concurrently with the bulk
            HIPCHK(ctx, hipEventRecord(ctx->ev_fork, ctx->stream));
            HIPCHK(ctx, hipStreamWaitEvent(ctx->stream2, ctx->ev_fork, 0));
            {
            Timer t(ctx, 6, ctx->stream2);
            // (a property of the function, process-wide: only ever raised, so a launch in flight from another context of this
            // process never sees its limit lowered)
            static std::atomic<size_t> heavy_attr{64u * 1024u};
            if (heavy_lds > heavy_attr.load()) {
                HIPCHK(ctx, hipFuncSetAttribute((const void *)k_chain_heavy, hipFuncAttributeMaxDynamicSharedMemorySize, (int)heavy_lds));
                heavy_attr.store(heavy_lds);
            }
            const uint32_t hb = n < 8192u ? n : 8192u;
            static const size_t heavy_pad = getenv("CM_CHEAVY_LDS_PAD") ? (size_t)atoi(getenv("CM_CHEAVY_LDS_PAD")) : 0;     // occupancy experiment
            hipLaunchKernelGGL(k_chain_heavy, dim3(hb), dim3(64), heavy_lds + heavy_pad, ctx->stream2, core, rd, pair0, S, sb.sstart, sb.scnt, sb.celloff,
                               ctx->d_dpscore, ctx->d_dpprev, ctx->d_pool, ctx->pool_bytes, sb.pool_cursor, rb.chains, rb.nchain, ctx->d_err,
                               rb.resid, sb.perm4, sb.cctr + CTR_BASE + CHAIN_LIGHT_CLS - 1, sb.cctr + 48, ctx->d_counters);
            ++ctx->launches[6];
            }
            HIPCHK(ctx, hipEventRecord(ctx->ev_join, ctx->stream2));
        }
        Timer t(ctx, 1);
        static const size_t light_pad = getenv("CM_CHAIN_LDS_PAD") ? (size_t)atoi(getenv("CM_CHAIN_LDS_PAD")) : 0;       // occupancy experiment
        hipLaunchKernelGGL(k_chain, dim3((n + BLK_CHAIN - 1) / BLK_CHAIN), dim3(BLK_CHAIN), light_pad, ctx->stream, core, rd, act, pair0, a, b, S,
                           sb.sstart, sb.scnt, sb.sraw, sb.celloff, base, ctx->d_dpscore, ctx->d_dpprev, ctx->d_pool,
                           ctx->pool_bytes, sb.pool_cursor, rb.chains, rb.nchain, rb.high, ctx->d_err, rb.resid,
                           use_split ? sb.perm4 : (const uint32_t *)nullptr, sb.cctr + CTR_BASE + CHAIN_LIGHT_CLS - 1, sb.cctr + CTR_SUM);
        if (use_split) HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0));       // timed with the light kernel: the stage ends here
        ++ctx->launches[1];
        HIPCHK(ctx, hipGetLastError());
        return CM_OK;
    };
    // did the group run out of log space?  (one small read-back per group; the pair stage must not start on truncated logs)
    auto pool_lost = [&](bool *lost) -> int {
        HIPCHK(ctx, hipMemcpyAsync(ctx->h_pin + 2, ctx->d_err, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        const int e = *(const int *)(ctx->h_pin + 2);
        *lost = (e & cmc::ERR_POOL) != 0;
        if (*lost) {                                          // the other flags stay for cm_sync to report (the pair stage of the
            hipLaunchKernelGGL(k_err_clear, dim3(1), dim3(64), 0, ctx->stream, ctx->d_err, (int)cmc::ERR_POOL);   // previous round may be setting some right now)
            HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        }
        return CM_OK;
    };
    const unsigned long long pool_max = getenv("CM_POOL_MAX") ? strtoull(getenv("CM_POOL_MAX"), nullptr, 10) : (48ull << 30);
    bool told = false;
    for (auto &rg : ranges) {
        unsigned long long base = 0;
        if (rg.first != 0) base = ctx->h_celloff[rg.first];
        int rc = launch_group(rg.first, rg.second, base, split);
        if (rc) return rc;
        if (!told && after_launch && (rc = after_launch())) return rc;
        told = true;
        bool lost = false;
        if ((rc = pool_lost(&lost))) return rc;
        // The reference's score2chain has no capacity limit.  When the log pool ran out: first a larger pool (x4 up to pool_max)
        // and the same group again; then the group in halves on the one-lane-per-problem kernel, every piece with the whole pool
        // to itself.  Every retry recomputes its problems from the seeds, so nothing of the failed attempt survives.
        while (lost && ctx->pool_bytes < pool_max) {
            unsigned long long want = ctx->pool_bytes * 4ull;
            if (want > pool_max) want = pool_max;
            HIPCHK(ctx, hipStreamSynchronize(ctx->stream2));
            if (ensure(ctx, ctx->d_pool, want) != hipSuccess) {            // not enough HBM for a larger pool: keep the old size, go to the split path
                (void)hipGetLastError();
                HIPCHK(ctx, ensure(ctx, ctx->d_pool, ctx->pool_bytes));
                break;
            }
            ctx->pool_bytes = want;
            if ((rc = launch_group(rg.first, rg.second, base, split))) return rc;
            if ((rc = pool_lost(&lost))) return rc;
        }
        if (lost) {
            std::vector<std::pair<uint32_t, uint32_t>> todo{{rg.first, rg.second}};
            while (!todo.empty()) {
                const auto pc = todo.back();
                todo.pop_back();
                if ((rc = launch_group(pc.first, pc.second, base, false))) return rc;
                if ((rc = pool_lost(&lost))) return rc;
                if (!lost) continue;
                if (pc.second - pc.first <= 1)
                    return fail(ctx, CM_ELIMIT, "one chaining problem's improvement log does not fit %llu bytes", ctx->pool_bytes);
                const uint32_t mid = pc.first + (pc.second - pc.first) / 2;
                todo.push_back({mid, pc.second});
                todo.push_back({pc.first, mid});
            }
        }
    }
    return CM_OK;
}

int check_dev_err(cm_ctx *ctx) {
    HIPCHK(ctx, hipMemcpyAsync(ctx->h_pin, ctx->d_err, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    const int e = *(const int *)ctx->h_pin;
    if (e) {
        (void)hipMemsetAsync(ctx->d_err, 0, sizeof(int), ctx->stream);       // reported once: the next batch starts clean
        return fail(ctx, CM_ELIMIT, "device capacity limit hit:%s%s%s", (e & cmc::ERR_POOL) ? " chain improvement-log pool exhausted;" : "",
                    (e & (cmc::ERR_SEEDS | cmc::ERR_BAND)) ? " DP string longer than the staging buffer;" : "",
                    (e & cmc::ERR_MEMO) ? " extension memo overflowed where the reference's memo would have served a colliding key" : "");
    }
    return CM_OK;
}

}  // namespace

// ================================================================== C-ABI
extern "C" {

int cm_create(const cm_params *p, cm_ctx **out) {
    if (!p || !out) return CM_EINVAL;
    *out = nullptr;
    const int c = p->kmer - CM_WINDOW_SIZE;
    if (p->kmer < CM_WINDOW_SIZE || c > 8) return CM_EINVAL;
    if (p->max_chain_len < 1 || p->max_chain_len > CM_BESTCHAINLIM) return CM_EINVAL;   // chain.h:14-17 fixed array
    if (p->band < 0 || p->band > cmc::MAX_BAND) return CM_EINVAL;
    if (p->seed_lim < 1 || p->seed_lim > 65535) return CM_EINVAL;
    if (p->max_ed < 0 || p->max_sc < 0 || p->max_read_len < p->kmer) return CM_EINVAL;
    // Five streams of one context work at the same time (seed / chain, heavy chains, pair stage, heavy pairs, H2D staging); the
    // runtime's default of 4 hardware queues makes two of them share one and serialises them.  Only effective when this is the
    // first HIP call of the process; callers that bring up HIP earlier (PyTorch) set the variable themselves (bench.py does).
    // When the variable was not set (or set lower) on entry and something else of this process -- PyTorch, an RCCL communicator --
    // has already brought HIP up, setting it now changes nothing and the schedule runs on the default 4 hardware queues: said
    // once on stderr (CM_QUIET=1 silences it) and left in cm_last_error() of the new context, instead of silently running slower.
    const char *hwq = getenv("GPU_MAX_HW_QUEUES");
    const bool hwq_low = !hwq || atoi(hwq) < 16;
    setenv("GPU_MAX_HW_QUEUES", "16", 0);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return CM_ENODEV;
    if (p->device < 0 || p->device >= ndev) return CM_ENODEV;
    if (hipSetDevice(p->device) != hipSuccess) return CM_ENODEV;
    cm_ctx *ctx = new cm_ctx();
    ctx->P = *p;
    {
        static std::atomic<unsigned> serial{0};
        ctx->id = ++serial;
    }
    // CM_STREAM_PRIO (tuning knob): bit 0 = seeding and pair-ordering streams at the highest priority, bit 1 = the heavy-pair stream too
    static const int prio_mask = getenv("CM_STREAM_PRIO") ? atoi(getenv("CM_STREAM_PRIO")) : 0;
    int prio_lo = 0, prio_hi = 0;
    if (prio_mask) (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
    // CM_STREAM_LOW: bit 0 = the light pair kernel's stream at the lowest priority, bit 1 = the heavy chaining stream.  Default 1: the light
    // kernel's persistent waves are the filler of a pair stage, and the short kernels of the heavy pairs' pipeline and of the next item's
    // seeding / chaining get their workgroups placed first (hg38-like step 74.3 -> 70.9 ms; 72.3 with the pipeline's stream raised
    // instead, 76.9 with both, 72.9 with the heavy chaining stream lowered).
    static const int low_mask = getenv("CM_STREAM_LOW") ? atoi(getenv("CM_STREAM_LOW")) : 1;
    if (low_mask && !prio_mask) (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
    auto mk_stream = [&](hipStream_t *st, bool high, bool low = false) {
        return high ? hipStreamCreateWithPriority(st, hipStreamNonBlocking, prio_hi)
                    : low ? hipStreamCreateWithPriority(st, hipStreamNonBlocking, prio_lo) : hipStreamCreateWithFlags(st, hipStreamNonBlocking);
    };
    // (CM_STREAM_PRIO bit 2 = the heavy chaining stream, bit 3 = the main stream)
    if (((prio_mask & 8) ? hipStreamCreateWithPriority(&ctx->stream, hipStreamDefault, prio_hi) : hipStreamCreate(&ctx->stream)) != hipSuccess ||
        ((low_mask & 2)    ? hipStreamCreateWithPriority(&ctx->stream2, hipStreamDefault, prio_lo)
         : (prio_mask & 4) ? hipStreamCreateWithPriority(&ctx->stream2, hipStreamDefault, prio_hi)
                           : hipStreamCreate(&ctx->stream2)) != hipSuccess ||
        hipStreamCreateWithFlags(&ctx->stream_copy, hipStreamNonBlocking) != hipSuccess ||
        mk_stream(&ctx->stream_p, false, (low_mask & 1) != 0) != hipSuccess ||
        mk_stream(&ctx->stream_p2, (prio_mask & 2) != 0) != hipSuccess ||
        hipStreamCreateWithFlags(&ctx->stream_p3, hipStreamNonBlocking) != hipSuccess ||
        mk_stream(&ctx->stream_o, (prio_mask & 1) != 0) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_order[0], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_order[1], hipEventDisableTiming) != hipSuccess ||
        mk_stream(&ctx->stream_s, (prio_mask & 1) != 0) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_seed[0], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_seed[1], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_first[0], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_first[1], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_join_p, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_prep[0], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_prep[1], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_pair[0], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_pair[1], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_tail, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_flags, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_staged, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_retired, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming) != hipSuccess) {
        cm_destroy(ctx);
        return CM_EHIP;
    }
    if (hipMalloc((void **)&ctx->d_pool_cursor, 2 * sizeof(unsigned long long)) != hipSuccess ||
        hipMalloc((void **)&ctx->d_err, sizeof(int)) != hipSuccess ||
        hipMalloc((void **)&ctx->d_counters, 32 * sizeof(unsigned long long)) != hipSuccess ||
        hipHostMalloc((void **)&ctx->h_pin, 128, hipHostMallocDefault) != hipSuccess) {
        cm_destroy(ctx);
        return CM_ENOMEM;
    }
    memset(ctx->h_pin, 0, 128);
    (void)hipMemsetAsync(ctx->d_err, 0, sizeof(int), ctx->stream);
    (void)hipMemsetAsync(ctx->d_counters, 0, 32 * sizeof(unsigned long long), ctx->stream);
    if (getenv("CM_ONE_STREAM")) {           // diagnostic: no concurrency between the light and the heavy kernels
        (void)hipStreamDestroy(ctx->stream2);
        ctx->stream2 = ctx->stream;
    }
    if (hwq_low) {
        ctx->err = "note: GPU_MAX_HW_QUEUES was " + std::string(hwq ? hwq : "unset") +
                   " when this context was created; if HIP was initialised earlier in this process the seven streams of a context share the "
                   "default 4 hardware queues (pipelined rounds run slower). Export GPU_MAX_HW_QUEUES=16 before the first HIP call.";
        static std::atomic<bool> said{false};
        if (hwq && !said.exchange(true) && !getenv("CM_QUIET")) fprintf(stderr, "[cmhot] %s\n", ctx->err.c_str());
    }
    *out = ctx;
    return CM_OK;
}

void cm_destroy(cm_ctx *ctx) {
    if (!ctx) return;
    report_hip(ctx, "hipSetDevice", hipSetDevice(ctx->P.device));
    // Every stream of the context is drained, and its status looked at, before anything it could still touch is released: an
    // asynchronous error of this context's work is reported here, under this context's id, instead of surfacing as an abort in
    // whichever context uses the device next.
    const std::pair<hipStream_t, const char *> streams[] = {{ctx->stream, "hipStreamSynchronize(main)"},        {ctx->stream2, "hipStreamSynchronize(heavy chains)"},
                                                            {ctx->stream_s, "hipStreamSynchronize(seeding)"},    {ctx->stream_p, "hipStreamSynchronize(pairs)"},
                                                            {ctx->stream_p2, "hipStreamSynchronize(heavy pairs)"}, {ctx->stream_p3, "hipStreamSynchronize(re-run)"}, {ctx->stream_o, "hipStreamSynchronize(pair ordering)"},
                                                            {ctx->stream_copy, "hipStreamSynchronize(copy)"}};
    for (const auto &st : streams)
        if (st.first) report_hip(ctx, st.second, hipStreamSynchronize(st.first));
    report_hip(ctx, "hipGetLastError at teardown", hipGetLastError());
    for (auto e : ctx->ev_free) report_hip(ctx, "hipEventDestroy", hipEventDestroy(e));
    ctx->ev_free.clear();
    for (auto &r : ctx->recs) {
        report_hip(ctx, "hipEventDestroy", hipEventDestroy(r.a));
        report_hip(ctx, "hipEventDestroy", hipEventDestroy(r.b));
    }
    ctx->recs.clear();
    free_reads(ctx);
    for (auto &s : ctx->slots) {
        free_all(ctx, s.idx_allocs);
        free_all(ctx, s.ann_allocs);
    }
    dfree(ctx, ctx->d_collect_idx);
    dfree(ctx, ctx->d_collect_st);
    dfree(ctx, ctx->d_collect_rec);        // grow-only output staging outlives a batch (collect_cap / collect_rec_cap go with it)
    dfree(ctx, ctx->d_pool_cursor);
    dfree(ctx, ctx->d_err);
    dfree(ctx, ctx->d_counters);
    if (ctx->h_pin) report_hip(ctx, "hipHostFree", hipHostFree(ctx->h_pin));
    for (hipEvent_t e : {ctx->ev_fork, ctx->ev_join, ctx->ev_join_p, ctx->ev_prep[0], ctx->ev_prep[1], ctx->ev_pair[0], ctx->ev_pair[1], ctx->ev_tail, ctx->ev_flags,
                         ctx->ev_first[0], ctx->ev_first[1], ctx->ev_order[0], ctx->ev_order[1], ctx->ev_seed[0], ctx->ev_seed[1], ctx->ev_staged, ctx->ev_retired})
        if (e) report_hip(ctx, "hipEventDestroy", hipEventDestroy(e));
    for (hipStream_t st : {ctx->stream_p3, ctx->stream_o, ctx->stream_s, ctx->stream_p, ctx->stream_p2, ctx->stream_copy})
        if (st) report_hip(ctx, "hipStreamDestroy", hipStreamDestroy(st));
    if (ctx->stream2 && ctx->stream2 != ctx->stream) report_hip(ctx, "hipStreamDestroy", hipStreamDestroy(ctx->stream2));
    if (ctx->stream) report_hip(ctx, "hipStreamDestroy", hipStreamDestroy(ctx->stream));
    delete ctx;
}

const char *cm_last_error(const cm_ctx *ctx) { return ctx ? ctx->err.c_str() : "null context"; }

// bucket descriptors + the final synchronisation of a contig load (s.X holds the device arrays)
static int finish_contig(cm_ctx *ctx, Slot &s) {
    const size_t nb = ((size_t)1 << (2 * CM_WINDOW_SIZE)) + 1;
    s.d_desc = nullptr;
    static const bool use_desc = !(getenv("CM_SEED_DESC") && getenv("CM_SEED_DESC")[0] == '0');
    if (use_desc) {          // 16 bytes per bucket (4 GiB per contig): a probe becomes one random read instead of two dependent ones
        const uint64_t n_buckets = nb - 1;
        uint32_t *d = nullptr;
        if (hipMalloc((void **)&d, n_buckets * cmc::DESC_WORDS * sizeof(uint32_t)) == hipSuccess) {
            s.idx_allocs.push_back(d);
            hipLaunchKernelGGL(k_build_desc, dim3((unsigned)((n_buckets + BLK - 1) / BLK)), dim3(BLK), 0, ctx->stream, s.X.bucket_off, s.X.checksum, n_buckets,
                               ctx->P.kmer, d);
            HIPCHK(ctx, hipGetLastError());
            s.d_desc = d;
        } else (void)hipGetLastError();      // not enough HBM: probes go through the arrays
    }
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    s.loaded = true;
    return CM_OK;
}

int cm_load_contig(cm_ctx *ctx, int slot, const cm_index_view *iv) {
    if (!ctx || !iv) return CM_EINVAL;
    if (slot < 0 || slot >= MAX_SLOTS) return fail(ctx, CM_EINVAL, "slot %d out of range", slot);
    if (!iv->genome || !iv->bucket_off || !iv->checksum || !iv->pos) return fail(ctx, CM_EINVAL, "null index array");
    HIPCHK(ctx, hipSetDevice(ctx->P.device));
    Slot &s = ctx->slots[slot];
    free_all(ctx, s.idx_allocs);
    s.d_desc = nullptr;
    s.loaded = false;
    ++s.gen;
    const size_t nb = ((size_t)1 << (2 * CM_WINDOW_SIZE)) + 1;
    s.X = *iv;
    int rc;
    {   // genome with CM_STAGE_PAD readable bytes on both sides (vector loads of the DP staging over-read)
        uint8_t *g = nullptr;
        const size_t pad = cmc::CM_STAGE_PAD;
        HIPCHK(ctx, hipMalloc((void **)&g, (size_t)iv->ref_len + 2 * pad));
        s.idx_allocs.push_back(g);
        HIPCHK(ctx, hipMemsetAsync(g, 0, (size_t)iv->ref_len + 2 * pad, ctx->stream));
        HIPCHK(ctx, hipMemcpyAsync(g + pad, iv->genome, (size_t)iv->ref_len, hipMemcpyHostToDevice, ctx->stream));
        s.X.genome = g + pad;
    }
    if ((rc = up(ctx, s.idx_allocs, iv->bucket_off, nb, &s.X.bucket_off))) return rc;
    if ((rc = up(ctx, s.idx_allocs, iv->checksum, (size_t)iv->n_entries, &s.X.checksum))) return rc;
    if ((rc = up(ctx, s.idx_allocs, iv->pos, (size_t)iv->n_entries, &s.X.pos))) return rc;
    return finish_contig(ctx, s);
}

// exclusive (or inclusive) scan of n uint32 items on ctx->stream; tmp: (n / S32_B + 2) words
static int scan32(cm_ctx *ctx, const uint32_t *in, uint64_t n, uint32_t *out, uint32_t *tmp, uint32_t add, int inclusive) {
    const uint32_t nb = (uint32_t)((n + S32_B - 1) / S32_B);
    if (nb == 0) return CM_OK;
    hipLaunchKernelGGL(k_scan32_a, dim3(nb), dim3(S32_T), 0, ctx->stream, in, n, out, tmp, add, inclusive);
    hipLaunchKernelGGL(k_scan32_b, dim3(1), dim3(1024), 0, ctx->stream, tmp, nb);
    hipLaunchKernelGGL(k_scan32_c, dim3(nb), dim3(S32_T), 0, ctx->stream, out, n, (const uint32_t *)tmp);
    HIPCHK(ctx, hipGetLastError());
    return CM_OK;
}

int cm_load_contig_raw(cm_ctx *ctx, int slot, const cm_index_raw *raw) {
    if (!ctx || !raw) return CM_EINVAL;
    if (slot < 0 || slot >= MAX_SLOTS) return fail(ctx, CM_EINVAL, "slot %d out of range", slot);
    if (!raw->genome || (raw->n_buckets && (!raw->hv || !raw->count14 || !raw->table))) return fail(ctx, CM_EINVAL, "null array in the raw record");
    if (raw->table_slots > 0xffffffffull) return fail(ctx, CM_ELIMIT, "table of %llu slots", (unsigned long long)raw->table_slots);
    HIPCHK(ctx, hipSetDevice(ctx->P.device));
    Slot &s = ctx->slots[slot];
    free_all(ctx, s.idx_allocs);
    s.d_desc = nullptr;
    s.loaded = false;
    ++s.gen;
    const uint64_t n_all = (uint64_t)1 << (2 * CM_WINDOW_SIZE);
    const uint32_t n_b = raw->n_buckets;
    s.X = cm_index_view{};
    s.X.contig_num = raw->contig_num;
    s.X.ref_len = raw->ref_len;
    int rc;
    {
        uint8_t *g = nullptr;
        const size_t pad = cmc::CM_STAGE_PAD;
        HIPCHK(ctx, hipMalloc((void **)&g, (size_t)raw->ref_len + 2 * pad));
        s.idx_allocs.push_back(g);
        HIPCHK(ctx, hipMemsetAsync(g, 0, (size_t)raw->ref_len + 2 * pad, ctx->stream));
        HIPCHK(ctx, hipMemcpyAsync(g + pad, raw->genome, (size_t)raw->ref_len, hipMemcpyHostToDevice, ctx->stream));
        s.X.genome = g + pad;
    }
    // temporaries: the table as in the file, the bucket list, the table offset of every bucket, scan block sums
    std::vector<void *> tmp;
    struct FreeTmp {
        cm_ctx *c;
        std::vector<void *> &v;
        ~FreeTmp() { free_all(c, v); }
    } free_tmp{ctx, tmp};
    const RawEntry *d_tab = nullptr;
    const uint32_t *d_hv = nullptr, *d_cnt = nullptr;
    if ((rc = up(ctx, tmp, (const RawEntry *)raw->table, (size_t)raw->table_slots, &d_tab))) return rc;
    if ((rc = up(ctx, tmp, raw->hv, (size_t)n_b, &d_hv))) return rc;
    if ((rc = up(ctx, tmp, raw->count14, (size_t)n_b, &d_cnt))) return rc;
    uint32_t *d_start = nullptr, *d_bs = nullptr, *d_off = nullptr;
    int *d_bad = nullptr;
    HIPCHK(ctx, hipMalloc((void **)&d_start, ((size_t)n_b + 1) * sizeof(uint32_t)));
    tmp.push_back(d_start);
    HIPCHK(ctx, hipMalloc((void **)&d_bs, ((size_t)(n_all / S32_B) + 4) * sizeof(uint32_t)));
    tmp.push_back(d_bs);
    HIPCHK(ctx, hipMalloc((void **)&d_bad, sizeof(int)));
    tmp.push_back(d_bad);
    HIPCHK(ctx, hipMemsetAsync(d_bad, 0, sizeof(int), ctx->stream));
    HIPCHK(ctx, hipMalloc((void **)&d_off, (n_all + 1) * sizeof(uint32_t)));
    s.idx_allocs.push_back(d_off);
    HIPCHK(ctx, hipMemsetAsync(d_off, 0, (n_all + 1) * sizeof(uint32_t), ctx->stream));
    if ((rc = scan32(ctx, d_cnt, n_b, d_start, d_bs, 1u, 0))) return rc;                 // slot of every bucket's header
    if (n_b) {
        hipLaunchKernelGGL(k_raw_counts, dim3((n_b + BLK - 1) / BLK), dim3(BLK), 0, ctx->stream, d_tab, (const uint32_t *)d_start, d_hv, d_cnt, n_b, n_all, d_off, d_bad);
        HIPCHK(ctx, hipGetLastError());
    }
    if ((rc = scan32(ctx, d_off, n_all + 1, d_off, d_bs, 0u, 1))) return rc;             // counts -> bucket offsets, in place
    unsigned long long landing[2] = {0, 0};
    HIPCHK(ctx, hipMemcpyAsync(&landing[0], d_off + n_all, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(&landing[1], d_bad, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if ((int)landing[1]) return fail(ctx, CM_EINVAL, "malformed index table: a bucket header is out of range");
    const uint64_t total = (uint32_t)landing[0];
    uint16_t *d_cs = nullptr;
    uint32_t *d_ps = nullptr;
    HIPCHK(ctx, hipMalloc((void **)&d_cs, (total ? total : 1) * sizeof(uint16_t)));
    s.idx_allocs.push_back(d_cs);
    HIPCHK(ctx, hipMalloc((void **)&d_ps, (total ? total : 1) * sizeof(uint32_t)));
    s.idx_allocs.push_back(d_ps);
    if (n_b) {
        hipLaunchKernelGGL(k_raw_scatter, dim3((n_b + BLK - 1) / BLK), dim3(BLK), 0, ctx->stream, d_tab, (const uint32_t *)d_start, d_hv, n_b, (const uint32_t *)d_off, d_cs, d_ps);
        HIPCHK(ctx, hipGetLastError());
    }
    s.X.bucket_off = d_off;
    s.X.checksum = d_cs;
    s.X.pos = d_ps;
    s.X.n_entries = total;
    return finish_contig(ctx, s);       // (synchronises the stream: the temporaries may go)
}

int cm_load_annotation(cm_ctx *ctx, int slot, const cm_annot_view *av) {
    if (!ctx || !av) return CM_EINVAL;
    if (slot < 0 || slot >= MAX_SLOTS) return fail(ctx, CM_EINVAL, "slot %d out of range", slot);
    if (av->n_iv == 0 || av->n_chr == 0) return fail(ctx, CM_EINVAL, "annotation needs >= 1 interval and >= 1 chromosome");
    HIPCHK(ctx, hipSetDevice(ctx->P.device));
    Slot &s = ctx->slots[slot];
    free_all(ctx, s.ann_allocs);
    s.has_annot = false;
    ++s.gen;
    cmc::AnnotAosHost aos;
    cmc::build_annot_aos(*av, aos);
    cmc::AnnotDev &A = s.A;
    A = cmc::AnnotDev{};
    A.n_iv = av->n_iv; A.n_seg = av->n_seg; A.n_trans = av->n_trans; A.n_gene = av->n_gene; A.n_chr = av->n_chr; A.n_bits = av->n_bits;
    A.pair_reach = aos.pair_reach;
    int rc;
    auto &al = s.ann_allocs;
    if ((rc = up(ctx, al, aos.iv.data(), aos.iv.size(), &A.iv))) return rc;
    if ((rc = up(ctx, al, av->iv_seg, (size_t)av->iv_seg_off[av->n_iv], &A.iv_seg))) return rc;
    if ((rc = up(ctx, al, aos.seg.data(), aos.seg.size(), &A.seg))) return rc;
    if ((rc = up(ctx, al, av->seg_tid, (size_t)av->seg_tid_off[av->n_seg], &A.seg_tid))) return rc;
    if ((rc = up(ctx, al, aos.tr.data(), aos.tr.size(), &A.tr))) return rc;
    if ((rc = up(ctx, al, av->t2s, (size_t)av->t2s_off[av->n_trans], &A.t2s))) return rc;
    if ((rc = up(ctx, al, aos.gene.data(), aos.gene.size(), &A.gene))) return rc;
    if ((rc = up(ctx, al, av->near_border_bits, (size_t)(av->n_bits / 64), &A.near_border_bits))) return rc;
    if ((rc = up(ctx, al, av->intronic_bits, (size_t)(av->n_bits / 64), &A.intronic_bits))) return rc;
    if ((rc = up(ctx, al, av->chr_shift, (size_t)av->n_chr, &A.chr_shift))) return rc;
    if ((rc = up(ctx, al, av->chr_id, (size_t)av->n_chr, &A.chr_id))) return rc;
    if (av->iv_bucket && av->n_iv_bucket >= 2) {
        if ((rc = up(ctx, al, av->iv_bucket, (size_t)av->n_iv_bucket, &A.iv_bucket))) return rc;
        A.iv_bucket_shift = av->iv_bucket_shift;
        A.n_iv_bucket = av->n_iv_bucket;
    }
    // the staging copies are asynchronous: the host vectors must outlive them
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    {   // every window bound upper_bound() can return lies within maxIntronLen of the queried hit (k_chain_heavy)
        const long long mi = ctx->P.max_intron;
        bool ok = mi >= (long long)ctx->P.max_read_len + ctx->P.max_ed && mi < (1ll << 30);
        for (uint32_t i = 0; i < av->n_iv && ok; ++i) {
            if (av->iv_seg_off[i + 1] == av->iv_seg_off[i]) continue;
            const long long far = std::max<long long>((long long)av->iv_max_next_exon[i] + ctx->P.kmer, (long long)av->iv_max_end[i]);
            if (far - (long long)av->iv_spos[i] > mi) ok = false;
        }
        s.chain_parallel_ok = ok;
    }
    s.has_annot = true;
    return CM_OK;
}

int cm_unload_contig(cm_ctx *ctx, int slot) {
    if (!ctx || slot < 0 || slot >= MAX_SLOTS) return CM_EINVAL;
    (void)hipSetDevice(ctx->P.device);
    (void)hipStreamSynchronize(ctx->stream);
    free_all(ctx, ctx->slots[slot].idx_allocs);
    free_all(ctx, ctx->slots[slot].ann_allocs);
    ctx->slots[slot].loaded = ctx->slots[slot].has_annot = false;
    ctx->slots[slot].d_desc = nullptr;
    ++ctx->slots[slot].gen;
    return CM_OK;
}

// Validation shared by cm_reads_upload / cm_reads_stage; returns the longest read through *max_len_out.
static int check_reads(cm_ctx *ctx, const cm_reads *rd, int *max_len_out) {
    const uint64_t n = rd->n_pairs;
    if (n > 0x3fffffffull) return fail(ctx, CM_ELIMIT, "more than 2^30 pairs in one batch");
    if (!rd->seq1 || !rd->seq2 || !rd->off1 || !rd->off2) return fail(ctx, CM_EINVAL, "null read arrays");
    int max_len = 0;
    const uint64_t lim = (uint64_t)ctx->P.max_read_len;
    uint64_t bad = 0, longest = 0;
    // (2^21 pairs: 1.4 ms on one core, with the GPU idle when the call sits between two batches -- cm_reads_stage; ranges on a few threads)
    auto scan = [&](uint64_t lo, uint64_t hi, uint64_t *bad_o, uint64_t *long_o) {
        uint64_t bd = 0, lg = 0;
        for (uint64_t i = lo; i < hi; ++i) {
            const uint64_t l1 = rd->off1[i + 1] - rd->off1[i], l2 = rd->off2[i + 1] - rd->off2[i];     // wraps to huge when not monotone
            const uint64_t m = l1 > l2 ? l1 : l2;
            if (m > lim && !bd) bd = i + 1;
            if (m > lg) lg = m;
        }
        *bad_o = bd;
        *long_o = lg;
    };
    const unsigned hw = std::thread::hardware_concurrency();
    const unsigned n_thr = n < (1u << 18) ? 1u : std::min(8u, hw ? hw : 1u);
    if (n_thr <= 1) {
        scan(0, n, &bad, &longest);
    } else {
        uint64_t bd[8] = {0}, lg[8] = {0};
        std::thread th[8];
        unsigned started = 1;
        try {
            for (; started < n_thr; ++started) th[started] = std::thread(scan, n * started / n_thr, n * (started + 1) / n_thr, &bd[started], &lg[started]);
        } catch (...) {                                   // no more threads to be had: the rest of the ranges on this one
        }
        scan(0, n / n_thr, &bd[0], &lg[0]);
        for (unsigned k = started; k < n_thr; ++k) scan(n * k / n_thr, n * (k + 1) / n_thr, &bd[k], &lg[k]);
        for (unsigned k = 1; k < started; ++k) th[k].join();
        for (unsigned k = 0; k < n_thr; ++k) {
            if (bd[k] && !bad) bad = bd[k];                   // the first offending pair
            if (lg[k] > longest) longest = lg[k];
        }
    }
    if (bad) {
        const uint64_t i = bad - 1;
        if (rd->off1[i + 1] < rd->off1[i] || rd->off2[i + 1] < rd->off2[i]) return fail(ctx, CM_EINVAL, "read offsets not monotone at pair %llu", (unsigned long long)i);
        return fail(ctx, CM_EINVAL, "pair %llu longer than max_read_len %d", (unsigned long long)i, ctx->P.max_read_len);
    }
    max_len = (int)longest;
    if (max_len / ctx->P.kmer > cmc::MAX_SEEDS) return fail(ctx, CM_ELIMIT, "%d seeds per read > %d supported", max_len / ctx->P.kmer, cmc::MAX_SEEDS);
    *max_len_out = max_len;
    return CM_OK;
}

// Pairs per tile (launch group) of a batch of n pairs.
static uint32_t tile_for(uint64_t n) {
    // A batch is walked in at least two tiles when it has more than 2^20 pairs (round-major order: a tile's seeding then sees the flags
    // its previous pair stage wrote), and the tiles are as large as the batch allows up to 2^21: an item costs 1 - 4 ms beyond what
    // grows with its pairs (tails of the persistent grids, ~40 launches, three host round trips) -- 2^22-pair batches 146.6 ms per step
    // in four tiles, 139.0 in two; 2^21-pair batches 73 ms in two tiles, 96 in four, 77 in one.
    uint32_t tile_cap = TILE_PAIRS;
    if (n > 2ull * TILE_PAIRS) {
        const uint64_t half = ((n + 1) / 2 + 65535) / 65536 * 65536;
        tile_cap = (uint32_t)std::min<uint64_t>(half, TILE_PAIRS_MAX);
    }
    if (const char *e = getenv("CM_TILE_PAIRS")) {       // tuning knob: pairs per launch group
        const long v = atol(e);
        if (v >= 64 && v <= (1l << 24)) tile_cap = (uint32_t)v;
    }
    return (uint32_t)(n < tile_cap ? n : tile_cap);
}

// Per-tile workspace and per-batch state of the batch that is becoming resident (n pairs, longest read max_len).
static int prepare_resident(cm_ctx *ctx, uint64_t n, int max_len) {
    ctx->n_seeds = max_len / ctx->P.kmer;
    ctx->max_len = max_len;
    HIPCHK(ctx, ensure(ctx, ctx->d_state, n * sizeof(cm_mapped_read)));
    HIPCHK(ctx, ensure(ctx, ctx->d_active, n));
    HIPCHK(ctx, ensure(ctx, ctx->d_active_b, n));
    HIPCHK(ctx, ensure(ctx, ctx->d_cat, n * sizeof(int32_t)));
    // workspace for one tile
    const uint32_t tile = tile_for(n);
    ctx->tile = tile;
    const size_t nprob = (size_t)tile * 4, nprobe = nprob * (size_t)(ctx->n_seeds ? ctx->n_seeds : 1);
    HIPCHK(ctx, ensure(ctx, ctx->d_sstart, nprobe * 4));
    HIPCHK(ctx, ensure(ctx, ctx->d_scnt, nprobe * 4));
    HIPCHK(ctx, ensure(ctx, ctx->d_sraw, nprobe * 4));
    HIPCHK(ctx, ensure(ctx, ctx->d_cells, nprob * 4));
    HIPCHK(ctx, ensure(ctx, ctx->d_celloff, (nprob + 2) * 8));
    HIPCHK(ctx, ensure(ctx, ctx->d_bmax, (nprob / SCAN_ELEMS + 2) * 4));
    HIPCHK(ctx, ensure(ctx, ctx->d_bsum, (nprob / SCAN_ELEMS + 2) * 8));
    HIPCHK(ctx, ensure(ctx, ctx->d_sstart_b, nprobe * 4));
    HIPCHK(ctx, ensure(ctx, ctx->d_scnt_b, nprobe * 4));
    HIPCHK(ctx, ensure(ctx, ctx->d_sraw_b, nprobe * 4));
    HIPCHK(ctx, ensure(ctx, ctx->d_celloff_b, (nprob + 2) * 8));
    HIPCHK(ctx, ensure(ctx, ctx->d_bmax_b, (nprob / SCAN_ELEMS + 2) * 4));
    HIPCHK(ctx, ensure(ctx, ctx->d_bsum_b, (nprob / SCAN_ELEMS + 2) * 8));
    // DP cells: room for 64 cells per problem on average, at least 8M (one worst-case problem is
    // n_seeds * seed_lim cells); larger tiles are split into ranges by run_chain_tile.
    unsigned long long cap = (unsigned long long)nprob * 64ull;
    const unsigned long long worst = (unsigned long long)ctx->P.seed_lim * (unsigned long long)(ctx->n_seeds ? ctx->n_seeds : 1);
    if (cap < (8ull << 20)) cap = 8ull << 20;
    if (cap < worst) cap = worst;
    if (cap > ctx->cells_cap || !ctx->d_dpscore) ctx->cells_cap = cap;        // grow-only, like the buffers behind it
    HIPCHK(ctx, ensure(ctx, ctx->d_dpscore, ctx->cells_cap * sizeof(double)));
    HIPCHK(ctx, ensure(ctx, ctx->d_dpprev, ctx->cells_cap * sizeof(int32_t)));
    HIPCHK(ctx, ensure(ctx, ctx->d_chains, nprob * CM_BESTCHAINLIM * sizeof(cm_chain)));
    HIPCHK(ctx, ensure(ctx, ctx->d_nchain, nprob * 4));
    HIPCHK(ctx, ensure(ctx, ctx->d_high, nprob * 4));
    HIPCHK(ctx, ensure(ctx, ctx->d_chains_b, nprob * CM_BESTCHAINLIM * sizeof(cm_chain)));
    HIPCHK(ctx, ensure(ctx, ctx->d_nchain_b, nprob * 4));
    HIPCHK(ctx, ensure(ctx, ctx->d_high_b, nprob * 4));
    HIPCHK(ctx, ensure(ctx, ctx->d_resid_b, (size_t)tile * 4 * sizeof(uint16_t)));
    HIPCHK(ctx, ensure(ctx, ctx->d_cctr, CTR_WORDS * sizeof(unsigned int)));
    HIPCHK(ctx, ensure(ctx, ctx->d_cblk, (size_t)N_CLS * (4 * (size_t)tile / CLS_T + 2) * sizeof(unsigned int)));
    HIPCHK(ctx, ensure(ctx, ctx->d_cls, (size_t)tile));
    HIPCHK(ctx, ensure(ctx, ctx->d_cls4, (size_t)tile * 4));
    HIPCHK(ctx, ensure(ctx, ctx->d_perm4, (size_t)tile * 4 * 4));
    HIPCHK(ctx, ensure(ctx, ctx->d_cctr_b, CTR_WORDS * sizeof(unsigned int)));
    HIPCHK(ctx, ensure(ctx, ctx->d_cblk_b, (size_t)N_CLS * (4 * (size_t)tile / CLS_T + 2) * sizeof(unsigned int)));
    HIPCHK(ctx, ensure(ctx, ctx->d_cls4_b, (size_t)tile * 4));
    HIPCHK(ctx, ensure(ctx, ctx->d_thigh, (size_t)tile * 4 * sizeof(int32_t)));
    HIPCHK(ctx, ensure(ctx, ctx->d_thigh_b, (size_t)tile * 4 * sizeof(int32_t)));
    HIPCHK(ctx, ensure(ctx, ctx->d_perm4_b, (size_t)tile * 4 * 4));
    HIPCHK(ctx, ensure(ctx, ctx->d_resid, (size_t)tile * 4 * sizeof(uint16_t)));
    HIPCHK(ctx, ensure(ctx, ctx->d_perm, (size_t)tile * 4 * 2));          // x 2: one per set of chain records, like the re-run list (run_pair_tile)
    HIPCHK(ctx, ensure(ctx, ctx->d_hlist, (size_t)tile * 4 * 2));
    HIPCHK(ctx, ensure(ctx, ctx->d_hres, (size_t)HEAVY_GRID_MAX * (64 * sizeof(HRes) + HEAVY_SCRATCH)));
    if (heavy_pipeline()) {
        // the pipeline's arrays: per heavy pair, per mate-pair task (~ 10 per heavy pair on the dense workload, room for 6 per pair of
        // the tile), per unpaired chain (room for 4 per pair of the tile); what does not fit goes to k_pair_heavy
        // (CM_HP_TASKS_CAP / CM_HP_UNP_CAP: test knobs, small capacities so that the fall-back path is taken)
        const size_t tc = getenv("CM_HP_TASKS_CAP") ? (size_t)std::max(1, atoi(getenv("CM_HP_TASKS_CAP"))) : std::max<size_t>((size_t)tile * 6, 4096);
        const size_t uc = getenv("CM_HP_UNP_CAP") ? (size_t)std::max(1, atoi(getenv("CM_HP_UNP_CAP"))) : std::max<size_t>((size_t)tile * 4, 4096);
        ctx->hp_tasks_cap = (uint32_t)tc;
        ctx->hp_unp_cap = (uint32_t)uc;
        HIPCHK(ctx, ensure(ctx, ctx->d_hp, (size_t)tile * sizeof(HPair)));
        HIPCHK(ctx, ensure(ctx, ctx->d_hp_list2, (size_t)tile * 4));
        HIPCHK(ctx, ensure(ctx, ctx->d_hp_fall, (size_t)tile * 4 * 2));
        HIPCHK(ctx, ensure(ctx, ctx->d_hp_fallctr, 2 * sizeof(unsigned int)));
        HIPCHK(ctx, ensure(ctx, ctx->d_hp_T, tc * sizeof(HTask)));
        HIPCHK(ctx, ensure(ctx, ctx->d_hp_tcls, tc));
        HIPCHK(ctx, ensure(ctx, ctx->d_hp_tperm, tc * 4));
        HIPCHK(ctx, ensure(ctx, ctx->d_hp_tblk, (size_t)N_CLS * (tc / CLS_T + 2) * sizeof(unsigned int)));
        HIPCHK(ctx, ensure(ctx, ctx->d_hp_tctr, CTR_WORDS * sizeof(unsigned int)));
        HIPCHK(ctx, ensure(ctx, ctx->d_hp_pre, tc * 4 * sizeof(cmc::PreDP)));
        HIPCHK(ctx, ensure(ctx, ctx->d_hp_q, tc * 4 * 4));
        HIPCHK(ctx, ensure(ctx, ctx->d_hp_res, tc * sizeof(HRes)));
        HIPCHK(ctx, ensure(ctx, ctx->d_hp_U, uc * sizeof(HUnp)));
        HIPCHK(ctx, ensure(ctx, ctx->d_hp_pre2, uc * 2 * sizeof(cmc::PreDP)));
        HIPCHK(ctx, ensure(ctx, ctx->d_hp_q2, uc * 2 * 4));
        HIPCHK(ctx, ensure(ctx, ctx->d_hp_lists, (size_t)HP_PLAN_GRID * HEAVY_LIST * 2));
        HIPCHK(ctx, ensure(ctx, ctx->d_hp_ctr, HC_WORDS * sizeof(unsigned int)));
    }
    {
        const uint32_t *before = ctx->d_pair_err;
        HIPCHK(ctx, ensure(ctx, ctx->d_pair_err, (size_t)tile * 4 * 2));      // one set per set of chain records (run_pair_tile)
        if (ctx->d_pair_err != before) HIPCHK(ctx, hipMemsetAsync(ctx->d_pair_err, 0, (size_t)tile * 4 * 2, ctx->stream_p));   // the kernels keep it zero
    }
    HIPCHK(ctx, ensure(ctx, ctx->d_retry_list, (size_t)tile * 4 * 2));
    HIPCHK(ctx, ensure(ctx, ctx->d_retry_ctr, 4 * sizeof(unsigned int)));
    HIPCHK(ctx, ensure(ctx, ctx->d_heavy_load, sizeof(unsigned long long)));
    HIPCHK(ctx, ensure(ctx, ctx->d_spill, (size_t)RETRY_GRID * BLK_PAIR * RETRY_SPILL * sizeof(cmc::MemoSpill)));
    HIPCHK(ctx, ensure(ctx, ctx->d_cls_ctr, 2 * CTR_WORDS * sizeof(unsigned int)));
    HIPCHK(ctx, ensure(ctx, ctx->d_cls_ctr2, CTR_WORDS * sizeof(unsigned int)));
    HIPCHK(ctx, ensure(ctx, ctx->d_cls_sub, (size_t)tile));
    HIPCHK(ctx, ensure(ctx, ctx->d_perm1, (size_t)tile * 4));
    HIPCHK(ctx, ensure(ctx, ctx->d_cls_ctr3, CTR_WORDS * sizeof(unsigned int)));
    HIPCHK(ctx, ensure(ctx, ctx->d_cls_sub2, (size_t)tile));
    HIPCHK(ctx, ensure(ctx, ctx->d_perm0, (size_t)tile * 4));
    HIPCHK(ctx, ensure(ctx, ctx->d_blk_cnt, (size_t)N_CLS * (4 * (size_t)tile / CLS_T + 2) * sizeof(unsigned int)));
    if (getenv("CM_LANE_CLK")) {
#if defined(CM_DIAG)
        const size_t clk_words = 16 * 2 + 1;  // per-pair rows + wave-level rows of k_pair and k_pair_heavy (diag)
#else
        const size_t clk_words = 1;
#endif
        HIPCHK(ctx, ensure(ctx, ctx->d_lane_clk, n * 8 * clk_words));
        HIPCHK(ctx, hipMemsetAsync(ctx->d_lane_clk, 0, n * 8 * clk_words, ctx->stream));
    }
    unsigned long long pool = (unsigned long long)nprob * 2048ull;       // improvement log
    if (pool < (256ull << 20)) pool = 256ull << 20;
    if (pool > (8ull << 30)) pool = 8ull << 30;
    if (const char *e = getenv("CM_POOL_BYTES")) {       // test knob: start with a small log pool to exercise the recovery path
        const unsigned long long v = strtoull(e, nullptr, 10);
        if (v >= 4096 && !ctx->d_pool) pool = v;
    }
    if (pool > ctx->pool_bytes || !ctx->d_pool) ctx->pool_bytes = pool;
    HIPCHK(ctx, ensure(ctx, ctx->d_pool, ctx->pool_bytes));
    return CM_OK;
}

// Copies of one batch into (grow-only) read buffers on stream `st`: only the CM_STAGE_PAD slack around the reads is
// cleared (cmc::stage over-reads into it), the reads themselves are overwritten by the copy.
static int copy_reads(cm_ctx *ctx, const cm_reads *rd, hipStream_t st, uint8_t *&seq1_base, uint8_t *&seq2_base, uint64_t *&off1, uint64_t *&off2) {
    const uint64_t n = rd->n_pairs;
    const size_t b1 = (size_t)rd->off1[n] - (size_t)rd->off1[0], b2 = (size_t)rd->off2[n] - (size_t)rd->off2[0];
    if (rd->off1[0] != 0 || rd->off2[0] != 0) return fail(ctx, CM_EINVAL, "read offsets must start at 0");
    const size_t pad = cmc::CM_STAGE_PAD;                  // readable slack around the reads (see cmc::stage)
    HIPCHK(ctx, ensure(ctx, seq1_base, b1 + 2 * pad));
    HIPCHK(ctx, ensure(ctx, seq2_base, b2 + 2 * pad));
    HIPCHK(ctx, ensure(ctx, off1, (n + 1) * sizeof(uint64_t)));
    HIPCHK(ctx, ensure(ctx, off2, (n + 1) * sizeof(uint64_t)));
    HIPCHK(ctx, hipMemsetAsync(seq1_base, 0, pad, st));
    HIPCHK(ctx, hipMemsetAsync(seq1_base + pad + b1, 0, pad, st));
    HIPCHK(ctx, hipMemsetAsync(seq2_base, 0, pad, st));
    HIPCHK(ctx, hipMemsetAsync(seq2_base + pad + b2, 0, pad, st));
    if (b1) HIPCHK(ctx, hipMemcpyAsync(seq1_base + pad, rd->seq1, b1, hipMemcpyHostToDevice, st));
    if (b2) HIPCHK(ctx, hipMemcpyAsync(seq2_base + pad, rd->seq2, b2, hipMemcpyHostToDevice, st));
    HIPCHK(ctx, hipMemcpyAsync(off1, rd->off1, (n + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, st));
    HIPCHK(ctx, hipMemcpyAsync(off2, rd->off2, (n + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, st));
    return CM_OK;
}

int cm_reads_upload(cm_ctx *ctx, const cm_reads *rd, const cm_mapped_read *prior) {
    if (!ctx || !rd) return CM_EINVAL;
    HIPCHK(ctx, hipSetDevice(ctx->P.device));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    const uint64_t n = rd->n_pairs;
    ctx->pre_ready = false;                        // a prefetched first round belonged to a batch that came in through cm_reads_swap
    ctx->n_pairs = 0;
    ctx->tile = 0;
    if (n == 0) {                                  // an empty batch releases the per-batch buffers
        free_reads(ctx);
        return CM_OK;
    }
    int max_len = 0;
    int rc = check_reads(ctx, rd, &max_len);
    if (rc) return rc;
    if ((rc = copy_reads(ctx, rd, ctx->stream, ctx->d_seq1_base, ctx->d_seq2_base, ctx->d_off1, ctx->d_off2))) return rc;
    ctx->d_seq1 = ctx->d_seq1_base + cmc::CM_STAGE_PAD;
    ctx->d_seq2 = ctx->d_seq2_base + cmc::CM_STAGE_PAD;
    if ((rc = prepare_resident(ctx, n, max_len))) return rc;
    ctx->n_pairs = n;
    KCore k{};
    k.P = ctx->P;
    hipLaunchKernelGGL(k_init_state, dim3((unsigned)((n + BLK - 1) / BLK)), dim3(BLK), 0, ctx->stream, k, ctx->d_state, ctx->d_active, ctx->d_cat, n);
    if (prior) {
        // carried state of an earlier round: a pair is active unless the caller marks it retired
        // by type < 0 (never produced by this library); active flags are otherwise all 1.
        HIPCHK(ctx, hipMemcpyAsync(ctx->d_state, prior, n * sizeof(cm_mapped_read), hipMemcpyHostToDevice, ctx->stream));
    }
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return CM_OK;
}

int cm_reads_stage(cm_ctx *ctx, const cm_reads *rd, const cm_mapped_read *prior) {
    if (!ctx || !rd) return CM_EINVAL;
    HIPCHK(ctx, hipSetDevice(ctx->P.device));
    if (rd->n_pairs == 0) return fail(ctx, CM_EINVAL, "cm_reads_stage: empty batch");
    int max_len = 0;
    static const bool trace = getenv("CM_STAGE_TRACE") != nullptr;          // diagnostic: host time of the three parts
    const auto t0 = std::chrono::steady_clock::now();
    int rc = check_reads(ctx, rd, &max_len);
    if (rc) return rc;
    const auto t1 = std::chrono::steady_clock::now();
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream_copy));          // an earlier staged batch that was never swapped in is dropped
    const auto t2 = std::chrono::steady_clock::now();
    ctx->staged = false;
    ctx->pre_launched = false;
    if ((rc = copy_reads(ctx, rd, ctx->stream_copy, ctx->st_seq1_base, ctx->st_seq2_base, ctx->st_off1, ctx->st_off2))) return rc;
    if (trace) {
        const auto t3 = std::chrono::steady_clock::now();
        auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
        fprintf(stderr, "cm_reads_stage: check %.3f ms, wait for the copy stream %.3f ms, copies issued %.3f ms\n", ms(t0, t1), ms(t1, t2), ms(t2, t3));
    }
    ctx->st_has_prior = prior != nullptr;
    if (prior) {
        HIPCHK(ctx, ensure(ctx, ctx->st_prior, rd->n_pairs * sizeof(cm_mapped_read)));
        HIPCHK(ctx, hipMemcpyAsync(ctx->st_prior, prior, rd->n_pairs * sizeof(cm_mapped_read), hipMemcpyHostToDevice, ctx->stream_copy));
    }
    HIPCHK(ctx, hipEventRecord(ctx->ev_staged, ctx->stream_copy));
    ctx->st_n_pairs = rd->n_pairs;
    ctx->st_max_len = max_len;
    ctx->staged = true;
    return CM_OK;
}

int cm_reads_swap(cm_ctx *ctx) {
    if (!ctx) return CM_EINVAL;
    HIPCHK(ctx, hipSetDevice(ctx->P.device));
    if (!ctx->staged) return fail(ctx, CM_ESTATE, "cm_reads_swap: no staged batch");
    // work already queued on the mapping stream still reads the old buffers: order the swap behind it on the device
    // (no host wait), and the new batch's first kernel behind the staged copies
    std::swap(ctx->d_seq1_base, ctx->st_seq1_base);
    std::swap(ctx->d_seq2_base, ctx->st_seq2_base);
    std::swap(ctx->d_off1, ctx->st_off1);
    std::swap(ctx->d_off2, ctx->st_off2);
    std::swap(ctx->caps[(const void *)&ctx->d_seq1_base], ctx->caps[(const void *)&ctx->st_seq1_base]);
    std::swap(ctx->caps[(const void *)&ctx->d_seq2_base], ctx->caps[(const void *)&ctx->st_seq2_base]);
    std::swap(ctx->caps[(const void *)&ctx->d_off1], ctx->caps[(const void *)&ctx->st_off1]);
    std::swap(ctx->caps[(const void *)&ctx->d_off2], ctx->caps[(const void *)&ctx->st_off2]);
    ctx->d_seq1 = ctx->d_seq1_base + cmc::CM_STAGE_PAD;
    ctx->d_seq2 = ctx->d_seq2_base + cmc::CM_STAGE_PAD;
    ctx->staged = false;
    const uint64_t n = ctx->st_n_pairs;
    ctx->pre_ready = ctx->pre_launched;                            // cm_map_rounds prepared this batch's first round already
    ctx->pre_launched = false;
    ctx->pre_n = n;
    ctx->n_pairs = 0;
    int rc = prepare_resident(ctx, n, ctx->st_max_len);
    if (rc) return rc;
    ctx->n_pairs = n;
    HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_staged, 0));
    KCore k{};
    k.P = ctx->P;
    hipLaunchKernelGGL(k_init_state, dim3((unsigned)((n + BLK - 1) / BLK)), dim3(BLK), 0, ctx->stream, k, ctx->d_state, ctx->d_active, ctx->d_cat, n);
    if (ctx->st_has_prior)
        HIPCHK(ctx, hipMemcpyAsync(ctx->d_state, ctx->st_prior, n * sizeof(cm_mapped_read), hipMemcpyDeviceToDevice, ctx->stream));
    // the next cm_reads_stage overwrites the buffers this swap retired: it must wait for the mapping stream's work on them
    HIPCHK(ctx, hipEventRecord(ctx->ev_retired, ctx->stream));
    HIPCHK(ctx, hipStreamWaitEvent(ctx->stream_copy, ctx->ev_retired, 0));
    HIPCHK(ctx, hipGetLastError());
    return CM_OK;
}

// The re-run launch of a pair stage and ev_pair[b] behind it.  Eight blocks that each ask for 47 - 64 KB of LDS behind kernels that
// fill the chip wait milliseconds for their turn -- on the hg38-like bench until the heavy chaining kernel of the NEXT item had
// drained, and the item after that (its seeding reads the flags, its chaining rewrites the chain records) behind them -- to find,
// nearly always, an empty list.  So with several tiles the decision is made late (`defer`): the pair stage copies the length of
// its re-run list to the host, and whoever first needs the stage to be complete calls settle_pair, which waits for the two pair
// kernels (ev_first[b]), launches the re-run only if something is queued, and records ev_pair[b].
static int launch_rerun(cm_ctx *ctx, int b) {
    const cm_ctx::Rerun &q = ctx->rerun[b];
    static const bool no_rerun = getenv("CM_NO_RERUN") != nullptr;      // diagnostic: what the launch costs (a queued pair would stay unmapped)
    if (no_rerun) return CM_OK;
    const RetryArgs ra2{q.pair_err, q.retry_list, q.retry_ctr, ctx->d_spill, RETRY_SPILL, 0};
    hipLaunchKernelGGL(k_pair_rerun, dim3(RETRY_GRID), dim3(BLK_PAIR), q.lds2, ctx->stream_p3, q.core, q.rd, q.p0, q.nt, q.chains, q.nchain, q.high,
                       ctx->d_state, q.act_out, ctx->d_cat, q.is_last, ctx->d_err, ctx->d_counters, q.cap2, (unsigned long long *)nullptr,
                       (const uint32_t *)q.retry_list, (const unsigned int *)q.retry_ctr, q.retry_ctr + 1, ra2);
    HIPCHK(ctx, hipGetLastError());
    return CM_OK;
}
// The pipeline's fall-back launch: k_pair_heavy over the pairs whose tasks or unpaired chains did not fit the pipeline's arrays (none on
// the bench workloads).  A launch that asks for 14 KB of LDS per wave to find an empty list waited 0.03 - 2.3 ms behind the chain kernels
// of the next item at the end of every pair stage: decided late like the re-run, from the list length the stage copied to the host.
static int launch_fall_back(cm_ctx *ctx, int b, hipStream_t st) {
    const cm_ctx::Rerun &q = ctx->rerun[b];
    const RetryArgs ra1{q.pair_err, q.retry_list, q.retry_ctr, nullptr, 0, 1};
    hipLaunchKernelGGL(k_pair_heavy, dim3(q.fall_grid), dim3(BLK_PAIR), q.lds_heavy, st, q.core, q.rd, q.p0, q.fall_list, (const unsigned int *)q.fall_ctr, q.chains,
                       q.nchain, q.high, ctx->d_state, q.act_out, ctx->d_cat, q.is_last, ctx->d_err, ctx->d_counters, q.str_cap, (unsigned long long *)nullptr,
                       ctx->d_hres, q.fall_cursor, ra1);
    HIPCHK(ctx, hipGetLastError());
    ++ctx->launches[4];
    return CM_OK;
}
// wait = false: without blocking the host (the end of cm_map_rounds, which stays asynchronous): the re-run is launched whatever the count
static int settle_pair(cm_ctx *ctx, int b, bool wait = true) {
    if (!ctx->rerun[b].deferred) return CM_OK;
    ctx->rerun[b].deferred = false;
    if (wait) HIPCHK(ctx, hipEventSynchronize(ctx->ev_first[b]));
    HIPCHK(ctx, hipStreamWaitEvent(ctx->stream_p3, ctx->ev_first[b], 0));
    bool fell = false;
    if (ctx->rerun[b].fall && (!wait || *(const volatile unsigned int *)(ctx->h_pin + 14 + b) != 0u)) {
        const int rc = launch_fall_back(ctx, b, ctx->stream_p3);        // (may queue pairs for the re-run: that one unconditionally then)
        if (rc) return rc;
        fell = true;
    }
    if (!wait || fell || *(const volatile unsigned int *)(ctx->h_pin + 12 + b) != 0u) {
        const int rc = launch_rerun(ctx, b);
        if (rc) return rc;
    }
    HIPCHK(ctx, hipEventRecord(ctx->ev_pair[b], ctx->stream_p3));
    return CM_OK;
}

// The pair stage of one tile and round on the pair streams: waits for that item's chains (ev_prep[b]), reads the flags
// act_in, writes act_out for every pair of the tile, signals ev_pair[b] when the chain buffers of set b are free again.
// same_tile_as_prev: the previous item was this tile's previous round -- its re-run launch (stream p3) wrote states and flags this
// item's pair kernels read; every other consumer is ordered behind the re-run through ev_pair[].
static int run_pair_tile(cm_ctx *ctx, const KCore &core, uint64_t p0, uint32_t nt, int is_last_round, const uint8_t *act_in, uint8_t *act_out,
                         const RoundBufs &rb, int b, bool same_tile_as_prev, bool defer) {
    const ReadsDev rd{ctx->d_seq1, ctx->d_seq2, ctx->d_off1, ctx->d_off2};
    hipStream_t sp = ctx->stream_p, sp2 = ctx->stream_p2, sp3 = ctx->stream_p3;
    {
        int rc;
        if ((rc = settle_pair(ctx, b))) return rc;                       // this set's previous stage (its re-run list is about to be reused)
        if (same_tile_as_prev && (rc = settle_pair(ctx, b ^ 1))) return rc;
    }
    // The work classes and ordered lists of this stage (ten small launches) go to a stream of their own, so -- every array the pair
    // kernels read from them existing once per set of chain records -- they are computed while the pair stage of the item before
    // still runs, as soon as this item's chains are complete; behind that stage on the pair stream they were 1.7 ms per item with
    // nothing else on the chip but the next item's seeding (12 % of the hg38-like step).
    hipStream_t so = ctx->stream_o;
    HIPCHK(ctx, hipStreamWaitEvent(so, ctx->ev_prep[b], 0));
    if (same_tile_as_prev && ctx->pair_pending[b ^ 1]) HIPCHK(ctx, hipStreamWaitEvent(so, ctx->ev_pair[b ^ 1], 0));
    uint32_t *const perm = ctx->d_perm + (size_t)b * ctx->tile, *const hlist = ctx->d_hlist + (size_t)b * ctx->tile;
    unsigned int *const cls_ctr = ctx->d_cls_ctr + (size_t)b * CTR_WORDS;
    // (the re-run list, its counters and the per-pair flags exist once per set of chain records, like those: item i + 1 leaves
    // item i's alone, item i + 2 starts after ev_pair[b])
    uint32_t *pair_err = ctx->d_pair_err + (size_t)b * ctx->tile, *retry_list = ctx->d_retry_list + (size_t)b * ctx->tile;
    unsigned int *retry_ctr = ctx->d_retry_ctr + 2 * b;
    // str_cap: chars per staged string (multiple of 8); LDS = 2 strings x lbuf_bytes(str_cap) x 64 lanes
    // a DP string is at most a read minus one seed, plus the band (extend_side: len + band; dp_fits() reports anything longer)
    // (+ 4 characters of slack: 144 characters = 76-byte rows for 2 x 150 bp at k = 20.  CM_PAIR_LDS_SLACK=0 gives 72-byte rows and
    // takes k_pair_heavy's LDS per wave from 12 to 11 allocation steps of 1 280 bytes: its kernel time -4 %, the step +1 %: NOTES 44)
    static const int str_slack = getenv("CM_PAIR_LDS_SLACK") ? atoi(getenv("CM_PAIR_LDS_SLACK")) : 4;
    const int str_cap = ((ctx->max_len - ctx->P.kmer + ctx->P.band + str_slack + 7) / 8) * 8;
    // The pair kernels' time hardly depends on their occupancy (16.0 / 16.2 / 16.4 ms per step at 4 / 3 / 2 waves per SIMD on the
    // hg38-like bench; 22 ms at 1).  CM_PAIR_OCC = 1..3 pads their LDS request to hold them at that many waves per SIMD, which
    // leaves registers and wave slots to the seeding / chaining of the next round; measured best overall: no padding (4).
    static const int pair_waves = getenv("CM_PAIR_OCC") ? atoi(getenv("CM_PAIR_OCC")) : 4;
    const size_t lds_need = (size_t)2 * lbuf_bytes(str_cap) * BLK_PAIR;
    size_t lds_bytes = lds_need;
    if (pair_waves >= 1 && pair_waves <= 3) {
        const size_t want = ((size_t)160 * 1024 / (size_t)(4 * pair_waves)) - HG * sizeof(HSlot) - 256;      // heavy adds its slots; keep clear of the next step
        if (want > lds_bytes && want <= 60 * 1024) lds_bytes = want;
    }
    const size_t lds_heavy = lds_bytes + HG * sizeof(HSlot);
    // the re-run launch of k_pair (RetryArgs): staging buffers for strings of any length a read of this batch can produce
    const int cap2 = std::min(((2 * ctx->max_len + 64 + 7) / 8) * 8, 1016);      // 1016: 64 KB of LDS per wave
    const size_t lds2 = (size_t)2 * lbuf_bytes(cap2) * BLK_PAIR;
    {   // dynamic-LDS limits of the three kernels: process-wide properties, only ever raised (see run_chain_tile)
        static std::atomic<size_t> lim[3] = {{48u * 1024u}, {48u * 1024u}, {48u * 1024u}};
        const void *fn[3] = {(const void *)k_pair, (const void *)k_pair_rerun, (const void *)k_pair_heavy};
        const size_t want[3] = {lds_bytes, lds2, lds_heavy};
        for (int k = 0; k < 3; ++k)
            if (want[k] > lim[k].load()) {
                HIPCHK(ctx, hipFuncSetAttribute(fn[k], hipFuncAttributeMaxDynamicSharedMemorySize, (int)want[k]));
                lim[k].store(want[k]);
            }
    }
    {
        Timer t(ctx, 5, so);
        const uint32_t nbk = (nt + CLS_T - 1) / CLS_T;
        static const bool fixed_cost = getenv("CM_HEAVY_COST") != nullptr;             // tuning knob: no adaptive threshold
        static const int heavy_cost = fixed_cost ? atoi(getenv("CM_HEAVY_COST")) : HEAVY_COST;
        unsigned long long *load = nullptr;
        if (!fixed_cost) {
            load = ctx->d_heavy_load;
            HIPCHK(ctx, hipMemsetAsync(load, 0, sizeof(unsigned long long), so));
            hipLaunchKernelGGL(k_pair_cost, dim3((nt + BLK - 1) / BLK), dim3(BLK), 0, so, rb.nchain, act_in, p0, nt, HEAVY_COST, load);
            // the host learns the load of a tile one or two items late (no wait): good enough to size the next heavy grid
            HIPCHK(ctx, hipMemcpyAsync((void *)(ctx->h_pin + 4), load, sizeof(unsigned long long), hipMemcpyDeviceToHost, so));
            ctx->h_pin_nt = nt;
        }
        hipLaunchKernelGGL(k_pair_cls, dim3((nt + BLK - 1) / BLK), dim3(BLK), 0, so, core, rb.chains, rb.resid, rb.nchain, act_in, p0, nt, ctx->d_cls,
                           ctx->d_cat, heavy_cost, ctx->d_cls_sub, ctx->d_cls_sub2, act_out, (const unsigned long long *)load);
        // three-pass LSD radix sort, 16 x 16 x 16 classes: by the longest residual, by the set of extensions a pair needs,
        // then (stable) by its class
        const uint32_t *no_order = nullptr;
        const unsigned int *no_count = nullptr;
        hipLaunchKernelGGL(k_cls_hist, dim3(nbk), dim3(CLS_W), 0, so, ctx->d_cls_sub2, nt, ctx->d_blk_cnt, nbk, no_order, no_count);
        hipLaunchKernelGGL(k_cls_scan, dim3(1), dim3(SCAN_CLS_T), 0, so, ctx->d_blk_cnt, nbk, ctx->d_cls_ctr3, -1, N_CLS);
        hipLaunchKernelGGL(k_cls_place, dim3(nbk), dim3(CLS_W), 0, so, ctx->d_cls_sub2, nt, ctx->d_blk_cnt, nbk, ctx->d_cls_ctr3, ctx->d_perm0,
                           (uint32_t *)nullptr, no_order, no_count);
        hipLaunchKernelGGL(k_cls_hist, dim3(nbk), dim3(CLS_W), 0, so, ctx->d_cls_sub, nt, ctx->d_blk_cnt, nbk, (const uint32_t *)ctx->d_perm0,
                           (const unsigned int *)(ctx->d_cls_ctr3 + CTR_SUM));
        hipLaunchKernelGGL(k_cls_scan, dim3(1), dim3(SCAN_CLS_T), 0, so, ctx->d_blk_cnt, nbk, ctx->d_cls_ctr2, -1, N_CLS);
        hipLaunchKernelGGL(k_cls_place, dim3(nbk), dim3(CLS_W), 0, so, ctx->d_cls_sub, nt, ctx->d_blk_cnt, nbk, ctx->d_cls_ctr2, ctx->d_perm1,
                           (uint32_t *)nullptr, (const uint32_t *)ctx->d_perm0, (const unsigned int *)(ctx->d_cls_ctr3 + CTR_SUM));
        hipLaunchKernelGGL(k_cls_hist, dim3(nbk), dim3(CLS_W), 0, so, ctx->d_cls, nt, ctx->d_blk_cnt, nbk, (const uint32_t *)ctx->d_perm1,
                           (const unsigned int *)(ctx->d_cls_ctr2 + CTR_SUM));
        hipLaunchKernelGGL(k_cls_scan, dim3(1), dim3(SCAN_CLS_T), 0, so, ctx->d_blk_cnt, nbk, cls_ctr, 1 << HEAVY_CLS, N_CLS);
        hipLaunchKernelGGL(k_cls_place, dim3(nbk), dim3(CLS_W), 0, so, ctx->d_cls, nt, ctx->d_blk_cnt, nbk, cls_ctr, perm,
                           hlist, (const uint32_t *)ctx->d_perm1, (const unsigned int *)(ctx->d_cls_ctr2 + CTR_SUM));
        ctx->launches[5] += 10;
    }
    HIPCHK(ctx, hipMemsetAsync(cls_ctr + CTR_NEXT, 0, 2 * sizeof(unsigned int), so));     // both work cursors
    HIPCHK(ctx, hipMemsetAsync(retry_ctr, 0, 2 * sizeof(unsigned int), so));                     // re-run count + cursor of this set
    HIPCHK(ctx, hipEventRecord(ctx->ev_order[b], so));
    HIPCHK(ctx, hipStreamWaitEvent(sp, ctx->ev_prep[b], 0));
    HIPCHK(ctx, hipStreamWaitEvent(sp, ctx->ev_order[b], 0));
    const RetryArgs ra1{pair_err, retry_list, retry_ctr, nullptr, 0, 1};
    {   // what the late launches of this stage need (settle_pair: the re-run, the pipeline's fall-back)
        cm_ctx::Rerun &q = ctx->rerun[b];
        q.core = core; q.rd = rd; q.p0 = p0; q.nt = nt;
        q.chains = rb.chains; q.nchain = rb.nchain; q.high = rb.high;
        q.act_out = act_out; q.is_last = is_last_round; q.cap2 = cap2; q.lds2 = lds2;
        q.pair_err = pair_err; q.retry_list = retry_list; q.retry_ctr = retry_ctr;
        q.fall = false;
    }
    // The heavy pairs go to a second stream: their kernels fit into the slots the light kernel leaves instead of queueing behind it.
    // That stream waits for this item's chains and lists itself, not for the light stream (which sits at the lowest priority, behind the
    // light kernel of the item before): what the heavy kernels share with the previous stage's light kernel exists once per set (re-run
    // list, cursors) or per tile (states, flags: another tile's, or ordered behind ev_pair through the ordering stream).
    HIPCHK(ctx, hipStreamWaitEvent(sp2, ctx->ev_prep[b], 0));
    HIPCHK(ctx, hipStreamWaitEvent(sp2, ctx->ev_order[b], 0));
    {
        Timer t(ctx, 4, sp2);
        // both pair kernels are persistent: together they fill `pair_waves` wave slots per SIMD (256 CUs x 4 SIMDs), half each
        static const unsigned slots_per_simd = (pair_waves >= 1 && pair_waves <= 3) ? (unsigned)pair_waves : 4u;
        const unsigned cap = 256u * 4u * slots_per_simd;
        // Heavy grid: half the wave capacity, or all of it when the tiles of this run carry a large heavy load (the value
        // k_pair_cost left for an earlier item, read without waiting): on the dense genome the heavy kernel has work for every
        // slot (19.9 vs 18.9 M pairs/s); with little heavy work the extra waves only sit on registers the next item's chain
        // stage is waiting for (round-2 genome: 43.3 vs 44.4 M pairs/s).
        static const unsigned heavy_div_env = getenv("CM_HEAVY_DIV") ? (unsigned)atoi(getenv("CM_HEAVY_DIV")) : 0u;      // tuning knob
        const volatile unsigned long long *seen = (const volatile unsigned long long *)(ctx->h_pin + 4);
        const bool loaded_run = ctx->h_pin_nt && *seen > HEAVY_LOAD * (unsigned long long)ctx->h_pin_nt;
        const unsigned heavy_div = heavy_div_env ? heavy_div_env : (loaded_run ? 1u : 2u);
        const unsigned heavy_cap = cap / heavy_div;
        static const unsigned heavy_fix = getenv("CM_HEAVY_GRID") ? (unsigned)atoi(getenv("CM_HEAVY_GRID")) : 0u;    // tuning knob
        const unsigned heavy_lim = std::min(heavy_fix ? heavy_fix : heavy_cap, HEAVY_GRID_MAX);     // d_hres is sized for HEAVY_GRID_MAX blocks
        const unsigned heavy_grid = nt < heavy_lim ? (nt ? nt : 1u) : heavy_lim;
        if (heavy_pipeline() && ctx->P.band == 3) {
            // the heavy pairs as a pipeline of full-width kernels (cm_heavy_pipe.h); what does not fit its arrays comes back in a
            // fall-back list and goes through k_pair_heavy behind it
            const HPipe hp{ctx->d_hp, ctx->d_hp_list2, ctx->d_hp_fall + (size_t)b * ctx->tile, ctx->d_hp_T, ctx->d_hp_pre, ctx->d_hp_q, ctx->d_hp_res, ctx->d_hp_U, ctx->d_hp_pre2,
                           ctx->d_hp_q2, ctx->d_hp_ctr, ctx->hp_tasks_cap, ctx->hp_unp_cap, ctx->d_hp_fallctr + b, ctx->d_hp_tcls};
            const unsigned int *n_heavy = cls_ctr + HEAVY_CLS, *n_list2 = ctx->d_hp_ctr + HC_LIST2;
            static const unsigned pipe_grid = getenv("CM_HP_GRID") ? (unsigned)atoi(getenv("CM_HP_GRID")) : 2048u;        // tuning knob: workgroups of the item kernels
            const size_t lds_slots = HG * sizeof(HSlot);
            HIPCHK(ctx, hipMemsetAsync(ctx->d_hp_ctr, 0, HC_WORDS * sizeof(unsigned int), sp2));
            HIPCHK(ctx, hipMemsetAsync(ctx->d_hp_fallctr + b, 0, sizeof(unsigned int), sp2));
            // Two passes (process_read's two attempts), the second over the few pairs whose other orientation has chains at all (k_hp_finish
            // settles the others in place).  CM_HP_ATTEMPTS=1 (diagnostic) sends those pairs whole to the fall-back kernel instead: its
            // long tail over a few hundred heavy pairs costs more than nine short launches (77.9 vs 75.9 ms per step).
            static const bool task_order = !(getenv("CM_HP_TASK_ORDER") && getenv("CM_HP_TASK_ORDER")[0] == '0');      // diagnostic: tasks in array order
            static const int n_attempts = (getenv("CM_HP_ATTEMPTS") && atoi(getenv("CM_HP_ATTEMPTS")) == 1) ? 1 : 2;
            for (int attempt = 0; attempt < n_attempts; ++attempt) {
                if (attempt) hipLaunchKernelGGL(k_hp_reset, dim3(1), dim3(64), 0, sp2, ctx->d_hp_ctr);
                hipLaunchKernelGGL(k_hp_plan, dim3(HP_PLAN_GRID), dim3(BLK_PAIR), lds_slots, sp2, core, rd, p0, (const uint32_t *)hlist, n_heavy,
                                   (const uint32_t *)ctx->d_hp_list2, n_list2, attempt, (const cm_chain *)rb.chains, (const int32_t *)rb.nchain,
                                   (const int32_t *)rb.high, (const cm_mapped_read *)ctx->d_state, hp, ctx->d_hp_lists, str_cap);
                hipLaunchKernelGGL(k_hp_dp, dim3(pipe_grid), dim3(BLK_PAIR), lds_bytes, sp2, core, rd, p0, attempt, hp, 0, str_cap);
                if (task_order && attempt == 0) {       // (the second attempt's handful: array order) the tasks by work class, heaviest first (16-class counting sort over the tile's task array)
                    const uint32_t nbt = (ctx->hp_tasks_cap + CLS_T - 1) / CLS_T;
                    const unsigned int *n_t = ctx->d_hp_ctr + HC_TASKS;
                    const uint32_t gt = std::min<uint32_t>(nbt, 2048u);        // (the count is on the device: a grid that walks the stretches in use)
                    hipLaunchKernelGGL(k_cls_hist, dim3(gt), dim3(CLS_W), 0, sp2, (const int8_t *)ctx->d_hp_tcls, ctx->hp_tasks_cap, ctx->d_hp_tblk, nbt,
                                       (const uint32_t *)nullptr, n_t);
                    hipLaunchKernelGGL(k_cls_scan, dim3(1), dim3(SCAN_CLS_T), 0, sp2, ctx->d_hp_tblk, nbt, ctx->d_hp_tctr, -1, N_CLS, n_t);
                    hipLaunchKernelGGL(k_cls_place, dim3(gt), dim3(CLS_W), 0, sp2, (const int8_t *)ctx->d_hp_tcls, ctx->hp_tasks_cap, ctx->d_hp_tblk, nbt,
                                       ctx->d_hp_tctr, ctx->d_hp_tperm, (uint32_t *)nullptr, (const uint32_t *)nullptr, n_t);
                }
                hipLaunchKernelGGL(k_hp_tasks, dim3(pipe_grid), dim3(BLK_PAIR), lds_bytes, sp2, core, rd, p0, attempt, (const cm_chain *)rb.chains,
                                   (const int32_t *)rb.nchain, hp, pair_err, str_cap, (task_order && attempt == 0) ? (const uint32_t *)ctx->d_hp_tperm : (const uint32_t *)nullptr,
                                   (const unsigned int *)(ctx->d_hp_tctr + CTR_SUM));
                hipLaunchKernelGGL(k_hp_fold, dim3(pipe_grid), dim3(BLK_PAIR), 0, sp2, core, p0, (const uint32_t *)ctx->d_hp_list2, n_list2, n_heavy, attempt, hp,
                                   ctx->d_counters);
                hipLaunchKernelGGL(k_hp_unp_req, dim3(pipe_grid), dim3(BLK_PAIR), 0, sp2, core, rd, p0, attempt, (const cm_chain *)rb.chains, hp, str_cap);
                hipLaunchKernelGGL(k_hp_dp, dim3(pipe_grid), dim3(BLK_PAIR), lds_bytes, sp2, core, rd, p0, attempt, hp, 1, str_cap);
                hipLaunchKernelGGL(k_hp_unp, dim3(pipe_grid), dim3(BLK_PAIR), lds_bytes, sp2, core, rd, p0, attempt, (const cm_chain *)rb.chains, hp, pair_err,
                                   str_cap);
                hipLaunchKernelGGL(k_hp_finish, dim3(pipe_grid), dim3(BLK_PAIR), 0, sp2, core, p0, (const uint32_t *)ctx->d_hp_list2, n_list2, n_heavy, attempt, hp,
                                   ctx->d_state, act_out, ctx->d_cat, is_last_round, ctx->d_counters, ra1, (const int32_t *)rb.nchain, n_attempts == 1 ? 1 : 0);
            }
            // what did not fit goes through k_pair_heavy: now (one tile), or when the stage is settled and the list is known to hold something
            // (its own work cursor, cls_ctr + CTR_NEXT + 1, was zeroed with the light kernel's)
            HIPCHK(ctx, hipMemcpyAsync((void *)(ctx->h_pin + 14 + b), ctx->d_hp_fallctr + b, sizeof(unsigned int), hipMemcpyDeviceToHost, sp2));
            {
                cm_ctx::Rerun &q = ctx->rerun[b];
                q.fall = true;
                q.fall_list = ctx->d_hp_fall + (size_t)b * ctx->tile;
                q.fall_ctr = ctx->d_hp_fallctr + b;
                q.fall_cursor = cls_ctr + CTR_NEXT + 1;
                q.lds_heavy = lds_heavy;
                q.str_cap = str_cap;
                q.fall_grid = std::min(heavy_grid, 256u);
            }
            if (!defer) {
                int rc;
                if ((rc = launch_fall_back(ctx, b, sp2))) return rc;
                ctx->rerun[b].fall = false;
            }
            ctx->launches[4] += 9 * n_attempts;
        } else {
        hipLaunchKernelGGL(k_pair_heavy, dim3(heavy_grid), dim3(BLK_PAIR), lds_heavy, sp2, core, rd, p0, hlist, cls_ctr + HEAVY_CLS, rb.chains,
                           rb.nchain, rb.high, ctx->d_state, act_out, ctx->d_cat, is_last_round, ctx->d_err, ctx->d_counters, str_cap,
                           ctx->d_lane_clk ? ctx->d_lane_clk + (size_t)nt * 16 + (size_t)(nt / 64 + 1) * 64 : nullptr, ctx->d_hres,
                           cls_ctr + CTR_NEXT + 1, ra1);
        ++ctx->launches[4];
        }
    }
    HIPCHK(ctx, hipEventRecord(ctx->ev_join_p, sp2));
    {
        Timer t(ctx, 2, sp);      // = the pair stage: the light kernel and the wait for the second stream
        static const unsigned slots_per_simd = (pair_waves >= 1 && pair_waves <= 3) ? (unsigned)pair_waves : 4u;
        static const unsigned light_fix = getenv("CM_PAIR_GRID") ? (unsigned)atoi(getenv("CM_PAIR_GRID")) : 0u;       // tuning knob
        const unsigned want = (nt + BLK_PAIR - 1) / BLK_PAIR, cap = light_fix ? light_fix
                                                                          : (heavy_pipeline() && ctx->P.band == 3 && slots_per_simd == 4u) ? 2048u      // the pipeline's kernels come and go: half the slots (78.7 -> 76.3 ms per step)
                                                                                                                                        : 256u * 4u * slots_per_simd;   // light takes the slots heavy leaves: full cap
        hipLaunchKernelGGL(k_pair, dim3(want < cap ? want : cap), dim3(BLK_PAIR), lds_bytes, sp, core, rd, p0, nt, rb.chains, rb.nchain, rb.high,
                           ctx->d_state, act_out, ctx->d_cat, is_last_round, ctx->d_err, ctx->d_counters, str_cap, ctx->d_lane_clk, perm,
                           cls_ctr + CTR_SUM, cls_ctr + CTR_NEXT, ra1);
        HIPCHK(ctx, hipStreamWaitEvent(sp, ctx->ev_join_p, 0));
        ++ctx->launches[2];
    }
    // The re-run of whatever the two kernels queued (usually nothing: the launch reads the count on the device and ends): the same
    // code over the re-run list, one pair per lane, memo spill area, staging buffers for strings of any length a read of this
    // batch can produce.  On a stream of its own: a launch of 8 blocks behind kernels that fill the chip can wait milliseconds
    // for its turn (2.5 ms on average on the hg38-like bench), and only the consumers of this item's results have to wait for
    // it -- ev_pair[b] (chain records of set b free, flags and states of the tile final) is recorded behind it.
    HIPCHK(ctx, hipMemcpyAsync((void *)(ctx->h_pin + 12 + b), retry_ctr, sizeof(unsigned int), hipMemcpyDeviceToHost, sp));
    HIPCHK(ctx, hipEventRecord(ctx->ev_first[b], sp));
    ctx->rerun[b].deferred = defer;
    if (!defer) {
        HIPCHK(ctx, hipStreamWaitEvent(sp3, ctx->ev_first[b], 0));
        int rc;
        if ((rc = launch_rerun(ctx, b))) return rc;
        HIPCHK(ctx, hipEventRecord(ctx->ev_pair[b], sp3));
    }
    ctx->pair_pending[b] = true;
    HIPCHK(ctx, hipGetLastError());
    return CM_OK;
}

// The rounds of cm_map_rounds; any failure (a HIP call, a stage) returns from here and is cleaned up by the caller below.
// *rounds_done counts the rounds whose pair stage was issued for every tile (the flags arrays have swapped roles that often).
static int map_rounds_issue(cm_ctx *ctx, const int *slots, int n_rounds, int last_is_final, int *items_done, int *rounds_done) {
    int rc;
    // Cross-batch prefetch (see the end of this function): possible when this call ends the batch and the next one is staged and fits
    // the workspace as it is sized now (nothing may be reallocated under the kernels in flight).
    static const bool prefetch_on = !(getenv("CM_PREFETCH") && getenv("CM_PREFETCH")[0] == '0');
    const bool prefetch = prefetch_on && last_is_final && ctx->staged && ctx->st_n_pairs <= ctx->n_pairs && ctx->st_max_len <= ctx->max_len;
    if (prefetch && ctx->ones_cap < ctx->st_n_pairs) {          // flags of a fresh batch: every pair active
        HIPCHK(ctx, ensure(ctx, ctx->d_ones, ctx->n_pairs));
        HIPCHK(ctx, hipMemsetAsync(ctx->d_ones, 1, ctx->n_pairs, ctx->stream));
        ctx->ones_cap = ctx->n_pairs;
    }
    // everything queued on the main stream so far (uploads, resets, collects of the previous batch) comes first
    HIPCHK(ctx, hipEventRecord(ctx->ev_tail, ctx->stream));
    HIPCHK(ctx, hipStreamWaitEvent(ctx->stream_p, ctx->ev_tail, 0));
    HIPCHK(ctx, hipStreamWaitEvent(ctx->stream_s, ctx->ev_tail, 0));
    HIPCHK(ctx, hipStreamWaitEvent(ctx->stream_o, ctx->ev_tail, 0));
    HIPCHK(ctx, hipEventRecord(ctx->ev_flags, ctx->stream));
    uint8_t *A[2] = {ctx->d_active, ctx->d_active_b};      // A[0] = flags before the first of these rounds
    // The work items: (tile, round).  One tile per batch: its rounds in order.  Several tiles: ROUND-major -- every tile through
    // round r, then every tile through round r + 1 -- so that between the pair stage of (tile, r) and the seeding of (tile, r + 1)
    // lies the work of the other tiles: the flags that pair stage wrote are final when the seeding starts, and seeds / chains are
    // computed for exactly the pairs still active (with one tile the seeding of round r + 1 runs UNDER the pair stage of round r
    // and has to take the flags from before it, a superset: 20 % more seeding and chaining on the hg38-like bench).
    struct Item { uint64_t p0; uint32_t nt; int r; };
    std::vector<Item> items;
    const uint64_t n_tiles = (ctx->n_pairs + ctx->tile - 1) / ctx->tile;
    static const bool tile_major_env = getenv("CM_TILE_MAJOR") && getenv("CM_TILE_MAJOR")[0] == '1';      // diagnostic: the round-2 order
    const bool round_major = n_tiles >= 2 && !tile_major_env;
    auto tile_nt = [&](uint64_t t) { return (uint32_t)((ctx->n_pairs - t * ctx->tile < ctx->tile) ? ctx->n_pairs - t * ctx->tile : ctx->tile); };
    if (round_major) {
        for (int r = 0; r < n_rounds; ++r)
            for (uint64_t t = 0; t < n_tiles; ++t) items.push_back(Item{t * ctx->tile, tile_nt(t), r});
    } else {
        for (uint64_t t = 0; t < n_tiles; ++t)
            for (int r = 0; r < n_rounds; ++r) items.push_back(Item{t * ctx->tile, tile_nt(t), r});
    }
    const int n_items = (int)items.size();
    std::vector<int> tiles_of_round((size_t)n_rounds, 0);
    // Was this batch's first item prepared while the previous batch was in its last pair stage (see the end of this function)?
    const bool use_pre = ctx->pre_ready && slots[0] == ctx->pre_slot && ctx->slots[slots[0]].gen == ctx->pre_gen && ctx->n_pairs == ctx->pre_n &&
                         items[0].nt == ctx->pre_nt && (ctx->item_base & 1) == ctx->pre_b;
    ctx->pre_ready = ctx->pre_launched = false;            // the items below reuse both sets of chain records
    if (use_pre) ++ctx->launches[7];
    const ReadsDev rd_cur = current_reads(ctx);
    // Seeds and chains of round r depend on the reads and the contig only; the flags merely skip pairs that are retired.
    // Round-major: A[r & 1], what the pair stage of (tile, r - 1) wrote -- item i - n_tiles, complete before item i - 2, for
    // which this item waits anyway (it reuses its chain records).  One tile: the pair stage of round r - 1 is still writing
    // A[r & 1], so the flags from before it (pairs it retires get chains nobody looks at).
    auto prep_flags = [&](int i) -> const uint8_t * {
        const int r = items[i].r;
        return round_major ? A[r & 1] : ((r == 0) ? A[0] : A[(r - 1) & 1]);
    };
    // Seeding runs one item ahead of chaining.  The chain stage of an item ends with a long tail of a few heavy problems, one
    // wave each, and the host waits for it (the pair stage must not start on truncated improvement logs); seeding of the next
    // item used to start after that wait, with the chip nearly idle through the tail and the next chain stage waiting for the
    // seeds.  Now seeding of item i + 1 is issued (stream_s, seed set (i + 1) & 1) as soon as the chain kernels of item i are
    // launched: it needs the flags of the pair stage of item i - 1 (two tiles; earlier with more), which ends during that chain
    // stage, so the seeds are computed under the tail and the next chain stage starts right behind this one.
    static const bool seed_ahead = !(getenv("CM_SEED_AHEAD") && getenv("CM_SEED_AHEAD")[0] == '0');       // diagnostic: the round-3a order
    std::vector<char> seeded((size_t)n_items, 0);
    bool pre_seeded = false;
    // With three or more tiles the flags item i + 1 reads were written by the pair stage of item i - 2 or earlier, complete before the
    // chain stage of item i starts: the seeds (not their classes, which go into chain records item i - 1's pair stage still reads) are
    // then computed under that chain stage instead of behind the pair stage of item i - 1.  Two tiles: seeding + chaining (8 + 14 ms at
    // 2^21 pairs) sat between the end of one pair stage and the start of the next but one, longer than the pair stage between them.
    // Two tiles: the flags are the ones the pair kernels of item i - 1 write, and the host cannot issue anything when those end -- it
    // sits in the wait for the chain stage of item i, which ends later: the seeding of item i + 1 started when THAT was over, and the
    // chip idled for 5 - 6 ms of every 22-ms item with nothing but the tail of k_chain_heavy on it.  The seeds are queued on the device
    // behind ev_first of that pair stage instead (its kernels and the heavy pairs' pipeline; a pair left to a late launch -- re-run,
    // fall-back -- counts as active until that launch writes its flag, a superset, and the classes below use the final flags).
    const bool early_ok = round_major && n_tiles >= 2 && !(getenv("CM_SEED_EARLY") && getenv("CM_SEED_EARLY")[0] == '0');
    auto issue_seed_early = [&](int i) -> int {
        HIPCHK(ctx, hipStreamWaitEvent(ctx->stream_s, ctx->ev_flags, 0));
        if (n_tiles == 2 && i >= 2) HIPCHK(ctx, hipStreamWaitEvent(ctx->stream_s, ctx->ev_first[(ctx->item_base + i) & 1], 0));
        // the whole seed stage: the classes too, into scratch of the seed set (the chain records they belong in are still being read;
        // run_chain_tile carries them over).  Under superset flags a pair the late launches retire gets chains nobody looks at.
        const RoundBufs rbe = round_bufs(ctx, (ctx->item_base + i) & 1);
        const int e = run_seed_tile(ctx, make_core(ctx, ctx->slots[slots[items[i].r]]), rd_cur, items[i].p0, items[i].nt, prep_flags(i), i & 1, ctx->stream_s, &rbe, true);
        seeded[(size_t)i] = 1;
        return e;
    };
    auto issue_seed = [&](int i, hipStream_t st) -> int {
        const int b = (ctx->item_base + i) & 1;
        // flags (and, for the chain stage behind it, the chain records of set b) are final once that pair stage is done; the main
        // stream queues the same wait before the chain stage and clears the mark
        int e;
        if ((e = settle_pair(ctx, b))) return e;
        if (ctx->pair_pending[b]) HIPCHK(ctx, hipStreamWaitEvent(st, ctx->ev_pair[b], 0));
        const RoundBufs rbi = round_bufs(ctx, b);
        e = run_seed_tile(ctx, make_core(ctx, ctx->slots[slots[items[i].r]]), rd_cur, items[i].p0, items[i].nt, prep_flags(i), i & 1, st, &rbi);
        seeded[(size_t)i] = 1;
        return e;
    };
    for (int i = 0; i < n_items; ++i) {
        const uint64_t p0 = items[i].p0;
        const uint32_t nt = items[i].nt;
        const int r = items[i].r, b = (ctx->item_base + i) & 1;
        const Slot &sl = ctx->slots[slots[r]];
        const KCore core = make_core(ctx, sl);
        const RoundBufs rb = round_bufs(ctx, b);
        const uint8_t *act_prep = prep_flags(i);
        // ... if the pair stage whose flags it reads (item i - 1 with two tiles) is over by then.  Otherwise the host would sit in
        // settle_pair until it is, and this item's pair stage -- whose work classes and lists can be computed under that same stage
        // (stream_o) -- would be issued late: then the seeding is issued behind this item's pair stage instead (`late`).
        auto ahead = [&]() -> int {
            if (!seed_ahead || i + 1 >= n_items || seeded[(size_t)i + 1]) return CM_OK;
            if (early_ok) return issue_seed_early(i + 1);      // queued on the device behind what it depends on: nothing left for later
            const int bn = (ctx->item_base + i + 1) & 1;
            if (ctx->rerun[bn].deferred && hipEventQuery(ctx->ev_first[bn]) != hipSuccess) {
                (void)hipGetLastError();                                  // not ready
                return CM_OK;
            }
            return issue_seed(i + 1, ctx->stream_s);
        };
        auto late = [&]() -> int { return (seed_ahead && i + 1 < n_items && !seeded[(size_t)i + 1]) ? issue_seed(i + 1, ctx->stream_s) : CM_OK; };
        // The last item has nothing of this batch to seed ahead: the seeds of the NEXT batch's first item instead (the prefetch below
        // then starts with its chain stage; without this the prefetched chains ended 3.7 ms after the batch's last pair stage and
        // held up the hand-over).  Seeds only: the chain records they will be chained into are still being read.
        if (i == n_items - 1 && prefetch && seed_ahead && !pre_seeded) {
            const Slot &sl0 = ctx->slots[slots[0]];
            const uint32_t nt0 = tile_for(ctx->st_n_pairs);                     // = that batch's first tile (<= this batch's: it has no more pairs)
            const ReadsDev rd_next{ctx->st_seq1_base + cmc::CM_STAGE_PAD, ctx->st_seq2_base + cmc::CM_STAGE_PAD, ctx->st_off1, ctx->st_off2};
            HIPCHK(ctx, hipStreamWaitEvent(ctx->stream_s, ctx->ev_staged, 0));
            if ((rc = run_seed_tile(ctx, make_core(ctx, sl0), rd_next, 0, nt0, ctx->d_ones, n_items & 1, ctx->stream_s, nullptr))) return rc;
            pre_seeded = true;
        }
        if (use_pre && i == 0) {                                          // set b holds this item's chains, ev_prep[b] is recorded
            if ((rc = ahead())) return rc;
        } else {
            if (!seeded[(size_t)i] && (rc = issue_seed(i, ctx->stream))) return rc;
            if ((rc = settle_pair(ctx, b))) return rc;
            if (ctx->pair_pending[b]) {                                   // chain buffers of set b: free once their pair stage is done
                HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_pair[b], 0));
                ctx->pair_pending[b] = false;
            }
            HIPCHK(ctx, hipEventRecord(ctx->ev_flags, ctx->stream));      // (pair stage of item i - 2 and everything before it)
            // CM_CHAIN_EXACT=1 (diagnostic, one tile): the chain kernels wait for the pair stage of round r - 1 and use its output
            // flags (chain kernels 8.4 -> 6.7 ms per step, step 24.0 -> 25.1 ms: the wait costs more than the work it saves)
            static const bool exact_flags = getenv("CM_CHAIN_EXACT") && getenv("CM_CHAIN_EXACT")[0] == '1';
            const bool wait_exact = exact_flags && !round_major && r > 0;
            if (wait_exact && (rc = settle_pair(ctx, b ^ 1))) return rc;
            if (wait_exact && ctx->pair_pending[b ^ 1]) {                 // item - 1 = the same tile's round r - 1
                HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_pair[b ^ 1], 0));
                ctx->pair_pending[b ^ 1] = false;
            }
            if ((rc = run_chain_tile(ctx, core, rd_cur, p0, nt, sl.chain_parallel_ok, wait_exact ? A[r & 1] : act_prep, rb, i & 1, ahead))) return rc;
            HIPCHK(ctx, hipEventRecord(ctx->ev_prep[b], ctx->stream));
        }
        const int is_last = (r == n_rounds - 1) ? (last_is_final != 0) : 0;
        const bool same_tile = i > 0 && items[i - 1].p0 == p0;
        if ((rc = run_pair_tile(ctx, core, p0, nt, is_last, A[r & 1], A[(r + 1) & 1], rb, b, same_tile, round_major))) return rc;
        if ((rc = late())) return rc;
        ++*items_done;
        if (++tiles_of_round[(size_t)r] == (int)n_tiles) ++*rounds_done;
        static const bool no_overlap = getenv("CM_PIPELINE") && getenv("CM_PIPELINE")[0] == '0';      // diagnostic: items back to back
        if (no_overlap) {
            if ((rc = settle_pair(ctx, b))) return rc;
            HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_pair[b], 0));
        }
    }
    ctx->item_base = (ctx->item_base + n_items) & 1;
    *items_done = 0;                                   // accounted for
    // Cross-batch prefetch.  A batch's first item cannot hide behind one of its own pair stages, and its last pair stage has no
    // later item of its own to cover.  So when this call ends the batch and the next one is already staged (cm_reads_stage),
    // that batch's first item (first tile, slots[0]) is seeded and chained now, from the staging buffers (every pair of a fresh
    // batch is active), into the set of chain records the running pair stage does not read.  cm_reads_swap keeps the result;
    // the next cm_map_rounds uses it if it starts with the same slot, still holding the same contig.  Condition: the staged
    // batch fits the workspace as it is sized now (nothing may be reallocated under the kernels in flight).
    if (prefetch) {
        const int b = ctx->item_base;
        const Slot &sl = ctx->slots[slots[0]];
        const KCore core = make_core(ctx, sl);
        const uint32_t nt = tile_for(ctx->st_n_pairs);     // = that batch's first tile (prepare_resident)
        HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_staged, 0));
        if ((rc = settle_pair(ctx, b))) return rc;
        if (ctx->pair_pending[b]) {
            HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_pair[b], 0));
            ctx->pair_pending[b] = false;
        }
        const RoundBufs rbn = round_bufs(ctx, b);
        const ReadsDev rd_next{ctx->st_seq1_base + cmc::CM_STAGE_PAD, ctx->st_seq2_base + cmc::CM_STAGE_PAD, ctx->st_off1, ctx->st_off2};
        if (pre_seeded) {                                      // seeds are there (or on their way): the classes, into the records now free
            HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_seed[n_items & 1], 0));
            if ((rc = seed_classes(ctx, 0, nt, ctx->d_ones, n_items & 1, ctx->stream, &rbn))) return rc;
        } else if ((rc = run_seed_tile(ctx, core, rd_next, 0, nt, ctx->d_ones, n_items & 1, ctx->stream, &rbn))) return rc;
        if ((rc = run_chain_tile(ctx, core, rd_next, 0, nt, sl.chain_parallel_ok, ctx->d_ones, rbn, n_items & 1))) return rc;
        HIPCHK(ctx, hipEventRecord(ctx->ev_prep[b], ctx->stream));
        ctx->pre_launched = true;
        ctx->pre_slot = slots[0];
        ctx->pre_gen = sl.gen;
        ctx->pre_b = b;
        ctx->pre_nt = nt;
    }
    // later work on the main stream (downloads, collects, the next batch) is ordered behind the last pair stage on the device
    // (set item_base ^ 1 = the last item's goes second: stream p3's last entry then waits for stream p's last)
    if ((rc = settle_pair(ctx, ctx->item_base, false)) || (rc = settle_pair(ctx, ctx->item_base ^ 1, false))) return rc;
    HIPCHK(ctx, hipEventRecord(ctx->ev_tail, ctx->stream_p3));               // (p3's last launch waits for p's last)
    HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_tail, 0));
    ctx->pair_pending[0] = ctx->pair_pending[1] = false;                     // covered by the wait above
    if (n_rounds & 1) std::swap(ctx->d_active, ctx->d_active_b);             // the current flags are in the other array now
    *rounds_done = 0;                                  // accounted for
    return CM_OK;
}

int cm_map_rounds(cm_ctx *ctx, const int *slots, int n_rounds, int last_is_final) {
    if (!ctx || (n_rounds > 0 && !slots) || n_rounds < 0) return CM_EINVAL;
    int rc;
    for (int r = 0; r < n_rounds; ++r)
        if ((rc = check_slot(ctx, slots[r], true))) return rc;
    HIPCHK(ctx, hipSetDevice(ctx->P.device));
    if (ctx->n_pairs == 0 || n_rounds == 0) return CM_OK;
    int items_done = 0, rounds_done = 0;
    rc = map_rounds_issue(ctx, slots, n_rounds, last_is_final, &items_done, &rounds_done);
    if (rc == CM_OK) return rc;
    // One exit for every failure inside: nothing of this call may still be running when the caller frees or reuses the batch
    // buffers, and the bookkeeping must describe what was actually issued -- the chain-record set parity (item_base), the
    // flags array that holds the latest flags (one swap per completed round), no prepared first item, no pending pair stage.
    // The states of the batch are unspecified after a failed call (upload or reset before mapping again).
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipStreamSynchronize(ctx->stream2);
    (void)hipStreamSynchronize(ctx->stream_p);
    (void)hipStreamSynchronize(ctx->stream_p2);
    (void)hipStreamSynchronize(ctx->stream_o);
    (void)hipStreamSynchronize(ctx->stream_p3);
    (void)hipStreamSynchronize(ctx->stream_s);
    ctx->rerun[0].deferred = ctx->rerun[1].deferred = false;
    ctx->cls_deferred[0] = ctx->cls_deferred[1] = false;
    ctx->item_base = (ctx->item_base + items_done) & 1;
    if (rounds_done & 1) std::swap(ctx->d_active, ctx->d_active_b);
    ctx->pair_pending[0] = ctx->pair_pending[1] = false;
    ctx->pre_ready = ctx->pre_launched = false;
    return rc;
}

int cm_map_round(cm_ctx *ctx, int slot, int is_last_round) { return cm_map_rounds(ctx, &slot, 1, is_last_round); }

int cm_sync(cm_ctx *ctx) {
    if (!ctx) return CM_EINVAL;
    HIPCHK(ctx, hipSetDevice(ctx->P.device));
    // the main stream is ordered behind every other stream's work at the end of each call (ev_tail); the staging copy of
    // cm_reads_stage is the one thing a caller can have in flight beside it
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream_copy));
    return check_dev_err(ctx);
}

int cm_reads_reset(cm_ctx *ctx) {
    if (!ctx) return CM_EINVAL;
    HIPCHK(ctx, hipSetDevice(ctx->P.device));
    if (ctx->n_pairs == 0) return CM_OK;
    KCore k{};
    k.P = ctx->P;
    hipLaunchKernelGGL(k_init_state, dim3((unsigned)((ctx->n_pairs + BLK - 1) / BLK)), dim3(BLK), 0, ctx->stream, k, ctx->d_state, ctx->d_active,
                       ctx->d_cat, ctx->n_pairs);
    HIPCHK(ctx, hipGetLastError());
    return CM_OK;
}

#ifdef CM_TRACE_HOST
#define TRACE_T0 auto tr_t = std::chrono::steady_clock::now()
#define TRACE_PT(name) do { auto n_ = std::chrono::steady_clock::now(); fprintf(stderr, "[trace] %s %.1f us\n", name, std::chrono::duration<double, std::micro>(n_ - tr_t).count()); tr_t = n_; } while (0)
#else
#define TRACE_T0
#define TRACE_PT(name)
#endif
// stable compaction of the active pairs (block histogram -> scan -> place): ascending pair index, no atomics, no
// host sort; leaves the permutation in d_col_perm and the count in d_col_ctr[0]
static int compact_active(cm_ctx *ctx) {
    const uint64_t n = ctx->n_pairs;
    if (n > 0xfffffff0ull) return fail(ctx, CM_ELIMIT, "cm_collect_*: too many pairs");
    const uint32_t nbk = (uint32_t)((n + CLS_T - 1) / CLS_T);
    // scratch sized for the whole batch, made on first use
    HIPCHK(ctx, ensure(ctx, ctx->d_col_cls, n));
    HIPCHK(ctx, ensure(ctx, ctx->d_col_perm, n * 4));
    HIPCHK(ctx, ensure(ctx, ctx->d_col_blk, (size_t)N_CLS * (nbk + 2) * sizeof(unsigned int)));
    HIPCHK(ctx, ensure(ctx, ctx->d_col_ctr, CTR_WORDS * sizeof(unsigned int)));
    hipLaunchKernelGGL(k_active_cls, dim3((unsigned)((n + BLK - 1) / BLK)), dim3(BLK), 0, ctx->stream, ctx->d_active, n, ctx->d_col_cls);
    hipLaunchKernelGGL(k_cls_hist, dim3(nbk), dim3(CLS_W), 0, ctx->stream, ctx->d_col_cls, (uint32_t)n, ctx->d_col_blk, nbk, (const uint32_t *)nullptr,
                       (const unsigned int *)nullptr);
    hipLaunchKernelGGL(k_cls_scan, dim3(1), dim3(SCAN_CLS_T), 0, ctx->stream, ctx->d_col_blk, nbk, ctx->d_col_ctr, -1, 1);
    hipLaunchKernelGGL(k_cls_place, dim3(nbk), dim3(CLS_W), 0, ctx->stream, ctx->d_col_cls, (uint32_t)n, ctx->d_col_blk, nbk, ctx->d_col_ctr,
                       ctx->d_col_perm, (uint32_t *)nullptr, (const uint32_t *)nullptr, (const unsigned int *)nullptr);
    return CM_OK;
}
// count + error flags through the pinned landing zone (one synchronisation)
static int read_count(cm_ctx *ctx, uint64_t cap, unsigned int *cnt, const char *what) {
    HIPCHK(ctx, hipMemcpyAsync(ctx->h_pin, ctx->d_col_ctr, sizeof(unsigned int), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(ctx->h_pin + 1, ctx->d_err, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    *cnt = *(const unsigned int *)ctx->h_pin;
    if (*(const int *)(ctx->h_pin + 1)) return check_dev_err(ctx);
    if (*cnt > cap) return fail(ctx, CM_ELIMIT, "%s: %u active pairs > cap %llu", what, *cnt, (unsigned long long)cap);
    return CM_OK;
}

int cm_collect_active(cm_ctx *ctx, uint64_t cap, uint64_t *out_idx, cm_mapped_read *out_state, uint64_t *out_n) {
    if (!ctx || !out_n || (cap && (!out_idx || !out_state))) return CM_EINVAL;
    HIPCHK(ctx, hipSetDevice(ctx->P.device));
    *out_n = 0;
    if (ctx->n_pairs == 0) return CM_OK;
    int rc = compact_active(ctx);
    if (rc) return rc;
    if (cap > ctx->collect_cap) {                 // grow-only output staging
        dfree(ctx, ctx->d_collect_idx);
        dfree(ctx, ctx->d_collect_st);
        ctx->collect_cap = 0;
        HIPCHK(ctx, hipMalloc((void **)&ctx->d_collect_idx, cap * sizeof(unsigned long long)));
        HIPCHK(ctx, hipMalloc((void **)&ctx->d_collect_st, cap * sizeof(cm_mapped_read)));
        ctx->collect_cap = cap;
    }
    if (cap)
        hipLaunchKernelGGL(k_gather_active, dim3((unsigned)((cap + BLK - 1) / BLK)), dim3(BLK), 0, ctx->stream, ctx->d_col_perm, ctx->d_col_ctr,
                           (unsigned long long)cap, ctx->d_state, ctx->d_collect_idx, ctx->d_collect_st);
    unsigned int cnt = 0;
    rc = read_count(ctx, cap, &cnt, "cm_collect_active");
    *out_n = cnt;
    if (rc) return rc;
    if (cnt) {
        HIPCHK(ctx, hipMemcpyAsync(out_idx, ctx->d_collect_idx, (size_t)cnt * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipMemcpyAsync(out_state, ctx->d_collect_st, (size_t)cnt * sizeof(cm_mapped_read), hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    }
    return CM_OK;
}

// compaction + record assembly into `d_dst` (device memory, cap records); complete on return
static int collect_records_into(cm_ctx *ctx, uint64_t index_base, uint64_t cap, cm_record *d_dst, uint64_t *out_n, const char *what) {
    *out_n = 0;
    if (ctx->n_pairs == 0) return CM_OK;
    int rc = compact_active(ctx);
    if (rc) return rc;
    if (cap)
        hipLaunchKernelGGL(k_gather_records, dim3((unsigned)((cap + BLK - 1) / BLK)), dim3(BLK), 0, ctx->stream, ctx->d_col_perm, ctx->d_col_ctr,
                           (unsigned long long)cap, ctx->d_state, (unsigned long long)index_base, d_dst);
    unsigned int cnt = 0;
    rc = read_count(ctx, cap, &cnt, what);
    *out_n = cnt;
    return rc;
}

int cm_collect_records(cm_ctx *ctx, uint64_t index_base, uint64_t cap, cm_record *out, uint64_t *out_n) {
    if (!ctx || !out_n || (cap && !out)) return CM_EINVAL;
    HIPCHK(ctx, hipSetDevice(ctx->P.device));
    if (cap > ctx->collect_rec_cap) {
        dfree(ctx, ctx->d_collect_rec);
        ctx->collect_rec_cap = 0;
        HIPCHK(ctx, hipMalloc((void **)&ctx->d_collect_rec, cap * sizeof(cm_record)));
        ctx->collect_rec_cap = cap;
    }
    int rc = collect_records_into(ctx, index_base, cap, ctx->d_collect_rec, out_n, "cm_collect_records");
    if (rc) return rc;
    if (*out_n) {
        HIPCHK(ctx, hipMemcpyAsync(out, ctx->d_collect_rec, (size_t)*out_n * sizeof(cm_record), hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    }
    return CM_OK;
}

int cm_collect_records_device(cm_ctx *ctx, uint64_t index_base, uint64_t cap, void *d_out, uint64_t *out_n) {
    if (!ctx || !out_n || (cap && !d_out)) return CM_EINVAL;
    HIPCHK(ctx, hipSetDevice(ctx->P.device));
    if (cap) {                                     // a host pointer here would fault inside the kernel: refuse it up front
        hipPointerAttribute_t at;
        if (hipPointerGetAttributes(&at, d_out) != hipSuccess || at.type != hipMemoryTypeDevice || at.device != ctx->P.device) {
            (void)hipGetLastError();
            return fail(ctx, CM_EINVAL, "cm_collect_records_device: d_out is not memory of device %d", ctx->P.device);
        }
    }
    return collect_records_into(ctx, index_base, cap, (cm_record *)d_out, out_n, "cm_collect_records_device");
}

int cm_host_alloc(cm_ctx *ctx, uint64_t bytes, void **out) {
    if (!ctx || !out) return CM_EINVAL;
    *out = nullptr;
    HIPCHK(ctx, hipSetDevice(ctx->P.device));
    HIPCHK(ctx, hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocDefault));
    return CM_OK;
}

int cm_host_free(cm_ctx *ctx, void *p) {
    if (!ctx) return CM_EINVAL;
    if (p) HIPCHK(ctx, hipHostFree(p));
    return CM_OK;
}

int cm_host_register(cm_ctx *ctx, void *p, uint64_t bytes) {
    if (!ctx || !p || !bytes) return CM_EINVAL;
    HIPCHK(ctx, hipSetDevice(ctx->P.device));
    HIPCHK(ctx, hipHostRegister(p, bytes, hipHostRegisterDefault));
    return CM_OK;
}

int cm_host_unregister(cm_ctx *ctx, void *p) {
    if (!ctx) return CM_EINVAL;
    if (p) HIPCHK(ctx, hipHostUnregister(p));
    return CM_OK;
}

int cm_type_histogram(cm_ctx *ctx, uint64_t out[14]) {
    if (!ctx || !out) return CM_EINVAL;
    HIPCHK(ctx, hipSetDevice(ctx->P.device));
    for (int i = 0; i < 14; ++i) out[i] = 0;
    if (ctx->n_pairs == 0) return CM_OK;
    HIPCHK(ctx, ensure(ctx, ctx->d_type_hist, 16 * sizeof(unsigned long long)));
    HIPCHK(ctx, hipMemsetAsync(ctx->d_type_hist, 0, 16 * sizeof(unsigned long long), ctx->stream));
    const unsigned grid = (unsigned)std::min<uint64_t>((ctx->n_pairs + BLK - 1) / BLK, 1024);
    hipLaunchKernelGGL(k_type_hist, dim3(grid), dim3(BLK), 0, ctx->stream, ctx->d_state, ctx->n_pairs, ctx->d_type_hist);
    unsigned long long h[16];
    HIPCHK(ctx, hipMemcpyAsync(h, ctx->d_type_hist, sizeof h, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    for (int i = 0; i < 14; ++i) out[i] = h[i];
    return check_dev_err(ctx);
}

int cm_reads_download(cm_ctx *ctx, cm_mapped_read *out_state, int32_t *out_category, uint8_t *out_active) {
    if (!ctx) return CM_EINVAL;
    HIPCHK(ctx, hipSetDevice(ctx->P.device));
    const uint64_t n = ctx->n_pairs;
    if (n) {
        if (out_state) HIPCHK(ctx, hipMemcpyAsync(out_state, ctx->d_state, n * sizeof(cm_mapped_read), hipMemcpyDeviceToHost, ctx->stream));
        if (out_category) HIPCHK(ctx, hipMemcpyAsync(out_category, ctx->d_cat, n * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
        if (out_active) HIPCHK(ctx, hipMemcpyAsync(out_active, ctx->d_active, n, hipMemcpyDeviceToHost, ctx->stream));
    }
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return check_dev_err(ctx);
}

int cm_map_batch(cm_ctx *ctx, int slot, int is_last_round, const cm_reads *reads, const cm_mapped_read *prior, cm_mapped_read *out_state,
                 int32_t *out_category) {
    int rc = cm_reads_upload(ctx, reads, prior);
    if (rc) return rc;
    if ((rc = cm_map_round(ctx, slot, is_last_round))) return rc;
    return cm_reads_download(ctx, out_state, out_category, nullptr);
}

int cm_seed_batch(cm_ctx *ctx, int slot, uint32_t *out_start, uint32_t *out_cnt, uint32_t *out_raw, uint32_t cap_probes, uint32_t *out_n_slots) {
    if (!ctx || !out_start || !out_cnt || !out_raw || !out_n_slots) return CM_EINVAL;
    int rc = check_slot(ctx, slot, false);
    if (rc) return rc;
    HIPCHK(ctx, hipSetDevice(ctx->P.device));
    *out_n_slots = (uint32_t)ctx->n_seeds;
    const uint64_t need = ctx->n_pairs * 4ull * (uint64_t)ctx->n_seeds;
    if (need > cap_probes) return fail(ctx, CM_EINVAL, "cm_seed_batch: need room for %llu probes", (unsigned long long)need);
    const KCore core = make_core(ctx, ctx->slots[slot]);
    for (uint64_t p0 = 0; p0 < ctx->n_pairs; p0 += ctx->tile) {
        const uint32_t nt = (uint32_t)((ctx->n_pairs - p0 < ctx->tile) ? ctx->n_pairs - p0 : ctx->tile);
        if ((rc = run_seed_tile(ctx, core, current_reads(ctx), p0, nt, ctx->d_active, 0, ctx->stream, nullptr))) return rc;
        const size_t cnt = (size_t)nt * 4 * ctx->n_seeds, o = (size_t)p0 * 4 * ctx->n_seeds;
        HIPCHK(ctx, hipMemcpyAsync(out_start + o, ctx->d_sstart, cnt * 4, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipMemcpyAsync(out_cnt + o, ctx->d_scnt, cnt * 4, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipMemcpyAsync(out_raw + o, ctx->d_sraw, cnt * 4, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    }
    return CM_OK;
}

int cm_chain_batch(cm_ctx *ctx, int slot, cm_chain *out_chains, int32_t *out_nchain, int32_t *out_high) {
    if (!ctx || !out_chains || !out_nchain || !out_high) return CM_EINVAL;
    int rc = check_slot(ctx, slot, true);
    if (rc) return rc;
    HIPCHK(ctx, hipSetDevice(ctx->P.device));
    const KCore core = make_core(ctx, ctx->slots[slot]);
    for (uint64_t p0 = 0; p0 < ctx->n_pairs; p0 += ctx->tile) {
        const uint32_t nt = (uint32_t)((ctx->n_pairs - p0 < ctx->tile) ? ctx->n_pairs - p0 : ctx->tile);
        HIPCHK(ctx, hipMemsetAsync(ctx->d_chains, 0, (size_t)nt * 4 * CM_BESTCHAINLIM * sizeof(cm_chain), ctx->stream));
        const RoundBufs rb0 = round_bufs(ctx, 0);
        if ((rc = run_seed_tile(ctx, core, current_reads(ctx), p0, nt, ctx->d_active, 0, ctx->stream, &rb0))) return rc;
        if ((rc = run_chain_tile(ctx, core, current_reads(ctx), p0, nt, ctx->slots[slot].chain_parallel_ok, ctx->d_active, rb0, 0))) return rc;
        const size_t np = (size_t)nt * 4, o = (size_t)p0 * 4;
        HIPCHK(ctx, hipMemcpyAsync(out_chains + o * CM_BESTCHAINLIM, ctx->d_chains, np * CM_BESTCHAINLIM * sizeof(cm_chain), hipMemcpyDeviceToHost,
                                   ctx->stream));
        HIPCHK(ctx, hipMemcpyAsync(out_nchain + o, ctx->d_nchain, np * 4, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipMemcpyAsync(out_high + o, ctx->d_high, np * 4, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    }
    return check_dev_err(ctx);
}

/* diagnostic: per-pair k_pair lane time in 100 MHz ticks (only when CM_LANE_CLK was set at upload) */
// diagnostic builds (-DCM_CHAIN_DIAG ...): the 32 raw counter words, [8..31] = whatever the build accumulates there
int cm_debug_counters(cm_ctx *ctx, unsigned long long *out) {
    if (!ctx || !out) return CM_EINVAL;
    HIPCHK(ctx, hipSetDevice(ctx->P.device));
    HIPCHK(ctx, hipMemcpyAsync(out, ctx->d_counters, 32 * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return CM_OK;
}

int cm_debug_lane_clk(cm_ctx *ctx, unsigned long long *out) {
    if (!ctx || !out || !ctx->d_lane_clk) return CM_EINVAL;
#if defined(CM_DIAG)
    HIPCHK(ctx, hipMemcpy(out, ctx->d_lane_clk, ctx->n_pairs * 8 * 33, hipMemcpyDeviceToHost));
#else
    HIPCHK(ctx, hipMemcpy(out, ctx->d_lane_clk, ctx->n_pairs * 8, hipMemcpyDeviceToHost));
#endif
    return CM_OK;
}

int cm_prof_enable(cm_ctx *ctx, int on) {
    if (!ctx) return CM_EINVAL;
    ctx->prof = on != 0;
    return CM_OK;
}

int cm_prof_reset(cm_ctx *ctx) {
    if (!ctx) return CM_EINVAL;
    HIPCHK(ctx, hipSetDevice(ctx->P.device));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    for (auto &r : ctx->recs) {
        ctx->ev_free.push_back(r.a);
        ctx->ev_free.push_back(r.b);
    }
    ctx->recs.clear();
    for (int i = 0; i < 8; ++i) {
        ctx->ms[i] = 0;
        ctx->launches[i] = 0;
    }
    HIPCHK(ctx, hipMemsetAsync(ctx->d_counters, 0, 32 * sizeof(unsigned long long), ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return CM_OK;
}

int cm_prof_get(cm_ctx *ctx, double ms[8], uint64_t launches[8]) {
    if (!ctx || !ms || !launches) return CM_EINVAL;
    HIPCHK(ctx, hipSetDevice(ctx->P.device));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    for (auto &r : ctx->recs) {
        float t = 0;
        if (hipEventElapsedTime(&t, r.a, r.b) == hipSuccess) ctx->ms[r.cls] += (double)t;
        ctx->ev_free.push_back(r.a);
        ctx->ev_free.push_back(r.b);
    }
    ctx->recs.clear();
    for (int i = 0; i < 8; ++i) {
        ms[i] = ctx->ms[i];
        launches[i] = ctx->launches[i];
    }
    return CM_OK;
}

int cm_prof_counters(cm_ctx *ctx, uint64_t c[8]) {
    if (!ctx || !c) return CM_EINVAL;
    HIPCHK(ctx, hipSetDevice(ctx->P.device));
    unsigned long long h[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    HIPCHK(ctx, hipMemcpyAsync(h, ctx->d_counters, sizeof h, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    for (int i = 0; i < 8; ++i) c[i] = h[i];
    return CM_OK;
}

}  // extern "C"
