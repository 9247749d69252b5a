// cm_heavy_pipe.h -- the heavy pairs of a tile as a PIPELINE of full-width kernels (included by cm_hot.hip, inside its anonymous
// namespace, behind k_pair_heavy whose slot layout and helpers it shares).
//
// k_pair_heavy maps six heavy pairs per wave through every phase of process_mates: its phases are as wide as six pairs make them
// (33 tasks of 64 lanes), its DPs end at different times (11 lanes per instruction in the DP's main loop, 65 % of the kernel's
// instructions: tests/diag/ablate.sh), and one wave-sized pool of DP requests is too small to refill from (NOTES 44).  Here every
// phase is a kernel over ALL heavy pairs of the tile, its items one per lane:
//
//   plan     (wave = HG pairs, as before)  chain ends, pairing predicates -> the pairs' task lists (HBM) and the four fall-back
//                                          DP requests of every task: exact matches answered in closed form, the rest queued
//   dp       (lane = one DP in flight)     the tile's queue of DP requests through cmc::XdropLane, lanes refilled as their DPs end
//   tasks    (lane = one mate-pair task)   extend_task with its answers at hand -> outcome (HBM)
//   fold     (lane = one pair)             the outcomes in (i, j) order into the pair's MatchedRead; which unpaired chains to extend
//   unp_req  (lane = one chain end)        their DP requests;  dp again;
//   unp      (lane = one unpaired chain)   chain_both_sides with its answers -> the pair's minima
//   finish   (lane = one pair)             leftover_type, the attempt's verdict: second attempt (other orientation) or final state
//
// run twice (process_read's two attempts; the second over the pairs that need it).  Results are those of k_pair_heavy by
// construction: the same device functions on the same inputs in the same per-pair order; a DP answer is used only when the
// identity of the request matches (cmc::PreDP).  A pair whose tasks or unpaired chains do not fit the pipeline's arrays goes, whole
// and untouched, to k_pair_heavy (fall-back list) -- so do all heavy pairs when the band is not 3.

struct HPair {                       // a heavy pair between the kernels (HBM, one per entry of the tile's heavy list)
    cm_mapped_read mr;
    uint32_t t;                      // the pair, tile-relative
    uint32_t task_off, unp_off;      // its tasks T[task_off .. + ntask), its unpaired chains U[unp_off .. + nfu + nbu) of the current attempt
    uint32_t fp, bp, fun, bun;
    int32_t ntask, exf, exb;
    int32_t min_ret1, min_ret2;
    int16_t len1, len2, nf, nb, nfu, nbu;
    int8_t st, first, over, a, g1, g2, gf, gb, do_f, do_b, pad[2];
};
struct HTask { uint32_t h, e; };     // e: idx | code << 10 (idx = i * nb + j)
struct HUnp { uint32_t h, x; };      // x: chain | back << 8 | (position among the side's unpaired chains) << 16
enum { HC_TASKS = 0, HC_Q1 = 1, HC_Q1CUR = 2, HC_UNP = 3, HC_Q2 = 4, HC_Q2CUR = 5, HC_LIST2 = 6, HC_FALL = 7, HC_CUR = 8 /* .. 15: work cursors */, HC_WORDS = 16 };
struct HPipe {
    HPair *hp;
    uint32_t *list2, *fall;
    HTask *T;
    cmc::PreDP *pre;
    uint32_t *q;
    HRes *res;
    HUnp *U;
    cmc::PreDP *pre2;
    uint32_t *q2;
    unsigned int *ctr;
    uint32_t tasks_cap, unp_cap;
    unsigned int *fall_ctr;          // entries of fall[] (one list per set of chain records: the fall-back launch may come late, see settle_pair)
    int8_t *Tcls;                    // work class of T[x] (-2: a hole), the key k_hp_tasks' order is sorted by
};
struct HReadsOf {                    // the two reads of a pair in one attempt's orientation
    cmc::g_u8 fseq, bseq;
    int flen, blen;
    uint32_t fset, bset;             // chain sets (problem index t * 4 + x) of the forward / backward read
};
__device__ inline HReadsOf hp_reads(const ReadsDev &rd, uint64_t pair0, uint32_t t, bool r1_fwd) {
    const uint64_t p = pair0 + t;
    const uint64_t a0 = rd.off1[p], a1 = rd.off1[p + 1], b0 = rd.off2[p], b1 = rd.off2[p + 1];
    HReadsOf r;
    r.fseq = (cmc::g_u8)(r1_fwd ? rd.seq1 + a0 : rd.seq2 + b0);
    r.bseq = (cmc::g_u8)(r1_fwd ? rd.seq2 + b0 : rd.seq1 + a0);
    r.flen = (int)(r1_fwd ? a1 - a0 : b1 - b0);
    r.blen = (int)(r1_fwd ? b1 - b0 : a1 - a0);
    r.fset = t * 4u + (r1_fwd ? 0u : 2u);
    r.bset = t * 4u + (r1_fwd ? 3u : 1u);
    return r;
}
// views of the fall-back DP of one end of chain `chp` of read `rdv` (what extend_side hands to local_alignment_sc)
__device__ inline bool hp_side_views(const cmc::Ext &ext, cmc::g_chain chp, const cmc::Read &rdv, bool right, int kmer, cmc::SV &sv, int &n, cmc::SV &tv,
                                     int &m) {
    const uint32_t clen = chp->chain_len;
    const cmc::SV seq = rdv.view();
    const uint32_t pos = right ? chp->rpos[clen - 1] + (uint32_t)kmer - 1u : chp->rpos[0];
    const int len = right ? rdv.len - (chp->qpos[clen - 1] + kmer) : chp->qpos[0];
    if (len <= 0) return false;
    return ext.fallback_views(pos, len, right ? seq.sub(rdv.len - len) : seq, right, sv, n, tv, m) && cmc::pre_keyable(sv, n, tv, m);
}
// answers a request in closed form or queues it (every lane of the wave calls this; in_range: the lane holds a request)
__device__ inline void hp_answer_or_queue(const cmc::DpMem &sm, bool in_range, bool have, const cmc::SV &sv, int n, const cmc::SV &tv, int m,
                                          cmc::PreDP *slot, uint32_t req, uint32_t *queue, unsigned int *tail, int lane) {
    int state = 0;
    cmc::PreDP e{0u, 0u, 0u, 0};
    if (have) {
        e.s_off = (uint32_t)sv.off;
        e.key = cmc::pre_key(sv, n, tv, m);
        if (cmc::sc_closed_form(sv, n, tv, m)) {
            e.res = cmc::pre_pack(0, 0, 0);
            e.score = m * cmc::SC_MAT;
            state = 1;
        } else if (n <= sm.a.cap && m <= sm.b.cap) state = 2;            // (too long for the staging buffers: left to the caller, who flags it)
    }
    if (in_range) *slot = e;
    const unsigned long long mq = __ballot(state == 2);
    if (mq) {
        const int cnt = __popcll(mq), first = __ffsll((long long)mq) - 1;
        unsigned int base = 0;
        if (lane == first) base = atomicAdd(tail, (unsigned int)cnt);
        base = (unsigned int)__shfl((int)base, first);
        if (state == 2) queue[base + (unsigned int)__popcll(mq & ((1ull << lane) - 1ull))] = req;
    }
}

// ---- plan ------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(BLK_PAIR, 6) k_hp_plan(KCore kc, ReadsDev rd, uint64_t pair0, const uint32_t *hlist, const unsigned int *n_heavy_p,
                                                        const uint32_t *lst, const unsigned int *n_lst_p, int attempt, const cm_chain *chains,
                                                        const int32_t *nchain, const int32_t *high, const cm_mapped_read *state, HPipe P,
                                                        uint16_t *lists, int str_cap) {
    extern __shared__ uint32_t lds_words[];
    const int lane = threadIdx.x;
    CM_L HSlot *S = (CM_L HSlot *)lds_words;
    __shared__ uint32_t slot_h[HG], slot_off[HG];
    CM_G uint16_t *list = (CM_G uint16_t *)(lists + (size_t)blockIdx.x * HEAVY_LIST);
    const Core c = cmc::to_core(kc);
    cmc::DpMem sm{cmc::LBuf{nullptr, str_cap}, cmc::LBuf{nullptr, str_cap}, nullptr};      // (capacities only: nothing is staged here)
    const cmc::Ext ext(c, sm);
    const int kmer = c.P.kmer;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    const unsigned int n_items = attempt == 0 ? *n_heavy_p : *n_lst_p;
    auto take = [&]() { return (unsigned int)__shfl((int)(lane == 0 ? atomicAdd(&P.ctr[HC_CUR + 0], (unsigned int)HG) : 0u), 0); };
    for (unsigned int h0 = take(); h0 < n_items; h0 = take()) {
        const bool owner = lane < HG && h0 + (unsigned int)lane < n_items;
        uint32_t h = 0, t = 0;
        cm_mapped_read mr{};
        int st = -1, len1 = 0, len2 = 0;
        bool first = true;
        if (owner) {
            h = attempt == 0 ? h0 + (unsigned int)lane : lst[h0 + lane];
            t = hlist[h];
            const uint64_t p = pair0 + t;
            if (attempt == 0) {
                len1 = (int)(rd.off1[p + 1] - rd.off1[p]);
                len2 = (int)(rd.off2[p + 1] - rd.off2[p]);
                int n[4], hh[4];
                for (int x = 0; x < 4; ++x) {
                    n[x] = nchain[(uint64_t)t * 4 + x];
                    hh[x] = high[(uint64_t)t * 4 + x];
                }
                mr = state[p];
                const int n1 = n[0] + n[1], n2 = n[2] + n[3];
                if (n1 + n2 <= 0) {             // unreachable for a pair classified heavy; kept for completeness
                    st = ((hh[0] + hh[1] > 0) && (hh[2] + hh[3] > 0)) ? CM_NOPROC_MANYHIT : CM_NOPROC_NOMATCH;
                    cmc::mr_update_type(mr, st);
                } else if (n1 <= 0 || n2 <= 0) {
                    st = CM_OEANCH;
                    cmc::mr_update_type(mr, st);
                } else {
                    auto sc0 = [&](int x) { return n[x] > 0 ? chains[((uint64_t)t * 4 + x) * CM_BESTCHAINLIM].score : 0.f; };
                    first = (sc0(0) + sc0(3)) >= (sc0(2) + sc0(1));
                }
            } else {
                const HPair &hp = P.hp[h];
                mr = hp.mr;
                len1 = hp.len1;
                len2 = hp.len2;
                first = hp.first != 0;
            }
        }
        const bool on = owner && st < 0;
        const bool r1_fwd = (attempt == 0) == first;
        if (lane < HG) {
            CM_L HSlot &s = S[lane];
            const HReadsOf R = hp_reads(rd, pair0, t, r1_fwd);
            const int nf = on ? nchain[R.fset] : 0, nb = on ? nchain[R.bset] : 0;
            s.fseq = R.fseq;
            s.bseq = R.bseq;
            s.fch_i = R.fset * CM_BESTCHAINLIM;
            s.bch_i = R.bset * CM_BESTCHAINLIM;
            s.t = t;
            s.flen = (int16_t)R.flen;
            s.blen = (int16_t)R.blen;
            s.nf = (int16_t)nf;
            s.nb = (int16_t)nb;
            s.inv_nb = nb > 0 ? (65536u + (unsigned int)nb - 1u) / (unsigned int)nb : 0u;
            s.saved_type = (int8_t)mr.type;
            s.fp = 0u;
            s.bp = 0u;
            s.ntask = 0;
            s.done = 0;
            s.nfu = 0;
            s.nbu = 0;
            slot_h[lane] = h;
            slot_off[lane] = 0xffffffffu;
        }
        __syncthreads();
        int pre[HG + 1];
        // chain ends: reference span + exon interval of the first fragment, one chain per lane
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
        // pairing predicate of every (i, j) of every slot; the accepted ones go to the wave's list in (slot, i, j) order
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
        // owners reserve their pair's stretch of the tile's task array; a pair that does not fit goes to the fall-back kernel
        bool over = false;
        if (on) {
            const unsigned int nt = (unsigned int)S[lane].ntask;
            const unsigned int off = nt ? atomicAdd(&P.ctr[HC_TASKS], nt) : 0u;
            over = nt && (unsigned long long)off + nt > (unsigned long long)P.tasks_cap;
            slot_off[lane] = off;
            if (over) {
                S[lane].done = 1;                              // (marks the slot: its tasks become holes, no requests)
                P.fall[atomicAdd(P.fall_ctr, 1u)] = t;
            }
        }
        __syncthreads();
        int t_lo[HG + 1];
        t_lo[0] = 0;
#pragma unroll
        for (int g = 0; g < HG; ++g) t_lo[g + 1] = t_lo[g] + S[g].ntask;
        for (int x = lane; x < n_task; x += 64) {
            const unsigned int e = list[x];
            const int g = (int)(e >> 12);
            const unsigned long long pos = (unsigned long long)slot_off[g] + (unsigned int)(x - t_lo[g]);
            if (pos < P.tasks_cap) {
                CM_L const HSlot &sl = S[g];
                P.T[pos] = HTask{sl.done ? 0xffffffffu : slot_h[g], e & 0xFFFu};
                // work class: mates in one transcript (exon walks) x the bases the four chain ends leave to extend; results do not depend on it
                const int idx = (int)(e & 1023u);
                const int i = (int)(((unsigned int)idx * sl.inv_nb) >> 16), j = idx - i * sl.nb;
                const cmc::g_chain f = slot_fch(sl, chains) + i, bk = slot_bch(sl, chains) + j;
                const int fl = (int)f->chain_len, bl = (int)bk->chain_len;
                const int resid = (int)f->qpos[0] + (sl.flen - ((int)f->qpos[fl - 1] + kmer)) + (int)bk->qpos[0] + (sl.blen - ((int)bk->qpos[bl - 1] + kmer));
                const int lv = resid <= 0 ? 0 : resid < 16 ? 1 : resid < 32 ? 2 : resid < 64 ? 3 : resid < 96 ? 4 : resid < 128 ? 5 : resid < 192 ? 6 : 7;
                P.Tcls[pos] = (int8_t)(sl.done ? -2 : ((((e >> 10) & 3u) == 1u ? 8 : 0) + lv));
            }
        }
        // the four fall-back DP requests of every task (both chains, both ends).  A task whose mates share a transcript walks
        // that first and may never ask (left to compute in place).
        for (int r0 = 0; r0 < 4 * n_task; r0 += 64) {
            const int r = r0 + lane;
            cmc::SV sv{}, tv{};
            int n = 0, m = 0;
            bool have = false, in_range = false;
            unsigned long long req = 0;
            if (r < 4 * n_task) {
                const unsigned int e = list[r >> 2];
                const int g = (int)(e >> 12);
                CM_L const HSlot &sl = S[g];
                if (!sl.done) {
                    in_range = true;
                    req = ((unsigned long long)slot_off[g] + (unsigned int)((r >> 2) - t_lo[g])) * 4ull + (unsigned int)(r & 3);
                    if (((e >> 10) & 3u) != 1u) {
                        const int idx = (int)(e & 1023u);
                        const int i = (int)(((unsigned int)idx * sl.inv_nb) >> 16), j = idx - i * sl.nb;
                        const bool back = (r & 2) != 0;
                        const cmc::Read rdv{back ? sl.bseq : sl.fseq, back ? sl.blen : sl.flen, back ? 1 : 0};
                        have = hp_side_views(ext, (back ? slot_bch(sl, chains) + j : slot_fch(sl, chains) + i), rdv, (r & 1) != 0, kmer, sv, n, tv, m);
                    }
                }
            }
            hp_answer_or_queue(sm, in_range, have, sv, n, tv, m, P.pre + req, (uint32_t)req, P.q, &P.ctr[HC_Q1], lane);
        }
        if (owner) {
            HPair &hp = P.hp[h];
            CM_L const HSlot &s = S[lane];
            if (attempt == 0) {
                hp.t = t;
                hp.len1 = (int16_t)len1;
                hp.len2 = (int16_t)len2;
                hp.first = first ? 1 : 0;
            }
            hp.mr = mr;
            hp.st = (int8_t)st;
            hp.over = over ? 1 : 0;
            hp.task_off = slot_off[lane];
            hp.ntask = s.ntask;
            hp.fp = s.fp;
            hp.bp = s.bp;
            hp.nf = s.nf;
            hp.nb = s.nb;
            hp.nfu = hp.nbu = 0;
            hp.unp_off = 0;
        }
        __syncthreads();                                           // the slots are rewritten by the next group
    }
}

// ---- the tile's DP queue ---------------------------------------------------------------------------------------------------
// mode 0: requests of tasks (item = request / 4 -> T -> pair; which & 2 = backward read);  mode 1: of unpaired chains (item =
// request / 2 -> U -> pair; the read is the chain's).
__global__ void __launch_bounds__(BLK_PAIR, 8) k_hp_dp(KCore kc, ReadsDev rd, uint64_t pair0, int attempt, HPipe P, int mode, int str_cap) {
    extern __shared__ uint32_t lds_words[];
    const int lane = threadIdx.x;
    CM_S uint8_t *lane_base = (CM_S uint8_t *)lds_words + 4 * lane;
    const int str_stride = lbuf_bytes(str_cap) * BLK_PAIR;
    cmc::DpMem sm{cmc::LBuf{lane_base, str_cap}, cmc::LBuf{lane_base + str_stride, str_cap}, nullptr};
    const Core c = cmc::to_core(kc);
    const uint32_t *queue = mode ? P.q2 : P.q;
    cmc::PreDP *pre = mode ? P.pre2 : P.pre;
    const unsigned int tail = P.ctr[mode ? HC_Q2 : HC_Q1];
    unsigned int *cursor = &P.ctr[mode ? HC_Q2CUR : HC_Q1CUR];
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    constexpr int REFILL = 16, BURST = 4;
    const int top = (sm.a.cap < sm.b.cap ? sm.a.cap : sm.b.cap) - 1;
    cmc::XdropLane L;
    L.go = false;
    bool busy = false, dry = false;
    uint32_t my_r = 0;
    for (;;) {
        const unsigned long long idle_m = __ballot(!busy);
        const int n_idle = __popcll(idle_m);
        if (!dry && (n_idle >= REFILL)) {
            unsigned int base = 0;
            if (lane == 0) base = atomicAdd(cursor, (unsigned int)n_idle);
            base = (unsigned int)__shfl((int)base, 0);
            if (base + (unsigned int)n_idle >= tail) dry = true;             // the queue has nothing beyond this hand-out
            const unsigned int mine = base + (unsigned int)__popcll(idle_m & lt_mask);
            if (!busy && mine < tail) {
                my_r = queue[mine];
                const cmc::PreDP e = pre[my_r];                            // the request's identity = the two views (cmc::pre_key)
                const int n = (int)(e.key & 1023u), m = (int)((e.key >> 10) & 1023u), t_off = (int)((e.key >> 20) & 1023u);
                const bool sneg = (e.key >> 30) & 1u, back = (e.key >> 31) & 1u;
                const uint32_t h = mode ? P.U[my_r >> 1].h : P.T[my_r >> 2].h;
                const HPair &hp = P.hp[h];
                const HReadsOf R = hp_reads(rd, pair0, hp.t, (attempt == 0) == (hp.first != 0));
                const cmc::SV sv{c.X.genome, (int32_t)e.s_off, sneg ? -1 : 1, 0};
                const cmc::SV tv{back ? R.bseq : R.fseq, t_off, (back != sneg) ? -1 : 1, back ? 1 : 0};
                cmc::stage(sv, n, sm.a, 4);
                cmc::stage(tv.rev(m), m, sm.b, 5);                        // the band-3 DP walks the read residual from its far end
                cmc::xdrop_w3_begin(L, sm.a, n, sm.b, m, top);
                busy = true;
            }
        } else if (n_idle == 64) break;                                  // nothing in flight, nothing left to hand out
        for (int it = 0; it < BURST; ++it) {
            if (busy && L.go) cmc::xdrop_w3_advance(L, sm.a, sm.b, top);
            if (__ballot(busy && L.go) == 0ull) break;
        }
        if (busy && !L.go) {                                             // ended: its answer into the table, the lane is free
            int sc_len, indel, score;
            const int ed = cmc::xdrop_w3_end(c, L, sc_len, indel, score);
            pre[my_r].res = cmc::pre_pack(ed, sc_len, indel);
            pre[my_r].score = score;
            busy = false;
        }
    }
}

// ---- tasks -----------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(BLK_PAIR, CM_PAIR_WAVES) k_hp_tasks(KCore kc, ReadsDev rd, uint64_t pair0, int attempt, const cm_chain *chains,
                                                                       const int32_t *nchain, HPipe P, uint32_t *pair_err, int str_cap,
                                                                       const uint32_t *order, const unsigned int *n_order) {
    extern __shared__ uint32_t lds_words[];
    const int lane = threadIdx.x;
    CM_S uint8_t *lane_base = (CM_S uint8_t *)lds_words + 4 * lane;
    const int str_stride = lbuf_bytes(str_cap) * BLK_PAIR;
    cmc::DpMem sm{cmc::LBuf{lane_base, str_cap}, cmc::LBuf{lane_base + str_stride, str_cap}, nullptr};
    const Core c = cmc::to_core(kc);
    const cmc::Ext ext(c, sm);
    const int kmer = c.P.kmer;
    const unsigned int n_tasks = P.ctr[HC_TASKS] < P.tasks_cap ? P.ctr[HC_TASKS] : P.tasks_cap;
    uint32_t tids[cmc::MAX_TID];
    auto take = [&]() { return (unsigned int)__shfl((int)(lane == 0 ? atomicAdd(&P.ctr[HC_CUR + 1], 64u) : 0u), 0); };
    const unsigned int n_work = order ? (*n_order < n_tasks ? *n_order : n_tasks) : n_tasks;       // (ordered: the holes are left out)
    for (unsigned int x0 = take(); x0 < n_work; x0 = take()) {
        const unsigned int y = x0 + (unsigned int)lane;
        if (y >= n_work) continue;
        const unsigned int x = order ? order[y] : y;
        if (x >= n_tasks) continue;
        const HTask tk = P.T[x];
        if (tk.h == 0xffffffffu) continue;
        const HPair &hp = P.hp[tk.h];
        const bool r1_fwd = (attempt == 0) == (hp.first != 0);
        const HReadsOf R = hp_reads(rd, pair0, hp.t, r1_fwd);
        const int nb = nchain[R.bset];
        const int idx = (int)(tk.e & 1023u), i = idx / nb, j = idx - i * nb;
        const uint32_t code = (tk.e >> 10) & 3u;
        const cmc::g_chain fch = (cmc::g_chain)(chains + (uint64_t)R.fset * CM_BESTCHAINLIM) + i, bch = (cmc::g_chain)(chains + (uint64_t)R.bset * CM_BESTCHAINLIM) + j;
        sm.err = (cmc::g_err)(pair_err + hp.t);
        cmc::TidList tl{tids, 0, -1, -1, false};
        if (code == 1) tl = cmc::common_tids(c, cmc::overlap(c, fch->rpos[0]), cmc::overlap(c, bch->rpos[0]), tids);
        const cmc::CH F{fch, kmer}, Rr{bch, kmer};
        const cmc::Read frd{R.fseq, R.flen, 0}, brd{R.bseq, R.blen, 1};
        cmc::MM r1, r2;
        bool il, ok;
        int row;
        sm.pre = (cmc::g_pre)(P.pre + (size_t)x * 4);
        sm.n_pre = 4;
        cmc::extend_task(c, ext, F, Rr, tl, frd, brd, r1, r2, il, ok, row);
        sm.pre = nullptr;
        sm.n_pre = 0;
        HRes &o = P.res[x];
        o.r1 = r1;
        o.r2 = r2;
        o.row = row;
        o.pair_type = (int)code - 1;
        o.ok = ok;
        o.is_left = il;
    }
}

// ---- fold ------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(BLK_PAIR, 8) k_hp_fold(KCore kc, uint64_t pair0, const uint32_t *lst, const unsigned int *n_lst_p, const unsigned int *n_heavy_p,
                                                        int attempt, HPipe P, unsigned long long *counters) {
    const Core c = cmc::to_core(kc);
    const unsigned int n_items = attempt == 0 ? *n_heavy_p : *n_lst_p;
    const unsigned int x = blockIdx.x * BLK_PAIR + threadIdx.x;
    for (unsigned int y = x; y < n_items; y += gridDim.x * BLK_PAIR) {
        const uint32_t h = attempt == 0 ? y : lst[y];
        HPair &hp = P.hp[h];
        if (hp.over || hp.st >= 0) continue;
        cm_mapped_read mr = hp.mr;
        const bool r1_fwd = (attempt == 0) == (hp.first != 0);
        int min_ret1 = CM_ORPHAN, min_ret2 = CM_ORPHAN, g1 = 0, g2 = 0;
        bool early = false;
        for (int k = 0; k < hp.ntask; ++k) {
            const HRes &o = P.res[(size_t)hp.task_off + k];
            const cmc::MM r1 = o.r1, r2 = o.r2;
            if (cmc::fold_task(c, r1, r2, o.is_left != 0, o.ok != 0, o.row, o.pair_type, r1_fwd, mr)) {
                early = true;
#if defined(CM_HP_DIAG)      // tasks the reference's sequential loop would have run (it returns from inside the loop here)
                atomicAdd(&counters[26], (unsigned long long)(k + 1));
                atomicAdd(&counters[27], 1ull);
                if (k == 0) atomicAdd(&counters[28], 1ull);
#endif
                break;
            }
            min_ret1 = r1.type < min_ret1 ? r1.type : min_ret1;
            min_ret2 = r2.type < min_ret2 ? r2.type : min_ret2;
            g1 = (r1.exons_spos >= 0) || (r1.exons_epos >= 0);
            g2 = (r2.exons_spos >= 0) || (r2.exons_epos >= 0);
        }
#if defined(CM_HP_DIAG)      // unpaired chains there are (counters[24]) against unpaired chains that get extended (HC_UNP)
        atomicAdd(&counters[24], (unsigned long long)(__popc(~hp.fp & (hp.nf >= 32 ? 0xffffffffu : ((1u << hp.nf) - 1u))) +
                                                      __popc(~hp.bp & (hp.nb >= 32 ? 0xffffffffu : ((1u << hp.nb) - 1u)))));
        atomicAdd(&counters[25], (unsigned long long)(hp.nf + hp.nb));
#endif
#if defined(CM_HP_DIAG)
        if (!early) atomicAdd(&counters[26], (unsigned long long)hp.ntask);
#endif
        // does the pair go on to the unpaired-chain extensions? (src/filter.cpp:344-393)
        int a = -1;
        bool do_f = false, do_b = false;
        uint32_t fun = 0, bun = 0;
        int nfu = 0, nbu = 0;
        if (early) a = CM_CONCRD;
        else if (mr.type == CM_CONCRD || mr.type == CM_DISCRD || mr.type == CM_CHIORF || mr.type == CM_CHIBSJ || mr.type == CM_CHI2BSJ) a = mr.type;
        else {
            fun = ~hp.fp & (hp.nf >= 32 ? 0xffffffffu : ((1u << hp.nf) - 1u));
            bun = ~hp.bp & (hp.nb >= 32 ? 0xffffffffu : ((1u << hp.nb) - 1u));
            do_f = min_ret1 != CM_CONCRD && fun != 0;
            do_b = min_ret2 != CM_CONCRD && bun != 0;
            if (!cmc::leftovers_matter(mr.type, min_ret1, do_f, min_ret2, do_b)) a = mr.type;
            else {
                nfu = do_f ? __popc(fun) : 0;
                nbu = do_b ? __popc(bun) : 0;
            }
        }
        uint32_t off = 0;
        if (nfu + nbu) {
            off = atomicAdd(&P.ctr[HC_UNP], (unsigned int)(nfu + nbu));
            if ((unsigned long long)off + (unsigned int)(nfu + nbu) > (unsigned long long)P.unp_cap) {          // does not fit: the whole pair to the fall-back kernel
                hp.over = 1;
                P.fall[atomicAdd(P.fall_ctr, 1u)] = hp.t;
                // (its stretch of U stays unwritten; the entries below the capacity are made holes)
                for (int k = 0; k < nfu + nbu; ++k)
                    if ((unsigned long long)off + (unsigned int)k < P.unp_cap) P.U[off + k] = HUnp{0xffffffffu, 0u};
                continue;
            }
            int k = 0;
            for (uint32_t mk = do_f ? fun : 0u, pos = 0; mk; mk &= mk - 1, ++pos, ++k) P.U[off + k] = HUnp{h, (uint32_t)(__ffs((int)mk) - 1) | (pos << 16)};
            for (uint32_t mk = do_b ? bun : 0u, pos = 0; mk; mk &= mk - 1, ++pos, ++k)
                P.U[off + k] = HUnp{h, (uint32_t)(__ffs((int)mk) - 1) | (1u << 8) | (pos << 16)};
        }
        hp.mr = mr;
        hp.min_ret1 = min_ret1;
        hp.min_ret2 = min_ret2;
        hp.g1 = (int8_t)g1;
        hp.g2 = (int8_t)g2;
        hp.a = (int8_t)a;
        hp.do_f = do_f ? 1 : 0;
        hp.do_b = do_b ? 1 : 0;
        hp.fun = fun;
        hp.bun = bun;
        hp.nfu = (int16_t)nfu;
        hp.nbu = (int16_t)nbu;
        hp.unp_off = off;
        hp.exf = 99;
        hp.exb = 99;
        hp.gf = 0;
        hp.gb = 0;
    }
}

// ---- unpaired chains: requests, extensions ----------------------------------------------------------------------------------
__global__ void __launch_bounds__(BLK_PAIR, 8) k_hp_unp_req(KCore kc, ReadsDev rd, uint64_t pair0, int attempt, const cm_chain *chains, HPipe P, int str_cap) {
    const int lane = threadIdx.x;
    const Core c = cmc::to_core(kc);
    cmc::DpMem sm{cmc::LBuf{nullptr, str_cap}, cmc::LBuf{nullptr, str_cap}, nullptr};
    const cmc::Ext ext(c, sm);
    const int kmer = c.P.kmer;
    const unsigned int n_unp = P.ctr[HC_UNP] < P.unp_cap ? P.ctr[HC_UNP] : P.unp_cap;
    const unsigned int n_req = 2u * n_unp;
    auto take = [&]() { return (unsigned int)__shfl((int)(lane == 0 ? atomicAdd(&P.ctr[HC_CUR + 2], 64u) : 0u), 0); };
    for (unsigned int r0 = take(); r0 < n_req; r0 = take()) {
        const unsigned int r = r0 + (unsigned int)lane;
        cmc::SV sv{}, tv{};
        int n = 0, m = 0;
        bool have = false, in_range = false;
        if (r < n_req) {
            const HUnp u = P.U[r >> 1];
            if (u.h != 0xffffffffu) {
                in_range = true;
                const HPair &hp = P.hp[u.h];
                const HReadsOf R = hp_reads(rd, pair0, hp.t, (attempt == 0) == (hp.first != 0));
                const bool back = (u.x >> 8) & 1u;
                const int ci = (int)(u.x & 0xFFu);
                const cmc::g_chain chp = (cmc::g_chain)(chains + (uint64_t)(back ? R.bset : R.fset) * CM_BESTCHAINLIM) + ci;
                const cmc::Read rdv{back ? R.bseq : R.fseq, back ? R.blen : R.flen, back ? 1 : 0};
                have = hp_side_views(ext, chp, rdv, (r & 1u) != 0u, kmer, sv, n, tv, m);
            }
        }
        hp_answer_or_queue(sm, in_range, have, sv, n, tv, m, P.pre2 + r, r, P.q2, &P.ctr[HC_Q2], lane);
    }
}
__global__ void __launch_bounds__(BLK_PAIR, CM_PAIR_WAVES) k_hp_unp(KCore kc, ReadsDev rd, uint64_t pair0, int attempt, const cm_chain *chains, HPipe P,
                                                                     uint32_t *pair_err, int str_cap) {
    extern __shared__ uint32_t lds_words[];
    const int lane = threadIdx.x;
    CM_S uint8_t *lane_base = (CM_S uint8_t *)lds_words + 4 * lane;
    const int str_stride = lbuf_bytes(str_cap) * BLK_PAIR;
    cmc::DpMem sm{cmc::LBuf{lane_base, str_cap}, cmc::LBuf{lane_base + str_stride, str_cap}, nullptr};
    const Core c = cmc::to_core(kc);
    const cmc::Ext ext(c, sm);
    const int kmer = c.P.kmer;
    const unsigned int n_unp = P.ctr[HC_UNP] < P.unp_cap ? P.ctr[HC_UNP] : P.unp_cap;
    auto take = [&]() { return (unsigned int)__shfl((int)(lane == 0 ? atomicAdd(&P.ctr[HC_CUR + 3], 64u) : 0u), 0); };
    for (unsigned int x0 = take(); x0 < n_unp; x0 = take()) {
        const unsigned int x = x0 + (unsigned int)lane;
        if (x >= n_unp) continue;
        const HUnp u = P.U[x];
        if (u.h == 0xffffffffu) continue;
        HPair &hp = P.hp[u.h];
        const HReadsOf R = hp_reads(rd, pair0, hp.t, (attempt == 0) == (hp.first != 0));
        const bool back = (u.x >> 8) & 1u;
        const int ci = (int)(u.x & 0xFFu), k = (int)(u.x >> 16);
        const cmc::CH ch{(cmc::g_chain)(chains + (uint64_t)(back ? R.bset : R.fset) * CM_BESTCHAINLIM) + ci, kmer};
        const cmc::Read frd{R.fseq, R.flen, 0}, brd{R.bseq, R.blen, 1};
        cmc::MM m = cmc::mm_init(c);
        sm.err = (cmc::g_err)(pair_err + hp.t);
        sm.pre = (cmc::g_pre)(P.pre2 + (size_t)x * 2);
        sm.n_pre = 2;
        const int ex = ext.chain_both_sides(ch, back ? brd : frd, m, back ? -1 : 1);
        sm.pre = nullptr;
        sm.n_pre = 0;
        atomicMin(back ? &hp.exb : &hp.exf, ex);
        if (k == 0) {          // the reference reuses one MatchedMate per side: only the first chain's exon look-ups ever happen
            cmc::overlap_to_spos(c, m);
            cmc::overlap_to_epos(c, m);
            (back ? hp.gb : hp.gf) = (int8_t)((m.exons_spos >= 0) || (m.exons_epos >= 0));
        }
    }
}

// ---- finish ----------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(BLK_PAIR, 8) k_hp_finish(KCore kc, uint64_t pair0, const uint32_t *lst, const unsigned int *n_lst_p, const unsigned int *n_heavy_p,
                                                          int attempt, HPipe P, cm_mapped_read *state, uint8_t *active, int32_t *cat, int is_last,
                                                          unsigned long long *counters, RetryArgs ra, const int32_t *nchain, int second_to_fall) {
    const Core c = cmc::to_core(kc);
    const unsigned int n_items = attempt == 0 ? *n_heavy_p : *n_lst_p;
    const unsigned int x = blockIdx.x * BLK_PAIR + threadIdx.x;
#if defined(CM_HP_DIAG)      // per attempt: pairs, tasks, unpaired chains, DP requests of both queues, fall-backs -> counters[8 + 8 * attempt ..]
    if (x == 0) {
        unsigned long long *o = counters + 8 + 8 * attempt;
        atomicAdd(&o[0], (unsigned long long)n_items);
        atomicAdd(&o[1], (unsigned long long)P.ctr[HC_TASKS]);
        atomicAdd(&o[2], (unsigned long long)P.ctr[HC_UNP]);
        atomicAdd(&o[3], (unsigned long long)P.ctr[HC_Q1]);
        atomicAdd(&o[4], (unsigned long long)P.ctr[HC_Q2]);
        atomicAdd(&o[5], (unsigned long long)*P.fall_ctr);
    }
#endif
    for (unsigned int y = x; y < n_items; y += gridDim.x * BLK_PAIR) {
        const uint32_t h = attempt == 0 ? y : lst[y];
        HPair &hp = P.hp[h];
        if (hp.over) {                                     // goes through the fall-back launch, which may come late: active until then
            if (attempt == 0) active[pair0 + hp.t] = 1;
            continue;
        }
        cm_mapped_read mr = hp.mr;
        int st = hp.st;
        if (st < 0) {                                  // the attempt ran: its verdict (the tail of process_mates + process_read's loop)
            int a = hp.a;
            if (a < 0) {
                int min_ret1 = hp.min_ret1, min_ret2 = hp.min_ret2, g1 = hp.g1, g2 = hp.g2;
                if (hp.do_f) {
                    min_ret1 = hp.exf < min_ret1 ? hp.exf : min_ret1;
                    g1 = hp.gf;
                }
                if (hp.do_b) {
                    min_ret2 = hp.exb < min_ret2 ? hp.exb : min_ret2;
                    g2 = hp.gb;
                }
                cmc::mr_update_type(mr, cmc::leftover_type(min_ret1, min_ret2, g1 != 0, g2 != 0));
                a = mr.type;
            }
            if (c.P.scan_level == 0 && a == CM_CONCRD) st = CM_CONCRD;
            if (st < 0 && attempt == 0) {              // the other orientation next
                // Most often (62 % of the heavy pairs of the dense workload go on, 1 % of those have anything to do) neither read has
                // a chain in that orientation: no task, no unpaired chain, and the attempt's verdict is the one k_hp_fold and this
                // kernel reach over empty lists -- taken here.
                const bool r1_fwd2 = hp.first == 0;
                const uint32_t fs = hp.t * 4u + (r1_fwd2 ? 0u : 2u), bs = hp.t * 4u + (r1_fwd2 ? 3u : 1u);
                if (nchain[fs] == 0 && nchain[bs] == 0) {
                    const bool decided = mr.type == CM_CONCRD || mr.type == CM_DISCRD || mr.type == CM_CHIORF || mr.type == CM_CHIBSJ || mr.type == CM_CHI2BSJ;
                    if (!decided && cmc::leftovers_matter(mr.type, CM_ORPHAN, false, CM_ORPHAN, false))
                        cmc::mr_update_type(mr, cmc::leftover_type(CM_ORPHAN, CM_ORPHAN, false, false));
                    st = (c.P.scan_level == 0 && mr.type == CM_CONCRD) ? CM_CONCRD : mr.type;
                } else if (second_to_fall) {           // the few others: whole, by the fall-back kernel behind the pipeline
                    P.fall[atomicAdd(P.fall_ctr, 1u)] = hp.t;
                    active[pair0 + hp.t] = 1;
                    continue;
                } else {
                    hp.mr = mr;
                    P.list2[atomicAdd(&P.ctr[HC_LIST2], 1u)] = h;
                    continue;
                }
            }
            if (st < 0) st = mr.type;
        }
        const uint32_t t = hp.t;
        const uint64_t p = pair0 + t;
        if (__hip_atomic_load(ra.pair_err + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
            ra.list[atomicAdd(ra.count, 1u)] = t;          // left as it was; the re-run launch of k_pair maps it (one lane, exact)
            active[p] = 1;                                 // (its flag: active until then)
        } else {
            uint8_t act = 1;
            cmc::finish_round(c, st, is_last, hp.len1, hp.len2, mr, act);
            state[p] = mr;
            active[p] = act;
            cat[p] = st;
            atomicAdd(&counters[3], 1ull);
        }
    }
}
// between the two attempts: the arrays are reused
__global__ void k_hp_reset(unsigned int *ctr) {
    if (threadIdx.x < HC_WORDS && threadIdx.x != HC_LIST2 && threadIdx.x != HC_FALL) ctr[threadIdx.x] = 0u;
}
