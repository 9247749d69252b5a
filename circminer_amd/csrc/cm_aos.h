// cm_aos.h — host-side repack of a caller's cm_annot_view (structure-of-arrays, the C-ABI layout) into the
// array-of-structs records the device code reads (cmc::IvRec / SegRec / TrRec / GeneRec, cm_core.h).
#pragma once
#include <vector>

#include "cm_core.h"

namespace cmc {
struct AnnotAosHost {
    std::vector<IvRec> iv;
    std::vector<SegRec> seg;
    std::vector<TrRec> tr;
    std::vector<GeneRec> gene;
    uint32_t pair_reach = 0xffffffffu;     // AnnotDev::pair_reach
};
inline void build_annot_aos(const cm_annot_view &v, AnnotAosHost &o) {
    o.iv.resize(v.n_iv);
    for (uint32_t i = 0; i < v.n_iv; ++i)
        o.iv[i] = IvRec{v.iv_spos[i], v.iv_epos[i], v.iv_max_end[i], v.iv_min_end[i], v.iv_max_next_exon[i], v.iv_seg_off[i],
                        v.iv_seg_off[i + 1] - v.iv_seg_off[i], 0u};
    o.seg.resize(v.n_seg);
    for (uint32_t i = 0; i < v.n_seg; ++i)
        o.seg[i] = SegRec{v.seg_start[i], v.seg_end[i], v.seg_next_exon_beg[i], v.seg_gene_id[i], v.seg_tid_off[i],
                          v.seg_tid_off[i + 1] - v.seg_tid_off[i], 0u, 0u};
    o.tr.resize(v.n_trans);
    for (uint32_t i = 0; i < v.n_trans; ++i) o.tr[i] = TrRec{v.trans_start_ind[i], v.t2s_off[i], v.t2s_off[i + 1] - v.t2s_off[i], 0u};
    o.gene.resize(v.n_gene);
    for (uint32_t i = 0; i < v.n_gene; ++i) o.gene[i] = GeneRec{v.gene_start[i], v.gene_end[i]};
    // pair_reach (cmc::pair_code): the largest (a) extent of the intervals that hold a segment of one transcript, (b) hull of an
    // interval and the span of a gene one of its segments belongs to
    uint64_t reach = 0;
    std::vector<uint32_t> lo(v.n_trans, 0xffffffffu), hi(v.n_trans, 0u);
    for (uint32_t i = 0; i < v.n_iv; ++i) {
        const uint32_t a = v.iv_spos[i], b = v.iv_epos[i];
        for (uint32_t k = v.iv_seg_off[i]; k < v.iv_seg_off[i + 1]; ++k) {
            const uint32_t s = v.iv_seg[k];
            if (s >= v.n_seg) continue;
            const uint32_t g = v.seg_gene_id[s];
            if (g < v.n_gene) {
                const uint32_t l = a < v.gene_start[g] ? a : v.gene_start[g], h = b > v.gene_end[g] ? b : v.gene_end[g];
                if ((uint64_t)h - l > reach) reach = (uint64_t)h - l;
            }
            for (uint32_t q = v.seg_tid_off[s]; q < v.seg_tid_off[s + 1]; ++q) {
                const uint32_t t = v.seg_tid[q];
                if (t >= v.n_trans) continue;
                if (a < lo[t]) lo[t] = a;
                if (b > hi[t]) hi[t] = b;
            }
        }
    }
    for (uint32_t t = 0; t < v.n_trans; ++t)
        if (hi[t] >= lo[t] && (uint64_t)hi[t] - lo[t] > reach) reach = (uint64_t)hi[t] - lo[t];
    o.pair_reach = reach > 0xfffffffeull ? 0xffffffffu : (uint32_t)reach;
}
// AnnotDev over host memory (host emulation): records from `o`, pass-through arrays from `v`
inline AnnotDev annot_dev_host(const cm_annot_view &v, const AnnotAosHost &o) {
    AnnotDev d{};
    d.n_iv = v.n_iv; d.n_seg = v.n_seg; d.n_trans = v.n_trans; d.n_gene = v.n_gene; d.n_chr = v.n_chr;
    d.iv_bucket_shift = v.iv_bucket_shift; d.n_iv_bucket = v.iv_bucket ? v.n_iv_bucket : 0; d.n_bits = v.n_bits;
    d.pair_reach = o.pair_reach;
    d.iv = o.iv.data(); d.iv_seg = v.iv_seg; d.seg = o.seg.data(); d.seg_tid = v.seg_tid; d.tr = o.tr.data(); d.t2s = v.t2s;
    d.gene = o.gene.data(); d.near_border_bits = v.near_border_bits; d.intronic_bits = v.intronic_bits;
    d.chr_shift = v.chr_shift; d.chr_id = v.chr_id; d.iv_bucket = v.iv_bucket;
    return d;
}
}  // namespace cmc
