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
}
// AnnotDev over host memory (host emulation): records from `o`, pass-through arrays from `v`
inline AnnotDev annot_dev_host(const cm_annot_view &v, const AnnotAosHost &o) {
    AnnotDev d{};
    d.n_iv = v.n_iv; d.n_seg = v.n_seg; d.n_trans = v.n_trans; d.n_gene = v.n_gene; d.n_chr = v.n_chr;
    d.iv_bucket_shift = v.iv_bucket_shift; d.n_iv_bucket = v.iv_bucket ? v.n_iv_bucket : 0; d.n_bits = v.n_bits;
    d.iv = o.iv.data(); d.iv_seg = v.iv_seg; d.seg = o.seg.data(); d.seg_tid = v.seg_tid; d.tr = o.tr.data(); d.t2s = v.t2s;
    d.gene = o.gene.data(); d.near_border_bits = v.near_border_bits; d.intronic_bits = v.intronic_bits;
    d.chr_shift = v.chr_shift; d.chr_id = v.chr_id; d.iv_bucket = v.iv_bucket;
    return d;
}
}  // namespace cmc
