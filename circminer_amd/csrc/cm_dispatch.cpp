// cm_dispatch.cpp — the public entry points of the device path.
//
// The kernels (cm_hot.hip) are compiled twice: once for reads of up to 16 seeds (floor(read length / k) <= 16: every 2x150 bp run,
// 300 bp down to k = 19) and once for up to 24 (300 bp at k = 14..18, reference src/commandline_parser.cpp:14,242-247).  The
// seed count sizes per-lane arrays and the chain records (136 vs 200 bytes each, 30 per problem), so the common case keeps the
// narrow build; cm_create picks the build from max_read_len / kmer and every other call follows the context.  Both builds
// export the same functions under a suffix (_k16 / _k24, renamed on the compiler command line by circminer_amd/_build.py).
#include <cstdlib>

#include "circminer_hot.h"

struct cm_ctx {
    int wide;        // 0: _k16 build, 1: _k24 build
    void *inner;
};

#define CM_VARIANTS(ret, name, params) extern "C" ret name##_k16 params; extern "C" ret name##_k24 params;
CM_VARIANTS(int, cm_create, (const cm_params *, void **))
CM_VARIANTS(void, cm_destroy, (void *))
CM_VARIANTS(const char *, cm_last_error, (const void *))
CM_VARIANTS(int, cm_load_contig, (void *, int, const cm_index_view *))
CM_VARIANTS(int, cm_load_annotation, (void *, int, const cm_annot_view *))
CM_VARIANTS(int, cm_unload_contig, (void *, int))
CM_VARIANTS(int, cm_reads_upload, (void *, const cm_reads *, const cm_mapped_read *))
CM_VARIANTS(int, cm_reads_stage, (void *, const cm_reads *, const cm_mapped_read *))
CM_VARIANTS(int, cm_reads_swap, (void *))
CM_VARIANTS(int, cm_map_rounds, (void *, const int *, int, int))
CM_VARIANTS(int, cm_map_round, (void *, int, int))
CM_VARIANTS(int, cm_sync, (void *))
CM_VARIANTS(int, cm_reads_reset, (void *))
CM_VARIANTS(int, cm_collect_active, (void *, uint64_t, uint64_t *, cm_mapped_read *, uint64_t *))
CM_VARIANTS(int, cm_collect_records, (void *, uint64_t, uint64_t, cm_record *, uint64_t *))
CM_VARIANTS(int, cm_collect_records_device, (void *, uint64_t, uint64_t, void *, uint64_t *))
CM_VARIANTS(int, cm_host_alloc, (void *, uint64_t, void **))
CM_VARIANTS(int, cm_host_free, (void *, void *))
CM_VARIANTS(int, cm_load_contig_raw, (void *, int, const cm_index_raw *))
CM_VARIANTS(int, cm_host_register, (void *, void *, uint64_t))
CM_VARIANTS(int, cm_host_unregister, (void *, void *))
CM_VARIANTS(int, cm_type_histogram, (void *, uint64_t *))
CM_VARIANTS(int, cm_reads_download, (void *, cm_mapped_read *, int32_t *, uint8_t *))
CM_VARIANTS(int, cm_map_batch, (void *, int, int, const cm_reads *, const cm_mapped_read *, cm_mapped_read *, int32_t *))
CM_VARIANTS(int, cm_seed_batch, (void *, int, uint32_t *, uint32_t *, uint32_t *, uint32_t, uint32_t *))
CM_VARIANTS(int, cm_chain_batch, (void *, int, void *, int32_t *, int32_t *))
CM_VARIANTS(int, cm_debug_lane_clk, (void *, unsigned long long *))
CM_VARIANTS(int, cm_debug_counters, (void *, unsigned long long *))
CM_VARIANTS(int, cm_prof_enable, (void *, int))
CM_VARIANTS(int, cm_prof_reset, (void *))
CM_VARIANTS(int, cm_prof_get, (void *, double *, uint64_t *))
CM_VARIANTS(int, cm_prof_counters, (void *, uint64_t *))

#define GO(name, ...) (ctx->wide ? name##_k24(ctx->inner, ##__VA_ARGS__) : name##_k16(ctx->inner, ##__VA_ARGS__))

extern "C" {

int cm_create(const cm_params *p, cm_ctx **out) {
    if (!p || !out) return CM_EINVAL;
    *out = nullptr;
    if (p->kmer < CM_WINDOW_SIZE) return CM_EINVAL;
    const int seeds = p->max_read_len / p->kmer;
    cm_ctx *c = new cm_ctx{seeds > 16 ? 1 : 0, nullptr};
    const int rc = c->wide ? cm_create_k24(p, &c->inner) : cm_create_k16(p, &c->inner);
    if (rc != CM_OK) {
        delete c;
        return rc;
    }
    *out = c;
    return CM_OK;
}
void cm_destroy(cm_ctx *ctx) {
    if (!ctx) return;
    if (ctx->wide) cm_destroy_k24(ctx->inner);
    else cm_destroy_k16(ctx->inner);
    delete ctx;
}
const char *cm_last_error(const cm_ctx *ctx) { return !ctx ? "null context" : (ctx->wide ? cm_last_error_k24(ctx->inner) : cm_last_error_k16(ctx->inner)); }
int cm_load_contig(cm_ctx *ctx, int slot, const cm_index_view *iv) { return ctx ? GO(cm_load_contig, slot, iv) : CM_EINVAL; }
int cm_load_annotation(cm_ctx *ctx, int slot, const cm_annot_view *av) { return ctx ? GO(cm_load_annotation, slot, av) : CM_EINVAL; }
int cm_unload_contig(cm_ctx *ctx, int slot) { return ctx ? GO(cm_unload_contig, slot) : CM_EINVAL; }
int cm_reads_upload(cm_ctx *ctx, const cm_reads *r, const cm_mapped_read *prior) { return ctx ? GO(cm_reads_upload, r, prior) : CM_EINVAL; }
int cm_reads_stage(cm_ctx *ctx, const cm_reads *r, const cm_mapped_read *prior) { return ctx ? GO(cm_reads_stage, r, prior) : CM_EINVAL; }
int cm_reads_swap(cm_ctx *ctx) { return ctx ? GO(cm_reads_swap) : CM_EINVAL; }
int cm_map_rounds(cm_ctx *ctx, const int *slots, int n, int last) { return ctx ? GO(cm_map_rounds, slots, n, last) : CM_EINVAL; }
int cm_map_round(cm_ctx *ctx, int slot, int is_last) { return ctx ? GO(cm_map_round, slot, is_last) : CM_EINVAL; }
int cm_sync(cm_ctx *ctx) { return ctx ? GO(cm_sync) : CM_EINVAL; }
int cm_reads_reset(cm_ctx *ctx) { return ctx ? GO(cm_reads_reset) : CM_EINVAL; }
int cm_collect_active(cm_ctx *ctx, uint64_t cap, uint64_t *idx, cm_mapped_read *st, uint64_t *n) { return ctx ? GO(cm_collect_active, cap, idx, st, n) : CM_EINVAL; }
int cm_collect_records(cm_ctx *ctx, uint64_t base, uint64_t cap, cm_record *out, uint64_t *n) { return ctx ? GO(cm_collect_records, base, cap, out, n) : CM_EINVAL; }
int cm_collect_records_device(cm_ctx *ctx, uint64_t base, uint64_t cap, void *d_out, uint64_t *n) {
    return ctx ? GO(cm_collect_records_device, base, cap, d_out, n) : CM_EINVAL;
}
int cm_host_alloc(cm_ctx *ctx, uint64_t bytes, void **out) { return ctx ? GO(cm_host_alloc, bytes, out) : CM_EINVAL; }
int cm_host_free(cm_ctx *ctx, void *p) { return ctx ? GO(cm_host_free, p) : CM_EINVAL; }
int cm_load_contig_raw(cm_ctx *ctx, int slot, const cm_index_raw *raw) { return ctx ? GO(cm_load_contig_raw, slot, raw) : CM_EINVAL; }
int cm_host_register(cm_ctx *ctx, void *p, uint64_t bytes) { return ctx ? GO(cm_host_register, p, bytes) : CM_EINVAL; }
int cm_host_unregister(cm_ctx *ctx, void *p) { return ctx ? GO(cm_host_unregister, p) : CM_EINVAL; }
int cm_type_histogram(cm_ctx *ctx, uint64_t out[14]) { return ctx ? GO(cm_type_histogram, out) : CM_EINVAL; }
int cm_reads_download(cm_ctx *ctx, cm_mapped_read *st, int32_t *cat, uint8_t *act) { return ctx ? GO(cm_reads_download, st, cat, act) : CM_EINVAL; }
int cm_map_batch(cm_ctx *ctx, int slot, int is_last, const cm_reads *reads, const cm_mapped_read *prior, cm_mapped_read *st, int32_t *cat) {
    return ctx ? GO(cm_map_batch, slot, is_last, reads, prior, st, cat) : CM_EINVAL;
}
int cm_seed_batch(cm_ctx *ctx, int slot, uint32_t *a, uint32_t *b, uint32_t *c, uint32_t cap, uint32_t *n_slots) {
    return ctx ? GO(cm_seed_batch, slot, a, b, c, cap, n_slots) : CM_EINVAL;
}
int cm_chain_batch(cm_ctx *ctx, int slot, cm_chain *out, int32_t *nchain, int32_t *high) {
    if (!ctx) return CM_EINVAL;
    if (ctx->wide) return CM_ELIMIT;          // cm_chain of this ABI holds 16 fragments; the mapping entry points are not affected
    return cm_chain_batch_k16(ctx->inner, slot, out, nchain, high);
}
int cm_debug_lane_clk(cm_ctx *ctx, unsigned long long *out) { return ctx ? GO(cm_debug_lane_clk, out) : CM_EINVAL; }
int cm_debug_counters(cm_ctx *ctx, unsigned long long *out) { return ctx ? GO(cm_debug_counters, out) : CM_EINVAL; }
int cm_prof_enable(cm_ctx *ctx, int on) { return ctx ? GO(cm_prof_enable, on) : CM_EINVAL; }
int cm_prof_reset(cm_ctx *ctx) { return ctx ? GO(cm_prof_reset) : CM_EINVAL; }
int cm_prof_get(cm_ctx *ctx, double ms[8], uint64_t launches[8]) { return ctx ? GO(cm_prof_get, ms, launches) : CM_EINVAL; }
int cm_prof_counters(cm_ctx *ctx, uint64_t c[8]) { return ctx ? GO(cm_prof_counters, c) : CM_EINVAL; }

}  // extern "C"
