#!/bin/bash
# Host-side code of the library (index / annotation builders, file formats, FASTQ I/O) under ASan + UBSan.
# The device entry points are replaced by stubs that report "no device": only the host paths are exercised.
set -e
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"
OUT="$ROOT/tests/_hostemu"; mkdir -p "$OUT"
cat > "$OUT/asan_stubs.cpp" <<'CPP'
#include "circminer_hot.h"
extern "C" {
int cm_create(const cm_params *, cm_ctx **out) { if (out) *out = nullptr; return CM_ENODEV; }
void cm_destroy(cm_ctx *) {}
const char *cm_last_error(const cm_ctx *) { return "asan host build: no device"; }
int cm_load_contig(cm_ctx *, int, const cm_index_view *) { return CM_ENODEV; }
int cm_load_annotation(cm_ctx *, int, const cm_annot_view *) { return CM_ENODEV; }
int cm_unload_contig(cm_ctx *, int) { return CM_ENODEV; }
int cm_reads_upload(cm_ctx *, const cm_reads *, const cm_mapped_read *) { return CM_ENODEV; }
int cm_map_round(cm_ctx *, int, int) { return CM_ENODEV; }
int cm_map_rounds(cm_ctx *, const int *, int, int) { return CM_ENODEV; }
int cm_reads_stage(cm_ctx *, const cm_reads *, const cm_mapped_read *) { return CM_ENODEV; }
int cm_reads_swap(cm_ctx *) { return CM_ENODEV; }
int cm_debug_lane_clk(cm_ctx *, unsigned long long *) { return CM_ENODEV; }
int cm_sync(cm_ctx *) { return CM_ENODEV; }
int cm_reads_reset(cm_ctx *) { return CM_ENODEV; }
int cm_collect_active(cm_ctx *, uint64_t, uint64_t *, cm_mapped_read *, uint64_t *) { return CM_ENODEV; }
int cm_collect_records(cm_ctx *, uint64_t, uint64_t, cm_record *, uint64_t *) { return CM_ENODEV; }
int cm_collect_records_device(cm_ctx *, uint64_t, uint64_t, void *, uint64_t *) { return CM_ENODEV; }
int cm_host_alloc(cm_ctx *, uint64_t, void **) { return CM_ENODEV; }
int cm_host_free(cm_ctx *, void *) { return CM_ENODEV; }
int cm_host_register(cm_ctx *, void *, uint64_t) { return CM_ENODEV; }
int cm_host_unregister(cm_ctx *, void *) { return CM_ENODEV; }
int cm_type_histogram(cm_ctx *, uint64_t *) { return CM_ENODEV; }
int cm_debug_counters(cm_ctx *, unsigned long long *) { return CM_ENODEV; }
int cm_reads_download(cm_ctx *, cm_mapped_read *, int32_t *, uint8_t *) { return CM_ENODEV; }
int cm_map_batch(cm_ctx *, int, int, const cm_reads *, const cm_mapped_read *, cm_mapped_read *, int32_t *) { return CM_ENODEV; }
int cm_seed_batch(cm_ctx *, int, uint32_t *, uint32_t *, uint32_t *, uint32_t, uint32_t *) { return CM_ENODEV; }
int cm_chain_batch(cm_ctx *, int, cm_chain *, int32_t *, int32_t *) { return CM_ENODEV; }
int cm_prof_enable(cm_ctx *, int) { return CM_ENODEV; }
int cm_prof_reset(cm_ctx *) { return CM_ENODEV; }
int cm_prof_get(cm_ctx *, double *, uint64_t *) { return CM_ENODEV; }
int cm_prof_counters(cm_ctx *, uint64_t *) { return CM_ENODEV; }
}
CPP
g++ -O1 -g -std=c++17 -fPIC -shared -fsanitize=address,undefined -fno-omit-frame-pointer -I "$ROOT/include" -I "$ROOT/circminer_amd/csrc" \
    "$OUT/asan_stubs.cpp" "$ROOT/circminer_amd/csrc/host_index.cpp" "$ROOT/circminer_amd/csrc/host_annot.cpp" \
    "$ROOT/circminer_amd/csrc/host_index_io.cpp" "$ROOT/circminer_amd/csrc/host_fastq.cpp" "$ROOT/circminer_amd/csrc/host_mapping.cpp" \
    "$ROOT/circminer_amd/csrc/host_circ.cpp" "$ROOT/circminer_amd/csrc/host_circ_call.cpp" -o "$OUT/libcmhost_asan.so" -lpthread -lz
cd "$ROOT"
LD_PRELOAD="$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so)" ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 \
UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 CM_LIB="$OUT/libcmhost_asan.so" \
    python -m pytest tests/test_index_files.py tests/test_fastq_io.py tests/test_host_builders.py tests/test_circ_stage2.py tests/test_circ_call.py -x -q "$@"
