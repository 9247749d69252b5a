#!/bin/bash
# Diagnostic build of the product library (-DCM_DIAG: phase cut points + per-section wave timers).
# DIAG_FLAGS / DIAG_NAME select other experiment macros and the output name.
# Not shipped, not used by tests; load it with CM_LIB=tests/_hostemu/libcmhot_diag.so.
set -e
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"
OUT="$ROOT/tests/_hostemu"; mkdir -p "$OUT"
for f in cm_hot.hip host_index.cpp host_annot.cpp host_index_io.cpp host_fastq.cpp host_mapping.cpp host_circ.cpp host_circ_call.cpp; do
  if [ "${f##*.}" = "hip" ]; then X="--offload-arch=gfx950"; else X="-x c++"; fi
  /opt/rocm/bin/hipcc $X -c -O3 -std=c++17 -fPIC -ffp-contract=off ${DIAG_FLAGS--DCM_DIAG} -I"$ROOT/include" -I"$ROOT/circminer_amd/csrc" \
      "$ROOT/circminer_amd/csrc/$f" -o "$OUT/${DIAG_NAME-diag}_${f%.*}.o"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT/libcmhot_${DIAG_NAME-diag}.so" "$OUT/${DIAG_NAME-diag}"_*.o -lpthread -lz
echo "$OUT/libcmhot_${DIAG_NAME-diag}.so"
