#!/bin/bash
# Diagnostic build of the product library (-DCM_DIAG: phase cut points + per-section wave timers).
# Not shipped, not used by tests; load it with CM_LIB=tests/_hostemu/libcmhot_diag.so.
set -e
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"
OUT="$ROOT/tests/_hostemu"; mkdir -p "$OUT"
for f in cm_hot.hip host_index.cpp host_annot.cpp; do
  /opt/rocm/bin/hipcc -c -O3 -std=c++17 -fPIC -ffp-contract=off -DCM_DIAG --offload-arch=gfx950 -I"$ROOT/include" -I"$ROOT/circminer_amd/csrc" \
      "$ROOT/circminer_amd/csrc/$f" -o "$OUT/diag_${f%.*}.o"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT/libcmhot_diag.so" "$OUT"/diag_*.o -lpthread
echo "$OUT/libcmhot_diag.so"
