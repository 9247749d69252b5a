#!/bin/bash
# Runs the host-emulation parity tests with the kernel bodies (cm_core.h) compiled under ASan + UBSan (CPU only).
set -e
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"
mkdir -p "$ROOT/tests/_hostemu"
g++ -O1 -g -std=c++17 -fPIC -shared -ffp-contract=off -fsanitize=address,undefined -fno-omit-frame-pointer \
    -I "$ROOT/include" -I "$ROOT/circminer_amd/csrc" "$ROOT/tests/hostemu.cpp" -o "$ROOT/tests/_hostemu/libcmemu_asan.so"
cd "$ROOT"
LD_PRELOAD="$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so)" ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 \
UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 CM_EMU_LIB="$ROOT/tests/_hostemu/libcmemu_asan.so" \
    python -m pytest tests/test_hostemu_parity.py -x -q "$@"
