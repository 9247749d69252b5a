"""Builds circminer_amd/csrc/libcmhot.so (gfx950 HIP kernels + C-ABI + host-side builders) in-tree.

hipcc cross-compiles for gfx950 without a GPU; the .so travels to the GPU box with the repo
snapshot.  `-ffp-contract=off` is required for parity: chain scores are fp64 sums whose rounding
order is part of the result (reference src/chain.cpp:189).
"""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INC = os.path.join(HERE, "..", "include")
OUT = os.path.join(CSRC, "libcmhot.so")
SOURCES = ["cm_hot.hip", "cm_dispatch.cpp", "host_index.cpp", "host_annot.cpp", "host_index_io.cpp", "host_fastq.cpp", "host_mapping.cpp", "host_circ.cpp",
           "host_circ_call.cpp"]
DEPS = SOURCES + sorted(f for f in os.listdir(CSRC) if f.endswith(".h")) + [os.path.join("..", "..", "include", "circminer_hot.h")]
# cm_hot.hip is compiled twice (reads of <= 16 seeds / <= 24 seeds, see cm_dispatch.cpp): its exported names get a suffix
KERNEL_EXPORTS = ['cm_create', 'cm_destroy', 'cm_last_error', 'cm_load_contig', 'cm_load_contig_raw', 'cm_load_annotation', 'cm_unload_contig', 'cm_reads_upload', 'cm_reads_stage', 'cm_reads_swap', 'cm_map_rounds', 'cm_map_round', 'cm_sync', 'cm_reads_reset', 'cm_collect_active', 'cm_collect_records', 'cm_collect_records_device', 'cm_host_alloc', 'cm_host_free', 'cm_host_register', 'cm_host_unregister', 'cm_type_histogram', 'cm_reads_download', 'cm_map_batch', 'cm_seed_batch', 'cm_chain_batch', 'cm_debug_lane_clk', 'cm_debug_counters', 'cm_prof_enable', 'cm_prof_reset', 'cm_prof_get', 'cm_prof_counters', 'cm_ctx', 'cm_chain']
VARIANTS = (("k16", []), ("k24", ["-DCM_MAX_CHAIN_FRAGS=24"]))


def needs_build(out: str = OUT) -> bool:
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(os.path.join(CSRC, d)) > t for d in DEPS)


def build(force: bool = False, verbose: bool = False, tag: str = "", flags=()) -> str:
    """tag / flags: a second library libcmhot_<tag>.so compiled with extra -D flags (test builds, e.g. a 2-entry extension
    memo that sends many pairs through the re-run launch); the product library has neither."""
    OUT = os.path.join(CSRC, f"libcmhot_{tag}.so") if tag else globals()["OUT"]
    if not force and not needs_build(OUT):
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    extra = os.environ.get("CM_EXTRA_FLAGS", "").split() + list(flags)
    sfx = ("_" + tag) if tag else ""
    objs = []
    jobs = []
    for s in SOURCES:
        base = [hipcc, "-c", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", *extra, "-I", INC, "-I", CSRC]
        if s.endswith(".hip"):
            arch = ["--offload-arch=gfx950", "-Rpass-analysis=kernel-resource-usage"] if verbose else ["--offload-arch=gfx950"]
            for vtag, vflags in VARIANTS:
                o = os.path.join(CSRC, s.rsplit(".", 1)[0] + "_" + vtag + sfx + ".o")
                ren = [f"-D{n}={n}_{vtag}" for n in KERNEL_EXPORTS + ["cmc"]]      # cmc: the kernel bodies' namespace (their structs differ in size)
                jobs.append((s + " [" + vtag + "]", [base[0]] + arch + base[1:] + vflags + ren + [os.path.join(CSRC, s), "-o", o]))
                objs.append(o)
        else:
            o = os.path.join(CSRC, s.rsplit(".", 1)[0] + sfx + ".o")
            jobs.append((s, [base[0], "-x", "c++"] + base[1:] + [os.path.join(CSRC, s), "-o", o]))   # host-only sources: plain C++
            objs.append(o)
    procs = [(name, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)) for name, cmd in jobs]   # in parallel
    failed = []
    for name, pr in procs:
        out, _ = pr.communicate()
        if verbose or pr.returncode != 0:
            sys.stderr.write(out)
        if pr.returncode != 0:
            failed.append(name)
    if failed:
        raise RuntimeError(f"hipcc failed on {failed}")
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs + ["-lpthread", "-lz"],
                       capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout + r.stderr)
        raise RuntimeError("link failed")
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="-v" in sys.argv))
