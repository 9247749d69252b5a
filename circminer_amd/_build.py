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
SOURCES = ["cm_hot.hip", "host_index.cpp", "host_annot.cpp", "host_index_io.cpp", "host_fastq.cpp", "host_mapping.cpp", "host_circ.cpp", "host_circ_call.cpp"]
DEPS = SOURCES + sorted(f for f in os.listdir(CSRC) if f.endswith(".h")) + [os.path.join("..", "..", "include", "circminer_hot.h")]


def needs_build() -> bool:
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(os.path.join(CSRC, d)) > t for d in DEPS)


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    extra = os.environ.get("CM_EXTRA_FLAGS", "").split()
    objs = []
    for s in SOURCES:
        o = os.path.join(CSRC, s.rsplit(".", 1)[0] + ".o")
        cmd = [hipcc, "-c", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", *extra, "-I", INC, "-I", CSRC,
               os.path.join(CSRC, s), "-o", o]
        if s.endswith(".hip"):
            cmd[1:1] = ["--offload-arch=gfx950", "-Rpass-analysis=kernel-resource-usage"] if verbose else ["--offload-arch=gfx950"]
        else:
            cmd[1:1] = ["-x", "c++"]                 # host-only sources: plain C++ (hipcc would otherwise treat .cpp as HIP)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if verbose or r.returncode != 0:
            sys.stderr.write(r.stdout + r.stderr)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {s}")
        objs.append(o)
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs + ["-lpthread", "-lz"],
                       capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout + r.stderr)
        raise RuntimeError("link failed")
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="-v" in sys.argv))
