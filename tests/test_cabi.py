"""C-ABI surface: the library loads, exports every symbol include/circminer_hot.h declares, and
fails loudly (never falls back to a CPU path) when no HIP device is present."""
import ctypes as C
import os
import re

import pytest

from circminer_amd import lib as cl

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "circminer_hot.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(cm_[a-z_]+)\s*\(", hdr)))


def test_header_symbols_exported():
    L = cl.load()
    names = _declared_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(L, n), f"libcmhot.so does not export {n}"
    assert set(names) == set(cl.EXPORTED_SYMBOLS), set(names) ^ set(cl.EXPORTED_SYMBOLS)


def test_struct_layouts_match_header():
    assert C.sizeof(cl.MappedRead) == 72 and cl.MAPPED_DTYPE.itemsize == 72
    assert cl.CHAIN_DTYPE.itemsize == 8 + 4 * 16 * 2
    assert C.sizeof(cl.Params) == 48


def test_no_cpu_fallback_without_device():
    import torch
    L = cl.load()
    h = C.c_void_p()
    P = cl.default_params()
    rc = L.cm_create(C.byref(P), C.byref(h))
    if torch.cuda.is_available():
        assert rc == 0
        L.cm_destroy(h)
    else:
        assert rc == -2 and not h          # CM_ENODEV: the hot path is HIP-only
        with pytest.raises(RuntimeError):
            cl.HotPath(P)


def test_bad_params_rejected():
    L = cl.load()
    h = C.c_void_p()
    for kw in (dict(kmer=13), dict(kmer=23), dict(max_chain_len=31), dict(band=9), dict(seed_lim=0)):
        P = cl.default_params(**kw)
        assert L.cm_create(C.byref(P), C.byref(h)) == -1      # CM_EINVAL before any device is touched


def test_product_does_not_reference_oracle():
    """The oracle is test infrastructure: nothing under circminer_amd/ may import, link or call it."""
    for dp, _, fs in os.walk(os.path.join(ROOT, "circminer_amd")):
        for f in fs:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                txt = open(os.path.join(dp, f)).read()
                assert "oracle_py" not in txt and "cm_oracle" not in txt and "libcmoracle" not in txt, f
