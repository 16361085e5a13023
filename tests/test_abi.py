"""The C-ABI library loads and exports every symbol include/esfm.h declares; host-only entry points
work without a GPU; compute entry points fail loudly (no CPU fallback).  CPU only."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def L():
    import easysfm_amd as E
    if not os.path.exists(E.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    return E.lib()


def test_header_symbols_all_exported(L):
    hdr = open(os.path.join(ROOT, "include", "esfm.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(esfm_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"esfm_allreduce_fn"}
    from easysfm_amd._lib import EXPORTED_SYMBOLS
    assert declared == set(EXPORTED_SYMBOLS), declared ^ set(EXPORTED_SYMBOLS)
    for s in declared:
        assert hasattr(L, s), s


def test_header_cites_reference_interfaces():
    hdr = open(os.path.join(ROOT, "include", "esfm.h")).read()
    for cite in ("feature_matching.h:17-18", "feature_matching.h:20-21", "ba.h:84", "sfm.cpp:140-161", "feature_matching.cpp:125", "ba.cpp:58-114"):
        assert cite in hdr, cite


def test_no_cpu_fallback_without_gpu(L):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    import easysfm_amd as E
    assert L.esfm_device_count() == 0
    with pytest.raises(E.EsfmError) as ei:
        E.Context(0)
    assert ei.value.status == -2 and "no CPU fallback" in str(ei.value)
    with pytest.raises(E.EsfmError):
        E.match_l2(np.zeros((2, 64), np.float32), np.zeros((3, 64), np.float32))


def test_product_never_imports_oracle():
    """The product tree must not import, include, link or dlopen the oracle (parity claims depend on it)."""
    pat = re.compile(r"^\s*(import|from)\s+oracle\b|#\s*include\s*[\"<][^\">]*oracle|libesfm_oracle|dlopen\([^)]*oracle", re.M)
    bad = []
    for base, _, files in os.walk(os.path.join(ROOT, "easysfm_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".hpp", ".h", "Makefile")):
                if pat.search(open(os.path.join(base, f), errors="replace").read()):
                    bad.append(os.path.join(base, f))
    assert not bad, bad


def test_options_default_match_reference_settings(L):
    import easysfm_amd as E
    o = E.default_options()
    assert o.max_num_iterations == 50           # ba.cpp:202
    assert o.cauchy_a == 0.5                    # ba.cpp:150
    assert (o.initial_trust_region_radius, o.min_relative_decrease, o.function_tolerance) == (1e4, 1e-3, 1e-6)
    assert (o.min_lm_diagonal, o.max_lm_diagonal, o.gradient_tolerance, o.parameter_tolerance) == (1e-6, 1e32, 1e-10, 1e-8)
