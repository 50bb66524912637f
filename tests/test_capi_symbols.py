"""CPU-only: the C-ABI library builds for gfx950, loads, and exports every symbol that
include/plonky_hip.h declares (no compute calls: there is no GPU here)."""
import ctypes
import os
import re

from plonky_amd import lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "plonky_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(plk_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_tables_agree():
    assert declared_symbols() == sorted(name for name, _, _ in lib.SYMBOLS)


def test_library_exports_every_declared_symbol():
    lib.build()
    L = ctypes.CDLL(lib.SO_PATH)
    for name in declared_symbols():
        assert hasattr(L, name), name
    lib.load()


def test_calls_fail_loudly_without_a_gpu():
    """No silent CPU fallback: with no device the entry points return PLK_ERR_NO_DEVICE."""
    import torch
    if torch.cuda.is_available():
        return
    import numpy as np
    L = lib.load()
    x = np.zeros((4, 4), dtype=np.uint64)
    rc = L.plk_ntt(0, 2, 0, x.ctypes.data, x.ctypes.data)
    assert rc in (lib.PLK_ERR_NO_DEVICE, lib.PLK_ERR_HIP)
    assert L.plk_field_limbs(3) == 6 and L.plk_curve_limbs(2) == 6 and L.plk_curve_scalar_field(0) == 1
    assert L.plk_field_limbs(9) < 0


def test_product_code_never_touches_the_oracle():
    pkg = os.path.join(ROOT, "plonky_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".cuh", ".h", "Makefile")):
                src = open(os.path.join(dirpath, fn), errors="replace").read()
                assert "oracle_lib" not in src and "liboracle" not in src and "bigint_ref import" not in src, fn
