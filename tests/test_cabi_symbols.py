"""The C-ABI library loads on a machine without a GPU and exports every symbol that
include/metalens_hip.h declares; no compute call is made."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    text = open(os.path.join(ROOT, 'include', 'metalens_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(ml_[a-z_0-9]+)\s*\(', text)))


def test_header_and_binding_agree():
    from metalens_amd import _lib
    assert header_symbols() == sorted(_lib.SYMBOLS)


def test_library_exports_every_declared_symbol():
    from metalens_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in header_symbols():
        assert hasattr(lib, name), name
    lib.ml_abi_version.restype = ctypes.c_int
    assert lib.ml_abi_version() == 1


def test_product_path_fails_loudly_without_gpu():
    """no CPU fallback: on a box without an MI355X the entry points raise"""
    from metalens_amd import _lib
    lib = _lib.load()
    n = ctypes.c_int(0)
    rc = lib.ml_device_count(ctypes.byref(n))
    if rc == 0 and n.value > 0:
        pytest.skip('a GPU is visible here')
    with pytest.raises(_lib.MetalensHipError):
        _lib.Context(0)


def test_product_does_not_import_the_oracle():
    """the oracle is test infrastructure; nothing under metalens_amd/ may reference it"""
    for dirpath, _, files in os.walk(os.path.join(ROOT, 'metalens_amd')):
        for f in files:
            if f.endswith(('.py', '.hip', '.h')):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle\b', src, flags=re.M), f


def test_pinned_pool_falls_back_to_pageable_memory():
    """`_lib.pinned.empty` hands out page-locked arrays where the host can pin them and ordinary
    NumPy arrays where it cannot (here: no HIP device at all) - a drop-in call must not fail for
    want of pinnable memory"""
    import numpy as np
    from metalens_amd import _lib
    a = _lib.pinned.empty((3, 5), np.complex128)
    assert a.shape == (3, 5) and a.dtype == np.complex128
    a[:] = 1 + 2j
    assert a.sum() == 15 * (1 + 2j)
    _lib.pinned.drain()
