"""The C-ABI library builds, loads without a GPU and exports every symbol include/murcl_amd.h declares."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "murcl_amd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(?:int|long)\s+(murcl_\w+)\s*\(", src)))


def test_header_declares_entry_points():
    names = _declared()
    assert "murcl_abmil_pool_fwd" in names and "murcl_ntxent_fwd_bwd" in names and len(names) >= 14


def test_library_builds_and_exports_every_declared_symbol():
    from murcl_amd import build
    lib_path = build.build()
    lib = ctypes.CDLL(lib_path)
    for name in _declared():
        assert hasattr(lib, name), f"{name} declared in include/murcl_amd.h but not exported"


def test_python_binding_covers_header():
    from murcl_amd import _lib
    assert sorted(_lib.SIGNATURES) == _declared()
    L = _lib.lib()                       # resolves all symbols, sets argtypes; no GPU call
    assert L is _lib.lib()


def test_workspace_queries_run_on_host():
    """Pure host entry points (no launch): chunking of the attention-pool work list, NT-Xent workspace."""
    from murcl_amd import ops, _lib
    for B, N, code, tr in [(128, 2048, _lib.BF16, 32), (4, 256, _lib.F32, 16), (1, 1, _lib.F32, 16), (2, 100000, _lib.BF16, 32)]:
        chunk, S = ops.pool_chunks(B, N, code)
        assert chunk % 16 == 0 and chunk <= 2048 and (S - 1) * chunk < N <= S * chunk
    assert _lib.lib().murcl_ntxent_workspace_bytes(128) >= 128 * 130 * 4


def test_unsupported_arguments_are_rejected_without_launching():
    from murcl_amd import _lib
    L = _lib.lib()
    # K not a multiple of 128 bytes -> -1 before any HIP call
    assert L.murcl_gemm_nt(None, None, None, 8, 8, 30, 30, 30, 8, _lib.F32, _lib.F32, 0, None, None, 0, None, None, 0,
                           None, 0, None) == -1
    assert L.murcl_abmil_pool_fwd(None, None, None, None, None, None, None, None, None, None, 1, 8, 256, 128,
                                  _lib.F32, 1, None) == -1          # L != 512
    assert L.murcl_ntxent_fwd_bwd(None, 7, 128, 1.0, None, None, None, 0, 1, 0, None, None) == -1   # odd n
    assert L.murcl_ntxent_fwd_bwd(None, 24, 128, 1.0, None, None, None, 0, 1, 5, None, None) == -1  # n not a multiple of 2*pair_stride
