"""The product never routes through the oracle or any CPU fallback."""
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _py_files(d):
    for base, _, files in os.walk(os.path.join(ROOT, d)):
        for f in files:
            if f.endswith(".py"):
                yield os.path.join(base, f)


def test_product_package_never_imports_oracle_or_reference():
    for path in _py_files("murcl_amd"):
        src = open(path).read()
        assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), path
        assert "/root/reference" not in src, path


def test_bench_uses_oracle_only_in_cpu_baseline():
    src = open(os.path.join(ROOT, "bench.py")).read()
    hits = [m.start() for m in re.finditer(r"from oracle import|import oracle", src)]
    assert len(hits) == 1
    fn_start = src.rfind("\ndef ", 0, hits[0])
    assert src[fn_start:hits[0]].lstrip().startswith("def cpu_baseline")
    assert "/root/reference" not in src


def test_gpu_tests_and_smoke_do_not_read_reference():
    for path in list(_py_files("tests")) + [os.path.join(ROOT, "__graft_entry__.py")]:
        if path.endswith("test_product_isolation.py"):
            continue
        assert "/root/reference" not in open(path).read(), path


def test_cpu_tensors_raise_instead_of_falling_back():
    from murcl_amd import ops
    from murcl_amd.models.abmil import ABMIL
    from murcl_amd.utils.losses import NT_Xent
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.gemm_nt(torch.zeros(4, 32), torch.zeros(4, 32))
    with pytest.raises(RuntimeError):
        ABMIL(512)(torch.zeros(2, 8, 512))
    with pytest.raises(RuntimeError):
        NT_Xent(2, 1.0)(torch.zeros(2, 128), torch.zeros(2, 128))


def test_missing_library_fails_loudly(monkeypatch):
    from murcl_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libmurcl_amd.so")
    with pytest.raises(RuntimeError, match="no CPU/PyTorch fallback"):
        _lib.lib()
