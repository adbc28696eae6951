"""CPU tier: the C-ABI library loads and exports every symbol include/fwgpu.h declares; the host-side
mirror fails loudly (no fallback) when there is no GPU.  No compute calls here."""
import os
import re

import pytest

import firewheel_amd as fa
from firewheel_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    fa.build_library()
    return fa.load_library()


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "fwgpu.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(fwgpu_[a-z0-9_]+)\s*\(", src)))


def test_header_and_binding_agree():
    syms = declared_symbols()
    assert len(syms) >= 35
    assert set(syms) == set(_lib.SIGNATURES.keys())


def test_library_exports_every_declared_symbol(lib):
    for s in declared_symbols():
        assert hasattr(lib, s), s


def test_no_torch_or_oracle_dependency():
    import subprocess

    out = subprocess.check_output(["ldd", fa.LIB_PATH]).decode()
    assert "torch" not in out and "oracle" not in out
    assert "amdhip64" in out


def test_product_never_references_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "firewheel_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "fw_oracle" not in txt and "oracle/" not in txt, f


def test_ctx_create_fails_loudly_without_gpu(lib):
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(fa.FwgpuError) as ei:
        fa.FirewheelGpuCtx()
    assert "no CPU fallback" in str(ei.value) or "HIP" in str(ei.value)


def test_header_is_plain_c99_and_the_c_host_builds(lib):
    # the boundary is a C ABI: the header must compile as C (not only as C++), and a C host must link against it
    import subprocess

    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-fsyntax-only", "-x", "c",
                           os.path.join(ROOT, "include", "fwgpu.h")])
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "examples", "host_c")])
    assert os.path.exists(os.path.join(ROOT, "examples", "host_c", "fw_host"))
