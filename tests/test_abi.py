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


def test_rust_binding_block_of_integration_md_matches_the_header():
    """INTEGRATION.md shows the `extern "C"` block a Firewheel maintainer would add (no rustc here to compile it): every
    function it declares must exist in include/fwgpu.h with the same number of parameters and a compatible return type."""
    import re

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    md = open(os.path.join(root, "INTEGRATION.md")).read()
    hdr = open(os.path.join(root, "include", "fwgpu.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    c_decl = {}
    for m in re.finditer(r"([A-Za-z_][A-Za-z_0-9 \*]*?)\b(fwgpu_[a-z_0-9]+)\s*\(([^;{]*?)\)\s*;", hdr, flags=re.S):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        c_decl[name] = (ret, 0 if args in ("", "void") else len(args.split(",")))
    block = md[md.index('extern "C" {'):]
    block = block[:block.index("\n}")]
    rust = re.findall(r"pub fn (fwgpu_[a-z_0-9]+)\s*\(([^)]*)\)\s*(?:->\s*([^;]+))?;", block, flags=re.S)
    assert len(rust) >= 25
    ret_ok = {"c_int": ("int",), "i64": ("int64_t",), "*const c_char": ("const char*", "const char *"),
              "*mut fwgpu_ctx": ("fwgpu_ctx*", "fwgpu_ctx *"), None: ("void",)}
    for name, args, ret in rust:
        assert name in c_decl, "%s is not in fwgpu.h" % name
        n_args = len([a for a in args.split(",") if a.strip()])
        assert n_args == c_decl[name][1], (name, n_args, c_decl[name][1])
        r = ret.strip() if ret else None
        assert c_decl[name][0] in ret_ok[r], (name, r, c_decl[name][0])


def test_the_product_never_loads_the_oracle_or_the_test_harnesses():
    """oracle/ and tests/*_harness are test infrastructure: nothing under firewheel_amd/ or include/ may import, link or
    dlopen them (comments that cite the oracle's arithmetic are fine), and libfwgpu's build recipe names product sources only."""
    import re

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pkg = os.path.join(root, "firewheel_amd")
    bad = []
    for d, _, files in os.walk(pkg):
        for f in files:
            if not f.endswith((".py", ".cpp", ".h", ".hip", "Makefile")):
                continue
            text = open(os.path.join(d, f), errors="replace").read()
            code = re.sub(r"//[^\n]*|/\*.*?\*/", "", text, flags=re.S) if not f.endswith(".py") else re.sub(r"#[^\n]*", "", text)
            if re.search(r"fw_oracle|libfw_oracle|oracle/|fwo_|host_harness|planner_harness|fakehip|fwh_", code):
                bad.append(os.path.join(d, f))
    assert not bad, bad
    mk = open(os.path.join(pkg, "csrc", "Makefile")).read()
    srcs = re.search(r"^SRCS\s*:=\s*(.*)$", mk, flags=re.M).group(1).split()
    assert sorted(srcs) == ["fwgpu_abi.cpp", "fwgpu_control_math.cpp", "fwgpu_graph.cpp", "fwgpu_kernels.hip",
                            "fwgpu_plan_detect.cpp", "fwgpu_plan_install.cpp", "fwgpu_run.cpp"]
