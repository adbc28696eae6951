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


def test_library_exports_nothing_else(lib):
    """... and nothing beyond it: every `fwgpu_*` symbol the library exports is one the header declares (a probe build —
    make EXTRA=-DFW_PROBE, scripts/placement_probe.py — left in the tree by accident would show up here)."""
    import subprocess

    out = subprocess.check_output(["nm", "-D", "--defined-only", fa.LIB_PATH]).decode()
    exported = {ln.split()[-1] for ln in out.splitlines() if ln.split() and ln.split()[-1].startswith("fwgpu_")}
    assert exported == set(declared_symbols()), sorted(exported ^ set(declared_symbols()))


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


def test_rust_crate_ffi_matches_the_header():
    """rust/firewheel-gpu is the reference-side binding shipped as a real source crate (no rustc here to compile it).  Its
    src/ffi.rs is generated from include/fwgpu.h: the committed file must be exactly what the generator produces now, must
    declare every function of the header with the same parameter count, and every `ffi::fwgpu_*` the hand-written wrapper
    modules call must be a declared function used with the declared number of arguments."""
    import re
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    crate = os.path.join(root, "rust", "firewheel-gpu")
    for f in ("Cargo.toml", "build.rs", "src/lib.rs", "src/ffi.rs", "src/nodes.rs", "src/sample.rs", "src/stream.rs", "src/exchange.rs", "src/host_node.rs"):
        assert os.path.exists(os.path.join(crate, f)), f
    assert subprocess.call([sys.executable, os.path.join(root, "scripts", "gen_rust_ffi.py"), "--check"]) == 0, \
        "rust/firewheel-gpu/src/ffi.rs is stale: run python scripts/gen_rust_ffi.py"
    hdr = re.sub(r"/\*.*?\*/", "", open(os.path.join(root, "include", "fwgpu.h")).read(), flags=re.S)
    c_decl = {}
    for m in re.finditer(r"^([A-Za-z_][A-Za-z_0-9 \*]*?)\b(fwgpu_[a-z_0-9]+)\s*\(([^;{]*?)\)\s*;", hdr, flags=re.S | re.M):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        if ret.startswith("typedef"):
            continue
        c_decl[name] = 0 if args in ("", "void") else len(args.split(","))
    assert set(c_decl) == set(_lib.SIGNATURES)
    ffi = open(os.path.join(crate, "src", "ffi.rs")).read()
    rust = dict((n, len([a for a in args.split(",") if a.strip()]))
                for n, args in re.findall(r"pub fn (fwgpu_[a-z_0-9]+)\s*\(([^)]*)\)", ffi))
    assert rust == c_decl
    # the wrappers: every call site names a declared function and passes as many arguments as it takes
    calls = 0
    for f in ("lib.rs", "nodes.rs", "sample.rs", "stream.rs", "exchange.rs", "host_node.rs"):
        src = open(os.path.join(crate, "src", f)).read()
        src = re.sub(r"//[^\n]*", "", src)
        for m in re.finditer(r"ffi::(fwgpu_[a-z_0-9]+)\s*\(", src):
            name = m.group(1)
            assert name in c_decl, (f, name)
            depth, i, n_args, any_tok = 1, m.end(), 0, False
            while depth:
                ch = src[i]
                if ch in "([{":
                    depth += 1
                elif ch in ")]}":
                    depth -= 1
                elif ch == "," and depth == 1:
                    n_args += 1
                    any_tok = False
                    i += 1
                    continue
                if depth and not ch.isspace():
                    any_tok = True
                i += 1
            n_args += 1 if any_tok else 0
            assert n_args == c_decl[name], (f, name, n_args, c_decl[name])
            calls += 1
    assert calls >= 25


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
    assert sorted(srcs) == ["fwgpu_abi.cpp", "fwgpu_control_math.cpp", "fwgpu_exchange.cpp", "fwgpu_graph.cpp", "fwgpu_kernels.hip",
                            "fwgpu_plan_detect.cpp", "fwgpu_plan_install.cpp", "fwgpu_rccl.cpp", "fwgpu_run.cpp"]
