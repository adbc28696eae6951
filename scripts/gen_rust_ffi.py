"""Generates rust/firewheel-gpu/src/ffi.rs from include/fwgpu.h (the job `bindgen` does on a machine with libclang and
a Rust toolchain; this image has neither).  tests/test_abi.py re-runs the generator and requires the committed file to
be identical, so the Rust binding cannot drift from the header.  usage: python scripts/gen_rust_ffi.py [--check]"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HDR = os.path.join(ROOT, "include", "fwgpu.h")
OUT = os.path.join(ROOT, "rust", "firewheel-gpu", "src", "ffi.rs")

BASE = {"int": "c_int", "uint32_t": "u32", "uint64_t": "u64", "int64_t": "i64", "float": "f32", "double": "f64", "void": "c_void",
        "char": "c_char", "uint8_t": "u8", "size_t": "usize", "fwgpu_ctx": "fwgpu_ctx", "fwgpu_stream": "fwgpu_stream",
        "fwgpu_sched_node": "fwgpu_sched_node", "fwgpu_bus_exchange": "fwgpu_bus_exchange", "fwgpu_rccl_comm": "fwgpu_rccl_comm",
        "fwgpu_host_process_fn": "fwgpu_host_process_fn"}


def rust_type(c):
    """C declarator type (no name) -> Rust.  Handles `const T*`, `T*`, `const T* const*`, `T* const*`."""
    c = " ".join(c.replace("*", " * ").split())
    toks = c.split()
    ptrs = []  # innermost first: (is_const_pointee)
    # parse: [const] base {* [const]}*
    i = 0
    const_base = False
    if toks[i] == "const":
        const_base = True
        i += 1
    base = BASE[toks[i]]
    i += 1
    cur_const = const_base
    out = base
    while i < len(toks):
        assert toks[i] == "*", c
        out = ("*const " if cur_const else "*mut ") + out
        i += 1
        cur_const = False
        if i < len(toks) and toks[i] == "const":
            cur_const = True
            i += 1
    return out


def parse_header():
    src = open(HDR).read()
    nocom = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    enums = []
    for m in re.finditer(r"enum\s+(fwgpu_[a-z_]+)\s*\{(.*?)\};", nocom, flags=re.S):
        items = []
        for it in m.group(2).split(","):
            it = it.strip()
            if not it:
                continue
            name, val = [x.strip() for x in it.split("=")]
            items.append((name, int(val)))
        enums.append((m.group(1), items))
    funcs = []
    for m in re.finditer(r"^([A-Za-z_][A-Za-z_0-9 \*]*?)\b(fwgpu_[a-z_0-9]+)\s*\(([^;{]*?)\)\s*;", nocom, flags=re.S | re.M):
        ret, name, args = " ".join(m.group(1).split()), m.group(2), " ".join(m.group(3).split())
        if ret.startswith("typedef"):
            continue
        params = []
        if args not in ("", "void"):
            for a in args.split(","):
                a = a.strip()
                mm = re.match(r"(.*?)([A-Za-z_][A-Za-z_0-9]*)$", a)
                params.append((mm.group(2), mm.group(1).strip()))
        funcs.append((name, ret, params))
    return enums, funcs


def generate():
    enums, funcs = parse_header()
    o = []
    o.append("// ffi.rs — GENERATED from include/fwgpu.h by scripts/gen_rust_ffi.py (do not edit; tests/test_abi.py keeps it in sync).")
    o.append("// The raw C ABI of libfwgpu, the MI355X executor behind Firewheel's AudioNodeProcessor / FirewheelProcessor.")
    o.append("#![allow(non_camel_case_types, dead_code)]")
    o.append("use std::os::raw::{c_char, c_int, c_void};")
    o.append("")
    o.append("#[repr(C)]\npub struct fwgpu_ctx {\n    _private: [u8; 0],\n}")
    o.append("#[repr(C)]\npub struct fwgpu_stream {\n    _private: [u8; 0],\n}")
    o.append("#[repr(C)]\npub struct fwgpu_bus_exchange {\n    _private: [u8; 0],\n}")
    o.append("pub const FWGPU_EXCHANGE_HANDLE_BYTES: usize = %d;" % int(re.search(r"#define FWGPU_EXCHANGE_HANDLE_BYTES (\d+)", open(HDR).read()).group(1)))
    o.append("#[repr(C)]\npub struct fwgpu_rccl_comm {\n    _private: [u8; 0],\n}")
    o.append("pub const FWGPU_RCCL_UNIQUE_ID_BYTES: usize = %d;" % int(re.search(r"#define FWGPU_RCCL_UNIQUE_ID_BYTES (\d+)", open(HDR).read()).group(1)))
    o.append("/// AudioNodeProcessor::process + ProcInfo (core/node.rs:37-53,94-118) as the C callback of a FWGPU_HOST_NODE")
    o.append("pub type fwgpu_host_process_fn = Option<\n    unsafe extern \"C\" fn(\n        user: *mut c_void,\n        frames: u64,\n        inputs: *const *const f32,\n"
             "        num_inputs: u32,\n        outputs: *const *mut f32,\n        num_outputs: u32,\n        in_silence_mask: u64,\n        out_silence_mask: *mut u64,\n"
             "        stream_time_secs: f64,\n        stream_status: u32,\n    ),\n>;")
    o.append("/// one ScheduledNode of Firewheel's CompiledSchedule (graph/graph/compiler/schedule.rs:12-30)")
    o.append("#[repr(C)]\npub struct fwgpu_sched_node {\n    pub node: i64,\n    pub num_inputs: u32,\n    pub num_outputs: u32,\n"
             "    pub in_buffer_index: *const u32,\n    pub in_should_clear: *const u8,\n    pub out_buffer_index: *const u32,\n}")
    o.append("")
    for ename, items in enums:
        o.append("// enum %s" % ename)
        for name, val in items:
            o.append("pub const %s: c_int = %d;" % (name, val))
        o.append("")
    o.append('#[link(name = "fwgpu")]')
    o.append('extern "C" {')
    for name, ret, params in funcs:
        ps = ", ".join("%s: %s" % ("r#%s" % n if n in ("type", "in", "fn", "loop", "ref") else n, rust_type(t)) for n, t in params)
        r = "" if ret == "void" else " -> %s" % rust_type(ret)
        o.append("    pub fn %s(%s)%s;" % (name, ps, r))
    o.append("}")
    return "\n".join(o) + "\n"


if __name__ == "__main__":
    text = generate()
    if "--check" in sys.argv:
        sys.exit(0 if open(OUT).read() == text else 1)
    open(OUT, "w").write(text)
    print("wrote", OUT)
