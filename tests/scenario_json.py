"""The parity scenarios as LANGUAGE-NEUTRAL data (VERDICT r3 "next" #4).

tests/scenarios.py holds the 50-odd parity scenarios as Python functions over an engine API.  The only engine that can pin
"parity unpinned" is the reference itself — Rust, on a machine with cargo — so the scenarios are also exported as JSON
(tests/golden/scenarios/<name>.json, written by tests/golden/make_scenarios_json.py): the exact sequence of primitive calls a
scenario makes on the engine underneath (`Recorder` below sits where the raw OracleEngine sits, UNDER scenarios.TaggedOracle, so
messages tagged `at_block` arrive here already flattened into "message, then process one block"), the sample data as generator
recipes, and the sha256 of every process call's output.

    replay(doc, engine)        runs a document on any engine with fwapi's primitive surface (CPU tier: the oracle;
                               rust/firewheel-gpu/tests/reference_digests.rs does the same on firewheel-graph)
    Recorder(engine)           the recording proxy

Document format (version 1):
  {"version": 1, "name": ..., "sample_rate": 48000, "max_block_frames": B, "num_graph_inputs": I, "num_graph_outputs": O,
   "node_kinds": [...kinds used...], "reference_kinds_only": bool,
   "ops": [ [opcode, args...], ... ], "sha256": digest of all process outputs concatenated}
  nodes are numbered in creation order (0, 1, ...); -1 = graph_in, -2 = graph_out; samples and edges likewise by creation order.
  ops:
   ["add_node", kind, n_in, n_out, [params...]]                       kinds / params: include/fwgpu.h FWGPU_KIND_* (= the reference's
                                                                      node constructors' arguments, nodes/*.rs)
   ["remove_node", node]   ["connect", src, src_port, dst, dst_port, check_for_cycles, expected_error_or_0]
   ["disconnect", src, sp, dst, dp]   ["disconnect_edge", edge]   ["update", expected_error_or_0]
   ["new_sample", fmt, channels, frames, data]                        fmt: 0 i16 interleaved, 1 u16 i., 2 f32 i., 3 i16 planar, 4 u16 p., 5 f32 p.
   ["set_param", node, param, value]   ["set_sample", node, sample, stop_playback]   ["play" | "pause" | "stop", node]
   ["set_playhead_secs", node, secs]   ["set_loop_range", node, mode, start_secs, end_secs]      mode 0 None, 1 Full, 2 RangeSecs
   ["process", frames, n_in_ch, n_out_ch, input_data_or_null, stream_time_secs, stream_status, sha256_of_output]
  data:  {"gen": "fmix32", "seed": S, "count": N, "quant": "f32" | "i16" | "u16"}  or  {"raw_b64": ..., "dtype": "f32" | "i16" | "u16"}
    fmix32 stream (tests/fwapi.py xorshift_uniform): element i (0-based) = f32(h(seed + (i + 1) * 0x9E3779B9) >> 8) * 2^-23 - 1 with
      h(x): x ^= x >> 16; x *= 0x85EBCA6B; x ^= x >> 13; x *= 0xC2B2AE35; x ^= x >> 16  (all mod 2^32)
    the stream is CHANNEL-MAJOR ([channels][frames]); quant i16 = round_half_even(x * 32767) in f32 arithmetic, u16 =
      round_half_even((x + 1) * 32767.5); an INTERLEAVED format stores its transpose ([frames][channels])
  sha256 is over the little-endian f32 bytes of the interleaved output.
"""
import base64
import hashlib

import numpy as np

import fwapi

VERSION = 1
# node kinds the reference implements (crates/firewheel-graph/src/basic_nodes/): dummy 0, beep 1, volume 2, sum 3, sampler 4,
# hard clip 5, mono->stereo 6, stereo->mono 7 — the SPEC kinds (8..14) exist only in this repository
REFERENCE_KINDS = {0, 1, 2, 3, 4, 5, 6, 7}
_DT = {"f32": np.float32, "i16": np.int16, "u16": np.uint16}
_FMT_QUANT = {0: "i16", 1: "u16", 2: "f32", 3: "i16", 4: "u16", 5: "f32"}


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a, dtype=np.float32).tobytes()).hexdigest()


def quantise(x, quant):
    if quant == "i16":
        return np.round(x * 32767).astype(np.int16)
    if quant == "u16":
        return np.round((x + 1) * 32767.5).astype(np.uint16)
    return x


def data_of(rec, fmt=None, channels=1):
    """a data record -> the flat array an engine's new_sample / process call takes"""
    if "raw_b64" in rec:
        return np.frombuffer(base64.b64decode(rec["raw_b64"]), dtype=_DT[rec["dtype"]]).copy()
    assert rec["gen"] == "fmix32"
    x = fwapi.xorshift_uniform(rec["seed"], rec["count"])
    q = quantise(x, rec["quant"])
    if fmt is not None and fmt <= 2:  # interleaved formats store [frames][channels]
        q = q.reshape(channels, -1).T.copy().reshape(-1)
    return q


def _recipe(a, fmt, channels):
    """the generator recipe of array `a` (a sample of format `fmt`), or its raw bytes"""
    flat = np.ascontiguousarray(a).reshape(-1)
    quant = _FMT_QUANT[fmt] if fmt is not None else "f32"
    for seed, n, x in reversed(fwapi._GEN_LOG):
        if n != flat.size:
            continue
        q = quantise(x, quant)
        if q.dtype != flat.dtype:
            continue
        if fmt is not None and fmt <= 2:
            q = q.reshape(channels, -1).T.reshape(-1)
        if np.array_equal(q.view(np.uint8), flat.view(np.uint8)):
            return {"gen": "fmix32", "seed": seed, "count": n, "quant": quant}
    name = {np.dtype(np.float32): "f32", np.dtype(np.int16): "i16", np.dtype(np.uint16): "u16"}[flat.dtype]
    return {"raw_b64": base64.b64encode(flat.tobytes()).decode(), "dtype": name}


class Recorder(fwapi.OracleEngine):
    """an OracleEngine that writes down every primitive call (scenarios wrap it in TaggedOracle like the plain one)"""

    def __init__(self, **kw):
        super().__init__(**kw)
        self.doc = {"version": VERSION, "sample_rate": self.sample_rate, "max_block_frames": self.max_block_frames,
                    "num_graph_inputs": kw.get("num_graph_inputs", 0), "num_graph_outputs": kw.get("num_graph_outputs", 2), "ops": []}
        self._nodes, self._edges, self._samples, self._kinds = {}, {}, 0, set()
        self._outs = []

    def _n(self, node_id):
        if node_id == self.graph_in_node:
            return -1
        if node_id == self.graph_out_node:
            return -2
        return self._nodes[node_id]

    def add_node(self, kind, n_in, n_out, params=()):
        nid = super().add_node(kind, n_in, n_out, params)
        self._nodes[nid] = len(self._nodes)
        self._kinds.add(int(kind))
        self.doc["ops"].append(["add_node", int(kind), int(n_in), int(n_out), [float(np.float32(p)) for p in params]])
        return nid

    def host_node(self, *a, **kw):
        raise NotImplementedError("custom host nodes are code, not data: not exportable")

    def remove_node(self, node):
        self.doc["ops"].append(["remove_node", self._n(node)])
        return super().remove_node(node)

    def connect(self, src, sp, dst, dp, check_for_cycles=False):
        op = ["connect", self._n(src), int(sp), self._n(dst), int(dp), bool(check_for_cycles), 0]
        self.doc["ops"].append(op)
        try:
            e = super().connect(src, sp, dst, dp, check_for_cycles)
        except fwapi.AddEdgeError as ex:
            op[6] = int(ex.args[0])
            raise
        self._edges[e] = len(self._edges)
        return e

    def disconnect(self, src, sp, dst, dp):
        self.doc["ops"].append(["disconnect", self._n(src), int(sp), self._n(dst), int(dp)])
        return super().disconnect(src, sp, dst, dp)

    def disconnect_by_edge_id(self, e):
        self.doc["ops"].append(["disconnect_edge", self._edges[e]])
        return super().disconnect_by_edge_id(e)

    def update(self):
        op = ["update", 0]
        self.doc["ops"].append(op)
        try:
            super().update()
        except fwapi.CompileGraphError as ex:
            op[1] = int(ex.args[0])
            raise

    def new_sample(self, fmt, channels, data):
        a = np.ascontiguousarray(np.asarray(data, dtype=fwapi._FMT_DTYPE[fmt]))
        self.doc["ops"].append(["new_sample", int(fmt), int(channels), int(a.size // channels), _recipe(a, fmt, channels)])
        self._samples += 1
        return super().new_sample(fmt, channels, data)

    def set_param(self, node, param, value, at_block=0):
        self.doc["ops"].append(["set_param", self._n(node), int(param), float(np.float32(value))])
        return super().set_param(node, param, value, at_block)

    def sampler_set_sample(self, node, sample, stop_playback=False, at_block=0):
        self.doc["ops"].append(["set_sample", self._n(node), int(sample), bool(stop_playback)])
        return super().sampler_set_sample(node, sample, stop_playback)

    def sampler_play(self, node, at_block=0):
        self.doc["ops"].append(["play", self._n(node)])
        return super().sampler_play(node)

    def sampler_pause(self, node, at_block=0):
        self.doc["ops"].append(["pause", self._n(node)])
        return super().sampler_pause(node)

    def sampler_stop(self, node, at_block=0):
        self.doc["ops"].append(["stop", self._n(node)])
        return super().sampler_stop(node)

    def sampler_set_playhead_secs(self, node, secs, at_block=0):
        self.doc["ops"].append(["set_playhead_secs", self._n(node), float(secs)])
        return super().sampler_set_playhead_secs(node, secs)

    def sampler_set_loop_range(self, node, mode, start=0.0, end=0.0, at_block=0):
        self.doc["ops"].append(["set_loop_range", self._n(node), int(mode), float(start), float(end)])
        return super().sampler_set_loop_range(node, mode, start, end)

    def process_interleaved(self, frames, n_out_ch=2, inp=None, n_in_ch=0, t=0.0, status=0):
        rec = None
        if inp is not None and n_in_ch > 0:
            rec = _recipe(np.ascontiguousarray(inp, dtype=np.float32), None, 1)
        out = super().process_interleaved(frames, n_out_ch, inp, n_in_ch, t, status)
        self.doc["ops"].append(["process", int(frames), int(n_in_ch), int(n_out_ch), rec, float(t), int(status), sha(out)])
        self._outs.append(np.array(out, dtype=np.float32))
        return out

    def process_blocks_flags(self, k, n_out_ch=2):
        raise NotImplementedError("not a scenario primitive")

    def finish(self, name, result):
        """`result` = what the scenario function returned (the concatenation the golden digests are made of)"""
        self.doc["name"] = name
        self.doc["node_kinds"] = sorted(self._kinds)
        self.doc["reference_kinds_only"] = self._kinds <= REFERENCE_KINDS
        self.doc["sha256"] = sha(result)
        # the digest of record: every process output in order.  (A scenario may return its calls' outputs in another arrangement —
        # then `sha256` differs from `sha256_calls` and a replayer checks the per-call digests and this one.)
        self.doc["sha256_calls"] = sha(np.concatenate(self._outs)) if self._outs else None
        return self.doc


def replay(doc, e, verify=True):
    """run a document on engine `e` (fwapi's primitive surface, created by the caller with the document's header); returns the
    concatenated process outputs after checking every call's digest (verify=False: the caller compares — BeepTest's `sinf` is
    the platform's, device and host libm differ in the last bits)"""
    assert doc["version"] == VERSION
    nodes, edges, samples, outs = [], [], [], []

    def n(i):
        return e.graph_in_node if i == -1 else (e.graph_out_node if i == -2 else nodes[i])

    for op in doc["ops"]:
        k = op[0]
        if k == "add_node":
            nodes.append(e.add_node(op[1], op[2], op[3], op[4]))
        elif k == "remove_node":
            e.remove_node(n(op[1]))
        elif k == "connect":
            try:
                edges.append(e.connect(n(op[1]), op[2], n(op[3]), op[4], op[5]))
                assert op[6] == 0, "connect was expected to fail with %d" % op[6]
            except fwapi.AddEdgeError as ex:
                assert int(ex.args[0]) == op[6], (op, ex)
        elif k == "disconnect":
            e.disconnect(n(op[1]), op[2], n(op[3]), op[4])
        elif k == "disconnect_edge":
            e.disconnect_by_edge_id(edges[op[1]])
        elif k == "update":
            try:
                e.update()
                assert op[1] == 0
            except fwapi.CompileGraphError as ex:
                assert int(ex.args[0]) == op[1], (op, ex)
        elif k == "new_sample":
            samples.append(e.new_sample(op[1], op[2], data_of(op[4], op[1], op[2])))
        elif k == "set_param":
            e.set_param(n(op[1]), op[2], op[3])
        elif k == "set_sample":
            e.sampler_set_sample(n(op[1]), samples[op[2]], op[3])
        elif k == "play":
            e.sampler_play(n(op[1]))
        elif k == "pause":
            e.sampler_pause(n(op[1]))
        elif k == "stop":
            e.sampler_stop(n(op[1]))
        elif k == "set_playhead_secs":
            e.sampler_set_playhead_secs(n(op[1]), op[2])
        elif k == "set_loop_range":
            e.sampler_set_loop_range(n(op[1]), op[2], op[3], op[4])
        elif k == "process":
            inp = data_of(op[4]) if op[4] is not None else None
            out = np.asarray(e.process_interleaved(op[1], op[3], inp, op[2], op[5], op[6]), dtype=np.float32)
            assert not verify or sha(out) == op[7], "%s: output of process call %d differs from the recorded digest" % (doc.get("name"), len(outs))
            outs.append(out)
        else:
            raise ValueError("unknown op %r" % (k,))
    return np.concatenate(outs) if outs else np.zeros(0, np.float32)
