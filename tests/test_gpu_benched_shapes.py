"""GPU tier (-m gpu): parity at the launch shapes bench.py actually times — the voice-bank plan at
768 and 256 blocks per call, the chain plan across its 64-block launch boundary and at config 3's full size, config 5's
block 1024 at K = 64, config 4 at 65 536 taps with K = 16 and rows in two MFMA row tiles — each against the oracle, bit
for bit; plus the device ends of the round-2 boundary work (ordered mix-bus kernel, headless stream, ReturnSample,
per-node B1 message queues, out-of-range loop guards)."""
import ctypes as C
import os

import numpy as np
import pytest

import fwapi
import scenarios
from fwapi import LOOP_FULL, LOOP_RANGE_SECS, PLANAR_F32, GpuEngine, OracleEngine, bits
from test_gpu_parity import assert_bits_equal, oracle

pytestmark = pytest.mark.gpu
f32 = np.float32


# ------------------------------------------------------------------ voice-bank plan at the benched batch sizes
def _bank_with_traffic(e, n_voices, blocks, radix=32, src_frames=3000):
    """config-2 voices under one call of `blocks` blocks with message traffic spread over it: gain changes at blocks 3,
    ~blocks/2 and blocks-2, a pan change, a pause / resume pair far into the call, a one-shot that ends inside it, a voice
    that never plays, a mute"""
    voices = scenarios.build_voice_bank(e, n_voices, radix=radix, src_frames=src_frames, mono_every=9)
    mid = blocks // 2 + 16
    for v, vc in enumerate(voices):
        if v % 7 != 5:
            e.sampler_set_loop_range(vc["sampler"], LOOP_FULL)     # v%7==5: one-shot, ends after src_frames
        if v % 11 != 4:
            e.sampler_play(vc["sampler"])                           # v%11==4: paused for the whole call
        if v % 5 == 0:
            e.set_param(vc["volume"], 0, 30.0 + v % 50, at_block=3)
        if v % 5 == 1:
            e.set_param(vc["volume"], 0, 95.0 - v % 40, at_block=mid)
        if v % 6 == 2:
            e.set_param(vc["pan"], 0, -0.7 + (v % 10) / 10.0, at_block=mid + 1)
        if v % 13 == 3:
            e.sampler_pause(vc["sampler"], at_block=mid - 9)
            e.sampler_play(vc["sampler"], at_block=blocks - 40)
        if v % 17 == 6:
            e.set_param(vc["sampler"], 0, 0.0, at_block=blocks - 30)   # ramps to 0, then muted
        if v % 19 == 8:
            e.set_param(vc["volume"], 0, 12.0, at_block=blocks - 2)
    return e.process_blocks(blocks)


@pytest.mark.parametrize("max_batch", [768, 256])
def test_voice_bank_plan_at_benched_batch_sizes_with_messages(max_batch):
    # 64 voices x 768 blocks: k_voice_control's 64-blocks-per-step closed-form fill, messages at block ~400, paused voices,
    # k_leaf_sum's (leaf, blocks/4) grid — 0.05 s of oracle time
    blocks = 768
    o = oracle(max_block_frames=256)
    g = GpuEngine(max_block_frames=256, max_batch=max_batch)
    oo = _bank_with_traffic(o, 64, blocks)
    og = _bank_with_traffic(g, 64, blocks)
    assert g.cx.plan_kind() == 1
    assert_bits_equal(oo, og, "64 voices x 768 blocks, max_batch %d" % max_batch)
    # the call after it starts from the same state on both sides
    assert_bits_equal(o.process_blocks(5), g.process_blocks(5), "follow-up call")


def test_config2_whole_benched_call_matches_the_oracle():
    # BASELINE configs[1] exactly as bench.py launches it: 1024 voices, ONE call of 768 blocks of 256 frames (0.7 s of oracle)
    V, blocks, src = 1024, 768, 4096
    o = oracle(max_block_frames=256)
    g = GpuEngine(max_block_frames=256, max_batch=768)
    oo = scenarios.scenario_voice_bank_steady(o, V, blocks, src_frames=src)
    og = scenarios.scenario_voice_bank_steady(g, V, blocks, src_frames=src)
    assert g.cx.plan_kind() == 1
    assert_bits_equal(oo, og, "config 2, 1024 voices x 768 blocks in one call")


def test_config5_shard_block_1024_at_k64_matches_the_oracle():
    # BASELINE configs[4], one GPU's shard, as benched: 8192 voices, block 1024, ONE call of 64 blocks (tree 256 + 8 + 1)
    V, blocks, src = 8192, 64, 3000
    o = oracle(max_block_frames=1024)
    g = GpuEngine(max_block_frames=1024, max_batch=64)
    oo = scenarios.scenario_voice_bank_steady(o, V, blocks, src_frames=src)
    og = scenarios.scenario_voice_bank_steady(g, V, blocks, src_frames=src)
    assert g.cx.plan_kind() == 1
    assert_bits_equal(oo, og, "config 5 shard, 8192 voices x 64 blocks of 1024")


# ------------------------------------------------------------------ chain plan
def test_chain_plan_130_block_call_spans_three_launches():
    # k_chain renders at most CH_FAST_KMAX = 64 blocks per launch: a 130-block call is 64 + 64 + 2, with messages landing in
    # each part and the ChainStart hand-over (delay position, coefficients) between them
    blocks = 130
    o = oracle(max_block_frames=128)
    g = GpuEngine(max_block_frames=128, max_batch=256)

    def run(e):
        voices = scenarios.build_chain_bank(e, 40, radix=32, src_frames=1700, mono_every=6, first_delay_frames=128,
                                            min_delay_frames=128, max_delay_frames=1400)
        for v, vc in enumerate(voices):
            e.sampler_set_loop_range(vc["sampler"], LOOP_FULL)
            if v % 9 != 2:
                e.sampler_play(vc["sampler"])
            if v % 4 == 0:
                e.set_param(vc["volume"], 0, 40.0 + v, at_block=5)
            if v % 4 == 1:
                e.set_param(vc["biquad"], 1, 600.0 + 20 * v, at_block=70)     # second launch
            if v % 4 == 2:
                e.set_param(vc["delay"], 1, 0.6, at_block=63)                  # last block of the first launch
                e.set_param(vc["delay"], 2, 0.8, at_block=64)                  # first block of the second
            if v % 4 == 3:
                e.set_param(vc["volume"], 0, 25.0, at_block=129)               # third launch
            if v % 9 == 2:
                e.sampler_play(vc["sampler"], at_block=100)
        return np.concatenate([e.process_blocks(blocks), e.process_blocks(70)])

    oo, og = run(o), run(g)
    assert g.cx.plan_kind() == 2
    assert_bits_equal(oo, og, "chain plan, 130-block call")


def test_config3_full_size_call_matches_the_oracle():
    # BASELINE configs[2] as benched: 4096 voices (sampler -> biquad -> delay -> gain), block 512, ONE call of 64 blocks on the
    # radix-32 tree (128 + 4 + 1); delays of 10..250 ms like bench.py (1.5 s of oracle time)
    V, blocks = 4096, 64
    kw = dict(radix=32, src_frames=2048, first_delay_frames=480, min_delay_frames=480, max_delay_frames=12000)
    o = oracle(max_block_frames=512)
    g = GpuEngine(max_block_frames=512, max_batch=64)
    oo = scenarios.scenario_chain_steady(o, V, blocks, **kw)
    og = scenarios.scenario_chain_steady(g, V, blocks, **kw)
    assert g.cx.plan_kind() == 2
    assert_bits_equal(oo, og, "config 3, 4096 voices x 64 blocks of 512")
    steady, general = g.cx.plan_chain_stats()
    assert steady + general == 2 * 128                      # one launch: 128 leaf groups x 2 channels
    assert_bits_equal(o.process_blocks(64), g.process_blocks(64), "second call (steady-call loop)")
    steady2, _ = g.cx.plan_chain_stats()
    assert steady2 - steady == 2 * 128                      # ... which every workgroup ran on the steady-call loop


# ------------------------------------------------------------------ FIR bank at the real size
def test_config4_real_size_65536_taps_k16_two_row_tiles_matches_the_oracle():
    # BASELINE configs[3]'s kernel shape: 65 536 taps = 17 split-K segments, K = 16 blocks per launch, 18 stereo voices = 36
    # rows = two 32-row MFMA tiles (the second one padded).  17 blocks: one K = 16 launch + one K = 1 launch.  The oracle
    # evaluates the same segment order with scalar fmaf: 18 voices x 17 blocks x 27 ms = 8 s.
    taps, frames, V, blocks = 65536, 256, 18, 17
    h = scenarios.reverb_ir(77, taps, 2, decay=16384.0)

    def run(e):
        ir = e.new_sample(PLANAR_F32, 2, h)
        m = e.sum(V)
        ss = []
        for v in range(V):
            s = e.sampler(60.0 + v)
            f = e.fir(ir)
            e.connect_stereo(s, f)
            e.connect_stereo(f, m, 2 * v)
            ss.append(s)
        e.connect_stereo(m, e.graph_out_node)
        e.update()
        for v, s in enumerate(ss):
            e.sampler_set_sample(s, e.new_sample(PLANAR_F32, 2, scenarios.voice_source(4400 + v, 1500)))
            if v % 3:
                e.sampler_set_loop_range(s, LOOP_FULL)      # v%3==0: one-shot, the tail keeps convolving zeros
            e.sampler_play(s)
        return e.process_blocks(blocks)

    g = GpuEngine(max_block_frames=frames, max_batch=16)
    og = run(g)
    assert g.cx.plan_kind() == 3 and g.cx.plan_fused_voices() == V   # (round 4: the samplers in front of the FIR nodes are solo voices of the hybrid plan)
    oo = run(OracleEngine(max_block_frames=frames))
    assert_bits_equal(oo, og, "65536-tap FIR bank, K = 16, 36 rows")


# ------------------------------------------------------------------ multi-GPU: the ordered mix-bus kernel
@pytest.mark.parametrize("parts,n", [(2, 4096), (4, 1000), (8, 12345), (64, 260), (3, 3)])
def test_bus_sum_ordered_kernel_is_the_port_ordered_sum(parts, n):
    import torch

    rng = np.random.default_rng(parts * 1000 + n)
    host = [(rng.standard_normal(n) * 10.0 ** rng.integers(-3, 4, n)).astype(f32) for _ in range(parts)]
    want = host[0].copy()
    for p in host[1:]:
        want = (want + p).astype(f32)      # sum.rs:117-131: out = in0; out += in_p, one rounding per port
    g = GpuEngine()
    dev = [torch.from_numpy(np.concatenate([h, np.zeros((-n) % 4, f32)])).cuda() for h in host]   # (16-byte aligned bases)
    out = torch.empty(n + (-n) % 4, dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    g.cx.bus_sum_ordered([d.data_ptr() for d in dev], out.data_ptr(), n)
    g.cx.synchronize()
    assert_bits_equal(want, out.cpu().numpy()[:n], "%d parts" % parts)
    # in place into part 0, as BusReducer does
    g.cx.bus_sum_ordered([d.data_ptr() for d in dev], dev[0].data_ptr(), n)
    g.cx.synchronize()
    assert_bits_equal(want, dev[0].cpu().numpy()[:n], "%d parts, in place" % parts)


# ------------------------------------------------------------------ boundary pieces on the device
def test_headless_stream_on_the_device_matches_oracle_audio_and_flags():
    o = scenarios.TaggedOracle(OracleEngine(max_block_frames=128))
    g = GpuEngine(max_block_frames=128, max_batch=4)
    for e in (o, g):
        voices = scenarios.build_voice_bank(e, 20, radix=8, src_frames=900)
        for vc in voices:
            e.sampler_set_loop_range(vc["sampler"], LOOP_FULL)
            e.sampler_play(vc["sampler"])
    st = g.cx.open_stream(0, 2)
    L = fwapi.oracle_lib()
    ost = L.fwo_stream_new(o.e.c, 48000, 0, 2)
    t, period = 7.0, 128 / 48000.0
    ref = np.zeros(128 * 2, f32)
    for i in range(60):
        t += period * (2.0 if i in (20, 41) else 1.0)          # two late callbacks
        out, status = st.callback(128, t)
        ostatus = L.fwo_stream_callback(ost, ref.ctypes.data_as(C.POINTER(C.c_float)), 128, t, None)
        assert status == ostatus
        assert_bits_equal(ref, out, "callback %d" % i)
    assert st.stats()[1] == 2 and g.cx.proc_info()[2] == 2
    L.fwo_stream_free(ost)


def test_stream_run_on_the_device_renders_what_the_callbacks_render():
    """fwgpu_stream_run (the backend thread's loop inside the library; what bench.py's realtime figure times) against the
    oracle driven callback by callback: the block it leaves in the output buffer is the oracle's block of that callback"""
    o = scenarios.TaggedOracle(OracleEngine(max_block_frames=256))
    g = GpuEngine(max_block_frames=256, max_batch=4)
    for e in (o, g):
        voices = scenarios.build_voice_bank(e, 40, radix=8, src_frames=3000)
        for vc in voices:
            e.sampler_set_loop_range(vc["sampler"], LOOP_FULL)
            e.sampler_play(vc["sampler"])
    st = g.cx.open_stream(0, 2)
    L = fwapi.oracle_lib()
    ost = L.fwo_stream_new(o.e.c, 48000, 0, 2)
    ref = np.zeros(256 * 2, f32)
    period = 256 / 48000.0
    done = 0
    for n in (1, 7, 30):
        out, secs = st.run(256, n, 3.0 + done * period)
        for i in range(n):
            L.fwo_stream_callback(ost, ref.ctypes.data_as(C.POINTER(C.c_float)), 256, 3.0 + (done + i) * period, None)
        done += n
        assert secs > 0.0
        assert_bits_equal(ref, out, "after %d callbacks" % done)
    assert st.stats()[:2] == (38, 0)
    L.fwo_stream_free(ost)


def test_b1_two_nodes_interleaved_messages_match_the_oracle():
    # node A's process() must not swallow node B's queued messages (one queue per node in the reference)
    o = OracleEngine(max_block_frames=64)
    g = GpuEngine(max_block_frames=64)
    x = [fwapi.xorshift_uniform(1, 64), fwapi.xorshift_uniform(2, 64)]
    nodes = {}
    for e in (o, g):
        a, b = e.volume(80.0), e.volume(40.0)
        e.connect_stereo(a, b)
        e.connect_stereo(b, e.graph_out_node)
        e.update()
        nodes[e.backend] = (a, b)

    def step(e, which, msgs=()):
        for node, val in msgs:
            e.set_param(nodes[e.backend][node], 0, val)
        return e.node_process(nodes[e.backend][which], 64, x, 2)

    script = [(0, [(1, 10.0)]), (0, []), (1, []), (0, [(0, 55.0), (1, 70.0)]), (1, []), (1, []), (0, []), (0, []), (1, [])]
    for i, (which, msgs) in enumerate(script):
        yo, mo = step(o, which, msgs)
        yg, mg = step(g, which, msgs)
        assert mo == mg
        assert_bits_equal(yo, yg, "step %d (node %d)" % (i, which))


def test_returned_samples_wait_for_the_device():
    g = GpuEngine(max_block_frames=256, max_batch=64)
    voices = scenarios.build_voice_bank(g, 8, radix=8, src_frames=2000)
    for vc in voices:
        g.sampler_set_loop_range(vc["sampler"], LOOP_FULL)
        g.sampler_play(vc["sampler"])
    g.process_blocks(4)
    assert g.cx.poll_returned_samples() == []
    first = voices[0]["sample"]
    assert not g.cx.sample_retired(first)
    new = g.new_sample(PLANAR_F32, 2, scenarios.voice_source(99, 500))
    g.sampler_set_sample(voices[0]["sampler"], new, at_block=2)
    out = g.process_blocks(8)                    # synchronous: the call has completed when it returns
    assert g.cx.poll_returned_samples() == [(voices[0]["sampler"], first)]
    assert g.cx.sample_retired(first)
    g.cx.destroy_sample(first)
    assert np.all(np.isfinite(out))
    g.process_blocks(4)


def test_loop_ranges_outside_the_sample_play_silence_and_never_fault():
    # ADVICE r1 (outside the parity domain, Q8: the reference panics): RangeSecs past the sample, start > end, a swap to a
    # shorter sample under a RangeSecs loop, a loop shorter than a block that would run past the sample end
    sr = 48000.0
    for force_generic in (False, True):
        g = GpuEngine(max_block_frames=256, max_batch=8, force_generic=force_generic)
        voices = scenarios.build_voice_bank(g, 6, radix=8, src_frames=1000)
        short = g.new_sample(PLANAR_F32, 2, scenarios.voice_source(5, 300))
        for vc in voices:
            g.sampler_set_loop_range(vc["sampler"], LOOP_FULL)
            g.sampler_play(vc["sampler"])
        g.process_blocks(2)
        s = [vc["sampler"] for vc in voices]
        g.sampler_set_loop_range(s[0], LOOP_RANGE_SECS, 100 / sr, 5000 / sr)        # end far past the 1000-frame sample
        g.sampler_set_loop_range(s[1], LOOP_RANGE_SECS, 900 / sr, 100 / sr)         # start > end
        g.sampler_set_loop_range(s[2], LOOP_RANGE_SECS, 200 / sr, 900 / sr)
        g.sampler_set_sample(s[2], short, at_block=1)                                # ... then a 300-frame sample under it
        g.sampler_set_loop_range(s[3], LOOP_RANGE_SECS, 990 / sr, 999 / sr)         # 9-frame loop: second copy runs past the end
        g.sampler_set_loop_range(s[4], LOOP_RANGE_SECS, 1e12, 2e12)                 # saturating conversions
        out = g.process_blocks(8)
        assert np.all(np.isfinite(out)) and np.max(np.abs(out)) < 8.0
        ok = g.process_blocks(4)                                                     # voice 5 keeps playing
        assert np.any(ok)
        # a valid range again: the voice comes back
        g.sampler_set_loop_range(s[1], LOOP_FULL)
        assert np.all(np.isfinite(g.process_blocks(3)))


# ------------------------------------------------------------------ digests generated from the INDEPENDENT numpy model
def _model_cases():
    import sys
    import os

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import make_golden_refmodel as mg

    return [n for n in mg.model_cases() if n not in mg.GPU_TOLERANCE_ONLY]


@pytest.mark.parametrize("name", _model_cases())
def test_hip_path_reproduces_the_independent_models_golden_digests(name):
    # tests/golden/refmodel_digests.json comes from tests/refmodel.py (numpy, written from the .rs files, no code shared with
    # the oracle): the HIP path must hit the same bits without the oracle in between
    import json
    import os

    from test_gpu_parity import run_case
    from test_scenarios_oracle import digest

    gold = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "refmodel_digests.json")))
    _, out_g, g = run_case(name)
    assert digest(out_g) == gold[name], name


def _replayable_docs():
    import glob
    import json
    import os

    d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "scenarios")
    return sorted(os.path.basename(p)[:-5] for p in glob.glob(os.path.join(d, "*.json"))
                  if not p.endswith("index.json") and json.load(open(p))["reference_kinds_only"])


@pytest.mark.parametrize("name", _replayable_docs())
def test_hip_path_replays_the_reference_replayable_documents_call_for_call(name):
    """the language-neutral documents a Rust test runs on the real firewheel-graph (tests/golden/scenarios, rust/firewheel-gpu/tests/
    reference_digests.rs) — the same documents through the C ABI on the device: every process call's sha256 must be the recorded
    one.  When tests/golden/reference_digests.json exists, those digests are the REFERENCE's: this is then GPU == reference, with
    nothing of this repository's making in between."""
    import json
    import os

    import scenario_json

    doc = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "scenarios", name + ".json")))
    g = GpuEngine(sample_rate=doc["sample_rate"], max_block_frames=doc["max_block_frames"], num_graph_inputs=doc["num_graph_inputs"],
                  num_graph_outputs=doc["num_graph_outputs"])
    beep = 1 in doc["node_kinds"]   # BeepTest: `sinf` of the device's libm vs the host's, 2e-6 absolute
    out = scenario_json.replay(doc, g, verify=not beep)
    if not beep:
        assert scenario_json.sha(out) == doc["sha256_calls"]
    else:
        o = OracleEngine(sample_rate=doc["sample_rate"], max_block_frames=doc["max_block_frames"], num_graph_inputs=doc["num_graph_inputs"],
                         num_graph_outputs=doc["num_graph_outputs"])
        ref = scenario_json.replay(doc, o)
        assert out.shape == ref.shape and float(np.max(np.abs(out - ref))) <= 2e-6


def test_lazy_records_calls_without_a_control_kernel_are_bit_exact_and_happen():
    """Round 4 (fwgpu_types.h LazyRec): message-free calls of a plain voice bank whose loops are whole blocks long are rendered without
    a control kernel — the leaf waves compute their records — until something happens: a message, a one-shot running out (the
    horizon the last control kernel reported), a one-block callback, a recompile.  Every call against the oracle, bit for bit; and
    the counters say the path was taken (a host that has SEEN the control kernel's report: every call here waits for its output)."""
    mbf = 64

    def run(e):
        voices = scenarios.build_voice_bank(e, 41, radix=8, src_frames=mbf * 9, mono_every=6, fmt_cycle=list(range(6)))  # every sample format
        for v, vc in enumerate(voices):
            if v % 7 != 3:
                e.sampler_set_loop_range(vc["sampler"], LOOP_FULL)       # v % 7 == 3: one-shots: they end after 9 blocks
            if v % 10 != 9:
                e.sampler_play(vc["sampler"])                            # v % 10 == 9: never started (silent, constant records)
        outs, marks = [], []
        for i, k in enumerate([3, 2, 2, 2, 5, 1, 4, 4, 9, 2, 2, 3, 3]):
            if i == 9:
                e.set_param(voices[4]["volume"], 0, 33.0)               # a glide: control until it has settled, then lazy again
            if i == 11:
                e.sampler_play(voices[9]["sampler"])
            outs.append(np.asarray(e.process_blocks(k)))
            if hasattr(e, "cx"):
                marks.append(e.cx.lazy_stats())
        return np.concatenate(outs), marks

    out_o, _ = run(scenarios.TaggedOracle(OracleEngine(max_block_frames=mbf)))
    g = GpuEngine(max_block_frames=mbf, max_batch=8)
    out_g, marks = run(g)
    assert g.cx.plan_kind() == 1
    assert np.array_equal(bits(out_g), bits(out_o))
    if os.environ.get("FWGPU_LAZY") == "0":
        assert marks[-1][0] == 0
        return
    lazy = [m[0] for m in marks]
    # call 0 carries the play messages, call 1 follows them (glides may continue): control; calls 2 and 3 are lazy; call 4 (5 blocks)
    # would cross the one-shots' end (block 9): the horizon says no -> control; call 5 is ONE block (its batch takes the ordinary
    # path here: no realtime flag on process_blocks... it is lazy too if the host may) ...
    assert lazy[1] == 0 and lazy[3] == 2, marks
    assert lazy[4] == 2, marks                      # the horizon held the control kernel in for the call the one-shots end in
    assert lazy[8] > lazy[4], marks                 # after it: lazy again, a 9-block call as two batches
    assert lazy[9] == lazy[8] and lazy[10] == lazy[9], marks    # the message and the call after it
    assert lazy[-1] >= lazy[10], marks


def test_node_state_after_lazily_rendered_blocks_is_the_control_paths_state():
    """k_lazy_flush wrote loop_start where the control path (tail_end_playhead) leaves playhead == loop_end when the last
    block of a call ends exactly on the loop end.  The two render alike until something READS the playhead: SetLoopRange keeps a
    playhead that lies inside the new range (sampler.rs:293-321, Q7) — loop_end of [0, 768) lies inside [0, 1536), loop_start plays
    from 0.  Loops of 12 blocks, lazy calls that end on the loop end, then the range is widened: every call against the oracle."""
    mbf = 64

    def run(e):
        voices = scenarios.build_voice_bank(e, 19, radix=8, src_frames=mbf * 24, fmt_cycle=list(range(6)))
        for vc in voices:
            e.sampler_set_loop_range(vc["sampler"], fwapi.LOOP_RANGE_SECS, 0.0, 0.016)   # [0, 768) frames = 12 blocks
            e.sampler_play(vc["sampler"])
        outs, marks = [], []
        for i, k in enumerate([6, 6, 12, 12, 3, 9, 12, 2, 5]):
            if i in (4, 7):   # after calls that ended on the loop end (lazily, if the host may): widen / narrow the range
                for vc in voices[::2]:
                    e.sampler_set_loop_range(vc["sampler"], fwapi.LOOP_RANGE_SECS, 0.0, 0.032 if i == 4 else 0.016)
            outs.append(np.asarray(e.process_blocks(k)))
            if hasattr(e, "cx"):
                marks.append(e.cx.lazy_stats())
        return np.concatenate(outs), marks

    out_o, _ = run(scenarios.TaggedOracle(OracleEngine(max_block_frames=mbf)))
    g = GpuEngine(max_block_frames=mbf, max_batch=16)
    out_g, marks = run(g)
    assert g.cx.plan_kind() == 1
    assert np.array_equal(bits(out_g), bits(out_o))
    if os.environ.get("FWGPU_LAZY") != "0":
        assert marks[3][0] > marks[1][0], marks    # the calls before the range change were lazy ones


@pytest.mark.parametrize("mbf,max_batch", [(256, 16), (64, 8), (100, 4), (512, 4), (90, 3)])
def test_resampler_bank_every_register_window_variant_bit_exact(mbf, max_batch):
    """Resampler-pure leaves (k_leaf_rs) over the ratio axis.  Written for round 5's register-window kernel (a lane owning four
    consecutive frames, chains as template variants by (floor(step), floor(2 step), floor(3 step)): scripts/experiments/
    r05_rs_register_window.patch — bit-exact, slower, not kept) and kept for the kernel that stayed: ratios in every such range and on
    its edges, ratios where window reads collide in the LDS banks (0.8, 1.0, 1.15, 4/3, 1.6), the last ratio a 256-frame piece's
    window fits (1.93) and ratios the general kernel takes (>= 2), loops that wrap inside a window, one-shots that end in it, mono
    sources; block lengths that are no multiple of 4 frames per lane (90), shorter than a piece (64, 100) or longer (512)."""
    ratios = [0.2, 1.0 / 3.0, 0.3334, 0.49999, 0.5, 0.61, 2.0 / 3.0, 0.6667, 0.8, 0.91875, 0.99999, 1.0, 1.00001, 1.15, 1.3333, 4.0 / 3.0,
              1.41, 1.5, 1.50001, 1.6, 5.0 / 3.0, 1.6667, 1.9, 1.93, 1.99999, 2.0, 2.5, 0.75, 1.25, 1.088, 0.0371, 1.75]

    def run(e):
        ends = []
        rng = np.random.default_rng(77)
        for v, ratio in enumerate(ratios):
            ch = 1 if v % 7 == 3 else 2
            frames = 1400 + 37 * v
            smp = e.new_sample(PLANAR_F32, ch, scenarios.voice_source(9300 + v, frames, ch))
            src = e.resampler(smp, ratio, loop=(v % 3 != 1), playing=True, n_out=2)
            vol = e.volume(float(rng.uniform(20, 110)))
            pan = e.pan(float(rng.uniform(-1, 1)))
            e.connect_stereo(src, vol)
            e.connect_stereo(vol, pan)
            ends.append(pan)
        mixers = []
        for i in range(0, len(ends), 16):
            m = e.sum(16)
            for p, n in enumerate(ends[i:i + 16]):
                e.connect_stereo(n, m, 2 * p)
            mixers.append(m)
        top = e.sum(len(mixers))
        for p, m in enumerate(mixers):
            e.connect_stereo(m, top, 2 * p)
        e.connect_stereo(top, e.graph_out_node)
        e.update()
        return np.concatenate([np.asarray(e.process_blocks(k)) for k in (3, max_batch, 2 * max_batch + 1, 5)])

    out_o = run(scenarios.TaggedOracle(OracleEngine(max_block_frames=mbf)))
    g = GpuEngine(max_block_frames=mbf, max_batch=max_batch)
    out_g = run(g)
    assert g.cx.plan_kind() == 1
    assert np.array_equal(bits(out_g), bits(out_o))


@pytest.mark.parametrize("shape,mbf,max_batch", [("bank", 64, 8), ("bank", 256, 1), ("chain", 128, 4), ("hybrid", 64, 8)])
def test_one_output_samplers_behind_the_mono_to_stereo_adapter_are_voices_of_the_fused_plans(shape, mbf, max_batch):
    """VERDICT r4 missing #4: `sampler(0 -> 1) -> MonoToStereoNode` (mono_to_stereo.rs:33-50, the reference's own adapter) in front of a
    gain chain fell off every fused plan.  Round 5: such a voice is a voice of the bank / chain / hybrid plans whose every block is
    VB_MONO — channel 0 of whatever sample it plays: mono and stereo samples, all six formats, one-shots that end, pauses, a sample
    swap to another format, gain glides; stereo samplers beside them under the same mixers.  Every call against the oracle."""
    import fwapi as fw

    def run(e):
        rng = np.random.default_rng(5)
        ends, voices = [], []
        fmts = [fw.PLANAR_F32, fw.INTERLEAVED_I16, fw.PLANAR_I16, fw.INTERLEAVED_F32, fw.PLANAR_U16, fw.INTERLEAVED_U16]
        samples = []
        for v in range(22):
            ch = 1 if v % 4 == 1 else 2
            fmt = fmts[v % 6]
            data = scenarios.voice_source(9900 + v, 1100 + 13 * v, ch)
            if fmt in (fw.PLANAR_F32, fw.INTERLEAVED_F32):
                raw = data if fmt == fw.PLANAR_F32 else data.T.copy()
            elif fmt in (fw.PLANAR_I16, fw.INTERLEAVED_I16):
                q = np.round(data * 32767).astype(np.int16)
                raw = q if fmt == fw.PLANAR_I16 else q.T.copy()
            else:
                q = (np.round(data * 32767).astype(np.int32) + 32768).astype(np.uint16)
                raw = q if fmt == fw.PLANAR_U16 else q.T.copy()
            samples.append(e.new_sample(fmt, ch, raw))
            mono_voice = v % 3 != 2
            s = e.sampler(float(rng.uniform(40, 100)), n_out=1 if mono_voice else 2)
            cur = s
            if mono_voice:
                m2s = e.add_node(fw.MONO_TO_STEREO, 1, 2)
                e.connect(s, 0, m2s, 0)
                cur = m2s
            if shape == "chain" and v % 5 == 2 and not mono_voice:   # (stereo voices with a filter: the plan is the chain plan, the adapter voices its dry ones)
                bq = e.biquad(0, 900.0 + 50 * v)
                e.connect_stereo(cur, bq)
                cur = bq
            vol = e.volume(float(rng.uniform(20, 110)))
            e.connect_stereo(cur, vol)
            pan = e.pan(float(rng.uniform(-1, 1)))
            e.connect_stereo(vol, pan)
            ends.append(pan)
            voices.append((s, vol))
        mixers = []
        for i in range(0, len(ends), 8):
            m = e.sum(8)
            for p, n in enumerate(ends[i:i + 8]):
                e.connect_stereo(n, m, 2 * p)
            mixers.append(m)
        top = e.sum(len(mixers) + (1 if shape == "hybrid" else 0))
        for p, m in enumerate(mixers):
            e.connect_stereo(m, top, 2 * p)
        if shape == "hybrid":   # a send off the first mixer through a delay: no fused shape as a whole
            dl = e.delay(0.01, 0.3, 0.5)
            e.connect_stereo(mixers[0], dl)
            e.connect_stereo(dl, top, 2 * len(mixers))
        e.connect_stereo(top, e.graph_out_node)
        e.update()
        for v, (s, vol) in enumerate(voices):
            e.sampler_set_sample(s, samples[v])
            if v % 5 != 4:
                e.sampler_set_loop_range(s, LOOP_FULL)      # v % 5 == 4: one-shots, they end inside the run
            if v % 7 != 6:
                e.sampler_play(s)
        outs = [np.asarray(e.process_blocks(3))]
        e.set_param(voices[0][1], 0, 15.0)                   # a glide behind an adapter voice
        e.sampler_pause(voices[3][0])
        outs.append(np.asarray(e.process_blocks(max_batch + 2)))
        e.sampler_set_sample(voices[1][0], samples[6])       # an adapter voice takes another voice's sample (another format)
        e.sampler_play(voices[3][0])
        e.sampler_play(voices[6][0])
        outs.append(np.asarray(e.process_blocks(2 * max_batch + 1)))
        outs.append(np.asarray(e.process_blocks(9)))
        return np.concatenate(outs)

    out_o = run(scenarios.TaggedOracle(OracleEngine(max_block_frames=mbf)))
    g = GpuEngine(max_block_frames=mbf, max_batch=max_batch)
    out_g = run(g)
    assert g.cx.plan_kind() == {"bank": 1, "chain": 2, "hybrid": 3}[shape]
    if shape != "hybrid":
        assert g.cx.plan_fused_voices() == 22
    assert np.array_equal(bits(out_g), bits(out_o))


def test_process_interleaved_begin_end_pipelined_calls_equal_the_oracle():
    """the host-buffer call split in two (fwgpu_process_interleaved_begin / _end): call n + 1 is begun before call n is ended — its
    rendering overlaps call n's copy back — with messages, a sample swap and a graph edit between the calls; every call's frames
    against the oracle's synchronous process_interleaved, bit for bit; graphs with inputs go through the same pair."""
    mbf = 128

    def script(e, i, voices):
        if i == 2:
            e.set_param(voices[3]["volume"], 0, 20.0)
        if i == 4:
            e.sampler_pause(voices[5]["sampler"])
        if i == 5:
            e.remove_node(voices[7]["pan"])
            e.update()
        if i == 7:
            e.sampler_play(voices[5]["sampler"])

    sizes = [3, 8, 1, 5, 17, 2, 2, 9, 4]

    def build(e):
        voices = scenarios.build_voice_bank(e, 29, radix=8, src_frames=mbf * 11 + 17, mono_every=6, fmt_cycle=list(range(6)))
        for v, vc in enumerate(voices):
            if v % 6 != 3:
                e.sampler_set_loop_range(vc["sampler"], LOOP_FULL)
            e.sampler_play(vc["sampler"])
        return voices

    o = scenarios.TaggedOracle(OracleEngine(max_block_frames=mbf))
    vo = build(o)
    want = []
    for i, k in enumerate(sizes):
        script(o, i, vo)
        want.append(np.asarray(o.process_interleaved(k * mbf)))
    g = GpuEngine(max_block_frames=mbf, max_batch=8)
    vg = build(g)
    got, prev = [], None
    for i, k in enumerate(sizes):
        script(g, i, vg)
        t = g.cx.process_interleaved_begin(None, 0, 2, k * mbf)
        if prev is not None:
            got.append(g.cx.process_interleaved_end(prev))
        prev = t
    got.append(g.cx.process_interleaved_end(prev))
    for i in range(len(sizes)):
        assert np.array_equal(bits(got[i]), bits(want[i])), "call %d" % i


@pytest.mark.parametrize("n_in,mbf,max_batch", [(2, 64, 8), (4, 128, 3), (3, 64, 64)])
def test_process_blocks_device_io_an_effects_rack_on_device_resident_stream_inputs(n_in, mbf, max_batch):
    """fwgpu_process_blocks_device_io (the device-resident call refused graphs with inputs): graph inputs read from
    DEVICE memory, whole blocks, asynchronously — an effects rack (width -> biquad -> delay -> volume on the first pair, a mix with the
    other inputs, a glide in between), calls of 1, max_batch and 2 max_batch + 3 blocks, against the oracle's process_interleaved."""
    import torch

    def build(e):
        gin = e.graph_in_node
        w = e.width(1.4)
        if n_in >= 2:
            e.connect_stereo(gin, w, 0, 0)
        else:
            m = e.add_node(fwapi.MONO_TO_STEREO, 1, 2)
            e.connect(gin, 0, m, 0)
            e.connect_stereo(m, w)
        bq = e.biquad(0, 2500.0, 0.9)
        e.connect_stereo(w, bq)
        dl = e.delay(0.004, 0.4, 0.5)
        e.connect_stereo(bq, dl)
        vol = e.volume(70.0)
        e.connect_stereo(dl, vol)
        mix = e.sum(2)
        e.connect_stereo(vol, mix, 0)
        if n_in >= 4:
            e.connect_stereo(gin, mix, 2, 2)
        elif n_in == 3:
            m = e.add_node(fwapi.MONO_TO_STEREO, 1, 2)
            e.connect(gin, 2, m, 0)
            e.connect_stereo(m, mix, 2)
        e.connect_stereo(mix, e.graph_out_node)
        e.update()
        return vol

    o = OracleEngine(max_block_frames=mbf, num_graph_inputs=n_in)
    g = GpuEngine(max_block_frames=mbf, num_graph_inputs=n_in, max_batch=max_batch)
    vo, vg = build(o), build(g)
    for i, k in enumerate([1, max_batch, 2 * max_batch + 3, 2]):
        if i == 2:
            o.set_param(vo, 0, 25.0)
            g.set_param(vg, 0, 25.0)
        frames = k * mbf
        inp = fwapi.xorshift_uniform(4100 + i, frames * n_in) if i != 3 else np.zeros(frames * n_in, dtype=np.float32)
        want = np.asarray(o.process_interleaved(frames, 2, inp=inp, n_in_ch=n_in))
        d_in = torch.from_numpy(inp.copy()).cuda()
        d_out = torch.full((frames * 2,), float("nan"), dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        g.cx.process_blocks_device_io(k, d_in.data_ptr(), n_in, d_out.data_ptr(), 2)
        g.cx.synchronize()
        assert np.array_equal(bits(d_out.cpu().numpy()), bits(want)), "call %d" % i


@pytest.mark.parametrize("fuse", ["1", "0"])
def test_level_executor_with_and_without_vertical_fusion_equals_the_oracle(fuse, monkeypatch):
    """the level executor's vertical fusion (k_generic.hip.h fz_links) on and off: a wide bank forced onto the levels — resting
    sampler -> volume -> pan [-> width -> clip] chains (links), a muted voice (a mute is no link), a glide that thaws one node of a chain
    in the middle of the run (its blocks go the slow way, its neighbours' stay fused), paused and never-started voices (silent heads: not
    fused), one-shots that end — 16-block batches so that the wide levels take 8 blocks per wave, every call against the oracle."""
    monkeypatch.setenv("FWGPU_LEVEL_FUSE", fuse)
    mbf = 64

    def run(e):
        def fx(e, v, rng):
            return [e.width(float(rng.uniform(0.5, 1.5))), e.hard_clip(-3.0)] if v % 3 == 0 else []
        voices = scenarios.build_voice_bank(e, 1100, radix=32, src_frames=mbf * 40 + 7, voice_fx=fx)
        for v, vc in enumerate(voices):
            if v % 6 != 5:
                e.sampler_set_loop_range(vc["sampler"], LOOP_FULL)
            if v % 11 != 10:
                e.sampler_play(vc["sampler"])
        e.set_param(voices[8]["volume"], 0, 0.0)        # a mute
        outs = [np.asarray(e.process_blocks(16))]
        outs.append(np.asarray(e.process_blocks(16)))
        e.set_param(voices[20]["pan"], 0, -0.4)         # a glide: the node thaws, then rests again
        e.sampler_pause(voices[33]["sampler"])
        outs.append(np.asarray(e.process_blocks(16)))
        outs.append(np.asarray(e.process_blocks(16)))
        e.sampler_play(voices[33]["sampler"])
        outs.append(np.asarray(e.process_blocks(7)))
        return np.concatenate(outs)

    out_o = run(scenarios.TaggedOracle(OracleEngine(max_block_frames=mbf)))
    g = GpuEngine(max_block_frames=mbf, max_batch=16, force_generic=True)
    out_g = run(g)
    assert g.cx.plan_kind() == 0
    assert np.array_equal(bits(out_g), bits(out_o))


def rs_quiet_run(e, mbf):
    """a bank of resampler voices (k_leaf_rs) with long message-free stretches: looping and one-shot sources (the one-shots run out: the
    horizon), ratios on both sides of 1 up to the last one a piece's window fits, mono sources, paused voices — one leaf of nothing but
    paused voices (clear_all_outputs) — and in between ratio changes, seeks, a pause / resume and gain glides inside control calls"""
    ratios = [1.0, 44100.0 / 48000.0, 1.5, 0.37, 1.93, 0.999, 1.0 / 3.0, 1.088, 0.75, 1.25]
    rng = np.random.default_rng(99)
    voices, ends = [], []
    for v in range(40):
        ch = 1 if v % 5 == 2 else 2
        smp = e.new_sample(PLANAR_F32, ch, scenarios.voice_source(8800 + v, 1100 + 13 * v, ch))
        paused = v % 7 == 3 or 32 <= v < 40          # (voices 32..39: a whole leaf silent)
        src = e.resampler(smp, ratios[v % len(ratios)], loop=(v % 4 != 1), playing=not paused, n_out=2)
        vol = e.volume(float(rng.uniform(20, 110)))
        pan = e.pan(float(rng.uniform(-1, 1)))
        e.connect_stereo(src, vol)
        e.connect_stereo(vol, pan)
        voices.append(dict(src=src, volume=vol, pan=pan))
        ends.append(pan)
    mixers = []
    for i in range(0, len(ends), 8):
        m = e.sum(8)
        for p, n in enumerate(ends[i:i + 8]):
            e.connect_stereo(n, m, 2 * p)
        mixers.append(m)
    top = e.sum(len(mixers))
    for p, m in enumerate(mixers):
        e.connect_stereo(m, top, 2 * p)
    e.connect_stereo(top, e.graph_out_node)
    e.update()
    outs, marks = [], []
    for c, k in enumerate([3, 2, 2, 3, 4, 1, 4, 4, 9, 2, 2, 3, 3, 5] + [8] * 8 + [3, 4]):
        if c == 8:
            for v, vc in enumerate(voices[:32]):
                if v % 6 == 0:
                    e.set_param(vc["src"], 1, [0.5, 1.25, 1.8][v % 3], at_block=1)     # ratio
                if v % 6 == 4:
                    e.set_param(vc["src"], 4, float(v * 7 % 900), at_block=2)          # seek
                if v % 9 == 5:
                    e.set_param(vc["src"], 3, 0.0, at_block=0)                         # pause ...
                    e.set_param(vc["src"], 3, 1.0, at_block=3)                         # ... and resume
                if v % 4 == 0:
                    e.set_param(vc["volume"], 0, 35.0 + v, at_block=2)
        if c == 12:
            e.set_param(voices[33]["src"], 3, 1.0, at_block=1)                         # one voice of the silent leaf starts
            e.set_param(voices[3]["src"], 3, 1.0, at_block=0)
        outs.append(np.asarray(e.process_blocks(k)))
        if hasattr(e, "cx"):
            marks.append(e.cx.lazy_stats())
    return np.concatenate(outs), marks


@pytest.mark.parametrize("mbf,max_batch", [(256, 8), (512, 4), (64, 16), (100, 3)])
def test_resampler_bank_calls_without_a_control_kernel_are_bit_exact_and_happen(mbf, max_batch):
    """Round 6: lazy records for resampler plans — a message-free call launches k_leaf_rs (+ the empty work-list
    kernel) alone: block j's 32.32 position is (pos + j * frames * step) mod (len << 32) from the voice's LazyRec, the rest the
    template the control kernel left beside it; k_lazy_flush writes the positions back.  Every call against the oracle, bit for bit."""
    ro, _ = rs_quiet_run(scenarios.TaggedOracle(OracleEngine(max_block_frames=mbf)), mbf)
    g = GpuEngine(max_block_frames=mbf, max_batch=max_batch)
    rg, marks = rs_quiet_run(g, mbf)
    assert g.cx.plan_kind() == 1
    assert np.array_equal(bits(rg), bits(ro))
    assert np.any(ro != 0)
    lazy = [m[0] for m in marks]
    if os.environ.get("FWGPU_LAZY") == "0":
        assert lazy[-1] == 0
        return
    assert lazy[-1] > lazy[13] >= lazy[9], marks   # ... and the quiet calls at the end run without a control kernel
    assert lazy[9] == lazy[8], marks            # the messages of call 8 and the call after it
