"""CPU tier: run every parity scenario on the oracle alone and pin the results to committed golden
digests (tests/golden/oracle_digests.json, made by tests/golden/make_golden.py from the oracle).  The
reference cannot run in the build image, so these digests pin the ORACLE against drift; they are not
reference outputs."""
import hashlib
import json
import os

import numpy as np
import pytest

import fwapi
import scenarios

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_digests.json")


def digest(a):
    return hashlib.sha256(np.ascontiguousarray(a, dtype=np.float32).tobytes()).hexdigest()


def oracle(**kw):
    return scenarios.TaggedOracle(fwapi.OracleEngine(**kw))


CASES = {
    "steady_96x32": lambda: scenarios.scenario_voice_bank_steady(oracle(max_block_frames=256), 96, 6),
    "steady_40x4_i16": lambda: scenarios.scenario_voice_bank_steady(oracle(max_block_frames=64), 40, 5, radix=4,
                                                                     fmt=fwapi.INTERLEAVED_I16),
    "steady_9x3_u16": lambda: scenarios.scenario_voice_bank_steady(oracle(max_block_frames=128), 9, 4, radix=3,
                                                                    fmt=fwapi.PLANAR_U16),
    "events_70": lambda: scenarios.scenario_voice_bank_events(oracle(max_block_frames=256), 70),
    "events_33_r2": lambda: scenarios.scenario_voice_bank_events(oracle(max_block_frames=128), 33, radix=2, src_frames=777),
    "mixed_generic": lambda: scenarios.scenario_mixed_generic(oracle(max_block_frames=256)),
    "mixed_generic_nobeep": lambda: scenarios.scenario_mixed_generic(oracle(max_block_frames=256), use_beep=False),
    "graph_inputs": lambda: scenarios.scenario_graph_inputs(oracle(max_block_frames=64, num_graph_inputs=3)),
}


@pytest.mark.parametrize("name", sorted(CASES))
def test_oracle_matches_golden_digest(name):
    out = CASES[name]()
    assert np.all(np.isfinite(out))
    assert np.any(out != 0)
    gold = json.load(open(GOLDEN))
    assert digest(out) == gold[name], "oracle output drifted from the committed digest"


def test_steady_bank_is_loop_periodic():
    # size-independent property: constant gains + full loops => output repeats with the loop period
    e = oracle(max_block_frames=256)
    out = scenarios.scenario_voice_bank_steady(e, 40, 16, src_frames=1024)
    frames = out.reshape(-1, 2)
    assert np.array_equal(frames[:1024], frames[1024:2048])
    assert np.array_equal(frames[:1024], frames[3072:4096])
