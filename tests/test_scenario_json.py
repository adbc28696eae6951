"""CPU tier: the language-neutral scenario documents (tests/golden/scenarios/*.json, format in tests/scenario_json.py).

 * every document replays on a PLAIN OracleEngine — through the generic interpreter, without tests/scenarios.py — to the digest of
   every process call recorded in it, and to the committed golden digest of the scenario (tests/golden/oracle_digests.json);
 * the set of documents is the set of scenarios;
 * when tests/golden/reference_digests.json exists — written by `cargo test` of rust/firewheel-gpu/tests/reference_digests.rs on a
   machine with a Rust toolchain (scripts/pin_parity.sh), from the REAL firewheel-graph — every digest in it must equal the
   oracle's.  That file is what turns "parity unpinned" into pinned; without it this test says so and passes."""
import glob
import json
import os

import numpy as np
import pytest

import fwapi
import scenario_json
import test_scenarios_oracle as t

HERE = os.path.dirname(os.path.abspath(__file__))
DOCS = os.path.join(HERE, "golden", "scenarios")
REFERENCE = os.path.join(HERE, "golden", "reference_digests.json")
NAMES = sorted(os.path.basename(p)[:-5] for p in glob.glob(os.path.join(DOCS, "*.json")) if not p.endswith("index.json"))


def load(name):
    return json.load(open(os.path.join(DOCS, name + ".json")))


def test_one_document_per_scenario():
    assert NAMES == sorted(t.CASES), "run python tests/golden/make_scenarios_json.py"
    index = json.load(open(os.path.join(DOCS, "index.json")))
    assert sorted(index) == NAMES
    replayable = [n for n in NAMES if index[n]["reference_kinds_only"]]
    assert len(replayable) >= 6, replayable   # what the Rust reference can be run on: keep some


@pytest.mark.parametrize("name", NAMES)
def test_document_replays_on_the_plain_oracle_to_its_recorded_and_golden_digests(name):
    doc = load(name)
    e = fwapi.OracleEngine(sample_rate=doc["sample_rate"], max_block_frames=doc["max_block_frames"],
                           num_graph_inputs=doc["num_graph_inputs"], num_graph_outputs=doc["num_graph_outputs"])
    out = scenario_json.replay(doc, e)   # (asserts every process call's digest)
    assert np.any(out != 0)
    assert scenario_json.sha(out) == doc["sha256_calls"]
    gold = json.load(open(t.GOLDEN))
    assert doc["sha256"] == gold[name], "the document was recorded from another oracle than the golden digests: regenerate both"
    if doc["sha256"] != doc["sha256_calls"]:   # the scenario returns its calls' outputs in another arrangement: same samples
        assert sorted(np.asarray(t.CASES[name](), dtype=np.float32).tolist()) == sorted(out.tolist())


def test_documents_only_use_the_documented_generator_or_raw_bytes():
    for name in NAMES:
        for op in load(name)["ops"]:
            rec = op[4] if op[0] in ("new_sample", "process") else None
            if isinstance(rec, dict):
                assert set(rec) in ({"gen", "seed", "count", "quant"}, {"raw_b64", "dtype"}), (name, rec.keys())


def test_reference_digests_equal_the_oracles_when_present():
    """the pin.  reference_digests.json: {scenario: {"calls": [sha256 per process call], "sha256_calls": ...}} from the Rust reference"""
    if not os.path.exists(REFERENCE):
        pytest.skip("no tests/golden/reference_digests.json: the Rust reference has not been run on the scenarios yet "
                    "(scripts/pin_parity.sh on a machine with cargo) — parity stays 'unpinned'")
    ref = json.load(open(REFERENCE))
    assert ref, "empty reference_digests.json"
    for name, ent in sorted(ref.items()):
        doc = load(name)
        assert doc["reference_kinds_only"], name
        mine = [op[7] for op in doc["ops"] if op[0] == "process"]
        assert ent["calls"] == mine, "%s: the reference's output differs from the oracle's at process call %d" % (
            name, next(i for i, (a, b) in enumerate(zip(ent["calls"], mine)) if a != b) if len(ent["calls"]) == len(mine) else -1)
        assert ent["sha256_calls"] == doc["sha256_calls"]


def test_the_rust_replayer_handles_everything_the_replayable_documents_use():
    """scripts/pin_parity.sh stays a ONE-command job: the first machine with cargo must not find that a document
    grew an operation, a node kind or a sample format rust/firewheel-gpu/tests/reference_digests.rs does not replay.  No Rust
    toolchain here, so the replayer's dispatch is read from its source: every `ops` verb of every reference-replayable document has a
    match arm, every node kind a constructor arm, set_param only goes to the three reference nodes that have a parameter, and the
    replayable set is what DESIGN.md says it is (9 documents)."""
    import re

    ROOT, SCEN = fwapi.ROOT, os.path.join(fwapi.ROOT, "tests", "golden", "scenarios")
    src = open(os.path.join(ROOT, "rust", "firewheel-gpu", "tests", "reference_digests.rs")).read()
    arms = set(re.findall(r'"([a-z_]+)"\s*(?:\||=>|\))', src))
    kind_arms = set(int(k) for k in re.findall(r"^\s*(\d+) => cx\.graph\.add_node", src, flags=re.M))
    docs = [json.load(open(p)) for p in sorted(glob.glob(os.path.join(SCEN, "*.json"))) if not p.endswith("index.json")]
    replayable = [d for d in docs if d["reference_kinds_only"]]
    assert len(replayable) == 9, [d["name"] for d in replayable]
    for d in replayable:
        kinds_of = []
        for op in d["ops"]:
            assert op[0] in arms, (d["name"], op[0])
            if op[0] == "add_node":
                assert op[1] in kind_arms, (d["name"], "node kind", op[1])
                kinds_of.append(op[1])
            if op[0] == "set_param":
                assert op[2] == 0 and op[1] >= 0 and kinds_of[op[1]] in (1, 2, 4), (d["name"], op)   # beep / volume / sampler, param 0
        assert set(d["node_kinds"]) <= kind_arms
    # and the crate the script copies into the reference's workspace names its path dependencies as the script lays them out
    cargo = open(os.path.join(ROOT, "rust", "firewheel-gpu", "Cargo.toml")).read()
    assert "../firewheel-core" in cargo and "../firewheel-graph" in cargo
    script = open(os.path.join(ROOT, "scripts", "pin_parity.sh")).read()
    assert "crates/firewheel-gpu" in script and "reference_digests" in script
