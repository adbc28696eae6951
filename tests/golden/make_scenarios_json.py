"""Exports every parity scenario of tests/test_scenarios_oracle.py::CASES as a language-neutral document
tests/golden/scenarios/<name>.json (format: tests/scenario_json.py) + index.json.  The documents are recorded from the ORACLE
running the scenario (the engine calls it receives, in order, and the sha256 of every process call's output) — they pin
nothing by themselves; they are what a machine with cargo replays on the real firewheel-graph (scripts/pin_parity.sh).
Run: python tests/golden/make_scenarios_json.py"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import scenario_json  # noqa: E402
import scenarios  # noqa: E402
import test_scenarios_oracle as t  # noqa: E402

OUT = os.path.join(HERE, "scenarios")
os.makedirs(OUT, exist_ok=True)
last = []


def recording_oracle(**kw):
    r = scenario_json.Recorder(**kw)
    last.append(r)
    return scenarios.TaggedOracle(r)


t.oracle = recording_oracle
index = {}
for name, fn in sorted(t.CASES.items()):
    del last[:]
    result = fn()
    assert len(last) == 1, "a scenario builds exactly one engine"
    doc = last[0].finish(name, result)
    path = os.path.join(OUT, name + ".json")
    with open(path, "w") as f:
        json.dump(doc, f, separators=(",", ":"))
        f.write("\n")
    index[name] = {"sha256": doc["sha256"], "sha256_calls": doc["sha256_calls"], "reference_kinds_only": doc["reference_kinds_only"],
                   "node_kinds": doc["node_kinds"], "ops": len(doc["ops"]), "bytes": os.path.getsize(path)}
    print("%-32s ops %6d  %8d bytes  %s" % (name, len(doc["ops"]), index[name]["bytes"], "REFERENCE-REPLAYABLE" if doc["reference_kinds_only"] else ""))
json.dump(index, open(os.path.join(OUT, "index.json"), "w"), indent=1, sort_keys=True)
print("total bytes", sum(v["bytes"] for v in index.values()), "reference-replayable:", sorted(k for k, v in index.items() if v["reference_kinds_only"]))
