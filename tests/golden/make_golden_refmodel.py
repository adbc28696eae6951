"""Regenerates tests/golden/refmodel_digests.json from the INDEPENDENT numpy model (tests/refmodel.py: a second
restatement of the reference's graph level, written from the .rs files, sharing no code with oracle/fw_oracle.cpp) — for
every parity scenario (the model covers every node kind).  The oracle must reproduce these
digests (tests/test_refmodel_differential.py, CPU tier) and so must the HIP path (tests/test_gpu_parity.py, GPU tier).
Run: python tests/golden/make_golden_refmodel.py"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import refmodel  # noqa: E402
import scenarios  # noqa: E402
import test_scenarios_oracle as t  # noqa: E402

UNSUPPORTED = ()
# scenarios the HIP path matches within a tolerance only (BeepTest: ocml's sinf is not glibc's, H6): the model and the oracle
# agree on them bit for bit, the GPU tier checks them its own way (test_gpu_parity.py)
GPU_TOLERANCE_ONLY = ("mixed_generic", "ref_desk_30", "ref_desk_21_b64")  # (the ref_desk pair: replayed on the GPU from their documents, tests/test_gpu_benched_shapes.py)


def model_cases():
    return sorted(n for n in t.CASES if n not in UNSUPPORTED)


def run_on_model(name):
    saved = t.oracle
    t.oracle = lambda **kw: scenarios.TaggedOracle(refmodel.RefEngine(**kw))  # the scenarios build on whatever `oracle()` returns
    try:
        return t.CASES[name]()
    finally:
        t.oracle = saved


if __name__ == "__main__":
    out = {name: t.digest(run_on_model(name)) for name in model_cases()}
    json.dump(out, open(os.path.join(HERE, "refmodel_digests.json"), "w"), indent=1, sort_keys=True)
    print(json.dumps(out, indent=1))
