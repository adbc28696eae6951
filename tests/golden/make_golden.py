"""Regenerates tests/golden/oracle_digests.json from the ORACLE (not from the reference: the Rust
reference cannot be built in this image — no cargo/rustc).  Run: python tests/golden/make_golden.py"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import test_scenarios_oracle as t  # noqa: E402

out = {name: t.digest(fn()) for name, fn in sorted(t.CASES.items())}
json.dump(out, open(os.path.join(HERE, "oracle_digests.json"), "w"), indent=1, sort_keys=True)
print(json.dumps(out, indent=1))
