#!/bin/bash
# Turns "parity unpinned" into pinned fixtures — ONE command, on any machine with cargo (no GPU, no ROCm needed):
#
#     scripts/pin_parity.sh /path/to/firewheel        # a checkout of BillyDM/firewheel @ 2024-10-16 (the reference)
#
# 1. copies rust/firewheel-gpu into <firewheel>/crates/firewheel-gpu (where its Cargo.toml expects to live: path deps
#    ../firewheel-core, ../firewheel-graph) and adds it to the workspace members if it is not there yet;
# 2. runs tests/reference_digests.rs: every scenario document of tests/golden/scenarios/ whose nodes are all reference nodes is
#    built on the REAL firewheel-graph (AudioGraph::add_node / connect, FirewheelProcessor::process_interleaved,
#    crates/firewheel-graph/src/processor.rs:61-165) and the sha256 of every process call's output is compared with the one the
#    oracle recorded — the test fails at the first difference;
# 3. writes tests/golden/reference_digests.json; commit it.  From then on the CPU test tier REQUIRES it to equal the oracle's
#    digests (tests/test_scenario_json.py::test_reference_digests_equal_the_oracles_when_present), and DESIGN.md section 5's
#    "parity unpinned" can go.
# Regenerating the documents (after a scenario changed): python tests/golden/make_golden.py && python tests/golden/make_scenarios_json.py
set -euo pipefail
REF=${1:?usage: scripts/pin_parity.sh /path/to/firewheel-checkout}
HERE="$(cd "$(dirname "$0")/.." && pwd)"
command -v cargo > /dev/null || { echo "cargo not found: this script is for a machine with a Rust toolchain" >&2; exit 2; }
[ -f "$REF/crates/firewheel-graph/Cargo.toml" ] || { echo "$REF is not a firewheel checkout" >&2; exit 2; }
rm -rf "$REF/crates/firewheel-gpu"
cp -r "$HERE/rust/firewheel-gpu" "$REF/crates/firewheel-gpu"
grep -q '"crates/firewheel-gpu"' "$REF/Cargo.toml" || sed -i 's|"crates/firewheel-cpal",|"crates/firewheel-cpal",\n    "crates/firewheel-gpu",|' "$REF/Cargo.toml"
cd "$REF"
FWGPU_NO_LINK=1 FWGPU_SCENARIOS="$HERE/tests/golden/scenarios" cargo test -p firewheel-gpu --test reference_digests -- --nocapture
echo
echo "wrote $HERE/tests/golden/reference_digests.json:"
python3 - "$HERE/tests/golden/reference_digests.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for k, v in sorted(d.items()):
    print("  %-28s %3d process calls  %s" % (k, len(v["calls"]), v["sha256_calls"][:16]))
PY
echo "now: (cd $HERE && python -m pytest tests/test_scenario_json.py -q) and commit tests/golden/reference_digests.json"
