#!/bin/bash
# The one command for the day an 8-GPU MI355X node is available.  Nothing here has run on more than one GPU: the
# builder's pool hands out single-GPU boxes; what HAS run is the same code as N processes on one device (bench.py --share-device,
# tests/test_bus_exchange.py: 8 processes x 1 000 skewed steps, bit-exact) and the launcher at world size 8 on the host-only
# harness (tests/test_bench_launch.py).
#
#   scripts/r04_multi_gpu.sh [N...]          default: 2 4 8
#
# For every N: the headline workload (configs[1] per GPU, weak scaling) under the three mix-bus reductions —
#   exchange   libfwgpu's one-shot exchange over peer-mapped slots (hipIpc + xGMI stores; rank-ordered: bit-exact)   [default]
#   ordered    RCCL all-gather + rank-ordered sum kernel (bit-exact)
#   allreduce  RCCL all-reduce (north_star's named path; re-associates the f32 sum for N > 2: tolerance, not bits)
#   allreduce_abi / ordered_abi (round 6)  the same two through libfwgpu's OWN C ABI (fwgpu_bus_allreduce_rccl / _allgather_ordered:
#              librccl dlopen'ed by the library, communicator from a unique id) — what a Rust / C host bound to include/fwgpu.h gets
# — each with its parity_check against the oracle's WHOLE graph, rccl_ranks_seen, the longest device-side wait per peer, and
# BASELINE configs[4] (8 192 voices per GPU, block 1024, the reduction after every step) in other_configs.  One line per
# (N, mode) in gpurun_out/r04/multi_gpu_N<N>.json, a table on stdout, scaling efficiency against N = 1 at the end.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r04
NS=${@:-2 4 8}
export HSA_ENABLE_IPC_MODE_LEGACY=0
python bench.py --gpus 1 --no-other-configs --contexts 3 > gpurun_out/r04/multi_gpu_N1.json 2> gpurun_out/r04/multi_gpu_N1.err
for N in $NS; do
  python bench.py --gpus $N > gpurun_out/r04/multi_gpu_N$N.json 2> gpurun_out/r04/multi_gpu_N$N.err || echo "N=$N failed: see gpurun_out/r04/multi_gpu_N$N.err"
done
python - $NS <<'PY'
import json, sys
def load(n):
    # (round 6: stdout carries a compact line; the whole record — every mode's parity object — is the side file it names)
    try:
        line = json.loads(open("gpurun_out/r04/multi_gpu_N%s.json" % n).read().strip().splitlines()[-1])
        try:
            return json.loads(open(line["full"]).read())
        except Exception:
            return line
    except Exception as ex:
        return {"error": repr(ex)}
one = load(1)
base = one.get("value")
print("N = 1: %.4g voice-samples/s, %.4f ms/step" % (base or 0, one.get("ms_per_step") or 0))
for n in sys.argv[1:]:
    d = load(n)
    if "error" in d and "value" not in d:
        print("N = %s: %s" % (n, d["error"])); continue
    rows = [("exchange" if d["config"]["bus_reduce"] == "exchange" else d["config"]["bus_reduce"], d)]
    rows += sorted((d.get("bus_reduce_modes") or {}).items())
    for mode, e in rows:
        pc = e.get("parity_check") or {}
        v = e.get("value")
        print("N = %s %-9s %.4g vs/s  %.4f ms/step  eff %.3f  parity: bit_exact=%s within_tol=%s expected=%s  rccl_ranks_seen=%s  fallback=%s  shared_device=%s" % (
            n, mode, v or 0, e.get("ms_per_step") or 0, (v / (int(n) * base)) if (v and base) else float("nan"), pc.get("bit_exact"),
            pc.get("within_tolerance"), pc.get("expected"), d.get("rccl_ranks_seen"), (e.get("config") or e).get("bus_reduce_fallback"),
            d.get("virtual_ranks_on_one_device", False)))
    c5 = (d.get("other_configs") or {}).get("cfg5") or {}
    if c5:
        print("N = %s configs[4] (8192 voices/GPU, block 1024, exchange every step): %.4g vs/s, parity bit_exact=%s" % (
            n, c5.get("value") or 0, (c5.get("parity_check") or {}).get("bit_exact")))
    waits = (d.get("config") or {}).get("bus_exchange_max_wait_us")
    if waits:
        print("N = %s longest device-side wait per peer (us): %s" % (n, waits))
PY
