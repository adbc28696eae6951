"""print the few fields of a bench.py JSON line that A/B runs compare: stdin = the line, argv[1:] = a label"""
import json
import sys

d = json.loads(sys.stdin.read())
r = d.get("roofline") or {}
print(" ".join(sys.argv[1:]), "value %.4g" % d["value"], "ms/step %.4f" % d["ms_per_step"],
      "kernel %s %.1f us frac %.3f" % (r.get("kernel"), r.get("avg_launch_us", 0.0), r.get("frac", 0.0)),
      "other", {k: round(v, 1) for k, v in (r.get("other_kernels_us_per_step") or {}).items()})
