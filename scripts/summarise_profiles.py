"""Turns the raw rocprofv3 output of scripts/collect_profiles.sh into the two small files per workload that are
committed under profiles/: the kernel-stats table and the per-launch HBM traffic (FETCH_SIZE + WRITE_SIZE from
separate --pmc passes; FETCH_SIZE doubled for gfx950's 16-B/lane streaming reads as MI355X_MICROARCH.md §HBM
prescribes, WRITE_SIZE as reported)."""
import collections
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

tag, cfg, steps, raw, outdir = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4], sys.argv[5]
suffix = sys.argv[6] if len(sys.argv) > 6 else ""
extra = sys.argv[7] if len(sys.argv) > 7 else ""
V, B, K, F, _ = bench.DEFAULTS[cfg]
if "--blocks-per-step" in extra.split():  # (the one shape flag the profile runs use)
    K = int(extra.split()[extra.split().index("--blocks-per-step") + 1])


def find(pattern):
    m = glob.glob(pattern, recursive=True)
    return m[0] if m else None


stats = find(raw + "_stats/**/*kernel_stats.csv")
if stats:
    rows = list(csv.reader(open(stats)))
    # the stats table averages EVERY launch of the process — the clock ramp of the first tens of milliseconds included (round 6:
    # --warm-ms).  bench.py's avg_launch_us is measured over the 20 steps right behind the timed region: the same launches are the last
    # 2 x steps of the trace, so their mean is written beside the table's
    warm_note = ""
    trace = find(raw + "_stats/**/*kernel_trace.csv")
    if trace:
        by = collections.defaultdict(list)
        for row in csv.DictReader(open(trace)):
            try:
                by[row["Kernel_Name"]].append((int(row["Start_Timestamp"]), int(row["End_Timestamp"]) - int(row["Start_Timestamp"])))
            except (KeyError, ValueError):
                pass
        fw = {k: sorted(v) for k, v in by.items() if "fwgpu::k_" in k}
        if fw:
            domk = max(fw, key=lambda k: sum(d for _, d in fw[k]))
            d = [x for _, x in fw[domk]]
            last = d[-2 * steps:]
            warm_note = ("# dominant kernel %s: %d launches, mean %.1f us over all of them (clock ramp included), mean %.1f us / min %.1f us over the last %d "
                         "(the timed region and the event pass behind it: what bench.py's avg_launch_us measures)\n"
                         % (domk.split("(")[0].replace("void ", ""), len(d), sum(d) / len(d) / 1e3, sum(last) / len(last) / 1e3, min(last) / 1e3, len(last)))
    with open(os.path.join(outdir, "%s_%s%s_kernel_stats.csv" % (tag, cfg, suffix)), "w") as f:
        f.write("# rocprofv3 --kernel-trace --stats -- python bench.py --workload %s --lean --steps %d --warmup 3 --warm-ms %s %s"
                "   (MI355X, %s; %d voices, block %d, %d blocks per step)\n" % (cfg, steps, os.environ.get("WARM_MS", "60"), extra, tag, V, B, K))
        f.write(warm_note)
        w = csv.writer(f)
        for r in rows[:14]:
            w.writerow(r)


def per_kernel(ctr):
    f = find(raw + "_%s/**/*counter_collection.csv" % ctr)
    acc = collections.defaultdict(list)
    if f:
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") == ctr:
                acc[row["Kernel_Name"].split("(")[0].replace("void ", "").replace("fwgpu::", "")].append(float(row["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items() if k.startswith("k_")}


fetch, write = per_kernel("FETCH_SIZE"), per_kernel("WRITE_SIZE")
per_vs = {"cfg2": 8.0, "cfg3": 24.0, "cfg5": 8.0}.get(cfg)
if per_vs and "i16" in extra.split():  # (--source-format i16: SURVEY 8d's 4 B row; config 3: 4 B of source + ring read + ring write)
    per_vs = 20.0 if cfg == "cfg3" else 4.0
dom = {"cfg2": "k_leaf_sum", "cfg3": "k_chain", "cfg5": "k_leaf_sum", "cfg4": "k_fir_gemm"}[cfg]
if "--rs-source" in extra.split():
    dom = "k_leaf_rs"
# (the render kernel of a voice-bank plan goes by three names: plain, without a control kernel — lazy records —, with spatialiser
#  stages; bench.py looks its traffic up under "k_leaf_sum")
aliases = {"k_leaf_sum": ("k_leaf_sum", "k_leaf_sum_lazy", "k_leaf_sum_sp")}
out = {
    "command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace -- python bench.py --workload %s --lean "
               "--no-kernel-timing --steps 4 --warmup 2 %s (one pass per counter)" % (cfg, extra),
    "workload": {"name": cfg + suffix, "voices": V, "block": B, "blocks_per_step": K},
    "units": "rocprofv3 reports KiB; gfx950 FETCH_SIZE counts 1/2 of wide coalesced reads (MI355X_MICROARCH.md §HBM) -> "
             "doubled; WRITE_SIZE as reported",
    "per_launch_raw_KiB": {k: {"FETCH_SIZE": fetch.get(k), "WRITE_SIZE": write.get(k)} for k in sorted(set(fetch) | set(write))},
}
for k in sorted(set(fetch) | set(write)):
    name = k.split("<")[0]
    if name not in aliases.get(dom, (dom,)):
        continue
    fb = 2.0 * 1024.0 * fetch.get(k, 0.0)
    wb = 1024.0 * write.get(k, 0.0)
    ent = {"kernel_name": k, "fetch_bytes_corrected": fb, "write_bytes": wb, "traffic_bytes": fb + wb}
    if per_vs:
        alg = per_vs * V * B * K
        ent["algorithmic_bytes"] = alg
        ent["traffic_over_algorithmic"] = (fb + wb) / alg
    if dom in out and out[dom]["traffic_bytes"] >= fb + wb:
        continue  # (two names in one run — a control launch's plain kernel beside the lazy one: keep the one with the traffic)
    out[dom] = ent
json.dump(out, open(os.path.join(outdir, "%s_%s%s_pmc_hbm_traffic.json" % (tag, cfg, suffix)), "w"), indent=1)
print(cfg, json.dumps({k: v for k, v in out.items() if k.startswith("k_")}))
