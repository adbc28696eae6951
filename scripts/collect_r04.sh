#!/bin/bash
# round 4: everything under profiles/r04_* in one gpurun call.
#  * cfg2 / cfg5: ONE context per profiled process, repeated until both HBM placement states of k_leaf_sum have been seen (at most 5
#    tries): r04_<cfg>_kernel_stats_{fast,slow}.csv, the state and the kernel-only roofline fraction in the header line — so that a
#    profile can be set beside the entry of the bench line that ran in the same state;
#  * cfg3, cfg4, cfg2 + resampler sources / spatialiser stages / variant B / i16 sources: kernel stats + PMC HBM traffic;
#  * SQ counters for the two LDS-heavy kernels; the full bench line; N = 2 on one device; config 2 on the level executor alone.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/profiles gpurun_out/raw
P=$GRAFT_REPO_ROOT/gpurun_out/profiles
state_run() {  # cfg
  cfg=$1
  for try in 1 2 3 4 5; do
    out=$GRAFT_REPO_ROOT/gpurun_out/raw/r04_${cfg}_try$try
    (cd /tmp && TMPDIR=/tmp timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d ${out}_stats -o s -- python $GRAFT_REPO_ROOT/bench.py --workload $cfg --lean --contexts 1 --steps 20 --warmup 3 > ${out}_stats.log 2>&1)
    python - $cfg $out $P $try <<'PY'
import csv, glob, os, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import bench
cfg, raw, outdir, tr = sys.argv[1:5]
V, B, K, F, _ = bench.DEFAULTS[cfg]
f = glob.glob(raw + "_stats/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.reader(open(f)))
leaf = [r for r in rows[1:] if "k_leaf_sum" in r[0]][0]
avg_ns = float(leaf[3])
frac = V * B * K * 8.0 / (avg_ns * 1e-9) / 8e12
state = "fast" if frac >= 0.76 else "slow"
dst = os.path.join(outdir, "r04_%s_kernel_stats_%s.csv" % (cfg, state))
if not os.path.exists(dst):
    with open(dst, "w") as o:
        o.write("# placement_state=%s  k_leaf_sum kernel-only roofline fraction %.3f (%.1f us per launch; >= 0.76 = fast)  |  rocprofv3 --kernel-trace --stats -- "
                "python bench.py --workload %s --lean --contexts 1 --steps 20 --warmup 3   (MI355X, r04, ONE fresh context; %d voices, block %d, %d blocks per step)\n"
                % (state, frac, avg_ns / 1e3, cfg, V, B, K))
        w = csv.writer(o)
        for r in rows[:12]:
            w.writerow([c[:160] for c in r])
print(cfg, "try", tr, state, "%.3f" % frac)
PY
    [ -f $P/r04_${cfg}_kernel_stats_fast.csv ] && [ -f $P/r04_${cfg}_kernel_stats_slow.csv ] && break
  done
}
state_run cfg2
state_run cfg5
EXTRA="--contexts 1" bash scripts/collect_profiles.sh r04 cfg2 cfg3 cfg4 cfg5 > gpurun_out/collect_r04_a.log 2>&1
EXTRA="--contexts 1 --rs-source" SUFFIX=_rs bash scripts/collect_profiles.sh r04 cfg2 > gpurun_out/collect_r04_b.log 2>&1
EXTRA="--contexts 1 --voice-spatial" SUFFIX=_spatial bash scripts/collect_profiles.sh r04 cfg2 > gpurun_out/collect_r04_c.log 2>&1
EXTRA="--contexts 1 --variant B" SUFFIX=_variantB bash scripts/collect_profiles.sh r04 cfg2 > gpurun_out/collect_r04_d.log 2>&1
EXTRA="--contexts 1 --source-format i16" SUFFIX=_i16 bash scripts/collect_profiles.sh r04 cfg2 > gpurun_out/collect_r04_e.log 2>&1
bash scripts/prof_sq.sh rs --rs-source --contexts 1 > $P/r04_cfg2_rs_sq_counters.txt 2>&1
bash scripts/prof_sq.sh spatial --voice-spatial --contexts 1 > $P/r04_cfg2_spatial_sq_counters.txt 2>&1
cd $GRAFT_REPO_ROOT
python bench.py > $P/r04_bench_line_full.json 2> gpurun_out/bench_full.err
python bench.py --gpus 2 --share-device > $P/r04_n2_virtual_ranks_one_device_line.json 2> gpurun_out/bench_n2.err
# config 2 on the level executor alone (--force-generic): kernel stats only
out=$GRAFT_REPO_ROOT/gpurun_out/raw/r04_cfg2_levels_only
(cd /tmp && TMPDIR=/tmp timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d ${out}_stats -o s -- python $GRAFT_REPO_ROOT/bench.py --workload cfg2 --lean --contexts 1 --force-generic --steps 5 --warmup 2 > ${out}_stats.log 2>&1)
f=$(find ${out}_stats -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then
  { echo "# rocprofv3 --kernel-trace --stats -- python bench.py --workload cfg2 --lean --contexts 1 --force-generic --steps 5 --warmup 2   (MI355X, r04; 1024 voices, block 256, 768 blocks per step: sampler / volume / pan / leaf sums / root levels)"; head -8 "$f" | cut -c1-220; } > $P/r04_cfg2_levels_only_kernel_stats.csv
fi
python bench.py --workload cfg2 --lean --contexts 1 --force-generic --steps 5 --warmup 2 > $P/r04_cfg2_levels_only_line.json 2> /dev/null
# (the edit race: profiles/r04_edit_race_cfg3.json is a same-box A/B against the old host code — scripts/r04_edit_ab.sh — and is not collected here)
ls -la $P | tail -40
