#!/bin/bash
# usage (on the GPU box, via gpurun):  [EXTRA="--voice-fx" SUFFIX=_voicefx] bash scripts/collect_profiles.sh r02 [cfg2 cfg3 cfg4 ...]
# For each workload: one `rocprofv3 --kernel-trace --stats` run of bench.py and two separate --pmc passes
# (FETCH_SIZE, WRITE_SIZE — TCC counters do not fit one pass; MI355X_MICROARCH.md §rocprofv3 PMC slots), then
# scripts/summarise_profiles.py turns them into gpurun_out/profiles/<tag>_<cfg>_{kernel_stats.csv,pmc_hbm_traffic.json}.
tag=$1; shift
cfgs=${@:-cfg2 cfg3 cfg4}
root=$GRAFT_REPO_ROOT
mkdir -p $root/gpurun_out/profiles $root/gpurun_out/raw
cd /tmp && export TMPDIR=/tmp
for cfg in $cfgs; do
  steps=20
  out=$root/gpurun_out/raw/${tag}_${cfg}${SUFFIX}
  # (--warm-ms: untimed steps until the device's clocks have settled, as bench.py's other_configs entries are timed — round 6)
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d ${out}_stats -o s -- python $root/bench.py --workload $cfg --lean --steps $steps --warmup 3 --warm-ms ${WARM_MS:-60} $EXTRA > ${out}_stats.log 2>&1
  for ctr in FETCH_SIZE WRITE_SIZE; do
    timeout 150 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d ${out}_$ctr -o p -- python $root/bench.py --workload $cfg --lean --no-kernel-timing --steps 4 --warmup 2 $EXTRA > ${out}_$ctr.log 2>&1
  done
  python $root/scripts/summarise_profiles.py $tag $cfg $steps $out $root/gpurun_out/profiles "$SUFFIX" "$EXTRA"
done
ls -la $root/gpurun_out/profiles
