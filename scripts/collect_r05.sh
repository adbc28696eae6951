#!/bin/bash
# round 5: everything under profiles/r05_* in one gpurun call.
#  * the whole GPU test tier once more (the library that is profiled is the library that is tested);
#  * cfg2 (headline), cfg3, cfg4, cfg5, cfg2 + resampler sources, cfg2 + spatialiser: kernel stats + the two PMC passes — the profiles
#    the bench line's roofline objects are checked against;
#  * cfg2 on the level executor alone (vertical fusion of frozen chains), fused and FWGPU_LEVEL_FUSE=0: kernel stats + line;
#  * cfg2 with resampler sources: kernel stats (round 4's kernel stays: scripts/experiments/r05_rs_register_window.patch);
#  * the full bench line (k_sweep, realtime per config, pipelined host buffers); N = 8 on one device.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/profiles gpurun_out/raw
P=$GRAFT_REPO_ROOT/gpurun_out/profiles
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 700 python -m pytest tests -m gpu -x -q -p no:cacheprovider > gpurun_out/r05_suite_final.log 2>&1; echo "suite rc $?" >> gpurun_out/r05_suite_final.log; tail -3 gpurun_out/r05_suite_final.log
EXTRA="--contexts 1" bash scripts/collect_profiles.sh r05 cfg2 cfg3 cfg4 cfg5 > gpurun_out/collect_r05_a.log 2>&1
EXTRA="--contexts 1 --rs-source" SUFFIX=_rs bash scripts/collect_profiles.sh r05 cfg2 > gpurun_out/collect_r05_b.log 2>&1
EXTRA="--contexts 1 --voice-spatial" SUFFIX=_spatial bash scripts/collect_profiles.sh r05 cfg2 > gpurun_out/collect_r05_c.log 2>&1
cd $GRAFT_REPO_ROOT
for f in 1 0; do
  out=$GRAFT_REPO_ROOT/gpurun_out/raw/r05_cfg2_levels_fuse$f
  (cd /tmp && TMPDIR=/tmp FWGPU_LEVEL_FUSE=$f timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d ${out}_stats -o s -- python $GRAFT_REPO_ROOT/bench.py --workload cfg2 --lean --contexts 1 --force-generic --steps 5 --warmup 2 > ${out}_stats.log 2>&1)
  g=$(find ${out}_stats -name "*kernel_stats.csv" | head -1)
  name=r05_cfg2_levels_only_kernel_stats.csv; [ $f = 0 ] && name=r05_cfg2_levels_only_unfused_kernel_stats.csv
  if [ -n "$g" ]; then
    { echo "# FWGPU_LEVEL_FUSE=$f rocprofv3 --kernel-trace --stats -- python bench.py --workload cfg2 --lean --contexts 1 --force-generic --steps 5 --warmup 2   (MI355X, r05; 1024 voices, block 256, 768 blocks per step: sampler / volume / pan / leaf sums / root levels; 1 = frozen 1:1 chains rendered in registers by the wave upstream)"; head -8 "$g" | cut -c1-220; } > $P/$name
  fi
done
FWGPU_LEVEL_FUSE=1 python bench.py --workload cfg2 --lean --contexts 1 --force-generic --steps 5 --warmup 2 > $P/r05_cfg2_levels_only_line.json 2> /dev/null
python bench.py > $P/r05_bench_line_full.json 2> gpurun_out/bench_full.err
timeout 400 python bench.py --gpus 8 --share-device --steps 10 --warmup 2 > $P/r05_n8_virtual_ranks_cfg5_line.json 2> gpurun_out/bench_n8.err
ls -la $P | tail -20
