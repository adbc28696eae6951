#!/bin/bash
# round 5, after the level executor's last change (32 blocks per wave on very wide levels, links rendered upstream leave at once): the
# GPU tier, the level-executor profiles and the full lines again on the final library (the other r05 profiles: scripts/collect_r05.sh —
# their kernels did not change in between)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/profiles gpurun_out/raw
P=$GRAFT_REPO_ROOT/gpurun_out/profiles
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 700 python -m pytest tests -m gpu -x -q -p no:cacheprovider > gpurun_out/r05_suite_final.log 2>&1; echo "suite rc $?" >> gpurun_out/r05_suite_final.log; tail -3 gpurun_out/r05_suite_final.log
FWGPU_POISON=2 timeout 700 python -m pytest tests -m gpu -x -q -p no:cacheprovider > gpurun_out/r05_suite_final_poison2.log 2>&1; echo "suite rc $?" >> gpurun_out/r05_suite_final_poison2.log; tail -2 gpurun_out/r05_suite_final_poison2.log
for f in 1 0; do
  out=$GRAFT_REPO_ROOT/gpurun_out/raw/r05_cfg2_levels_fuse$f
  rm -rf ${out}_stats
  (cd /tmp && TMPDIR=/tmp FWGPU_LEVEL_FUSE=$f timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d ${out}_stats -o s -- python $GRAFT_REPO_ROOT/bench.py --workload cfg2 --lean --contexts 1 --force-generic --steps 5 --warmup 2 > ${out}_stats.log 2>&1)
  g=$(find ${out}_stats -name "*kernel_stats.csv" | head -1)
  name=r05_cfg2_levels_only_kernel_stats.csv; [ $f = 0 ] && name=r05_cfg2_levels_only_unfused_kernel_stats.csv
  if [ -n "$g" ]; then
    { echo "# FWGPU_LEVEL_FUSE=$f rocprofv3 --kernel-trace --stats -- python bench.py --workload cfg2 --lean --contexts 1 --force-generic --steps 5 --warmup 2   (MI355X, r05 final library; 1024 voices, block 256, 768 blocks per step: sampler / volume / pan / leaf sums / root levels; 1 = frozen 1:1 chains rendered in registers by the wave upstream)"; head -8 "$g" | cut -c1-220; } > $P/$name
  fi
done
FWGPU_LEVEL_FUSE=1 python bench.py --workload cfg2 --lean --contexts 1 --force-generic --steps 5 --warmup 2 > $P/r05_cfg2_levels_only_line.json 2> /dev/null
FWGPU_LEVEL_FUSE=0 python bench.py --workload cfg2 --lean --contexts 1 --force-generic --steps 5 --warmup 2 > $P/r05_cfg2_levels_only_unfused_line.json 2> /dev/null
python bench.py > $P/r05_bench_line_full.json 2> gpurun_out/bench_full.err
timeout 400 python bench.py --gpus 8 --share-device --steps 10 --warmup 2 > $P/r05_n8_virtual_ranks_cfg5_line.json 2> gpurun_out/bench_n8.err
ls -la $P | grep r05 | tail -8
