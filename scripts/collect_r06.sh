#!/bin/bash
# round 6: everything under profiles/r06_* in one gpurun call.
#  * the whole GPU test tier once more (the library that is profiled is the library that is tested);
#  * cfg2 (headline), cfg3, cfg3 reordered (gain -> biquad -> biquad -> delay -> pan on 16-bit sources), cfg4, cfg5, cfg2 + resampler sources,
#    cfg2 + spatialiser: kernel stats + the two PMC passes — the profiles the bench line's roofline objects are checked against;
#  * cfg2 on the level executor alone: kernel stats + line;
#  * the bench line as the driver runs it (compact) and the whole record; the C ABI's RCCL reductions at world 1.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/profiles gpurun_out/raw
P=$GRAFT_REPO_ROOT/gpurun_out/profiles
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider > gpurun_out/r06_suite_final.log 2>&1; echo "suite rc $?" >> gpurun_out/r06_suite_final.log; grep -E "passed|failed|rc" gpurun_out/r06_suite_final.log | tail -3
EXTRA="--contexts 1" bash scripts/collect_profiles.sh r06 cfg2 cfg3 cfg4 cfg5 > gpurun_out/collect_r06_a.log 2>&1
EXTRA="--contexts 1 --chain-reordered --source-format i16" SUFFIX=_reordered bash scripts/collect_profiles.sh r06 cfg3 > gpurun_out/collect_r06_d.log 2>&1
EXTRA="--contexts 1 --rs-source" SUFFIX=_rs bash scripts/collect_profiles.sh r06 cfg2 > gpurun_out/collect_r06_b.log 2>&1
EXTRA="--contexts 1 --voice-spatial" SUFFIX=_spatial bash scripts/collect_profiles.sh r06 cfg2 > gpurun_out/collect_r06_c.log 2>&1
cd $GRAFT_REPO_ROOT
out=$GRAFT_REPO_ROOT/gpurun_out/raw/r06_cfg2_levels
(cd /tmp && TMPDIR=/tmp timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d ${out}_stats -o s -- python $GRAFT_REPO_ROOT/bench.py --workload cfg2 --lean --contexts 1 --force-generic --steps 5 --warmup 2 --warm-ms 60 > ${out}_stats.log 2>&1)
g=$(find ${out}_stats -name "*kernel_stats.csv" | head -1)
if [ -n "$g" ]; then
  { echo "# rocprofv3 --kernel-trace --stats -- python bench.py --workload cfg2 --lean --contexts 1 --force-generic --steps 5 --warmup 2   (MI355X, r06; 1024 voices, block 256, 768 blocks per step: the level executor alone)"; head -8 "$g" | cut -c1-220; } > $P/r06_cfg2_levels_only_kernel_stats.csv
fi
python bench.py --workload cfg2 --lean --contexts 1 --force-generic --steps 5 --warmup 2 --warm-ms 60 > $P/r06_cfg2_levels_only_line.json 2> /dev/null
python bench.py --gpus 1 --steps 20 --warmup 5 > $P/r06_bench_line.json 2> gpurun_out/bench_line.err
cp gpurun_out/bench_full.json $P/r06_bench_full.json
for m in allreduce_abi ordered_abi; do
  FWGPU_BENCH_FORCE_DIST=1 timeout 300 python bench.py --lean --steps 10 --warmup 2 --contexts 1 --bus-reduce $m > $P/r06_world1_${m}_line.json 2> gpurun_out/bench_$m.err || tail -5 gpurun_out/bench_$m.err
done
timeout 500 python bench.py --gpus 8 --share-device --steps 10 --warmup 2 > $P/r06_n8_virtual_ranks_line.json 2> gpurun_out/bench_n8.err; cp gpurun_out/bench_full_n8.json $P/r06_n8_virtual_ranks_full.json 2>/dev/null
ls -la $P | tail -30
