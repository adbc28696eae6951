#!/bin/bash
# round 3: everything under profiles/r03_* in one gpurun call (≈ 6 GPU-minutes)
cd $GRAFT_REPO_ROOT
bash scripts/collect_profiles.sh r03 cfg2 cfg3 cfg4 cfg5 > gpurun_out/collect_r03_a.log 2>&1
EXTRA="--rs-source" SUFFIX=_rs bash scripts/collect_profiles.sh r03 cfg2 > gpurun_out/collect_r03_b.log 2>&1
EXTRA="--voice-spatial" SUFFIX=_spatial bash scripts/collect_profiles.sh r03 cfg2 > gpurun_out/collect_r03_c.log 2>&1
EXTRA="--variant B" SUFFIX=_variantB bash scripts/collect_profiles.sh r03 cfg2 > gpurun_out/collect_r03_d.log 2>&1
bash scripts/prof_sq.sh rs --rs-source > gpurun_out/profiles/r03_cfg2_rs_sq_counters.txt 2>&1
bash scripts/prof_sq.sh spatial --voice-spatial > gpurun_out/profiles/r03_cfg2_spatial_sq_counters.txt 2>&1
python bench.py > gpurun_out/profiles/r03_bench_line_full.json 2> gpurun_out/bench_full.err
python bench.py --gpus 2 --share-device > gpurun_out/profiles/r03_n2_virtual_ranks_one_device_line.json 2> gpurun_out/bench_n2.err
make -C examples/host_c > /dev/null 2>&1 && ./examples/host_c/fw_edit_race 4096 512 300 30 > gpurun_out/profiles/r03_edit_race_cfg3.json 2> gpurun_out/profiles/r03_edit_race_cfg3_by_update_phase.txt
./examples/host_c/fw_edit_race 4096 512 300 30 1000 > gpurun_out/profiles/r03_edit_race_cfg3_paced_1ms.json 2> /dev/null
FWGPU_QUIET_WAIT_US=0 ./examples/host_c/fw_edit_race 4096 512 300 30 > gpurun_out/profiles/r03_edit_race_cfg3_all_at_once.json 2> /dev/null
ls -la gpurun_out/profiles
