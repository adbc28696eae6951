#!/bin/bash
# round 4: the edit race as a same-box A/B of two libraries (profiles/r04_edit_race_cfg3.json).
#   usage (via gpurun):  bash scripts/r04_edit_ab.sh /path/to/old_libfwgpu.so
# `old` is preloaded (LD_PRELOAD) in front of the product library the example links against; runs are interleaved old / new / new with
# 256 KiB groups, three times, then one paced run each.  FWGPU_UPDATE_PROF=1 prints the new library's host time per update phase.
old=$1
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/edit2
make -C examples/host_c > /dev/null 2>&1
cd examples/host_c
for i in 1 2 3; do
  LD_PRELOAD=$old ./fw_edit_race 4096 512 300 30 > ../../gpurun_out/edit2/old_$i.json 2>/dev/null
  FWGPU_UP_PIECE=131072 FWGPU_UPDATE_PROF=1 ./fw_edit_race 4096 512 300 30 > ../../gpurun_out/edit2/new_$i.json 2> ../../gpurun_out/edit2/new_$i.err
  ./fw_edit_race 4096 512 300 30 > ../../gpurun_out/edit2/new256_$i.json 2>/dev/null
done
LD_PRELOAD=$old ./fw_edit_race 4096 512 300 30 1000 > ../../gpurun_out/edit2/old_paced.json 2>/dev/null
./fw_edit_race 4096 512 300 30 1000 > ../../gpurun_out/edit2/new_paced.json 2>/dev/null
