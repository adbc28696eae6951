#!/bin/bash
# fw_edit_race (config-3 bank, 4096 voices, 512-frame callbacks, 30 voice replacements) in four modes:
#   back to back / paced at 1 ms, each with the build as it is by default (job list, changed chunks, grouped k_build_apply in quiet
#   windows) and with everything issued at once (FWGPU_QUIET_WAIT_US=0)
mkdir -p gpurun_out
make -C examples/host_c > /dev/null 2>&1
for q in 100 0; do
  for per in 0 1000; do
    for i in 1 2; do
      FWGPU_QUIET_WAIT_US=$q ./examples/host_c/fw_edit_race 4096 512 300 30 $per > gpurun_out/er_q${q}_p${per}_$i.json 2> gpurun_out/er_q${q}_p${per}_$i.err
    done
  done
done
