#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r05
for mode in 0 1; do
  for k in "96-4-256" "200-8-64" "70-4-1024" "33-2-128"; do
    FWGPU_RT_PERSIST=$mode timeout 60 python -m pytest tests/test_rt_resident.py -m gpu -x -q -p no:cacheprovider -k "any_depth and $k" > gpurun_out/r05/rt_tree_${mode}_$k.log 2>&1
    echo "persist=$mode $k rc=$? $(tail -1 gpurun_out/r05/rt_tree_${mode}_$k.log | cut -c1-150)"
  done
done
