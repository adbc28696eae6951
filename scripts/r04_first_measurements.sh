#!/bin/bash
# First GPU call of the next round: what round 3 built and could not measure any more.
#   gpurun --timeout 600 -- scripts/r04_first_measurements.sh
# 1. parity of the whole GPU tier in the default mode and with every plan adopted by a process call + poisoned memory
# 2. the edit race (config 3, 4 096 voices) back to back and paced, default against FWGPU_BUILD_STREAM=audio — the build's
#    k_build_apply groups launched into the AUDIO stream: if the fixed ~27 us a callback pays per operation on the build's stream is the
#    wake-up of an idle hardware queue, the saturated stream's p99 drops to steady + a few microseconds
# 3. the default bench line
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r04
timeout 200 python -m pytest tests -m gpu -x -q -p no:cacheprovider > gpurun_out/r04/suite.log 2>&1; tail -1 gpurun_out/r04/suite.log
FWGPU_LAZY_ADOPT=1 FWGPU_POISON=1 timeout 200 python -m pytest tests -m gpu -x -q -p no:cacheprovider > gpurun_out/r04/suite_lazy_poison.log 2>&1; tail -1 gpurun_out/r04/suite_lazy_poison.log
make -C examples/host_c > /dev/null 2>&1
for mode in default audio; do
  for per in 0 1000; do
    for i in 1 2 3; do
      if [ $mode = audio ]; then export FWGPU_BUILD_STREAM=audio; else unset FWGPU_BUILD_STREAM; fi
      ./examples/host_c/fw_edit_race 4096 512 300 30 $per > gpurun_out/r04/edit_race_${mode}_p${per}_$i.json 2> gpurun_out/r04/edit_race_${mode}_p${per}_$i.err
    done
  done
done
unset FWGPU_BUILD_STREAM
FWGPU_BUILD_STREAM=audio timeout 200 python -m pytest tests -m gpu -x -q -p no:cacheprovider > gpurun_out/r04/suite_build_stream_audio.log 2>&1; tail -1 gpurun_out/r04/suite_build_stream_audio.log
python - <<'PY'
import glob, json
for f in sorted(glob.glob('gpurun_out/r04/edit_race_*.json')):
    try:
        d = json.load(open(f))
    except Exception as ex:
        print(f, 'unreadable', ex); continue
    s, b = d['callback_us_steady'], d['callback_us_while_the_plan_is_built']
    print(f.split('/')[-1][:-5].ljust(28), 'update %.2f ms | steady p99 %.1f max %.1f | built n %d median %.1f p99 %.1f max %.1f | longest adoption %.1f us'
          % (d['update_ms_mean'], s['p99'], s['max'], b['n'], b['median'], b['p99'], b['max'], d['longest_adoption_us']))
PY
python bench.py > gpurun_out/r04/bench_line.json 2> gpurun_out/r04/bench.err; tail -c 400 gpurun_out/r04/bench_line.json
