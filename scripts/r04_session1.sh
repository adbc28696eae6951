#!/bin/bash
# GPU session 1 of round 4: control-kernel A/B on one box, the GPU suite, the edit race with the build's groups on the audio stream
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r04
CTX=7 bash scripts/r04_ab_ctl.sh
timeout 600 python -m pytest tests -m gpu -x -q -p no:cacheprovider > gpurun_out/r04/suite.log 2>&1; tail -3 gpurun_out/r04/suite.log
make -C examples/host_c > /dev/null 2>&1
for mode in default audio; do
  for per in 0 1000; do
    for i in 1 2; do
      if [ $mode = audio ]; then export FWGPU_BUILD_STREAM=audio; else unset FWGPU_BUILD_STREAM; fi
      timeout 120 ./examples/host_c/fw_edit_race 4096 512 300 30 $per > gpurun_out/r04/edit_race_${mode}_p${per}_$i.json 2> gpurun_out/r04/edit_race_${mode}_p${per}_$i.err
    done
  done
done
unset FWGPU_BUILD_STREAM
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
