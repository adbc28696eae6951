#!/bin/bash
# Round 4, first GPU call: what the control kernel costs the headline step, mode by mode, on ONE box.
#   FWGPU_CTL_AHEAD = 0 (in-stream), 1 (round 3: every call a batch ahead), 2 (round 4 default: only calls with messages / glides)
# variants A (steady) and B (a gain glide per voice per run), medians of $CTX fresh contexts; then one timed pass for the kernels' own durations.
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r04
CTX=${CTX:-7}
for mode in 0 1 2; do for var in A B; do
FWGPU_CTL_AHEAD=$mode timeout 300 python bench.py --variant $var --lean --steps 30 --no-kernel-timing --contexts $CTX 2>gpurun_out/r04/ab_ctl_${mode}_$var.err | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('AHEAD=$mode $var', '%.4g'%d['value'], 'median %.4f'%d['ms_per_step'], ['%.3f'%x for x in sorted(d['contexts']['ms_per_step_runs'])])"
done; done | tee gpurun_out/r04/ab_ctl.txt
FWGPU_CTL_AHEAD=2 timeout 200 python bench.py --lean --steps 30 --contexts 3 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('timed pass:', d['ms_per_step'], json.dumps(d['roofline']['other_kernels_us_per_step']), d['roofline']['avg_launch_us'], d['contexts']['kernel_us_runs'])" | tee -a gpurun_out/r04/ab_ctl.txt
