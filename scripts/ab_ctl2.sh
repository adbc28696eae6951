# control-ahead mode x control kernel build (FWGPU_CTL_OCC: 1 = 240 VGPRs, 3 = 168) on variants A and B, medians of 9 fresh contexts
for cfg in "0 1" "1 1" "1 3"; do set -- $cfg; for var in A B; do
FWGPU_CTL_AHEAD=$1 FWGPU_CTL_OCC=$2 python bench.py --variant $var --lean --steps 30 --no-kernel-timing --contexts 9 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('AHEAD=$1 OCC=$2 $var', '%.4g'%d['value'], '%.4f'%d['ms_per_step'], ['%.3f'%x for x in sorted(d['contexts']['ms_per_step_runs'])])"
done; done
