# A/B of the control kernel's workgroup size and of control-ahead mode on variants A and B (medians of 5 fresh contexts each)
for wpb in 4 1; do for ah in 0 1; do
for var in A B; do
FWGPU_CTL_WPB=$wpb FWGPU_CTL_AHEAD=$ah python bench.py --variant $var --lean --steps 40 --no-kernel-timing --contexts 5 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('WPB=$wpb AHEAD=$ah $var', '%.4g'%d['value'], '%.4f'%d['ms_per_step'], [round(x['value']/1e9) for x in d.get('contexts',{}).get('runs',[])] if isinstance(d.get('contexts'),dict) else '')"
done; done; done
