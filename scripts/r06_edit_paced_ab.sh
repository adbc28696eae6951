# round 6: paced edit race (a callback every millisecond), canonical plan order against the reference's: median update time and the worst callback that began while a plan was being built
make -C examples/host_c > /dev/null 2>&1
for rep in 1 2 3 4 5 6; do
  for o in "canonical 60" "reference 60"; do
    set -- $o
    FWGPU_PLAN_ORDER=$1 FWGPU_QUIET_MARGIN_US=$2 ./examples/host_c/fw_edit_race 4096 512 300 30 1000 2> /dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
s, b = d['callback_us_steady'], d['callback_us_while_the_plan_is_built']
print('$o', 'update median %.3f ms' % d['update_ms_median'], 'steady p99 %.1f max %.1f' % (s['p99'], s['max']), 'built n %d median %.1f p99 %.1f max %.1f' % (b['n'], b['median'], b['p99'], b['max']))
"
  done
done
