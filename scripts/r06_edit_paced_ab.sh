# round 6: paced edit race (a callback every millisecond): the worst callback that began while a plan was being built, by plan order and
# by how the build's device work is grouped (stderr of fw_edit_race names the slow callbacks: which edit, which update phase)
make -C examples/host_c > /dev/null 2>&1
for rep in 1 2 3 4 5 6; do
  for o in "canonical 262144" "reference 262144"; do
    set -- $o
    FWGPU_PLAN_ORDER=$1 FWGPU_UP_PIECE=$2 ./examples/host_c/fw_edit_race 4096 512 300 30 1000 2> /tmp/er.err | python -c "
import sys, json
d = json.loads(sys.stdin.read())
s, b = d['callback_us_steady'], d['callback_us_while_the_plan_is_built']
print('$o', 'update median %.3f ms' % d['update_ms_median'], 'steady p99 %.1f max %.1f' % (s['p99'], s['max']), 'built n %d median %.1f p99 %.1f max %.1f' % (b['n'], b['median'], b['p99'], b['max']))
"
    grep "slow callback" /tmp/er.err | sed 's/^/      /'
  done
done
