# round 6: examples/host_c/fw_edit_race — one voice after another of config 3's bank replaced while callbacks run — with build_plan's canonical
# table order (the default) against the reference's Kahn order (FWGPU_PLAN_ORDER=reference), same box, runs interleaved.
# FWGPU_UPDATE_PROF=1: host microseconds per update phase ([1] graph compile, [22] activation + plan detection, [23] node tables, [27] voice tables,
# [3] the build's device work: upload groups and their waits, [4] commit) and what an update puts on the stream.
make -C examples/host_c > /dev/null 2>&1
one() {  # order voices period
  FWGPU_PLAN_ORDER=$1 FWGPU_UPDATE_PROF=1 ./examples/host_c/fw_edit_race $2 512 300 30 $3 2> /tmp/er.err | python -c "
import sys, json
d = json.loads(sys.stdin.read())
s, b, a = d['callback_us_steady'], d['callback_us_while_the_plan_is_built'], d['callback_us_adoption_and_the_two_after']
print('$1 voices $2 period ${3:-0} us: update median %.3f mean %.3f max %.3f ms | callbacks steady median %.1f p99 %.1f max %.1f | while built n %d median %.1f p99 %.1f max %.1f | adoption+2 median %.1f max %.1f' % (
    d['update_ms_median'], d['update_ms_mean'], d['update_ms_max'], s['median'], s['p99'], s['max'], b['n'], b['median'], b['p99'], b['max'], a['median'], a['max']))
"
  grep "fwgpu update profile" /tmp/er.err | sed 's/^/    /'
}
for rep in 1 2 3; do for o in canonical reference; do one $o 4096 ""; done; done
for o in canonical reference; do one $o 8192 ""; done
for rep in 1 2 3; do for o in canonical reference; do one $o 4096 1000; done; done
