"""The edit race (examples/host_c/fw_edit_race): graph edits on a control thread WHILE an audio thread runs callbacks.  A timing
test — its bounds have room, and it lives in the file that sorts last so that a miss on a noisy box cannot hide (pytest -x) the
parity tests of the rest of the GPU tier."""
import os

import pytest

import fwapi


@pytest.mark.gpu
def test_graph_edits_while_the_audio_thread_runs_cost_the_callbacks_microseconds_not_milliseconds():
    """VERDICT r2 missing #3: examples/host_c/fw_edit_race (plain C + pthreads through the C ABI) replaces voices of the config-3
    graph — 4 096 voices of sampler -> biquad -> delay -> gain — one after another while an audio thread runs one-block
    callbacks.  Each fwgpu_update recompiles the whole launch plan and uploads what changed of it (~1 ms since round 4, 2-6 before) ON
    THE CONTROL THREAD, off to the side; the callback that follows adopts it (graph/processor.rs:167-206).  The bar: an adoption holds its callback up for
    microseconds (measured 17-48), and the callbacks' median does not move.  Every timing bound below has room — the measured values
    are in profiles/r04_edit_race_cfg3.json and DESIGN.md section 1; a miss here would hide the parity tests that run after it."""
    import json
    import subprocess

    exe = os.path.join(fwapi.ROOT, "examples", "host_c", "fw_edit_race")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.dirname(exe)])
    env = {k: v for k, v in os.environ.items() if k not in ("FWGPU_LAZY_ADOPT", "FWGPU_POISON", "FWGPU_POISON_ONLY")}  # (test modes, not the product's)
    def run(*extra):
        r = subprocess.run([exe, "4096", "512", "300", "30"] + list(extra), capture_output=True, text=True, timeout=300, env=env)
        assert r.returncode == 0, r.stderr[-2000:]
        return json.loads(r.stdout.strip().splitlines()[-1])

    runs = [run() for _ in range(3)]   # three runs, the middle one of each figure: one outlier callback in 9 000 is not the design
    mid = lambda f: sorted(f(d) for d in runs)[1]
    for d in runs:
        assert d["launch_plan"] == 2 and d["edits"] == 30
        assert d["adopted_by_a_callback"] >= 20, d      # the audio thread was running: (nearly) every plan was picked up by a callback
        assert d["longest_adoption_us"] < 200.0, d      # (measured 17-48 us; a build is 2-5 ms)
        assert d["update_ms_mean"] > 0.3, d             # ... while each update really was a millisecond of work
    assert mid(lambda d: d["update_ms_median"]) <= 1.6, runs  # (measured 0.94-1.00; the old host code on the box of the A/B: 2.1-2.5.  The mean
                                                            #  carries the first edit's device allocations: 2-25 ms depending on the box)
    assert mid(lambda d: d["callback_us_while_the_plan_is_built"]["median"] - d["callback_us_steady"]["median"]) <= 10.0, runs
    # VERDICT r3's bar for a SATURATED audio thread (callbacks back to back, no gap for the build's groups to use): p99 <= steady + 30 us,
    # maximum <= steady maximum + 50 us.  Round 4 meets it with the build's job groups launched into the audio stream (measured +3..+12 /
    # -110..+11, profiles/r04_edit_race_cfg3.json; on the build's own stream — FWGPU_BUILD_STREAM=own — it was +40-65 / +60-75).
    if os.environ.get("FWGPU_QUIET_WAIT_US", "100") != "0" and os.environ.get("FWGPU_BUILD_STREAM", "audio") != "own":
        assert mid(lambda d: d["callback_us_while_the_plan_is_built"]["p99"] - d["callback_us_steady"]["p99"]) <= 30.0, runs
        assert mid(lambda d: d["callback_us_while_the_plan_is_built"]["max"] - d["callback_us_steady"]["max"]) <= 50.0, runs
        # ... and a paced stream (a callback every millisecond) does not see a build: same p99 (+15: the resolution of a 30-sample tail)
        # (three runs and the middle one here too, since round 6: ~30 callbacks begin while a plan is built, so the p99 IS the maximum, and
        #  one callback of +35 us — in the SECOND edit of a run, the first rebuild of the first plan image with its device allocations on the
        #  control thread — turns up in one run of three with either plan order: scripts/r06_edit_paced_ab.sh, fw_edit_race's stderr)
        paced = [run("1000") for _ in range(3)]
        for d in paced:
            assert d["callback_period_us"] == 1000 and d["callback_us_while_the_plan_is_built"]["n"] >= 15, d
        pmid = lambda f: sorted(f(d) for d in paced)[1]
        assert pmid(lambda d: d["callback_us_while_the_plan_is_built"]["p99"] - d["callback_us_steady"]["p99"]) <= 15.0, paced
        assert pmid(lambda d: d["callback_us_while_the_plan_is_built"]["max"] - d["callback_us_steady"]["max"]) <= 50.0, paced
