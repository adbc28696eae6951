"""one-block callbacks of the config-2 graph PACED like an audio device would (a gap between them), host time per callback"""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from fwapi import GpuEngine, LOOP_FULL
import scenarios
gap_ms = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0
e = GpuEngine(max_block_frames=256)
voices = scenarios.build_voice_bank(e, 1024, src_frames=200000)
for vc in voices:
    e.sampler_set_loop_range(vc["sampler"], LOOP_FULL); e.sampler_play(vc["sampler"])
ts = []
for i in range(400):
    t = time.perf_counter(); e.process_interleaved(256); ts.append((time.perf_counter() - t) * 1e6)
    end = time.perf_counter() + gap_ms * 1e-3
    while time.perf_counter() < end: pass
ts = np.array(ts[50:])
print("gap %.1f ms: median %.1f us  p90 %.1f  p99 %.1f  (resident stats %s)" % (gap_ms, np.median(ts), np.percentile(ts, 90), np.percentile(ts, 99), e.cx.rt_resident_stats()))
