"""what the first callback after a plan adoption costs (cold steady caches) against the callbacks around it"""
import sys, time, os
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from fwapi import GpuEngine, LOOP_FULL
import scenarios
which = sys.argv[1] if len(sys.argv) > 1 else "chain"
if which == "chain":
    e = GpuEngine(max_block_frames=512)
    voices = scenarios.build_chain_bank(e, 4096, src_frames=20000)
else:
    e = GpuEngine(max_block_frames=256)
    voices = scenarios.build_voice_bank(e, 1024, src_frames=20000)
for vc in voices:
    e.sampler_set_loop_range(vc["sampler"], LOOP_FULL)
    e.sampler_play(vc["sampler"])
mbf = e.max_block_frames
def cb():
    t = time.perf_counter(); e.process_interleaved(mbf); return (time.perf_counter() - t) * 1e6
for _ in range(50): cb()
rows = []
for ed in range(6):
    pre = [cb() for _ in range(5)]
    v = voices[100 + ed]
    e.set_param(v["volume"], 0, 50.0 + ed) if False else None
    n = e.volume(55.0); e.remove_node(n); e.update()       # an edit that leaves every voice as it was
    post = [cb() for _ in range(5)]
    rows.append((np.median(pre), post))
for pre, post in rows:
    print("steady %.1f us | after the edit: %s" % (pre, " ".join("%.1f" % x for x in post)))
