"""CPU tier: the oracle (oracle/fw_oracle.cpp) against a SECOND, independent restatement of the reference's graph level
(tests/refmodel.py, numpy float32, written from the .rs files — no shared code), bit for bit:

* 320 seeds of the GPU fuzz families' own generators (random voice banks with gain / pan / biquad / delay chains, every
  sample format, master chains, message traffic tagged at random blocks, graph edits between calls; effects racks on
  stream inputs with calls of arbitrary length), spread over worker processes;
* every parity scenario the model covers, through digests generated FROM THE MODEL (tests/golden/refmodel_digests.json).

What this buys: a misreading of smoother.rs / sampler.rs / sum.rs / volume.rs, or a slip in the
SPEC nodes' control math, would have to be made twice, independently, in two languages, to go unnoticed.  What it cannot
buy: both restatements were written by the same builder from the same reading of the Rust source — the Rust reference
itself has still never run here."""
import json
import multiprocessing as mp
import os
import sys

import numpy as np
import pytest

import fwapi
import scenarios

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
N_BANK, N_STREAM = 240, 80


def _pair(kind, seed):
    """one seed on both restatements -> (ok, message)"""
    import refmodel
    import test_fuzz_gpu as F

    if kind == "bank":
        pick = np.random.default_rng(10_000 + seed)
        mbf = int(pick.choice([64, 128, 256]))
        want = F.fuzz_run(scenarios.TaggedOracle(fwapi.OracleEngine(max_block_frames=mbf)), seed)
        got = F.fuzz_run(scenarios.TaggedOracle(refmodel.RefEngine(max_block_frames=mbf)), seed)
    else:
        pick = np.random.default_rng(95_000 + seed)
        mbf = int(pick.choice([16, 64, 100, 256]))
        n_in = int(pick.choice([1, 2, 3, 4]))
        want = F.fuzz_stream(scenarios.TaggedOracle(fwapi.OracleEngine(max_block_frames=mbf, num_graph_inputs=n_in)), seed, n_in)
        got = F.fuzz_stream(scenarios.TaggedOracle(refmodel.RefEngine(max_block_frames=mbf, num_graph_inputs=n_in)), seed, n_in)
    a, b = fwapi.bits(want), fwapi.bits(got)
    if a.shape == b.shape and np.array_equal(a, b):
        return True, ""
    bad = np.nonzero(a != b)[0] if a.shape == b.shape else np.array([0])
    return False, "%s seed %d: %d of %d samples differ, first at %d" % (kind, seed, bad.size, a.size, bad[0])


def _work(job):
    try:
        return _pair(*job)
    except Exception as ex:  # a crash in a worker is a failure of that seed, with its name on it
        return False, "%s seed %d: %r" % (job[0], job[1], ex)


def test_oracle_and_independent_model_agree_on_320_fuzz_seeds():
    jobs = [("bank", s) for s in range(N_BANK)] + [("stream", s) for s in range(N_STREAM)]
    fwapi.oracle_lib()  # built once, before the workers start
    procs = max(1, min(8, (os.cpu_count() or 2)))
    with mp.get_context("fork").Pool(procs) as pool:
        res = pool.map(_work, jobs, chunksize=4)
    fails = [m for ok, m in res if not ok]
    assert not fails, fails[:10]


def test_fma_emulation_is_correctly_rounded():
    """refmodel.fma32 (f64 TwoSum + round-to-odd) against exact rational arithmetic on adversarial operands"""
    from fractions import Fraction

    import refmodel

    rng = np.random.default_rng(7)
    a = rng.standard_normal(4000).astype(np.float32)
    b = rng.standard_normal(4000).astype(np.float32)
    c = (-(a.astype(np.float64) * b.astype(np.float64))).astype(np.float32)      # near-total cancellation
    c[::3] = rng.standard_normal(c[::3].size).astype(np.float32) * np.float32(1e-6)
    c[1::7] = np.float32(2.0) ** rng.integers(-60, 60, c[1::7].size).astype(np.float32)  # halfway-prone magnitudes
    got = refmodel.fma32(a, b, c)
    for i in range(a.size):
        exact = Fraction(float(a[i])) * Fraction(float(b[i])) + Fraction(float(c[i]))
        want = np.float32(exact)  # Fraction -> float is correctly rounded to f64; to f32 needs care: do it exactly
        lo, hi = np.nextafter(want, np.float32(-np.inf)), np.nextafter(want, np.float32(np.inf))
        best = min((abs(Fraction(float(x)) - exact), abs(int(np.float32(x).view(np.uint32)) & 1), float(x)) for x in (lo, want, hi))
        assert float(got[i]) == best[2], (i, a[i], b[i], c[i], got[i], best[2])


def test_model_golden_digests_are_current_and_the_oracle_reproduces_them():
    import make_golden_refmodel as mg
    import test_scenarios_oracle as t

    gold = json.load(open(os.path.join(HERE, "golden", "refmodel_digests.json")))
    assert sorted(gold) == mg.model_cases()
    for name in mg.model_cases():
        assert t.digest(t.CASES[name]()) == gold[name], "oracle differs from the independent model's golden digest: " + name
    # the committed digests really are what the model produces now (spot-checked: the full regeneration is the script)
    for name in ("events_70", "chain_events_37", "master_chain_fx", "graph_inputs"):
        assert t.digest(mg.run_on_model(name)) == gold[name], "refmodel_digests.json is stale: " + name
