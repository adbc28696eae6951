"""CPU tier, host logic of the PRODUCT: firewheel_amd/csrc/fwgpu_graph.cpp (graph mirror + launch planner, plain C++)
built with g++ behind tests/planner_harness.  (1) the reference's five routing KATs + the AddEdgeError variants
(graph/graph/compiler/schedule.rs:407-710, graph.rs:407-446) on the product's planner; (2) seeded random graphs and
edit sequences cross-checked against the oracle's restated compiler: same schedule order, same should_clear sets, same
errors; every edge routed, levels consistent, no buffer written twice."""
import random

import pytest

import fwapi
from fwapi import DUMMY, SUM, AddEdgeError, CompileGraphError, OracleEngine, PlannerEngine


def _find(sched, node):
    for s in sched:
        if s["id"] == node:
            return s
    raise AssertionError("node not scheduled")


def verify_node(sched, node, n_in, n_out, should_clear):
    # schedule.rs:600-635, for a planner whose unconnected inputs all read the constant zero buffer (id 0)
    s = _find(sched, node)
    assert len(s["in"]) == n_in and len(s["out"]) == n_out
    assert [c for _, c in s["in"]] == list(should_clear)
    seen = set()
    for buf, clear in s["in"]:
        if clear:
            assert buf == 0
            continue
        assert buf != 0 and buf not in seen
        seen.add(buf)
    for buf in s["out"]:
        assert buf != 0 and buf not in seen
        seen.add(buf)


def reference_order(g):
    """the KATs below restate the reference's own tests of ITS schedule order (schedule.rs:407-710): they run on the walk that
    reproduces it; the order build_plan writes the plan's tables in by default (round 6: by where a node is connected) is checked by
    the random-graph family and the voice-replacement test further down"""
    g.set_canonical_order(False)
    return g


def verify_edge(sched, src, sp, dst, dp):
    assert _find(sched, src)["out"][sp] == _find(sched, dst)["in"][dp][0]  # schedule.rs:637-660


def test_simplest_graph_compile():
    g = reference_order(PlannerEngine(num_graph_inputs=1, num_graph_outputs=1))
    n0, n1 = g.graph_in_node, g.graph_out_node
    g.connect(n0, 0, n1, 0)
    g.update()
    s = g.schedule()
    assert [x["id"] for x in s] == [n0, n1] and g.num_buffers() > 0
    verify_node(s, n0, 0, 1, [])
    verify_node(s, n1, 1, 0, [False])
    verify_edge(s, n0, 0, n1, 0)


def test_graph_compile_1():
    g = reference_order(PlannerEngine(num_graph_inputs=2, num_graph_outputs=2))
    n0 = g.graph_in_node
    n1, n2, n3, n4, n5 = (g.add_node(DUMMY, *p) for p in ((1, 2), (1, 1), (2, 2), (2, 2), (5, 2)))
    n6 = g.graph_out_node
    edges = [(n0, 0, n1, 0), (n0, 1, n2, 0), (n1, 0, n3, 0), (n1, 1, n4, 1), (n3, 0, n5, 0), (n3, 1, n5, 1),
             (n4, 0, n5, 2), (n4, 1, n5, 3), (n2, 0, n5, 4), (n5, 0, n6, 0), (n5, 1, n6, 1)]
    for e in edges:
        g.connect(*e)
    g.update()
    s = g.schedule()
    ids = [x["id"] for x in s]
    assert len(s) == 7 and g.num_buffers() > 6
    assert ids[0] == n0 and set(ids[1:3]) == {n1, n2} and set(ids[3:5]) == {n3, n4} and ids[5:] == [n5, n6]
    for n, a, b, clr in ((n0, 0, 2, []), (n1, 1, 2, [False]), (n2, 1, 1, [False]), (n3, 2, 2, [False, True]),
                         (n4, 2, 2, [True, False]), (n5, 5, 2, [False] * 5), (n6, 2, 0, [False, False])):
        verify_node(s, n, a, b, clr)
    for e in edges:
        verify_edge(s, *e)


def test_graph_compile_2():
    g = reference_order(PlannerEngine(num_graph_inputs=2, num_graph_outputs=2))
    n0 = g.graph_in_node
    n1, n2, n3, n4 = (g.add_node(DUMMY, *p) for p in ((1, 1), (2, 2), (2, 2), (5, 4)))
    n5 = g.graph_out_node
    n6 = g.add_node(DUMMY, 1, 1)
    edges = [(n0, 0, n2, 0), (n0, 0, n3, 1), (n2, 0, n4, 0), (n3, 1, n4, 3), (n1, 0, n4, 4), (n4, 0, n5, 0), (n4, 2, n6, 0)]
    for e in edges:
        g.connect(*e)
    g.update()
    s = g.schedule()
    ids = [x["id"] for x in s]
    assert len(s) == 7 and g.num_buffers() > 7
    # the reference's BFS yields {n5, n6} last in either order; the product always closes the schedule with graph_out
    assert set(ids[0:2]) == {n0, n1} and set(ids[2:4]) == {n2, n3} and ids[4] == n4 and set(ids[5:]) == {n5, n6}
    for e in edges:
        verify_edge(s, *e)
    for n, a, b, clr in ((n0, 0, 2, []), (n1, 1, 1, [True]), (n2, 2, 2, [False, True]), (n3, 2, 2, [True, False]),
                         (n4, 5, 4, [False, True, True, False, False]), (n5, 2, 0, [False, True]), (n6, 1, 1, [False])):
        verify_node(s, n, a, b, clr)


def test_many_to_one_detection():
    g = reference_order(PlannerEngine(num_graph_inputs=2, num_graph_outputs=1))
    g.connect(g.graph_in_node, 0, g.graph_out_node, 0)
    with pytest.raises(AddEdgeError) as ei:
        g.connect(g.graph_in_node, 1, g.graph_out_node, 0)
    assert ei.value.name == "InputPortAlreadyConnected"


def test_cycle_detection():
    g = reference_order(PlannerEngine(num_graph_inputs=0, num_graph_outputs=2))
    n1, n2, n3 = g.add_node(DUMMY, 1, 1), g.add_node(DUMMY, 2, 1), g.add_node(DUMMY, 1, 1)
    g.connect(n1, 0, n2, 0)
    g.connect(n2, 0, n3, 0)
    e3 = g.connect(n3, 0, n1, 0)
    assert g.cycle_detected()
    with pytest.raises(CompileGraphError) as ei:
        g.update()
    assert ei.value.name == "CycleDetected"
    g.disconnect_by_edge_id(e3)
    assert not g.cycle_detected()
    g.connect(n3, 0, n2, 1)
    assert g.cycle_detected()


def test_add_edge_error_variants():
    g = reference_order(PlannerEngine())
    a, b = g.add_node(DUMMY, 1, 1), g.add_node(DUMMY, 1, 1)
    for args, name in [((a, 1, b, 0), "OutPortOutOfRange"), ((a, 0, b, 1), "InPortOutOfRange"),
                       ((a, 0, a, 0), "CycleDetected"), ((12345 << 32 | 99, 0, b, 0), "SrcNodeNotFound"),
                       ((a, 0, 12345 << 32 | 99, 0), "DstNodeNotFound")]:
        with pytest.raises(AddEdgeError) as ei:
            g.connect(*args)
        assert ei.value.name == name
    g.connect(a, 0, b, 0)
    with pytest.raises(AddEdgeError) as ei:
        g.connect(a, 0, b, 0)
    assert ei.value.name == "EdgeAlreadyExists"
    with pytest.raises(AddEdgeError) as ei:
        g.connect(b, 0, a, 0, check_for_cycles=True)
    assert ei.value.name == "CycleDetected"


def test_rejected_checked_connect_is_rolled_back_completely_unlike_the_reference():
    # graph.rs:466-471 removes the edge but leaves `existing_edges` / `connected_input_ports` populated: in the reference
    # (restated by the oracle) the input port is wedged afterwards.  The product rolls the edit back.
    def setup(g):
        a, b, c = g.add_node(DUMMY, 1, 1), g.add_node(DUMMY, 1, 1), g.add_node(DUMMY, 1, 1)
        g.connect(a, 0, b, 0)
        with pytest.raises(AddEdgeError) as ei:
            g.connect(b, 0, a, 0, check_for_cycles=True)
        assert ei.value.name == "CycleDetected"
        assert not g.cycle_detected()
        return a, b, c

    g = reference_order(PlannerEngine())
    a, b, c = setup(g)
    g.connect(c, 0, a, 0)  # the port is free again
    g.update()
    o = OracleEngine()
    a, b, c = setup(o)
    with pytest.raises(AddEdgeError) as ei:
        o.connect(c, 0, a, 0)
    assert ei.value.name == "InputPortAlreadyConnected"


# ---------------------------------------------------------------- random graphs / edit sequences vs the oracle's compiler
class Pair(object):
    """the same edit applied to the product's planner and to the oracle; handles are kept side by side"""

    def __init__(self, gin, gout):
        self.p = PlannerEngine(num_graph_inputs=gin, num_graph_outputs=gout)
        self.o = OracleEngine(max_block_frames=64, num_graph_inputs=gin, num_graph_outputs=gout)
        self.nodes = [(self.p.graph_in_node, self.o.graph_in_node, 0, gin), (self.p.graph_out_node, self.o.graph_out_node, gout, 0)]
        self.edges = {}  # (src index, sp, dst index, dp) -> (planner edge id, oracle edge id)

    def add(self, kind, n_in, n_out):
        self.nodes.append((self.p.add_node(kind, n_in, n_out), self.o.add_node(kind, n_in, n_out), n_in, n_out))
        # same NodeID (slot + generation) as the reference's thunderdome arena hands out, through removals and slot reuse:
        # what lets a binding pass the reference's ids straight to fwgpu_schedule_upload
        assert self.nodes[-1][0] == self.nodes[-1][1]
        return len(self.nodes) - 1

    def both(self, fp, fo):
        """run the same call on both engines; both succeed with a result, or both fail with the same error name"""
        res = []
        for f in (fp, fo):
            try:
                res.append(("ok", f()))
            except (AddEdgeError, CompileGraphError) as e:
                res.append(("err", e.name))
        assert res[0][0] == res[1][0], res
        if res[0][0] == "err":
            assert res[0][1] == res[1][1], res
            return None
        return res[0][1], res[1][1]

    def connect(self, si, sp, di, dp, check):
        # check_for_cycles is exercised as connect + cycle_detected + clean disconnect: a REJECTED checked connect leaves
        # the reference's bookkeeping maps populated (graph.rs:466-471 — the port stays "connected" for good), which the
        # oracle restates and the product deliberately does not (pinned by the test below)
        r = self.both(lambda: self.p.connect(self.nodes[si][0], sp, self.nodes[di][0], dp, False),
                      lambda: self.o.connect(self.nodes[si][1], sp, self.nodes[di][1], dp, False))
        if r is None:
            return None
        if check:
            cyc = self.p.cycle_detected()
            assert cyc == self.o.cycle_detected()
            if cyc:
                assert self.p.disconnect_by_edge_id(r[0]) == 1 and self.o.disconnect_by_edge_id(r[1]) == 1
                return None
        assert r[0] == r[1]  # ... and the same EdgeID
        self.edges[(si, sp, di, dp)] = r
        return r

    def compare_schedules(self):
        assert self.p.cycle_detected() == self.o.cycle_detected()
        self.p.set_canonical_order(False)      # the walk itself: the reference's order
        r = self.both(self.p.update, self.o.update)
        if r is None:
            # both refused (a cycle, a node that cannot be activated): the canonical order's own walk — it does not run the Kahn walk —
            # must refuse with the same error
            self.p.set_canonical_order(True)
            assert self.both(self.p.update, self.o.update) is None
            return False
        self.check(self.p.schedule(), self.o.schedule(), kahn=True)
        self.p.set_canonical_order(True)       # ... and the order the plan's tables are written in (round 6, the default)
        self.p.update()
        self.check(self.p.schedule(), self.o.schedule(), kahn=False)
        return True

    def check(self, sp, so, kahn):
        pidx = {n[0]: i for i, n in enumerate(self.nodes) if n is not None}
        oidx = {n[1]: i for i, n in enumerate(self.nodes) if n is not None}
        order_p = [pidx[x["id"]] for x in sp]
        order_o = [oidx[x["id"]] for x in so]
        assert order_p[-1] == 1 and sorted(order_p) == sorted(order_o)  # graph_out closes the product's schedule
        if kahn:  # same order as the reference's Kahn BFS, except that graph_out is moved to the end
            assert [i for i in order_p if i != 1] == [i for i in order_o if i != 1]
        else:     # a schedule: graph_in first, every node behind all of its producers
            assert order_p[0] == 0
            at = {i: k for k, i in enumerate(order_p)}
            for (si, _, di, _) in self.edges:
                assert at[si] < at[di]
        by_p = {pidx[x["id"]]: x for x in sp}
        by_o = {oidx[x["id"]]: x for x in so}
        written = set()
        for i, x in by_p.items():
            assert [c for _, c in x["in"]] == [c for _, c in by_o[i]["in"]], "should_clear differs"
            for b in x["out"]:
                assert b != 0 and b not in written  # one buffer per output port, never the zero buffer
                written.add(b)
        for (si, spt, di, dpt) in self.edges:
            assert by_p[si]["out"][spt] == by_p[di]["in"][dpt][0]
            assert by_p[di]["level"] >= by_p[si]["level"] + 1
        for i, x in by_p.items():  # a level is exactly one more than the deepest producer (graph_out: at least)
            srcs = [by_p[si]["level"] for (si, _, di, _) in self.edges if di == i]
            want = 1 + max(srcs) if srcs else 0
            assert x["level"] == want or (i == 1 and x["level"] >= want)
        assert self.p.num_levels() == 1 + max(x["level"] for x in sp)


@pytest.mark.parametrize("seed", range(300))
def test_random_graphs_and_edit_sequences_match_the_oracle_compiler(seed):
    rng = random.Random(0xF1EE + seed)
    g = Pair(rng.randint(0, 3), rng.randint(1, 3))
    for _ in range(rng.randint(2, 14)):
        if rng.random() < 0.3:
            ch = rng.choice([1, 2])
            g.add(SUM, ch * rng.randint(2, 5), ch)
        else:
            g.add(DUMMY, rng.randint(0, 4), rng.randint(0, 4))
    compiled = 0
    for step in range(rng.randint(10, 60)):
        alive = [i for i, n in enumerate(g.nodes) if n is not None]
        r = rng.random()
        if r < 0.62:  # connect (mostly forward edges so that acyclic graphs are common; some bad ports, some cycles)
            si, di = rng.choice(alive), rng.choice(alive)
            if rng.random() < 0.8 and si > di and di != 0 and si != 1:
                si, di = di, si
            n_out, n_in = g.nodes[si][3], g.nodes[di][2]
            sp = rng.randrange(n_out + 1) if rng.random() < 0.05 else rng.randrange(max(n_out, 1))
            dp = rng.randrange(n_in + 1) if rng.random() < 0.05 else rng.randrange(max(n_in, 1))
            g.connect(si, sp, di, dp, rng.random() < 0.5)
        elif r < 0.74 and g.edges:  # disconnect by ports or by edge id
            key = rng.choice(sorted(g.edges))
            pe, oe = g.edges.pop(key)
            if rng.random() < 0.5:
                assert g.p.disconnect_by_edge_id(pe) == 1 and g.o.disconnect_by_edge_id(oe) == 1  # graph.rs: true = removed
                assert g.p.disconnect_by_edge_id(pe) == 0 and g.o.disconnect_by_edge_id(oe) == 0  # already gone
            else:
                si, sp, di, dp = key
                assert g.p.disconnect(g.nodes[si][0], sp, g.nodes[di][0], dp) == 1
                assert g.o.disconnect(g.nodes[si][1], sp, g.nodes[di][1], dp) == 1
        elif r < 0.80 and len(alive) > 2:  # remove a node (never graph_in / graph_out), its edges go with it
            i = rng.choice([a for a in alive if a > 1])
            assert g.p.remove_node(g.nodes[i][0]) == 0
            g.o.remove_node(g.nodes[i][1])
            g.nodes[i] = None
            g.edges = {k: v for k, v in g.edges.items() if k[0] != i and k[2] != i}
        elif r < 0.86:
            g.add(DUMMY, rng.randint(1, 3), rng.randint(1, 3))
        else:
            compiled += g.compare_schedules()
    compiled += g.compare_schedules()
    # removing graph_in / graph_out is refused by both
    assert g.p.remove_node(g.nodes[0][0]) != 0 and g.p.remove_node(g.nodes[1][0]) != 0


def test_replacing_a_voice_moves_only_that_voices_entries_in_the_plan_tables():
    """Round 6: build_plan writes the plan's tables in an order that depends on where a node is CONNECTED — the order
    in which the levels above, walked from graph_out down with input ports ascending, name it — not on the edge arena (the reference's
    Kahn walk: a voice whose edges were made last is scheduled last in its level) and not on graph slots (a replacement's slots
    differ).  A voice put into the mixer port of the voice it replaces takes that voice's positions and buffer ids; nothing else moves:
    a one-voice edit of a large graph then differs in O(1) 4 KiB chunks of the uploaded tables (fwgpu_plan_install.cpp up())."""
    SAMPLER, VOLUME, SUM = 3, 1, 4
    g = PlannerEngine()

    def voice():
        s = g.add_node(SAMPLER, 0, 2)
        a = g.add_node(VOLUME, 2, 2)
        b = g.add_node(VOLUME, 2, 2)
        for p in (0, 1):
            g.connect(s, p, a, p)
            g.connect(a, p, b, p)
        return [s, a, b]

    voices = [voice() for _ in range(64)]
    leaves = [g.add_node(SUM, 32, 2) for _ in range(4)]
    top = g.add_node(SUM, 8, 2)
    for v, vc in enumerate(voices):
        for p in (0, 1):
            g.connect(vc[2], p, leaves[v // 16], 2 * (v % 16) + p)
    for i, m in enumerate(leaves):
        for p in (0, 1):
            g.connect(m, p, top, 2 * i + p)
    for p in (0, 1):
        g.connect(top, p, g.graph_out_node, p)
    g.update()
    before = g.schedule()
    pos = {x["id"]: i for i, x in enumerate(before)}
    # voices sit in mixer-port order inside their levels, whatever order they were made and connected in
    assert [pos[vc[0]] for vc in voices] == sorted(pos[vc[0]] for vc in voices)
    for rnd, v in enumerate((37, 5, 37, 63, 0)):
        old = voices[v]
        spare = g.add_node(VOLUME, 2, 2)      # (other slots come and go meanwhile: the replacement's slots are not the old ones)
        for n in old:
            g.remove_node(n)
        new = voice()
        g.remove_node(spare)
        for p in (0, 1):
            g.connect(new[2], p, leaves[v // 16], 2 * (v % 16) + p)
        voices[v] = new
        g.update()
        after = g.schedule()
        assert len(after) == len(before)
        swapped = dict(zip(old, new))
        for i, (x, y) in enumerate(zip(before, after)):
            assert y["id"] == swapped.get(x["id"], x["id"]), (rnd, i)          # same position: the same node, or the replacement of the node that sat there
            assert (y["in"], y["out"], y["level"]) == (x["in"], x["out"], x["level"]), (rnd, i)   # same buffers, same level
        before = after


def test_canonical_order_finds_cycles_wherever_they_sit():
    """build_plan's canonical walk is its own cycle check (round 6: the reference's Kahn walk is not run in that mode): cycles behind
    graph_out, cycles in a part of the graph graph_out does not reach, a two-node loop, a long ring — CycleDetected every time, and the
    same graph compiles once the closing edge is gone"""
    VOLUME, SUM = 1, 4
    for where in ("reachable", "island", "pair", "ring"):
        for canonical in (True, False):
            g = PlannerEngine()
            g.set_canonical_order(canonical)
            a, b, c, d = (g.add_node(VOLUME, 2, 2) for _ in range(4))
            m = g.add_node(SUM, 4, 2)
            g.connect(a, 0, b, 0)
            g.connect(b, 0, m, 0)
            g.connect(m, 0, g.graph_out_node, 0)
            if where == "reachable":
                g.connect(m, 1, a, 1)          # m -> a -> b -> m
                closing = (m, 1, a, 1)
            elif where == "island":
                g.connect(c, 0, d, 0)
                g.connect(d, 0, c, 0)          # c <-> d, nobody listens
                closing = (d, 0, c, 0)
            elif where == "pair":
                g.connect(b, 1, a, 1)          # a <-> b in front of the mixer
                closing = (b, 1, a, 1)
            else:
                ring = [g.add_node(VOLUME, 2, 2) for _ in range(50)]
                for x, y in zip(ring, ring[1:]):
                    g.connect(x, 0, y, 0)
                g.connect(ring[-1], 0, ring[0], 0)
                g.connect(ring[7], 1, m, 2)
                closing = (ring[-1], 0, ring[0], 0)
            assert g.cycle_detected()
            with pytest.raises(CompileGraphError) as ei:
                g.update()
            assert ei.value.name == "CycleDetected", (where, canonical, ei.value.name)
            assert g.disconnect(*closing) == 1
            g.update()
            ids = [x["id"] for x in g.schedule()]
            assert len(ids) == len(set(ids)) and ids[-1] == g.graph_out_node
