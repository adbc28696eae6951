"""The reference's own known-answer tests for this path, ported 1:1 against the ORACLE's restated
graph compiler (graph/graph/compiler/schedule.rs:407-710).  These five tests are the only golden
vectors the reference ships; they pin routing (order, should_clear, buffer sharing, alias freedom,
error variants), not sample values."""
import pytest

import fwapi
from fwapi import DUMMY, AddEdgeError, OracleEngine


def _find(sched, node):
    for s in sched:
        if s["id"] == node:
            return s
    raise AssertionError("node not scheduled")


def verify_node(eng, sched, node, n_in, n_out, in_ports_that_should_clear):
    # schedule.rs:600-635
    s = _find(sched, node)
    assert len(s["in"]) == n_in
    assert len(s["out"]) == n_out
    assert len(in_ports_that_should_clear) == n_in
    for (buf, clear), want in zip(s["in"], in_ports_that_should_clear):
        assert clear == want
    seen = set()
    for buf, _ in s["in"]:
        assert buf not in seen
        seen.add(buf)
    for buf in s["out"]:
        assert buf not in seen
        seen.add(buf)


def verify_edge(sched, src, sp, dst, dp):
    # schedule.rs:637-660
    assert _find(sched, src)["out"][sp] == _find(sched, dst)["in"][dp][0]


def test_simplest_graph_compile():
    # schedule.rs:407-436
    g = OracleEngine(max_block_frames=128, num_graph_inputs=1, num_graph_outputs=1)
    n0, n1 = g.graph_in_node, g.graph_out_node
    g.connect(n0, 0, n1, 0)
    g.update()
    s = g.schedule()
    assert len(s) == 2
    assert g.num_buffers() > 0
    assert s[0]["id"] == n0 and s[1]["id"] == n1
    verify_node(g, s, n0, 0, 1, [])
    verify_node(g, s, n1, 1, 0, [False])
    verify_edge(s, n0, 0, n1, 0)


def test_graph_compile_1():
    # schedule.rs:451-524
    g = OracleEngine(max_block_frames=128, num_graph_inputs=2, num_graph_outputs=2)
    n0 = g.graph_in_node
    n1 = g.add_node(DUMMY, 1, 2)
    n2 = g.add_node(DUMMY, 1, 1)
    n3 = g.add_node(DUMMY, 2, 2)
    n4 = g.add_node(DUMMY, 2, 2)
    n5 = g.add_node(DUMMY, 5, 2)
    n6 = g.graph_out_node
    edges = [(n0, 0, n1, 0), (n0, 1, n2, 0), (n1, 0, n3, 0), (n1, 1, n4, 1), (n3, 0, n5, 0), (n3, 1, n5, 1),
             (n4, 0, n5, 2), (n4, 1, n5, 3), (n2, 0, n5, 4), (n5, 0, n6, 0), (n5, 1, n6, 1)]
    for e in edges:
        g.connect(*e)
    g.update()
    s = g.schedule()
    assert len(s) == 7
    assert g.num_buffers() > 6
    assert s[0]["id"] == n0
    assert {s[1]["id"], s[2]["id"]} == {n1, n2}
    assert {s[3]["id"], s[4]["id"]} == {n3, n4}
    assert s[5]["id"] == n5
    assert s[6]["id"] == n6
    verify_node(g, s, n0, 0, 2, [])
    verify_node(g, s, n1, 1, 2, [False])
    verify_node(g, s, n2, 1, 1, [False])
    verify_node(g, s, n3, 2, 2, [False, True])
    verify_node(g, s, n4, 2, 2, [True, False])
    verify_node(g, s, n5, 5, 2, [False] * 5)
    verify_node(g, s, n6, 2, 0, [False, False])
    for e in edges:
        verify_edge(s, *e)


def test_graph_compile_2():
    # schedule.rs:539-598
    g = OracleEngine(max_block_frames=128, num_graph_inputs=2, num_graph_outputs=2)
    n0 = g.graph_in_node
    n1 = g.add_node(DUMMY, 1, 1)
    n2 = g.add_node(DUMMY, 2, 2)
    n3 = g.add_node(DUMMY, 2, 2)
    n4 = g.add_node(DUMMY, 5, 4)
    n5 = g.graph_out_node
    n6 = g.add_node(DUMMY, 1, 1)
    edges = [(n0, 0, n2, 0), (n0, 0, n3, 1), (n2, 0, n4, 0), (n3, 1, n4, 3), (n1, 0, n4, 4), (n4, 0, n5, 0),
             (n4, 2, n6, 0)]
    for e in edges:
        g.connect(*e)
    g.update()
    s = g.schedule()
    assert len(s) == 7
    assert g.num_buffers() > 7
    assert {s[0]["id"], s[1]["id"]} == {n0, n1}
    assert {s[2]["id"], s[3]["id"]} == {n2, n3}
    assert s[4]["id"] == n4
    assert {s[5]["id"], s[6]["id"]} == {n5, n6}
    for e in edges:
        verify_edge(s, *e)
    verify_node(g, s, n0, 0, 2, [])
    verify_node(g, s, n1, 1, 1, [True])
    verify_node(g, s, n2, 2, 2, [False, True])
    verify_node(g, s, n3, 2, 2, [True, False])
    verify_node(g, s, n4, 5, 4, [False, True, True, False, False])
    verify_node(g, s, n5, 2, 0, [False, True])
    verify_node(g, s, n6, 1, 1, [False])


def test_many_to_one_detection():
    # schedule.rs:662-683
    g = OracleEngine(max_block_frames=128, num_graph_inputs=2, num_graph_outputs=1)
    n1, n2 = g.graph_in_node, g.graph_out_node
    g.connect(n1, 0, n2, 0)
    with pytest.raises(AddEdgeError) as ei:
        g.connect(n1, 1, n2, 0)
    assert ei.value.name == "InputPortAlreadyConnected"


def test_cycle_detection():
    # schedule.rs:685-710
    g = OracleEngine(max_block_frames=128, num_graph_inputs=0, num_graph_outputs=2)
    n1 = g.add_node(DUMMY, 1, 1)
    n2 = g.add_node(DUMMY, 2, 1)
    n3 = g.add_node(DUMMY, 1, 1)
    g.connect(n1, 0, n2, 0)
    g.connect(n2, 0, n3, 0)
    e3 = g.connect(n3, 0, n1, 0)
    assert g.cycle_detected()
    g.disconnect_by_edge_id(e3)
    assert not g.cycle_detected()
    g.connect(n3, 0, n2, 1)
    assert g.cycle_detected()


def test_add_edge_error_variants():
    # graph.rs:407-446
    g = OracleEngine()
    a = g.add_node(DUMMY, 1, 1)
    b = g.add_node(DUMMY, 1, 1)
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


def test_level_order_and_lifo_buffers():
    # Q26: Kahn BFS gives level order; free list is LIFO (compiler.rs:110-130, 252-292)
    g = OracleEngine()
    s0, s1 = g.sampler(), g.sampler()
    v0, v1 = g.volume(50), g.volume(60)
    m = g.sum(2)
    g.connect_stereo(s0, v0)
    g.connect_stereo(s1, v1)
    g.connect_stereo(v0, m, 0)
    g.connect_stereo(v1, m, 2)
    g.connect_stereo(m, g.graph_out_node)
    g.update()
    s = g.schedule()
    ids = [x["id"] for x in s]
    assert ids == [g.graph_in_node, s0, s1, v0, v1, m, g.graph_out_node]
    # samplers take 0..3; v0 acquires 4,5 then frees its inputs (0,1); v1 pops 1 then 0 (LIFO)
    assert _find(s, s0)["out"] == [0, 1] and _find(s, s1)["out"] == [2, 3]
    assert _find(s, v0)["out"] == [4, 5]
    assert _find(s, v1)["out"] == [1, 0]
