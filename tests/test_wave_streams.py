"""Stream-level hazards of the wavefront engine's tick (rc_api.cpp: run_wave2_segment), on an abstract model (no GPU).

tests/test_wave_plan.py checks WHICH tick every step of every row runs at. This file checks that, given those ticks, the way
a tick is issued is race free: kernels are nodes on streams, stream order and the engine's event waits are the only ordering,
the problems of one merged launch run concurrently. Every pair of accesses to one buffer with a write among them must be
ordered by happens-before -- for the two-stream tick (batch >= 48: {rnn6, rnn4 (+ init_net)} on the caller's stream,
{H = 512 nets, linear1} on a third one, per-row kernels + linear2 on the second) and for the one-stream tick with the late
wait. The same model shows what the third copy of the hidden state (RC_HBUF) and the third relu(linear1) buffer are for:
with two copies each the corresponding race appears.

Buffers: per ring slot (16) the frame's inputs / inter-stage vectors / flags; per sub-net relu(linear1) [x1 copies], h[layer]
[hbuf copies] and c[layer]; the init_net chain. A frame started at tick e is step e of every sub-net (all rows visible from
tick 0 on; riders' inputs `xl` are modelled as always present). Row-disjoint accesses are not modelled: the tail's init_net
write into rnn2's state (the plan keeps the row's next rnn2 step behind it, test_wave_plan.py) and the per-row init_net rows."""
import itertools

import numpy as np
import pytest

NETS_A, NETS_B = ("rnn2", "rnn4"), ("rnn6", "rnn3", "rnn7", "rnn8")
G0_NETS = ("rnn6", "rnn4")                       # caller's stream; the others ride the {linear1} launch
RING, TICKS = 16, 44


def frame_ops(e, hbuf, x1buf, init=True):
    """{stage: [(kernel, net or None, reads, writes)]} of the frame started at tick e (stage s runs at tick e + s)."""
    sl = e % RING
    S = lambda name: ("slot", sl, name)
    ops = {s: [] for s in range(9)}
    ops[0].append(("prep", None, [], [S("flags"), S("xA")]))

    def net_stages(nets, s1, xin):
        for n in nets:
            ops[s1].append(("lin1", n, [S(xin), S("flags"), S("xl")], [("x1", n, e % x1buf)]))
            ops[s1 + 1].append(("l0", n, [("x1", n, e % x1buf), S("flags"), ("h", n, 0, (e - 1) % hbuf), ("c", n, 0)],
                                [("h", n, 0, e % hbuf), ("c", n, 0)]))
            ops[s1 + 2].append(("l1", n, [("h", n, 0, e % hbuf), S("flags"), ("h", n, 1, (e - 1) % hbuf), ("c", n, 1)],
                                [("h", n, 1, e % hbuf), ("c", n, 1)]))
    net_stages(NETS_A, 1, "xA")
    net_stages(NETS_B, 5, "xB")
    ops[4].append(("lin2", None, [("h", n, 1, e % hbuf) for n in NETS_A] + [S("flags")], [S("yA")]))
    ops[4].append(("fuse", None, [S("yA"), S("flags")], [S("xB"), S("xi")]))
    if init:                                                              # ticks with init_net problems (a row's one-shot reach)
        ops[5].append(("init0", None, [S("xi"), S("flags")], []))        # (the chain's own buffers are row-disjoint)
    ops[8].append(("lin2", None, [("h", n, 1, e % hbuf) for n in NETS_B] + [S("flags")], [S("yB")]))
    # the tail also posts riders into the slot that starts at ITS tick
    ops[8].append(("tail", None, [S("yB"), S("flags")], [S("out"), ("slot", (e + 8) % RING, "xl"), ("slot", (e + 8) % RING, "flags")]))
    return ops


def build_tri(hbuf, x1buf, init=True):
    """Round 6, contexts on the shared-weight kernel: THREE streams of layer steps -- G4 {rnn4} on the caller's, G6 {rnn6} and G5 {the
    H = 512 nets} on streams of the context -- and L1 {linear1 of every net, init_net} at the head of the second stream's tick, in
    front of that stream's waits for the layer steps of the previous tick (rc_api.cpp: run_wave2_segment, `tri`)."""
    names = ("L1", "prep", "lin2", "fuse", "tail", "G4", "G6", "G5")
    per_tick = {k: {n: [] for n in names} for k in range(TICKS)}
    for e in range(TICKS):
        for s, lst in frame_ops(e, hbuf, x1buf, init).items():
            k = e + s
            if k >= TICKS:
                continue
            for kern, net, rd, wr in lst:
                if kern in ("prep", "lin2", "fuse", "tail"):
                    per_tick[k][kern].append((rd, wr))
                elif kern in ("init0", "lin1"):
                    per_tick[k]["L1"].append((rd, wr))
                else:
                    per_tick[k]["G4" if net == "rnn4" else ("G6" if net == "rnn6" else "G5")].append((rd, wr))
    nodes, idx = [], {}
    for k in range(TICKS):
        for name in names:
            idx[(k, name)] = len(nodes)
            nodes.append((k, name, per_tick[k][name]))
    n = len(nodes)
    hb = np.zeros((n, n), bool)
    edge = lambda a, b: hb.__setitem__((idx[a], idx[b]), True)
    for k in range(TICKS):
        for a, b in (("L1", "prep"), ("prep", "lin2"), ("lin2", "fuse"), ("fuse", "tail")):   # second stream, in order
            edge((k, a), (k, b))
        if k:
            edge((k - 1, "tail"), (k, "L1"))
            for g in ("G4", "G6", "G5"):
                edge((k - 1, g), (k, "lin2"))                                          # linear2 (and fuse, tail behind it) waits for every layer step of the previous tick;
                edge((k - 1, g), (k, g))                                               # prep stands in front of that wait (round 6: it reads none of them). Each stream in order
            edge((k - 1, "L1"), (k, "G4"))
            edge((k - 1, "L1"), (k, "G6"))
            # The H = 512 nets' stream: what it NEEDS is linear1 like the other two, and the END of the second stream's tick only behind an
            # init_net state write of its tail -- the edges modelled here (RC_SEQ_H5_EARLY=1). The default waits for the end on every tick
            # (measured faster): more ordering, never less.
            edge((k - 1, "L1"), (k, "G5"))
            if init:
                edge((k - 1, "tail"), (k, "G5"))
    for m in range(n):
        hb |= np.outer(hb[:, m], hb[m, :])
    return nodes, hb


def build(mode, hbuf, x1buf, init=True, g0_nets=G0_NETS, init_in="G0"):
    """nodes = launches; returns (accesses per node, happens-before matrix)."""
    if mode == "tri":
        return build_tri(hbuf, x1buf, init)
    per_tick = {k: {"prep": [], "lin2": [], "fuse": [], "tail": [], "G0": [], "G2": []} for k in range(TICKS)}
    for e in range(TICKS):
        for s, lst in frame_ops(e, hbuf, x1buf, init).items():
            k = e + s
            if k >= TICKS:
                continue
            for kern, net, rd, wr in lst:
                if kern in ("prep", "lin2", "fuse", "tail"):
                    per_tick[k][kern].append((rd, wr))
                elif kern == "init0":
                    per_tick[k][init_in].append((rd, wr))
                else:
                    per_tick[k]["G0" if (kern in ("l0", "l1") and net in g0_nets) else "G2"].append((rd, wr))
    nodes, idx = [], {}
    for k in range(TICKS):
        for name in ("prep", "lin2", "fuse", "tail", "G0", "G2"):
            idx[(k, name)] = len(nodes)
            nodes.append((k, name, per_tick[k][name]))
    n = len(nodes)
    hb = np.zeros((n, n), bool)
    edge = lambda a, b: hb.__setitem__((idx[a], idx[b]), True)
    for k in range(TICKS):
        for a, b in (("prep", "lin2"), ("lin2", "fuse"), ("fuse", "tail")):           # second stream, in order
            edge((k, a), (k, b))
        if k:
            edge((k - 1, "tail"), (k, "prep"))
            edge((k - 1, "G0"), (k, "prep"))                                           # it waits for the previous tick's wide launches
            edge((k - 1, "G2"), (k, "prep"))
        if mode == "two":
            if k:
                edge((k - 1, "G2"), (k, "G2"))                                         # third stream, in order
                edge((k - 1, "tail"), (k, "G2"))                                       # ... behind the second stream's previous tick
                edge((k - 1, "G0"), (k, "G0"))                                         # caller's stream, in order
                edge((k - 1, "G2"), (k, "G0"))                                         # ... behind the previous linear1
                if init and init_in == "G0":
                    edge((k - 1, "tail"), (k, "G0"))                                   # init_net problems read the previous fuse
        else:                                                                           # one stream: G0 then G2, the wait in front of G2 ...
            edge((k, "G0"), (k, "G2"))
            if k:
                edge((k - 1, "G2"), (k, "G0"))
                edge((k - 1, "tail"), (k, "G0") if init else (k, "G2"))                 # ... or of G0 in a tick with init_net problems
    for m in range(n):                                                                  # transitive closure (nodes are in tick order)
        hb |= np.outer(hb[:, m], hb[m, :])
    return nodes, hb


def races(mode, hbuf=3, x1buf=3, init=True, g0_nets=G0_NETS, init_in="G0"):
    nodes, hb = build(mode, hbuf, x1buf, init, g0_nets, init_in)
    touched = {}
    out = []
    for i, (k, name, probs) in enumerate(nodes):
        for pi, (rd, wr) in enumerate(probs):
            for buf in rd:
                touched.setdefault(buf, []).append((i, pi, False))
            for buf in wr:
                touched.setdefault(buf, []).append((i, pi, True))
    for buf, acc in touched.items():
        for (i, pi, wi), (j, pj, wj) in itertools.combinations(acc, 2):
            if not (wi or wj) or (i == j and pi == pj):
                continue
            if i == j or not (hb[i, j] or hb[j, i]):                                    # two problems of one launch, or unordered launches
                out.append((buf, nodes[i][:2], nodes[j][:2]))
    return out


@pytest.mark.parametrize("mode,init", [("two", True), ("two", False), ("one", True), ("one", False), ("tri", True), ("tri", False)])
def test_a_tick_as_issued_has_no_race(mode, init):
    assert races(mode, init=init) == []


def test_the_third_state_copy_and_the_third_linear1_buffer_are_what_make_it_so():
    r = races("two", hbuf=2, init=False)
    assert r and all(b[0] == "h" for b, _, _ in r)                                      # a wide launch overwrites h that linear2 still reads
    assert any({x[1], y[1]} == {"G0", "lin2"} for _, x, y in r)
    r = races("one", hbuf=2, init=False)                                                # the late wait of the one-stream tick needs it too
    assert r and all(b[0] == "h" for b, _, _ in r)
    r = races("two", x1buf=2)                                                           # {linear1} runs up to a tick ahead of {rnn6, rnn4}
    assert r and all(b[0] == "x1" and b[1] in G0_NETS for b, _, _ in r)
    assert races("one", x1buf=2, init=False) == []                                      # on one stream two buffers were enough


def test_the_regrouped_two_stream_tick_and_the_three_stream_tick_need_the_same_copies():
    """Round 6: {rnn4} alone on the caller's stream with rnn6 + init_net beside the H = 512 nets (RC_SEQ_TRI=0), and the three-stream tick."""
    assert races("two", g0_nets=("rnn4",), init_in="G2") == []
    r = races("tri", hbuf=2, init=False)
    assert r and all(b[0] == "h" for b, _, _ in r)
    r = races("tri", x1buf=2)
    assert r and all(b[0] == "x1" for b, _, _ in r)
