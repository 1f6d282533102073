"""Frame-stepped launch planner of rc_sequence (host logic of librobustcap_hip.so, no GPU): which frames need the three
transition launches. Reference semantics: the vision updater fires at c <= lo (net/sig_mp.py:264-271); its two sub-net
steps are deferred to the next frame and need their own launches only when the row steps on camera keypoints there
(c > lo, or first_frame: L149-156). (The wavefront engine's planner is tests/test_wave_plan.py.)"""
import ctypes as C

import numpy as np

from robustcap_amd import _lib

STEP_TR, STEP = 0, 1


def plan(codes, pend=None, first_frame=False, vis=True):
    lib = _lib.load()
    codes = np.ascontiguousarray(np.asarray(codes, np.int8))
    T, B = codes.shape
    pd = np.ascontiguousarray(np.zeros(B, np.int32) if pend is None else np.asarray(pend, np.int32))
    out = np.full(T, 255, np.uint8)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    rc = lib.rc_plan_sequence(p(codes), B, T, p(pd), 1 if first_frame else 0, int(vis), p(out))
    assert rc == 0
    return out.tolist()


def test_visible_rows_never_need_transition_launches():
    assert plan(np.full((12, 3), 2)) == [STEP] * 12
    assert plan(np.full((12, 3), 1), first_frame=True) == [STEP] * 12


def test_occlusion_marks_the_frame_after_it():
    codes = np.full((16, 2), 2)
    codes[5:8, 0] = 0                                                    # row 0 occluded on frames 5..7
    # frame 8: row 0 carries a pending updater step INTO a visible frame -> transition launches
    assert plan(codes) == [STEP] * 8 + [STEP_TR] + [STEP] * 7
    assert plan(codes, vis=False) == [STEP] * 16                         # no updater -> nothing pends
    # pending state carried in from the previous call
    assert plan(np.full((6, 2), 2), pend=[1, 0]) == [STEP_TR] + [STEP] * 5
    # an occluded row with a pending step and no camera step: merged launch, no transition
    codes = np.zeros((3, 1))
    assert plan(codes, pend=[1]) == [STEP, STEP, STEP]
    assert plan(codes, pend=[1], first_frame=True) == [STEP_TR, STEP, STEP]      # first_frame steps rnn4 anyway
    assert plan(np.zeros((0, 1))) == []


def test_random_plans_respect_the_invariants():
    rng = np.random.default_rng(7)
    for _ in range(50):
        T, B = int(rng.integers(1, 60)), int(rng.integers(1, 6))
        codes = rng.choice([0, 1, 2], size=(T, B), p=[0.15, 0.25, 0.6])
        pd = rng.integers(0, 2, B)
        got = plan(codes, pend=pd)
        pend = pd.copy()
        for t in range(T):
            need_tr = bool(((pend == 1) & (codes[t] >= 1)).any())
            assert got[t] == (STEP_TR if need_tr else STEP)
            pend = (codes[t] == 0).astype(int)
