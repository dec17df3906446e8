"""Randomised call sequences through the step-wise C-ABI (the drop-in boundary: CSLAM::SLAM(), SLAM.cpp:87-112 — predictMotion / predictMeasurement /
data association / KalmanUpdate per frame, with the host free to look at the state, change the map or abandon a frame at any point in between).

The fast path of that API (srukf_step.hip) runs the staged replay's launch sequence cut at the association step and keeps a dozen flags about what it has
submitted ahead (the next frame's checkpoint copy, frame scalars, first launch; a projection by the previous tail; an uncommitted motion step ...).  Round 5's net
under it was four hand-picked interleavings.  Here a seeded generator draws ~60 calls per sequence from
    announce the next odometry pair (the right one / a wrong one / not at all; before the predict or between predict and update),
    srukf_predict_motion (also twice in a row: a frame that is predicted and never updated keeps its motion step),
    every state getter at any point (state, robot, landmark block, frame view, covariance),
    srukf_predict_measurement, srukf_associate on a gray frame, srukf_update with random match sets (including none and one),
    srukf_set_state, srukf_delete_landmark, srukf_add_landmarks (+ the NEED_REORDER update behind it), srukf_set_storage
and runs every sequence on TWO filters — the default and one with the fast path switched off ("step_fast" 0: the launch sequences of round 4) — which must agree
after every call that returns numbers (equal visibility, |dX| <= 1e-9, |dP| <= 1e-11 per update; accumulated over a sequence: the bounds scale with the frame count),
and, for the sequences without map or storage changes, on the ORACLE (oracle/srukf_oracle.c) with the same calls."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

N0 = 40                     # n = 244: rank-aware form, "fused tail" mode (plan 21 <= N <= 159)
CALLS = 60


def _P(S):
    return S.T @ S


class _Pair:
    """the two device filters (+ the oracle) behind one set of calls"""

    def __init__(self, srukf, oracle, synth, seed, with_oracle):
        self.srukf, self.O = srukf, oracle
        self.p = synth.scene_params()
        self.rng = np.random.default_rng(1000 + seed)
        self.sc = synth.make_scene(N0, CALLS + 4, seed=100 + seed, p=self.p)
        self.a = srukf.Filter(N0, self.p)
        self.b = srukf.Filter(N0, self.p); self.b.debug_set("step_fast", 0)
        self.fs = (self.a, self.b)
        for f in self.fs:
            f.set_state(self.sc["X0"], self.sc["S0"])
        self.o = None
        if with_oracle:
            self.o = oracle.Oracle(N0, self.p); self.o.set_state(self.sc["X0"], self.sc["S0"])
        self.t = 0               # next odometry pair: (odo[t], odo[t + 1])
        self.phase = 0
        self.updates = 0
        self.need_reorder = 0
        self.h = self.vis = None
        self.log = []
        self.has_app = False
        self.f32_touched = False     # fp32 storage was on at some point: the two filters' states are float roundings of nearly equal values from then on

    def close(self):
        for f in self.fs:
            f.close()
        if self.o:
            self.o.close()

    # tolerances grow with the number of updates behind the two filters (each within 1e-13 / 1e-14 of the other per frame in practice)
    def tx(self):
        return (2e-6 if self.f32_touched else 1e-9) * max(1, self.updates)

    def tp(self):
        return (2e-8 if self.f32_touched else 1e-11) * max(1, self.updates)

    def th(self):
        return (1e-3 if self.f32_touched else 1e-7) * max(1, self.updates)

    def same(self, what, xa, xb, tol):
        d = float(np.abs(np.asarray(xa) - np.asarray(xb)).max()) if np.asarray(xa).size else 0.0
        assert d <= tol, f"{what}: {d:.3e} > {tol:.1e} after {self.log[-12:]}"

    # ---- calls ----
    def announce(self, right):
        nxt = (self.sc["odo"][self.t + 1], self.sc["odo"][self.t + 2]) if self.phase == 0 else (self.sc["odo"][self.t], self.sc["odo"][self.t + 1])
        # (phase 0: the frame about to be predicted is (t, t + 1), the one behind it (t + 1, t + 2); after the predict self.t has advanced already)
        if not right:
            nxt = (nxt[0] + np.array([2e-3, -1e-3, 4e-4]), nxt[1])
        for f in self.fs:
            f.predict_motion_next(*nxt)
        self.log.append("announce_right" if right else "announce_wrong")

    def predict_motion(self):
        o0, o1 = self.sc["odo"][self.t], self.sc["odo"][self.t + 1]
        for f in self.fs:
            f.predict_motion(o0, o1)
        if self.o:
            self.o.predict_motion(o0, o1)
        self.t += 1
        self.phase = 1
        self.log.append("predict_motion")

    def predict_measurement(self):
        (ha, Sia, va), (hb, Sib, vb) = self.a.predict_measurement(), self.b.predict_measurement()
        assert np.array_equal(va, vb), f"visibility differs after {self.log[-12:]}"
        self.same("h", ha, hb, self.th())
        self.same("Si^T Si", np.einsum("kab,kac->kbc", Sia, Sia), np.einsum("kab,kac->kbc", Sib, Sib), self.th())
        if self.o:
            ho, Sio, vo = self.o.predict_measurement()
            assert np.array_equal(va, vo), f"visibility differs from the oracle's after {self.log[-12:]}"
            self.same("h vs oracle", ha, ho, self.th())
        self.h, self.vis = ha, va
        self.phase = 2
        self.log.append("predict_measurement")

    def getter(self, which):
        a, b = self.a, self.b
        if which == 0:
            (Xa, Sa), (Xb, Sb) = a.get_state(), b.get_state()
            self.same("get_state X", Xa, Xb, self.tx()); self.same("get_state P", _P(Sa), _P(Sb), self.tp())
            assert np.all(np.tril(Sa, -1) == 0.0) and np.all(np.tril(Sb, -1) == 0.0)
            if self.o and self.phase == 0:
                Xo, So = self.o.get_state()
                self.same("X vs oracle", Xa, Xo, self.tx()); self.same("P vs oracle", _P(Sa), _P(So), self.tp())
        elif which == 1:
            (pa, Pa), (pb, Pb) = a.get_robot(), b.get_robot()
            self.same("get_robot pose", pa, pb, self.tx()); self.same("get_robot P4", Pa, Pb, self.tp())
        elif which == 2 and a.N > 0:
            k = int(self.rng.integers(a.N))
            (xa, Pa), (xb, Pb) = a.get_landmark_block(k), b.get_landmark_block(k)
            self.same("landmark X6", xa, xb, self.tx()); self.same("landmark P66", Pa, Pb, self.tp())
        elif which == 3:
            va, vb = a.get_frame_view(), b.get_frame_view()
            self.same("view X", va[0], vb[0], self.tx()); self.same("view xyz", va[1], vb[1], 100 * self.tx()); self.same("view cov", va[2], vb[2], 1000 * self.tp())
            self.same("view pose", va[3], vb[3], self.tx()); self.same("view P4", va[4], vb[4], self.tp())
        elif which == 4:
            self.same("get_covariance", a.get_covariance(), b.get_covariance(), self.tp())
        self.log.append(f"get{which}")

    def associate(self):
        """srukf_associate between predict_measurement and update: commits the motion step on demand, uses the accessors' staging buffer and the export flag.  Its
        matches depend on correlation thresholds: the two filters' outputs are compared where both matched; the update that follows takes its z from the test."""
        if not self.has_app:
            for f in self.fs:
                for k in range(min(f.N, 6)):
                    f.set_landmark_appearance(k, self.rng_patch[k % 6], np.eye(3), np.zeros(3), np.array([320.0 + 7 * k, 240.0 - 5 * k]))
            self.has_app = True
        (za, ma, ca), (zb, mb, cb) = self.a.associate(self.gray), self.b.associate(self.gray)
        both = (ma != 0) & (mb != 0)
        if both.any():
            idx = np.repeat(both, 2)
            self.same("associate z", za[idx], zb[idx], 1e-6)
        self.log.append("associate")

    def update(self):
        N = self.a.N
        style = self.rng.integers(8)
        m = np.asarray(self.vis, dtype=np.int32).copy()
        if style == 0:
            m[:] = 0                                              # no match: the frame ends with its motion step (SLAM.cpp:2050-2051)
        elif style == 1 and m.sum() > 0:
            keep = self.rng.choice(np.flatnonzero(m)); m[:] = 0; m[keep] = 1      # a single match
        elif style < 5:
            m = m * (self.rng.random(N) < 0.7)
        z = self.h + self.rng.normal(0.0, 0.5, self.h.shape)
        reorder = self.srukf.NEED_REORDER if self.need_reorder else self.srukf.NEEDNOT_REORDER
        if reorder == self.srukf.NEED_REORDER and m.sum() == 0:
            m = np.asarray(self.vis, dtype=np.int32).copy()     # (the frame behind an addition matches: that is why the reference added)
        for f in self.fs:
            f.update(z, m.astype(np.int32), reorder=reorder)
        if self.o:
            self.o.update(z, m.astype(np.int32), mode=self.O.Oracle.BATCHED)
        self.need_reorder = 0
        self.phase = 0
        if m.sum() > 0:
            self.updates += 1
        self.log.append(f"update[{int(m.sum())}{'R' if reorder == self.srukf.NEED_REORDER else ''}]")
        (Xa, Sa), (Xb, Sb) = self.a.get_state(), self.b.get_state()
        self.same("X after update", Xa, Xb, self.tx()); self.same("P after update", _P(Sa), _P(Sb), self.tp())
        if self.o:
            Xo, So = self.o.get_state()
            self.same("X vs oracle after update", Xa, Xo, self.tx()); self.same("P vs oracle after update", _P(Sa), _P(So), self.tp())

    def set_state_roundtrip(self):
        for f in self.fs:
            X, S = f.get_state(); f.set_state(X, S)
        self.phase = 0
        self.log.append("set_state")

    def delete(self):
        k = int(self.rng.integers(self.a.N))
        for f in self.fs:
            f.delete_landmark(k)
        self.phase = 0; self.has_app = False
        self.log.append(f"delete[{k}]")

    def add(self):
        K = int(self.rng.integers(1, 4))
        uv = np.column_stack([self.rng.uniform(60, 580, K), self.rng.uniform(60, 420, K)])
        for f in self.fs:
            f.add_landmarks(uv)
        self.phase = 0; self.need_reorder = K
        self.log.append(f"add[{K}]")

    def storage(self):
        st = self.srukf.STORAGE_F32 if self.rng.integers(2) else self.srukf.STORAGE_F64
        for f in self.fs:
            f.set_storage(st)
        self.f32_touched = self.f32_touched or st == self.srukf.STORAGE_F32
        self.phase = 0
        self.log.append(f"storage[{st}]")


def _run(srukf, oracle, synth, seed, changes, storage=False):
    q = _Pair(srukf, oracle, synth, seed, with_oracle=not changes)
    rng = q.rng
    q.rng_patch = rng.integers(0, 255, size=(6, 21, 21), dtype=np.uint8)
    q.gray = rng.integers(0, 255, size=(480, 640), dtype=np.uint8)
    try:
        for _ in range(CALLS):
            r = rng.random()
            if q.a.N < 24 and changes:                           # keep the map inside the rank-aware sizes
                q.add(); continue
            if r < 0.22:
                q.getter(int(rng.integers(5)))
            elif r < 0.30:
                q.announce(bool(rng.random() < 0.7))
            elif q.phase == 0:
                if changes and r < 0.36 and not q.need_reorder:
                    q.delete() if rng.random() < 0.6 and q.a.N > 26 else q.add()
                elif storage and r < 0.39 and not q.need_reorder:
                    q.storage()
                elif r < 0.43 and not q.need_reorder:
                    q.set_state_roundtrip()
                else:
                    q.predict_motion()
            elif q.phase == 1:
                if r < 0.36:
                    q.predict_motion()                           # the frame is abandoned: its motion step stands, the next pair follows
                else:
                    q.predict_measurement()
            else:
                if r < 0.40 and changes:
                    q.associate()
                q.update()
        fa, sa, fb = q.a.debug_get("step_fast"), q.a.debug_get("step_slow"), q.b.debug_get("step_fast")
        assert fb == 0
        return fa, sa, q.updates, q.log
    finally:
        q.close()


@pytest.mark.parametrize("seed", range(12))
def test_random_call_sequences_without_map_changes_against_the_oracle(srukf, oracle, synth, seed):
    fa, sa, ups, log = _run(srukf, oracle, synth, seed, changes=False)
    assert ups >= 3, log
    assert fa >= 1, (fa, sa, log)                              # the fast path was taken (a state from outside starts on the other one)


@pytest.mark.parametrize("seed", range(12, 20))
def test_random_call_sequences_with_map_changes(srukf, oracle, synth, seed):
    fa, sa, ups, log = _run(srukf, oracle, synth, seed, changes=True)
    assert ups >= 2, log


@pytest.mark.parametrize("seed", range(20, 24))
def test_random_call_sequences_with_map_and_storage_changes(srukf, oracle, synth, seed):
    fa, sa, ups, log = _run(srukf, oracle, synth, seed, changes=True, storage=True)
    assert ups >= 2, log
