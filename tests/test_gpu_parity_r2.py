"""GPU parity tests added in round 2 (-m gpu): the cases the round-1 review found untested —
weight types 1 / 2 (calculateSampleParameter, SLAM.cpp:1077-1103), a whole frame with the reference's SHIPPED
process-noise constants a1..a4 = 8 (SLAM.cpp:195-198) where the theta clamp of the modified Cholesky fires, and whole
frames against the oracle at the benchmark sizes N = 200 and N = 500 (fp64 and fp32 storage)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _step_both(f, o, sc, t, mode, srukf):
    f.predict_motion(sc["odo"][t], sc["odo"][t + 1]); o.predict_motion(sc["odo"][t], sc["odo"][t + 1])
    h, Si, vis = f.predict_measurement(); ho, Sio, viso = o.predict_measurement()
    f.update(sc["z"][t], sc["matched"][t], mode=mode); o.update(sc["z"][t], sc["matched"][t], 1, 0, mode)
    X, S = f.get_state(); Xo, So = o.get_state()
    return (h, Si, vis), (ho, Sio, viso), (X, S.T @ S), (Xo, So.T @ So)


@pytest.mark.parametrize("weight_type,N,F,mode", [(1, 8, 4, 0), (1, 8, 4, 1), (1, 20, 3, 1), (2, 8, 4, 0), (2, 8, 4, 1), (2, 20, 3, 1)])
def test_weight_types_against_oracle(srukf, oracle, synth, weight_type, N, F, mode):
    """FLAG_4_WEIGHT2 (type 1: Julier-2000, wm0 = 1 - 1/alpha^2 = -999 999, wc0 = wm0 + 3 - alpha^2) and FLAG_4_WEIGHT3
    (type 2: wm0 = wc0 = 1/3), SEQUENTIAL (the reference's structure) and BATCHED, against the oracle.

    Type 1 is the only one with wc0 != wm0: calculateOneFeatureCrossCovariance (SLAM.cpp:2030-2036) then depends on
    the running state through the centre column, which the device adds back in k_gain_center.  Its weights cancel six
    digits in every weighted mean (h = wm0 Z0 + wi sum Z_c with |wm0 Z0| ~ 3e8), in the reference as in the oracle, so
    its tolerance is 1e6 * the fp64 tolerance of the other types: the oracle's own h carries ~5e-7 px of rounding."""
    p = synth.scene_params(); p["weight_type"] = weight_type
    sc = synth.make_scene(N, F, seed=200 + N, p=p)
    f = srukf.Filter(N, p); f.set_state(sc["X0"], sc["S0"])
    o = oracle.Oracle(N, p); o.set_state(sc["X0"], sc["S0"])
    tol_h, tol_x, tol_p = (5e-6, 1e-7, 5e-9) if weight_type == 1 else (1e-8, 1e-9, 1e-11)
    for t in range(F):
        (h, Si, vis), (ho, Sio, viso), (X, P), (Xo, Po) = _step_both(f, o, sc, t, mode, srukf)
        assert np.array_equal(vis, viso)
        np.testing.assert_allclose(h, ho, rtol=0, atol=tol_h)
        np.testing.assert_allclose(np.abs(Si), np.abs(Sio), rtol=0, atol=tol_h)
        np.testing.assert_allclose(X, Xo, rtol=0, atol=tol_x)
        np.testing.assert_allclose(P, Po, rtol=0, atol=tol_p)


def test_weight_type1_centre_term_matters(srukf, oracle, synth):
    """The running-state centre term of type 1 is not a rounding-level effect: the device agrees with the oracle
    (which recentres on the running m_X_k like the reference, SLAM.cpp:2030) far better than a cross covariance centred
    on the state at the start of KalmanUpdate would.  The gap is measured with the oracle's test knob frozen_center."""
    p = synth.scene_params(); p["weight_type"] = 1
    N = 12
    sc = synth.make_scene(N, 2, seed=212, p=p)
    Xs = []
    for frozen in (0, 1):
        o = oracle.Oracle(N, p); o.set_state(sc["X0"], sc["S0"]); o.set_frozen_center(frozen)
        for t in range(2):
            o.predict_motion(sc["odo"][t], sc["odo"][t + 1]); o.predict_measurement(); o.update(sc["z"][t], sc["matched"][t], 1, 0, 1)
        Xs.append(o.get_state()[0])
    f = srukf.Filter(N, p); f.set_state(sc["X0"], sc["S0"])
    for t in range(2):
        f.predict_motion(sc["odo"][t], sc["odo"][t + 1]); f.predict_measurement(); f.update(sc["z"][t], sc["matched"][t])
    gap = np.abs(Xs[0] - Xs[1]).max()                                    # what the running-state term changes in X
    err = np.abs(Xs[0] - f.get_state()[0]).max()                         # device vs the reference-structured oracle
    print(f"centre term: gap {gap:.3e}, device error {err:.3e}")
    assert gap > 20 * err, (gap, err)


def test_default_params_frames_theta_clamp(srukf, oracle, synth):
    """The reference's SHIPPED constants a1..a4 = 8 (SLAM.cpp:195-198), N = 8, SEQUENTIAL (the reference's structure):
    S^T S - u u^T turns indefinite, the theta clamp of modifiedCholeskyDecomposition (2279-2285) fires and srukf_update
    repeats those refactorisations on the exact column path (k_gmw_col_*).  Whole frames against orc_update."""
    p = synth.default_params()
    N = 8
    sc = synth.make_scene(N, 3, seed=1, p=p)
    f = srukf.Filter(N, p); f.set_state(sc["X0"], sc["S0"])
    o = oracle.Oracle(N, p); o.set_state(sc["X0"], sc["S0"])
    theta_before = 0
    for t in range(3):
        (h, Si, vis), (ho, Sio, viso), (X, P), (Xo, Po) = _step_both(f, o, sc, t, srukf.UPDATE_SEQUENTIAL, srukf)
        theta = o.clamp_stats()["theta"]
        assert theta > theta_before                                      # the clamp really was active in this frame
        theta_before = theta
        assert np.array_equal(vis, viso)
        np.testing.assert_allclose(h, ho, rtol=0, atol=1e-8)
        np.testing.assert_allclose(X, Xo, rtol=0, atol=1e-9)
        # the filter diverges with these constants (|P| grows 1e-1 -> 1e8 in three frames):
        # the clamped pivots amplify rounding by theta^2 / beta^2, so P is compared relative to its size
        np.testing.assert_allclose(P, Po, rtol=0, atol=1e-5 * np.abs(Po).max() if t == 2 else 1e-10 * max(1.0, np.abs(Po).max()))
    # and BATCHED with the same constants: the clamp fires in frame 3 and the exact path takes over
    f = srukf.Filter(N, p); f.set_state(sc["X0"], sc["S0"])
    o = oracle.Oracle(N, p); o.set_state(sc["X0"], sc["S0"])
    for t in range(3):
        _, _, (X, P), (Xo, Po) = _step_both(f, o, sc, t, srukf.UPDATE_BATCHED, srukf)
        np.testing.assert_allclose(X, Xo, rtol=0, atol=1e-9)
        np.testing.assert_allclose(P, Po, rtol=0, atol=1e-11 * max(1.0, np.abs(Po).max()))
    assert o.clamp_stats()["theta"] > 0


def test_oracle_frames_n200(srukf, oracle, synth):
    """BASELINE configs[2] (N = 200, n = 1204): two whole batched frames against the oracle (~2 s of CPU each),
    through the step-wise API and through the staged replay the benchmark times."""
    p = synth.scene_params()
    N, F = 200, 2
    sc = synth.make_scene(N, F, seed=0, p=p)
    f = srukf.Filter(N, p); f.set_state(sc["X0"], sc["S0"])
    o = oracle.Oracle(N, p); o.set_state(sc["X0"], sc["S0"])
    g = srukf.Filter(N, p); g.set_state(sc["X0"], sc["S0"]); g.stage_sequence(sc["odo"], sc["z"], sc["matched"])
    traj = g.run_frames(0, F)
    for t in range(F):
        (h, Si, vis), (ho, Sio, viso), (X, P), (Xo, Po) = _step_both(f, o, sc, t, srukf.UPDATE_BATCHED, srukf)
        assert np.array_equal(vis, viso)
        np.testing.assert_allclose(h, ho, rtol=0, atol=1e-8)
        np.testing.assert_allclose(X, Xo, rtol=0, atol=1e-9)
        np.testing.assert_allclose(P, Po, rtol=0, atol=1e-11)
        np.testing.assert_allclose(traj[t, :4], Xo[-4:], rtol=0, atol=1e-9)
        np.testing.assert_allclose(traj[t, 4:], Po[-4:-2, -4:-2].ravel(), rtol=0, atol=1e-12)
    Xg, Sg = g.get_state()
    np.testing.assert_allclose(Xg, Xo, rtol=0, atol=1e-9)
    np.testing.assert_allclose(Sg.T @ Sg, Po, rtol=0, atol=1e-11)


ORACLE_JOBS = [("one_frame", dict(N=500, seed=0, storage=st, eps=None, mode=1)) for st in ("f64", "f32")]


@pytest.mark.parametrize("storage", ["f64", "f32"])
def test_oracle_frame_n500(srukf, oracle_pool, synth, storage):
    """BASELINE configs[4] (N = 500, n = 3004): one whole batched frame against the oracle (~30 s of CPU).
    fp32 storage: the oracle starts from the same float-rounded state; after the frame the device state is the
    float rounding of its fp64 result, so it is held to one fp32 ulp of the oracle's."""
    p = synth.scene_params()
    N = 500
    sc = synth.make_scene(N, 1, seed=0, p=p)
    f = srukf.Filter(N, p)
    X0, S0 = sc["X0"], sc["S0"]
    if storage == "f32":
        f.set_storage(srukf.STORAGE_F32)
        X0, S0 = X0.astype(np.float32).astype(np.float64), np.triu(S0).astype(np.float32).astype(np.float64)
    f.set_state(X0, S0)
    f.predict_motion(sc["odo"][0], sc["odo"][1])
    h, Si, vis = f.predict_measurement()
    f.update(sc["z"][0], sc["matched"][0], mode=srukf.UPDATE_BATCHED)
    X, S_ = f.get_state(); P = S_.T @ S_
    r = oracle_pool.get("one_frame", N=N, seed=0, storage=storage, eps=None, mode=1)      # the oracle's frame (~30 - 60 s of one core): started with the session, tests/oracle_jobs.py
    ho, viso, Xo, Po = r["h"], r["vis"], r["Xo"], r["So"].T @ r["So"]
    assert np.array_equal(vis, viso)
    # h = wm0 Z0 + wi sum_c Z_c as the reference accumulates it (SLAM.cpp:1678-1681; wm0 = -1002 at N = 500, 6018 terms,
    # running sum ~3e5) carries ~4e-8 px of rounding in the ORACLE; the device sums deviations from Z0 and is the more
    # accurate of the two.  1e-8 holds up to N = 200.
    np.testing.assert_allclose(h, ho, rtol=0, atol=2e-7)
    if storage == "f64":
        np.testing.assert_allclose(X, Xo, rtol=0, atol=1e-9)
        np.testing.assert_allclose(P, Po, rtol=0, atol=1e-11)
    else:
        eps32 = float(np.finfo(np.float32).eps)
        np.testing.assert_allclose(X, Xo, rtol=eps32, atol=1e-9)
        # P = S^T S with every entry of S rounded to float: |dP_ij| <= eps32 * sum_k |S_ki||S_kj| (+ fp64 noise)
        S = f.get_state()[1]
        bound = eps32 * (np.abs(S).T @ np.abs(S)) * 1.5 + 1e-11
        assert np.all(np.abs(P - Po) <= bound)


def test_replay_recovers_from_theta_clamp_frame(srukf, oracle, synth):
    """Staged replay with the shipped a1..a4 = 8: the third frame needs the theta clamp.  The asynchronous API reports
    SRUKF_ERR_CLAMP_PENDING and names the frame (the frames before it are valid); the synchronous srukf_run_frames
    rewinds to the state before the block, replays the good frames, runs the flagged frame on the exact path and
    continues — and matches the oracle's batched frames (whose orc_gmw always evaluates the clamp)."""
    p = synth.default_params()
    N, F = 8, 4
    sc = synth.make_scene(N, F, seed=1, p=p)
    o = oracle.Oracle(N, p); o.set_state(sc["X0"], sc["S0"])
    to = o.run_frames(sc["odo"], sc["z"], sc["matched"], mode=oracle.Oracle.BATCHED)
    assert o.clamp_stats()["theta"] > 0
    Xo, So = o.get_state()
    f = srukf.Filter(N, p); f.set_state(sc["X0"], sc["S0"]); f.stage_sequence(sc["odo"], sc["z"], sc["matched"])
    f.run_frames_async(0, F)
    with pytest.raises(srukf.SrukfError) as e:
        f.synchronize()
    assert e.value.rc == -7
    frame, row = f.clamp_info()
    assert frame == 2 and 0 <= row < f.n                          # frames 0 and 1 are clean (scripts/r2_parity_probe.py)
    g = srukf.Filter(N, p); g.set_state(sc["X0"], sc["S0"]); g.stage_sequence(sc["odo"], sc["z"], sc["matched"])
    traj = g.run_frames(0, F)
    X, S = g.get_state()
    scale = max(1.0, np.abs(So.T @ So).max())
    np.testing.assert_allclose(traj[:, :4], to[:, :4], rtol=0, atol=1e-8)
    assert np.all(np.abs(traj[:3, 4:] - to[:3, 4:]) <= 1e-9 * np.maximum(1.0, np.abs(to[:3, 4:])))
    np.testing.assert_allclose(X, Xo, rtol=0, atol=1e-8)
    np.testing.assert_allclose(S.T @ S, So.T @ So, rtol=0, atol=1e-6 * scale)    # the filter diverges with these constants (|P| ~ 1e8)


@pytest.mark.parametrize("N", [20, 50, 200])
def test_mixed_precision_downdate(srukf, oracle, synth, N):
    """SRUKF_STORAGE_F32_MIXED (BASELINE configs[4]): fp32 state, S^T S - U U^T on the fp32 matrix pipe (fp32 accumulators flushed into FP64 every 32 rows),
    pivots and trailing updates FP64 — at the reference's EPSILON = 1e-13 (refused until round 6: the mode diverged there; DESIGN.md row g).  N = 20: no null set is
    taken (n < 128), the product is formed in state order; N = 50 / 200: the rank-aware form — kept rows only, robot / shared-anchor tiles in FP64.  One frame from the
    float-rounded state matches the oracle's fp64 frame to what single-precision products imply, and the trajectory stays within 1e-6 m of the fp64 run."""
    p = synth.scene_params()
    F = 12 if N < 200 else 6
    sc = synth.make_scene(N, F, seed=3, p=p)
    X0 = sc["X0"].astype(np.float32).astype(np.float64); S0 = np.triu(sc["S0"]).astype(np.float32).astype(np.float64)
    f = srukf.Filter(N, p); f.set_storage(srukf.STORAGE_F32_MIXED); f.set_state(X0, S0)
    o = oracle.Oracle(N, p); o.set_state(X0, S0)
    f.predict_motion(sc["odo"][0], sc["odo"][1]); o.predict_motion(sc["odo"][0], sc["odo"][1])
    f.predict_measurement(); o.predict_measurement()
    f.update(sc["z"][0], sc["matched"][0]); o.update(sc["z"][0], sc["matched"][0], 1, 0, 1)
    X, S = f.get_state(); Xo, So = o.get_state()
    assert np.all(np.tril(S, -1) == 0.0)
    eps32 = float(np.finfo(np.float32).eps)
    np.testing.assert_allclose(X, Xo, rtol=eps32, atol=1e-9)
    P, Po = S.T @ S, So.T @ So
    # entries of P in units of what single-precision products of the stored factors imply, eps32 (|S|^T |S|)_ij, where that scale is above 1e-10; absolutely below it
    # (the structurally null directions: an fp32-formed product leaves ~1e-7 of the neighbouring variances there, the reference's clamp 1e-13)
    B = eps32 * (np.abs(So).T @ np.abs(So))
    d, big = np.abs(P - Po), B > 1e-10
    ratio, small = float((d[big] / B[big]).max()), float(d[~big].max()) if (~big).any() else 0.0
    print(f"mixed downdate N = {N}, one frame: max |dP| / (eps32 |S|^T|S|) = {ratio:.2f}, max |dP| below that scale = {small:.2e}")
    # (measured, round 6: ratio 0.76 - 0.85; below the scale 2e-11 in the rank-aware form and 5.6e-7 at N = 20, where every pivot is factored and the null ones divide fp32 noise)
    assert ratio <= 16.0 and small <= (2e-9 if N > 20 else 4e-6), (ratio, small)
    tr = {}
    for st in (srukf.STORAGE_F64, srukf.STORAGE_F32_MIXED):
        g = srukf.Filter(N, p); g.set_storage(st); g.set_state(sc["X0"], sc["S0"]); g.stage_sequence(sc["odo"], sc["z"], sc["matched"])
        tr[st] = g.run_frames(0, F)
    d = np.abs(tr[srukf.STORAGE_F32_MIXED][:, :2] - tr[srukf.STORAGE_F64][:, :2]).max()
    assert 0 < d < 1e-6, d
    assert np.abs(tr[srukf.STORAGE_F32_MIXED][:, :2] - sc["odo"][1:F + 1, :2]).max() < 2e-4      # still tracks the truth


@pytest.mark.parametrize("N", [20, 50, 200])
def test_rank_aware_refactor(srukf, oracle, synth, N):
    """The rank-aware refactorisation (structurally null pivots permuted to the end and not factored) against the plain one
    and against the oracle: 3 (N - 1) directions are skipped on a jointly initialised map (+ the three robot-position rows that
    are still copies of the anchors at frame 0), X and P agree with the full factorisation to what the EPSILON clamp leaves
    (1e-13-level entries), S stays upper triangular with sqrt(EPSILON) on the skipped pivots, and the two paths give the same
    trajectory over a staged replay."""
    p = synth.scene_params()
    F = 8 if N < 200 else 4
    sc = synth.make_scene(N, F, seed=5, p=p)
    res = {}
    for on in (True, False):
        f = srukf.Filter(N, p); f.set_rank_aware(on); f.set_state(sc["X0"], sc["S0"])
        nd = f.null_directions()
        n = 6 * N + 4                                     # skipping pays only when it removes a whole 64-row panel
        pays = -(-(n - 3 * (N - 1)) // 64) < -(-n // 64)
        assert nd == (3 * (N - 1) if on and pays else 0), nd
        for t in range(2):
            f.predict_motion(sc["odo"][t], sc["odo"][t + 1]); f.predict_measurement(); f.update(sc["z"][t], sc["matched"][t])
        X, S = f.get_state()
        g = srukf.Filter(N, p); g.set_rank_aware(on); g.set_state(sc["X0"], sc["S0"]); g.stage_sequence(sc["odo"], sc["z"], sc["matched"])
        res[on] = (X, S, g.run_frames(0, F))
    (Xa, Sa, ta), (Xb, Sb, tb) = res[True], res[False]
    assert np.all(np.tril(Sa, -1) == 0.0)
    np.testing.assert_allclose(Xa, Xb, rtol=0, atol=1e-10)
    np.testing.assert_allclose(Sa.T @ Sa, Sb.T @ Sb, rtol=0, atol=2e-12)
    np.testing.assert_allclose(ta, tb, rtol=0, atol=1e-10)
    # the skipped pivots: sqrt(EPSILON) on the diagonal, nothing else in the row; the kept rows are the full factorisation's rows
    if srukf.Filter(N, p).null_directions() == 0 and N == 20:
        return                                            # 124 states: two panels either way, the plain path ran twice
    dropped = [6 * k + e for k in range(1, N) for e in range(3)]
    assert np.all(np.diag(Sa)[dropped] == np.sqrt(1e-13))
    off = Sa[dropped].copy(); off[np.arange(len(dropped)), dropped] = 0.0
    assert np.all(off == 0.0)
    kept = np.setdiff1d(np.arange(6 * N + 4), dropped)
    np.testing.assert_allclose(np.abs(Sa[kept]), np.abs(Sb[kept]), rtol=0, atol=1e-9)
    o = oracle.Oracle(N, p); o.set_state(sc["X0"], sc["S0"])
    for t in range(2):
        o.predict_motion(sc["odo"][t], sc["odo"][t + 1]); o.predict_measurement(); o.update(sc["z"][t], sc["matched"][t], 1, 0, 1)
    Xo, So = o.get_state()
    np.testing.assert_allclose(Xa, Xo, rtol=0, atol=1e-9)
    np.testing.assert_allclose(Sa.T @ Sa, So.T @ So, rtol=0, atol=1e-11)


def test_rank_aware_follows_map_changes(srukf, oracle, synth):
    """The set of skipped pivots is re-derived whenever the map changes: deleting the landmark whose anchor carried the
    batch's pivots hands them to the next landmark of the batch; a new batch of K landmarks adds 3 (K - 1) null anchors
    (+3: the new anchors are copies of the robot position until the next motion step separates them); and the frames after
    each change agree with the oracle started from the same state."""
    p = synth.scene_params()
    N, K = 60, 12
    sc = synth.make_scene(N, 4, seed=8, p=p)
    f = srukf.Filter(N, p); f.set_state(sc["X0"], sc["S0"])
    assert f.null_directions() == 3 * (N - 1)
    f.predict_motion(sc["odo"][0], sc["odo"][1]); f.predict_measurement(); f.update(sc["z"][0], sc["matched"][0])
    f.delete_landmark(0)
    X, S = f.get_state()
    e = np.sum(S * S, axis=1)
    # the set is read off the state itself (row energy of S < 1e-12); the reference's six rank-one updates of the deletion
    # (SLAM.cpp:2654-2668) may leave a few anchors slightly above that, which are then simply factored
    assert f.null_directions() == int(np.sum(e[:-4] < 1e-12))
    assert 3 * (N - 6) <= f.null_directions() <= 3 * (N - 2)
    assert np.all(e[0:3] > 1e-6)                          # landmark 1's anchor carries the batch's pivots now
    sel = np.arange(1, N)
    o = oracle.Oracle(N - 1, p); o.set_state(X, S)
    z, m = sc["z"][1].reshape(N, 2)[sel].ravel(), sc["matched"][1][sel]
    f.predict_motion(sc["odo"][1], sc["odo"][2]); f.predict_measurement(); f.update(z, m)
    o.predict_motion(sc["odo"][1], sc["odo"][2]); o.predict_measurement(); o.update(z, m, mode=oracle.Oracle.BATCHED)
    X1, S1 = f.get_state(); Xo, So = o.get_state()
    np.testing.assert_allclose(X1, Xo, atol=1e-9)
    np.testing.assert_allclose(S1.T @ S1, So.T @ So, atol=1e-10)
    # a second batch
    rng = np.random.default_rng(4)
    uv = np.column_stack([rng.uniform(60, 580, K), rng.uniform(60, 420, K)])
    f.add_landmarks(uv)
    Na = N - 1 + K
    X2, S2 = f.get_state()
    e2 = np.sum(S2 * S2, axis=1)
    assert f.null_directions() == int(np.sum(e2[:-4] < 1e-12)) >= 3 * (N - 6) + 3 * (K - 1)
    o2 = oracle.Oracle(Na, p); o2.set_state(X2, S2)
    f.predict_motion(sc["odo"][2], sc["odo"][3]); h, Si, vis = f.predict_measurement()
    o2.predict_motion(sc["odo"][2], sc["odo"][3]); ho, _, _ = o2.predict_measurement()
    np.testing.assert_allclose(h, ho, atol=1e-8)
    z = h + rng.normal(0, 0.5, h.shape); m = np.asarray(vis, dtype=np.int32)
    f.update(z, m, reorder=srukf.NEED_REORDER, mode=srukf.UPDATE_SEQUENTIAL)      # the frame right after an augmentation (2126-2131)
    o2.update(z, m, reorder=oracle.Oracle.NEED_REORDER, k_new=K, mode=oracle.Oracle.SEQUENTIAL)
    X3, S3 = f.get_state(); Xo3, So3 = o2.get_state()
    np.testing.assert_allclose(X3, Xo3, atol=1e-8)
    np.testing.assert_allclose(S3.T @ S3, So3.T @ So3, atol=1e-9)
    # and ordinary batched frames on the enlarged map as a staged replay, straight from the state the reorder frame left
    # (no set_state in between: the null set and the permuted copy must have followed the reorder path by themselves)
    o3 = oracle.Oracle(Na, p); o3.set_state(X3, S3)
    e3 = np.sum(S3 * S3, axis=1)
    assert f.null_directions() == int(np.sum(e3[:-4] < 1e-12)) > 0
    F2 = 3
    odo2 = np.vstack([sc["odo"][3], sc["odo"][4], sc["odo"][4] + (sc["odo"][4] - sc["odo"][3]), sc["odo"][4] + 2 * (sc["odo"][4] - sc["odo"][3])])
    zs, ms = np.zeros((F2, 2 * Na)), np.zeros((F2, Na), dtype=np.int32)
    for t in range(F2):
        o3.predict_motion(odo2[t], odo2[t + 1]); ho, _, viso = o3.predict_measurement()
        zs[t] = ho + rng.normal(0, 0.5, ho.shape); ms[t] = np.asarray(viso, dtype=np.int32)
        o3.update(zs[t], ms[t], mode=oracle.Oracle.BATCHED)
    f.stage_sequence(odo2, zs, ms)
    f.run_frames(0, F2)
    X4, S4 = f.get_state(); Xo4, So4 = o3.get_state()
    np.testing.assert_allclose(X4, Xo4, atol=1e-8)
    np.testing.assert_allclose(S4.T @ S4, So4.T @ So4, atol=1e-9)


@pytest.mark.parametrize("N", [100, 200])
def test_three_filters_share_the_gpu(srukf, synth, N):
    """Three filters replaying concurrently in GPU_SHARED mode (persistent launches of half the CUs, at most two admitted at a
    time by k_gmw_gate): every one reproduces, bit for bit, what it computes alone with the GPU to itself, and nothing is flagged.
    N = 200 is the size bench.py's multi_sequence leg runs (there the exclusive filter uses the head fold and the XCD-aware tile
    order, the shared ones k_syrk's head tiles and seven workers with two tiles: the same arithmetic)."""
    p = synth.scene_params()
    F, B = 24, 3
    scs = [synth.make_scene(N, F, seed=0, p=p, obs_seed=7000 + b) for b in range(B)]
    alone = []
    for sc in scs:
        f = srukf.Filter(N, p); f.set_state(sc["X0"], sc["S0"]); f.stage_sequence(sc["odo"], sc["z"], sc["matched"])
        alone.append((f.run_frames(0, F),) + f.get_state()); f.close()
    import torch
    fs, ds = [], []
    for sc in scs:
        f = srukf.Filter(N, p); f.set_exclusive(srukf.GPU_SHARED); f.set_state(sc["X0"], sc["S0"]); f.stage_sequence(sc["odo"], sc["z"], sc["matched"])
        fs.append(f); ds.append(torch.zeros(F, 8, dtype=torch.float64, device="cuda"))
    for k0 in range(0, F, 8):
        for f, dt in zip(fs, ds):
            f.run_frames_async(k0, 8, d_traj_ptr=dt.data_ptr() + 8 * 8 * k0)
    for f in fs:
        f.synchronize()
    for f, dt, ref in zip(fs, ds, alone):
        assert f.clamp_info() == (-1, -1)
        X, S = f.get_state()
        np.testing.assert_array_equal(dt.cpu().numpy(), ref[0])
        np.testing.assert_array_equal(X, ref[1]); np.testing.assert_array_equal(S, ref[2])
        f.close()
    # the same through the C entry point for B filters (srukf_run_frames_batch): filters created with the default (exclusive)
    # setting are switched to the shared form by the call
    fs = []
    for sc in scs:
        f = srukf.Filter(N, p); f.set_state(sc["X0"], sc["S0"]); f.stage_sequence(sc["odo"], sc["z"], sc["matched"]); fs.append(f)
    traj = srukf.run_frames_batch(fs, 0, F)
    for b, (f, ref) in enumerate(zip(fs, alone)):
        X, S = f.get_state()
        np.testing.assert_array_equal(traj[b], ref[0])
        np.testing.assert_array_equal(X, ref[1]); np.testing.assert_array_equal(S, ref[2])
        f.close()


def test_rank_aware_notices_a_direction_that_is_not_null(srukf, oracle, synth):
    """A skipped direction that has energy after all (here: poked into S behind the filter's back, 1e-3 on the diagonal of a
    duplicate anchor row) is caught by the per-frame check G_aa - sum Sp[k][a]^2 <= 1e-12: the refactorisation is flagged and
    repeated on the exact column path, the null set is re-derived without that row, and the result is the oracle's for the
    poked state.  (Step-wise API, which forms G from S itself; the replay path works from the permuted copy, which is in step
    with S by construction — every writer of S goes through k_rank_expand or k_rank_shadow.)"""
    p = synth.scene_params()
    N, F = 50, 3
    sc = synth.make_scene(N, F, seed=12, p=p)
    f = srukf.Filter(N, p); f.set_state(sc["X0"], sc["S0"])
    nd = f.null_directions()
    assert nd == 3 * (N - 1)
    row = 6 * 7 + 1                                           # y anchor of landmark 7: a copy of landmark 0's
    f.debug_poke_S(row, row, 1e-3)
    assert f.null_directions() == nd                          # nobody told the filter
    S0 = sc["S0"].copy(); S0[row, row] = 1e-3
    o = oracle.Oracle(N, p); o.set_state(sc["X0"], S0)
    for t in range(F):
        f.predict_motion(sc["odo"][t], sc["odo"][t + 1]); f.predict_measurement(); f.update(sc["z"][t], sc["matched"][t])
        o.predict_motion(sc["odo"][t], sc["odo"][t + 1]); o.predict_measurement(); o.update(sc["z"][t], sc["matched"][t], 1, 0, 1)
        assert f.null_directions() == nd - 1                  # noticed in the first frame: the row is factored from now on
        X, S = f.get_state(); Xo, So = o.get_state()
        np.testing.assert_allclose(X, Xo, rtol=0, atol=1e-9)
        np.testing.assert_allclose(S.T @ S, So.T @ So, rtol=0, atol=1e-11)


def test_prepared_block_graph_gives_the_same_frames(srukf, synth):
    """srukf_prepare_frames: a block of frames captured as ONE graph replays to the bit what the 8-frame / single-frame graphs
    and the eager launches compute; a later call with another count falls back to the default graphs."""
    p = synth.scene_params()
    N, F = 50, 27
    sc = synth.make_scene(N, F, seed=3, p=p)
    res = []
    for prep in (0, 19):
        f = srukf.Filter(N, p); f.set_state(sc["X0"], sc["S0"]); f.stage_sequence(sc["odo"], sc["z"], sc["matched"])
        if prep:
            f.prepare_frames(prep)
        import torch
        dt = torch.zeros(F, 8, dtype=torch.float64, device="cuda")
        f.run_frames_async(0, 19, d_traj_ptr=dt.data_ptr()); f.synchronize()            # the prepared count
        f.run_frames_async(19, 8, d_traj_ptr=dt.data_ptr() + 8 * 8 * 19); f.synchronize()  # another count: default graphs
        res.append((dt.cpu().numpy(),) + f.get_state()); f.close()
    for a, b in zip(res[0], res[1]):
        np.testing.assert_array_equal(a, b)
