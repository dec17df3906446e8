"""GPU test of the drop-in boundary in the reference's own language: the CSLAM-shaped C++ facade
(cv-monoslam_amd/host) driven by a small C++ host the way the MFC view drives the reference
class, checked against the oracle's golden trajectory."""
import os
import struct
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REPLAY = os.path.join(ROOT, "cv-monoslam_amd", "cslam_replay.bin")


def _write_inputs(tmp, sc, p):
    N, F = sc["N"], sc["F"]
    with open(os.path.join(tmp, "scene.bin"), "wb") as f:
        f.write(struct.pack("ii", N, F))
        f.write(np.array([p["a1"], p["a2"], p["a3"], p["a4"]], dtype=np.float64).tobytes())
        f.write(np.ascontiguousarray(sc["X0"]).tobytes())
        f.write(np.ascontiguousarray(sc["S0"]).tobytes())
        f.write(np.ascontiguousarray(sc["z"]).tobytes())
    # the reference's odometry text format: "%d : %*lf %lf %lf %lf"  (SLAM.cpp:475)
    with open(os.path.join(tmp, "odo.txt"), "w") as f:
        for i, (x, y, th) in enumerate(sc["odo"]):
            f.write(f"{i + 1} : {0.1 * i:.3f} {float(x)!r} {float(y)!r} {float(th)!r}\n")


@pytest.mark.parametrize("mode", ["batched", "sequential"])
def test_cslam_facade_replay_matches_golden(tmp_path, golden, synth, mode):
    assert os.path.exists(REPLAY), "run __graft_entry__.build() first"
    g = golden["g6_trajectory_n20"]
    p = synth.scene_params()
    F = 50 if mode == "batched" else 12
    sc = synth.make_scene(20, 50, seed=int(g["seed"]), p=p)
    sc = dict(sc, F=F, z=sc["z"][:F], odo=sc["odo"][:F + 1])
    tmp = str(tmp_path)
    _write_inputs(tmp, sc, p)
    args = [REPLAY, f"{tmp}/scene.bin", f"{tmp}/odo.txt", f"{tmp}/RobotPath.txt", f"{tmp}/traj.bin"]
    if mode == "sequential":
        args.append("sequential")
    out = subprocess.run(args, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    traj = np.fromfile(f"{tmp}/traj.bin").reshape(F, 8)
    np.testing.assert_allclose(traj[:, :4], g["traj_sequential"][:F, :4], atol=1e-9)
    np.testing.assert_allclose(traj[:, 4:], g["traj_sequential"][:F, 4:], atol=1e-13)
    # RobotPath.txt: idx, odoX, odoY, x, y, P00, P01, P10, P11 with %f (SLAM.cpp:3546-3561)
    rows = [l.rstrip("\n").split("\t") for l in open(f"{tmp}/RobotPath.txt")]
    assert len(rows) == F and all(len(r) >= 9 for r in rows)
    rp = np.array([[float(v) for v in r[:9]] for r in rows])
    assert rp[0, 0] == 2 and rp[-1, 0] == F + 1                       # m_showCounter starts at 1 and is bumped before recording
    np.testing.assert_allclose(rp[:, 1:3], sc["odo"][1:F + 1, :2], atol=1e-6)
    np.testing.assert_allclose(rp[:, 3:5], traj[:, :2], atol=1e-6)
    assert f"frames {F}  landmarks 20  predicts 20  matches 20" in out.stdout
    # display accessors (getFeatureCartesianInformation / get3DdisplayInformation for every landmark of the map)
    feat = np.fromfile(f"{tmp}/traj.bin.features").reshape(20, 19)
    import __graft_entry__ as ge
    srukf = ge.load_package().srukf
    flt = srukf.Filter(20, p); flt.set_state(sc["X0"], sc["S0"])
    for k in range(F):
        flt.predict_motion(sc["odo"][k], sc["odo"][k + 1]); flt.predict_measurement()
        flt.update(sc["z"][k], sc["matched"][k], mode=srukf.UPDATE_BATCHED if mode == "batched" else srukf.UPDATE_SEQUENTIAL)
    X, S = flt.get_state(); P = S.T @ S
    for k in range(20):
        xi, yi, zi, th, ph, rho = X[6 * k:6 * k + 6]
        xyz = np.array([xi + np.cos(ph) * np.sin(th) / rho, yi - np.sin(ph) / rho, zi + np.cos(ph) * np.cos(th) / rho])
        J = np.zeros((3, 6)); J[:, :3] = np.eye(3)
        J[:, 3:] = [[np.cos(ph) * np.cos(th) / rho, -np.sin(ph) * np.sin(th) / rho, -np.cos(ph) * np.sin(th) / rho ** 2],
                    [0.0, -np.cos(ph) / rho, np.sin(ph) / rho ** 2],
                    [-np.cos(ph) * np.sin(th) / rho, -np.sin(ph) * np.cos(th) / rho, -np.cos(ph) * np.cos(th) / rho ** 2]]
        cov = J @ P[6 * k:6 * k + 6, 6 * k:6 * k + 6] @ J.T
        np.testing.assert_allclose(feat[k, :3], xyz, atol=1e-9)
        np.testing.assert_allclose(feat[k, 3:12].reshape(3, 3), cov, rtol=1e-9, atol=1e-9 * np.abs(cov).max())
        lam = np.linalg.eigvalsh(feat[k, 3:12].reshape(3, 3))
        np.testing.assert_allclose(np.sort(feat[k, 16:19] ** 2), np.maximum(lam, 0.0), rtol=1e-6, atol=1e-9 * lam.max())   # Jacobi stops at |offdiag| < EPSILON
        # ellipsoid orientation (calculateEigenvaluesAndEigenvectors 2815-2892 + matrix2Quaternion 2903-2948): the Jacobi
        # eigenvector matrix V is a proper rotation (identity times plane rotations); the quaternion encodes V^T in the
        # tr > 0 branch ((m23 - m32) ordering, 2916-2921) and V in the three other branches (2925-2947).  Either way the
        # rotation of the quaternion must diagonalise the covariance with the reported semi-axes, in their order.
        q = feat[k, 12:16]
        assert abs(np.linalg.norm(q) - 1.0) < 1e-9
        r, x, y, z = q
        Rq = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * r), 2 * (x * z + y * r)],
                       [2 * (x * y + z * r), 1 - 2 * (x * x + z * z), 2 * (y * z - x * r)],
                       [2 * (x * z - y * r), 2 * (y * z + x * r), 1 - 2 * (x * x + y * y)]])
        C3 = feat[k, 3:12].reshape(3, 3)
        D = np.diag(feat[k, 16:19] ** 2)
        rec = min(np.abs(M @ D @ M.T - C3).max() for M in (Rq, Rq.T))
        assert rec < 1e-6 * np.abs(C3).max() + 1e-12, (k, rec)
        # and against numpy's eigen-decomposition: every axis is an eigenvector (up to sign) of its eigenvalue
        lam_np, V_np = np.linalg.eigh(C3)
        V = Rq if np.abs(Rq @ D @ Rq.T - C3).max() <= np.abs(Rq.T @ D @ Rq - C3).max() else Rq.T
        for a in range(3):
            j = int(np.argmin(np.abs(lam_np - feat[k, 16 + a] ** 2)))
            gap = np.min(np.abs(np.delete(lam_np, j) - lam_np[j]))
            if gap > 1e-3 * lam_np.max():                               # direction defined only for a separated eigenvalue
                assert abs(abs(V[:, a] @ V_np[:, j]) - 1.0) < 1e-5, (k, a)


def test_cslam_facade_redirection_restart(tmp_path, synth):
    """The redirection branch of predictMotion (SLAM.cpp:1354-1428) through the C++ host: at the flagged odometry sample
    the map is archived in m_featuresAllInfo, a fresh 4-state filter starts at the current position, the host's key points
    are joint-initialised on the device (NEED_REORDER update next), one odometry sample is consumed, and the filter keeps
    tracking.  No oracle counterpart (the oracle restates the numeric path, not the bookkeeping): checked on the
    reference's own invariants."""
    assert os.path.exists(REPLAY), "run __graft_entry__.build() first"
    p = synth.scene_params()
    N, F, R = 12, 16, 6
    sc = synth.make_scene(N, F, seed=5, p=p)
    sc = dict(sc, F=F)
    tmp = str(tmp_path)
    _write_inputs(tmp, sc, p)
    out = subprocess.run([REPLAY, f"{tmp}/scene.bin", f"{tmp}/odo.txt", f"{tmp}/RobotPath.txt", f"{tmp}/traj.bin", "batched", f"redirect={R}"],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    assert f"redirection: archived {N}  stored map {N}  show map {2 * N}" in out.stdout      # 1357-1378, 1408, 1422
    assert f"landmarks {N} " in out.stdout                                                    # the N key points became the new map
    traj = np.fromfile(f"{tmp}/traj.bin").reshape(F, 8)[:F - 1]                               # one sample consumed by the restart
    assert np.all(np.isfinite(traj))
    # before the restart: frame f ends at odometry sample f + 1; from the restart on: at sample f + 2
    np.testing.assert_allclose(traj[:R - 1, :2], sc["odo"][1:R, :2], atol=2e-4)
    after = traj[R - 1:, :2] - sc["odo"][R + 1:F + 1, :2]
    # The new sub-map is anchored at the position of the LAST frame while its key points come from the image of the flagged
    # frame (1381-1406), and the heading is re-seeded with the NEXT odometry sample before the motion step adds that same
    # increment again (1427-1428 then 1518-1523) — the reference's own sequence.  On this scene's tight figure-8
    # (0.2 rad of heading per frame, a prior heading sigma of 0.02) that leaves a heading error the filter works off
    # slowly: the track stays within a few steps of the truth and bounded, which is what is asserted.
    assert np.abs(after).max() < 0.05
    assert np.all(np.abs(np.diff(after, axis=0)) < 0.004)
    dth = sc["odo"][R + 1, 2] - sc["odo"][R, 2]
    assert abs(traj[R - 1, 3] - sc["odo"][R + 1, 2]) < abs(dth) + 0.05   # re-seeded from the odometry, then incremented once more (see above)
    # the fresh filter starts from the initial robot sqrt covariance diag(sigma_x, ...) (1402-1406): P_xx is back near sigma_x^2
    # (the anchors of jointly initialised landmarks are copies of the robot position, so measurements do not shrink it)
    assert 0.5 * p["sigma_x"] ** 2 < traj[R - 1, 4] < 2.0 * p["sigma_x"] ** 2


def _map_change_scenario(synth):
    """N = 10 landmarks, two of them far enough from the principal point that the heading sweep of the figure-8 carries their
    predicted pixel inside the DIST_2_BORDER band; in frame F_STARVE only three landmarks are matched, so K = 3 new ones are added."""
    p = synth.scene_params()
    F, F_STARVE, KEEP, K = 20, 10, 3, 3
    rng = np.random.default_rng(1)
    r = 80 * np.sqrt(rng.uniform(0, 1, 8)); a = rng.uniform(0, 2 * np.pi, 8)
    uv = np.column_stack([p["cam_cy"] + r * np.cos(a), p["cam_cx"] + r * np.sin(a)])
    uv = np.vstack([uv, [[p["cam_cy"] + 165 * np.cos(0.3), p["cam_cx"] + 165 * np.sin(0.3)], [p["cam_cy"] + 166, p["cam_cx"]]]])
    N = len(uv)
    odo = synth.figure8_odometry(F)

    def truth_of(uvd, pose):
        th, ph = synth.pixel_to_angles(uvd, pose[2], p)
        return np.column_stack([np.full(len(uvd), pose[0]), np.full(len(uvd), pose[1]), np.zeros(len(uvd)), th, ph, np.cos(ph) * np.cos(th) / 3.0])

    def measure(truth, noise):
        z = np.zeros((F, 2 * len(truth)))
        for t in range(F):
            x, y, psi = odo[t + 1]
            z[t] = (synth.project(truth, np.broadcast_to([x, y, 0.0], (len(truth), 3)), np.full(len(truth), psi), np.zeros((len(truth), 2)), p, iters=12) + noise[t]).ravel()
        return z
    nrng = np.random.default_rng(77)
    z = measure(truth_of(uv, odo[0]), nrng.normal(0, 0.5, (F, N, 2)))
    X0, S0 = synth.joint_init(np.zeros(4), np.diag([p["sigma_x"], p["sigma_y"], p["sigma_z"], p["sigma_theta"]]), uv, p)
    # the new key points: where three new ceiling points show in the image of frame F_STARVE (pose odo[F_STARVE + 1])
    uv_new = np.array([[p["cam_cy"] + 40.0, p["cam_cx"] - 30.0], [p["cam_cy"] - 55.0, p["cam_cx"] + 20.0], [p["cam_cy"] + 10.0, p["cam_cx"] + 60.0]])
    z_new = measure(truth_of(uv_new, odo[F_STARVE + 1]), nrng.normal(0, 0.5, (F, K, 2)))
    return dict(p=p, N=N, F=F, F_STARVE=F_STARVE, KEEP=KEEP, K=K, odo=odo, z=z, X0=X0, S0=S0, uv_new=uv_new, z_new=z_new)


def _oracle_replay_with_policy(oracle, sc):
    """The same sequence through the CPU oracle with the deletion policy of updateFeaturesInformation (SLAM.cpp:2443-2460, traversal
    2554-2615) and the addFeatures trigger (552-562) restated here: returns (trajectory rows, event list)."""
    p, N, F, K = sc["p"], sc["N"], sc["F"], sc["K"]
    W, H, B = p["image_w"], p["image_h"], 20
    ids = list(range(1, N + 1)); next_id = N + 1
    npred = {i: 0 for i in ids}; nmatch = {i: 0 for i in ids}; pred = {i: (0.0, 0.0) for i in ids}
    o = oracle.Oracle(N, p); o.set_state(sc["X0"], sc["S0"])
    k_new, events, traj = 0, [], []
    for f in range(F):
        n_l = len(ids)
        o.predict_motion(sc["odo"][f], sc["odo"][f + 1])
        h, Si, vis = o.predict_measurement()
        zf = np.zeros(2 * n_l); mf = np.zeros(n_l, dtype=np.int32)
        for pos, i in enumerate(ids):
            zz = sc["z"][f, 2 * (i - 1):2 * i] if i <= N else sc["z_new"][f, 2 * (i - N - 1):2 * (i - N)]
            zf[2 * pos:2 * pos + 2] = zz
            if vis[pos]:
                npred[i] += 1; pred[i] = (h[2 * pos], h[2 * pos + 1])
                if not (f == sc["F_STARVE"] and pos >= sc["KEEP"]):
                    mf[pos] = 1; nmatch[i] += 1
        nm = int(mf.sum())
        if nm:
            # NEED_REORDER on the frame after an addition (2083-2090): the oracle runs that path in the reference's per-column form
            o.update(zf, mf, 0 if k_new else 1, k_new, oracle.Oracle.SEQUENTIAL if k_new else oracle.Oracle.BATCHED)
        X, S = o.get_state()
        # deletion policy, the reference's traversal: the node behind a deleted one is skipped in this call
        pos = 0
        while pos < len(ids):
            i = ids[pos]; dim = len(X)
            zi, th, ph, rho = X[6 * pos + 2], X[6 * pos + 3], X[6 * pos + 4], X[6 * pos + 5]
            hz = rho * (zi - X[dim - 2]) + np.cos(ph) * np.cos(th)
            px, py = pred[i]
            dele = (npred[i] > 2 * nmatch[i] and npred[i] >= 10) or rho < 0.01 or hz < 0 or px < B or py < B or W - px < B or H - py < B
            if mf[pos] if pos < len(mf) else False:
                mx, my = zf[2 * pos], zf[2 * pos + 1]
                dele = dele or mx < B or my < B or W - mx < B or H - my < B
            if dele:
                events.append((f, "delete", i))
                X, S = oracle.delete_feature(p, X, S, pos)
                ids.pop(pos); zf = np.delete(zf, [2 * pos, 2 * pos + 1]); mf = np.delete(mf, pos)
                if pos == len(ids):
                    break
            pos += 1
        if len(ids) != n_l:
            o = oracle.Oracle(len(ids), p); o.set_state(X, S)
        traj.append(np.concatenate([X[-4:], (S.T @ S)[-4:-2, -4:-2].ravel()]))
        k_new = 0
        if int(mf.sum()) < 5:                                                        # m_nMatches < m_minNUM (556); deletions of matched landmarks count (2497, 2509)
            events.append((f, "add", K))
            X, S = oracle.joint_init(p, X, S, sc["uv_new"])
            for j in range(K):
                ids.append(next_id); npred[next_id] = 0; nmatch[next_id] = 0; pred[next_id] = (0.0, 0.0); next_id += 1
            o = oracle.Oracle(len(ids), p); o.set_state(X, S)
            k_new = K
    return np.array(traj), events


def test_cslam_facade_map_changes_mid_sequence(tmp_path, synth, oracle):
    """SLAM() with the reference's map management: two landmarks drift into the border band and are deleted by
    updateFeaturesInformation's policy (SLAM.cpp:2443-2460 -> deleteOneFeature), a frame with fewer than m_minNUM matches
    triggers addFeatures (552-562) -> integrateFeaturesInformation on the device -> a NEED_REORDER update.  Events and
    trajectory are held to an oracle-driven replay of the same sequence in which the policy is restated independently."""
    assert os.path.exists(REPLAY), "run __graft_entry__.build() first"
    sc = _map_change_scenario(synth)
    p, N, F, K = sc["p"], sc["N"], sc["F"], sc["K"]
    otraj, oevents = _oracle_replay_with_policy(oracle, sc)
    dels = [e for e in oevents if e[1] == "delete"]; adds = [e for e in oevents if e[1] == "add"]
    assert sorted(e[2] for e in dels) == [9, 10] and all(e[0] < sc["F_STARVE"] for e in dels)     # the two outer landmarks, before the additions
    assert adds == [(sc["F_STARVE"], "add", K)]
    tmp = str(tmp_path)
    _write_inputs(tmp, dict(N=N, F=F, X0=sc["X0"], S0=sc["S0"], z=sc["z"], odo=sc["odo"]), p)
    with open(f"{tmp}/extra.bin", "wb") as f:
        f.write(struct.pack("iii", K, sc["F_STARVE"], sc["KEEP"]))
        f.write(np.ascontiguousarray(sc["uv_new"]).tobytes()); f.write(np.ascontiguousarray(sc["z_new"]).tobytes())
    out = subprocess.run([REPLAY, f"{tmp}/scene.bin", f"{tmp}/odo.txt", f"{tmp}/RobotPath.txt", f"{tmp}/traj.bin", "batched", f"extra={tmp}/extra.bin"],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    events = []
    for ln in out.stdout.splitlines():
        if ln.startswith("event frame"):
            w = ln.split(); events.append((int(w[2]), w[3], int(w[4])))
    assert sorted(events) == sorted(oevents), (events, oevents)
    assert f"landmarks {N - 2 + K} " in out.stdout
    traj = np.fromfile(f"{tmp}/traj.bin").reshape(F, 8)
    R = sc["F_STARVE"] + 2                                                           # frames up to and including the NEED_REORDER update
    np.testing.assert_allclose(traj[:R, :4], otraj[:R, :4], rtol=0, atol=1e-9)
    np.testing.assert_allclose(traj[:R, 4:], otraj[:R, 4:], rtol=0, atol=1e-11)
    # behind it the 3K null directions of the new anchors are pivoted with EPSILON by both sides from factors that differ by rounding noise there, which the
    # reference algorithm divides by 1e-13 (test_need_reorder_matches_oracle).  Round 3 held these frames to 1e-6 / 1e-8 without knowing how much of that
    # was used; measured in round 4: max |dpose| 1.2e-11, max |dP| 2.0e-13 — the tolerance of every other frame holds here too.
    print(f"map-change scenario, frames behind the NEED_REORDER update: max |dpose| {np.abs(traj[R:, :4] - otraj[R:, :4]).max():.3e}, max |dP| {np.abs(traj[R:, 4:] - otraj[R:, 4:]).max():.3e}")
    np.testing.assert_allclose(traj[R:, :4], otraj[R:, :4], rtol=0, atol=1e-9)
    np.testing.assert_allclose(traj[R:, 4:], otraj[R:, 4:], rtol=0, atol=1e-11)
    assert np.abs(traj[:, :2] - sc["odo"][1:, :2]).max() < 2e-3                      # and the filter keeps tracking through the changes
