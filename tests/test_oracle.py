"""CPU tests: the oracle against its golden vectors, against numpy/scipy factorizations and
against the independent numpy restatement (tests/np_filter.py, cv-monoslam_amd/synth.py)."""
import numpy as np
import pytest
import scipy.linalg

from np_filter import NpFilter, gmw as np_gmw
from g7_check import g7_check as _g7_check


def test_g1_weights(oracle, golden, synth):
    tab = golden["g1_weights"]["table"]
    for row in tab:
        wt, Na = int(row[0]), int(row[1])
        w = oracle.sample_parameter(Na, wt)
        got = [w[k] for k in ("wm0", "wc0", "wi", "wi_sr", "gamma", "wm0_sr", "wc0_sr")]
        assert np.array_equal(got, row[2:])
        wm0, wc0, wi, wi_sr, gamma = synth.ut_weights(Na, wt)
        np.testing.assert_allclose([wm0, wc0, wi, wi_sr, gamma], row[2:7], rtol=1e-14)
        assert abs(wm0 + 2 * Na * wi - 1.0) < 1e-9 * max(1, abs(wm0))      # weights sum to one
        assert abs(wi * gamma ** 2 - 0.5) < 1e-12                          # => motion QR keeps S11
    assert golden["g1_weights"]["table"][3][2] == pytest.approx(1 - 1209 / 3.0)   # N=200: wm0 = -402


def test_g2_projection_golden_and_numpy(oracle, golden, synth):
    g = golden["g2_projection"]
    p = synth.default_params()
    uv = oracle.project(p, g["feat"], g["pos"], g["psi"], g["err"], early_exit=0)
    assert np.array_equal(uv, g["uv"])                       # bit-stable restatement
    uv_fast = oracle.project(p, g["feat"], g["pos"], g["psi"], g["err"], early_exit=1)
    assert np.array_equal(uv, uv_fast)                       # early Newton exit is bit-exact
    uv_np = synth.project(g["feat"], g["pos"], g["psi"], g["err"], p)
    np.testing.assert_allclose(uv_np, uv, rtol=0, atol=1e-9)
    # a zeroed undistorted pixel still passes through the distortion model (quirk): not (0, 0)
    assert np.allclose(g["uv"][1], g["uv"][2]) and 0 < g["uv"][1][0] < 0.1


def test_newton_early_exit_is_bit_exact_on_random_points(oracle, synth):
    """The 100 Newton iterations of distortOnePointRW (SLAM.cpp:3186-3193) either reach a fixed point or fall into a 2-cycle
    between neighbouring doubles; both exits (the device's srukf_project takes them too) reproduce the full loop bit for bit."""
    p = synth.default_params()
    rng = np.random.default_rng(7)
    M = 20000
    feat = np.column_stack([rng.uniform(-2, 2, M), rng.uniform(-1, 1, M), rng.uniform(-2, 2, M), rng.uniform(-0.8, 0.8, M),
                            rng.uniform(-0.5, 0.5, M), rng.uniform(0.05, 1.0, M)])
    pos = np.column_stack([rng.uniform(-0.5, 0.5, M), rng.uniform(-0.2, 0.2, M), rng.uniform(-0.5, 0.5, M)])
    psi = rng.uniform(-0.6, 0.6, M); err = rng.normal(0, 1.0, (M, 2))
    full = oracle.project(p, feat, pos, psi, err, early_exit=0)
    fast = oracle.project(p, feat, pos, psi, err, early_exit=1)
    assert np.array_equal(full, fast)
    assert (full != 0).any(axis=1).mean() > 0.2                         # a good share of the points is inside the image


def test_projection_quirks(oracle, synth):
    p = synth.default_params()
    feat = np.array([[0, 0, 0, 0.05, -0.02, 1 / 3.0]])
    z = np.zeros((1, 3))
    uv = oracle.project(p, feat, z, [0.0], np.zeros((1, 2)))[0]
    # x/y swap (SLAM.cpp:3338-3339): world +x (theta > 0) moves the SECOND undistorted coordinate,
    # which is centred on cam_cx; the first is centred on cam_cy
    assert uv[1] > p["cam_cx"] and abs(uv[0] - p["cam_cy"]) < 20
    # pixel noise rows add to (uvu.y, uvu.x) = (err0, err1)
    uv2 = oracle.project(p, feat, z, [0.0], np.array([[2.0, 0.0]]))[0]
    assert uv2[1] - uv[1] == pytest.approx(2.0, abs=1e-2) and abs(uv2[0] - uv[0]) < 1e-3


def test_qr_matches_numpy_up_to_row_signs(oracle):
    rng = np.random.default_rng(0)
    for m, k in ((7, 3), (40, 12), (130, 60), (5, 5), (3, 6)):
        A = rng.normal(size=(m, k))
        R = oracle.qr_r(A)
        assert np.allclose(np.tril(R, -1), 0)
        np.testing.assert_allclose(R.T @ R, (A.T @ A) if m >= k else R.T @ R, atol=1e-11)
        if m >= k:
            Rn = np.linalg.qr(A, mode="r")
            np.testing.assert_allclose(np.abs(R), np.abs(Rn), atol=1e-11)
            # GSL sign rule: R_jj = -sign(leading element) * norm
            assert np.sign(R[0, 0]) == -np.sign(A[0, 0])
    # tau = 0 when the sub-column is already zero: R_jj keeps the element
    A = np.array([[2.0, 1.0], [0.0, 3.0], [0.0, 0.0]])
    R = oracle.qr_r(A)
    assert R[0, 0] == 2.0 and R[0, 1] == 1.0


def test_g3_gmw(oracle, golden):
    g = golden["g3_gmw"]
    for name in ("spd", "psd", "ind"):
        S, D, L, ce, ct = oracle.gmw(g[name + "_G"])
        assert np.array_equal(S, g[name + "_S"]) and np.array_equal(D, g[name + "_D"]) and np.array_equal(L, g[name + "_L"])
        assert [ce, ct] == list(g[name + "_clamps"])
        Sn, Dn, ne, nt, _ = np_gmw(g[name + "_G"])
        # rows of S that belong to clamped (null) pivots are rounding noise / sqrt(EPSILON):
        # compare on P = S^T S, never on S element-wise
        np.testing.assert_allclose(Sn.T @ Sn, S.T @ S, atol=1e-12)
        if name != "psd":
            np.testing.assert_allclose(Sn, S, atol=1e-12)
        assert (ne, nt) == (ce, ct)
    # SPD: equals the Cholesky factor; nothing is clamped
    G = g["spd_G"]
    np.testing.assert_allclose(g["spd_S"], scipy.linalg.cholesky(G), atol=1e-12)
    # PSD rank-deficient: S^T S = G + E with diagonal E >= 0 small
    P = g["psd_S"].T @ g["psd_S"]
    E = P - g["psd_G"]
    assert np.abs(E - np.diag(np.diag(E))).max() < 1e-12 and np.diag(E).min() > -1e-12
    assert g["psd_clamps"][0] >= 5 and g["psd_clamps"][1] == 0      # EPSILON clamp on the null pivots
    # indefinite: theta clamp active, result is PD
    assert g["ind_clamps"][1] > 0 and np.linalg.eigvalsh(g["ind_S"].T @ g["ind_S"]).min() > 0


def test_g4_joint_init_rank_and_numpy(oracle, golden, synth):
    g = golden["g4_joint_init"]
    p = synth.default_params()
    for K in (1, 2, 8):
        X, S = oracle.joint_init(p, np.zeros(4), np.diag([.02, .02, .005, .02]), g[f"K{K}_uv"])
        assert np.array_equal(X, g[f"K{K}_X"]) and np.array_equal(S, g[f"K{K}_S"])
        P = S.T @ S
        assert np.linalg.matrix_rank(P, tol=1e-12) == 4 + 3 * K
        Xn, Sn = synth.joint_init(np.zeros(4), np.diag([.02, .02, .005, .02]), g[f"K{K}_uv"], p)
        np.testing.assert_allclose(Xn, X, atol=1e-13)
        np.testing.assert_allclose(Sn.T @ Sn, P, atol=1e-15)
        # anchors are copies of the robot position
        for k in range(K):
            assert np.array_equal(X[6 * k:6 * k + 3], np.zeros(3))
            assert X[6 * k + 5] == pytest.approx(1 / 3.0)


def test_g5_one_frame(oracle, golden, synth):
    g = golden["g5_frame_n8"]
    p = synth.scene_params()
    o = oracle.Oracle(8, p)
    o.set_state(g["X0"], g["S0"])
    o.predict_motion(g["odo"][0], g["odo"][1])
    X, S = o.get_state()
    assert np.array_equal(X, g["X_motion"])
    np.testing.assert_allclose(S.T @ S, g["P_motion"], atol=1e-18)
    # motion re-triangularisation only changes the last four columns of S
    np.testing.assert_allclose(np.abs(S[:-4, :-4]), np.abs(np.triu(g["S0"])[:-4, :-4]), atol=1e-16)
    h, Si, vis = o.predict_measurement()
    assert np.array_equal(h, g["h"]) and np.array_equal(Si, g["Si"]) and np.array_equal(vis, g["vis"])
    o.update(g["z"][0], g["matched"][0], 1, 0, 0)
    X, S = o.get_state()
    assert np.array_equal(X, g["X_post"])
    np.testing.assert_allclose(S.T @ S, g["P_post"], atol=1e-18)
    # independent numpy restatement of the same frame
    f = NpFilter(8, p, synth)
    f.set_state(g["X0"], g["S0"])
    f.predict_motion(g["odo"][0], g["odo"][1])
    np.testing.assert_allclose(f.X, g["X_motion"], atol=1e-13)
    np.testing.assert_allclose(f.S.T @ f.S, g["P_motion"], atol=1e-15)
    hn, Sin, visn = f.predict_measurement()
    np.testing.assert_allclose(hn, g["h"], atol=1e-9)
    np.testing.assert_allclose(np.abs(Sin), np.abs(g["Si"]), atol=1e-10)
    f.update(g["z"][0], g["matched"][0], mode=0)
    np.testing.assert_allclose(f.X, g["X_post"], atol=1e-11)
    np.testing.assert_allclose(f.S.T @ f.S, g["P_post"], atol=1e-13)


def test_g6_trajectory(oracle, golden, synth):
    g = golden["g6_trajectory_n20"]
    p = synth.scene_params()
    N, F = int(g["N"]), int(g["F"])
    sc = synth.make_scene(N, F, seed=int(g["seed"]), p=p)
    o = oracle.Oracle(N, p)
    o.set_state(sc["X0"], sc["S0"])
    traj = o.run_frames(sc["odo"][:11], sc["z"][:10], sc["matched"][:10], oracle.Oracle.SEQUENTIAL)
    np.testing.assert_allclose(traj, g["traj_sequential"][:10], rtol=0, atol=1e-12)
    # the filter tracks: estimated pose within 0.1 mm of the true pose, all 50 frames
    assert np.abs(g["traj_sequential"][:, :2] - sc["odo"][1:, :2]).max() < 1e-4
    # one batched refactor per frame == 2M sequential ones, N = 20
    assert np.abs(g["traj_sequential"][:, :4] - g["traj_batched"][:, :4]).max() < 1e-11
    assert np.abs(g["traj_sequential"][:, 4:] - g["traj_batched"][:, 4:]).max() < 1e-15
    # the theta clamp never fires on this scene, the eps clamp fires on every refactor (rank-deficient P)
    assert g["clamps_seq"][1] == 0 and g["clamps_bat"][1] == 0 and g["clamps_bat"][0] >= F
    # numpy restatement, batched, first 10 frames
    f = NpFilter(N, p, synth)
    f.set_state(sc["X0"], sc["S0"])
    tn = f.run(sc, mode=1, frames=10)
    np.testing.assert_allclose(tn[:, :4], g["traj_batched"][:10, :4], atol=1e-10)
    np.testing.assert_allclose(tn[:, 4:], g["traj_batched"][:10, 4:], atol=1e-13)


def test_need_reorder_path(oracle, synth):
    """Frame right after landmarks were added (m_nAddings != 0): rank-aware pivoted refactor
    (SLAM.cpp:2122-2138, 2158-2179).  P must stay PSD with rank <= 4 + 3K + 3K and close to the
    NEEDNOT_REORDER result in X (the gains are the same)."""
    p = synth.scene_params()
    N = 4
    sc = synth.make_scene(N, 1, seed=9, p=p)
    res = {}
    for reorder in (0, 1):
        o = oracle.Oracle(N, p)
        o.set_state(sc["X0"], sc["S0"])
        o.predict_motion(sc["odo"][0], sc["odo"][1])
        o.predict_measurement()
        o.update(sc["z"][0], sc["matched"][0], reorder, N, 0)
        res[reorder] = o.get_state()
    np.testing.assert_allclose(res[0][0], res[1][0], atol=1e-12)
    P0, P1 = res[0][1].T @ res[0][1], res[1][1].T @ res[1][1]
    assert np.linalg.eigvalsh(P0).min() > -1e-12
    assert np.abs(P0 - P1).max() < 1e-6


def test_sequential_equals_batched_small(oracle, synth):
    p = synth.scene_params()
    for N, seed in ((2, 1), (8, 2)):
        sc = synth.make_scene(N, 12, seed=seed, p=p)
        out = []
        for mode in (0, 1):
            o = oracle.Oracle(N, p)
            o.set_state(sc["X0"], sc["S0"])
            out.append(o.run_frames(sc["odo"], sc["z"], sc["matched"], mode))
        assert np.abs(out[0][:, :4] - out[1][:, :4]).max() < 1e-11
        assert np.abs(out[0][:, 4:] - out[1][:, 4:]).max() < 1e-15


def test_default_gain_structure_diverges_with_large_process_noise(oracle, synth):
    """Documents WHY the synthetic scene uses the reference's alternative a1..a4 (SLAM.cpp:191-194):
    with a = 8 and >= 8 landmarks matched per frame the reference update over-subtracts and the
    theta clamp / negative pivots appear (DESIGN.md 'Synthetic scene')."""
    p = synth.default_params()          # a1..a4 = 8
    sc = synth.make_scene(8, 8, seed=1, p=p)
    sc["odo"] = synth.circle_odometry(8) if hasattr(synth, "circle_odometry") else sc["odo"]
    o = oracle.Oracle(8, p)
    o.set_state(sc["X0"], sc["S0"])
    o.run_frames(sc["odo"], sc["z"], sc["matched"], 1)
    assert o.clamp_stats()["theta"] > 0


def test_delete_feature_is_the_marginal(oracle, synth):
    """deleteOneFeature (SLAM.cpp:2637-2668): the six rank-1 updates with the removed rows rebuild exactly the
    remaining block of P = S^T S (numpy cross-check of the oracle restatement)."""
    p = synth.scene_params()
    for N, idx in ((5, 0), (5, 2), (5, 4), (1, 0)):
        sc = synth.make_scene(N, 1, seed=40 + N, p=p, init="fullrank")
        X, S = sc["X0"], sc["S0"]
        Xn, Sn = oracle.delete_feature(p, X, S, idx)
        keep = np.r_[0:6 * idx, 6 * idx + 6:6 * N + 4]
        assert np.array_equal(Xn, X[keep]) and np.all(np.tril(Sn, -1) == 0.0)
        np.testing.assert_allclose(Sn.T @ Sn, (S.T @ S)[np.ix_(keep, keep)], rtol=0, atol=1e-14)


def _texture(rng, h=480, w=640):
    """smooth random gray texture (box-filtered noise) so that correlation peaks are well defined"""
    t = rng.uniform(0, 255, (h, w))
    k = 5
    c = np.cumsum(np.cumsum(np.pad(t, ((k, k), (k, k)), mode="wrap"), axis=0), axis=1)
    t = (c[2 * k:, 2 * k:] - c[:-2 * k, 2 * k:] - c[2 * k:, :-2 * k] + c[:-2 * k, :-2 * k]) / (2 * k) ** 2
    t = (t - t.min()) / (t.max() - t.min()) * 255
    return t.astype(np.uint8)


def test_association_oracle_against_numpy(oracle, synth):
    """calculateCrossCorrelation / dataAssociation (SLAM.cpp:1915-2009, 3141-3166) restated in the oracle, pinned on
    numpy: the best normalised cross correlation inside the chi-square gate, first maximum in row-major order."""
    p = synth.scene_params()
    rng = np.random.default_rng(0)
    img = _texture(rng)
    for trial in range(6):
        u, v = rng.uniform(40, 600), rng.uniform(40, 440)
        true = (int(u) + rng.integers(-3, 4), int(v) + rng.integers(-3, 4))
        tmpl = img[true[1] - 8:true[1] + 9, true[0] - 8:true[0] + 9].copy()          # rows = y, like the image ROI
        Si = np.array([[rng.uniform(2.5, 4.0), rng.uniform(-0.5, 0.5)], [0.0, rng.uniform(2.5, 4.0)]])
        ok, best, loc = oracle.associate_one(p, img, [u, v], Si, tmpl)
        # numpy restatement
        pi = Si.T @ Si; pinv = np.linalg.inv(pi)
        hx = min(10, max(8, int(np.ceil(2 * Si[0, 0])))); hy = min(10, max(8, int(np.ceil(2 * Si[1, 1]))))
        bestn, locn = 0.0, None
        for j in range(int(v) - hy, int(v) + hy + 1):
            for i in range(int(u) - hx, int(u) + hx + 1):
                e = np.array([i - u, j - v])
                if e @ pinv @ e < 5.99146454710798:
                    roi = img[j - 8:j + 9, i - 8:i + 9].astype(np.float64); a = roi - roi.mean(); b = tmpl - tmpl.astype(np.float64).mean()
                    c = float((a * b).sum() / np.sqrt((a * a).sum()) / np.sqrt((b * b).sum()))
                    if c > bestn: bestn, locn = c, (i - int(u) + u, j - int(v) + v)
        assert ok and abs(best - bestn) < 1e-12 and best > 0.999
        np.testing.assert_allclose(loc, locn, atol=1e-12)
        assert (round(loc[0] - (u - int(u))), round(loc[1] - (v - int(v)))) == true   # found where the template was cut


def test_warp_patch_identity_view(oracle, synth):
    """wrapPatch (SLAM.cpp:1803-1906) from the very pose and pixel the landmark was created at reproduces the centre
    of the init patch (x/y transposed as the reference indexes it; the truncating uchar cast may lose one grey level)."""
    p = synth.scene_params()
    rng = np.random.default_rng(1)
    img = _texture(rng)
    robot = np.array([0.3, -0.2, 0.0, 0.4])
    R = np.array([[np.cos(robot[3]), -np.sin(robot[3]), 0], [np.sin(robot[3]), np.cos(robot[3]), 0], [0, 0, 1.0]])
    for trial in range(4):
        px = np.array([rng.uniform(60, 580), rng.uniform(60, 420)])
        cu, cv = int(round(px[0])), int(round(px[1]))
        patch = img[cv - 10:cv + 11, cu - 10:cu + 11].copy()
        xyz = np.array([1.0, 0.5, 3.0])
        out = oracle.warp_patch(p, robot, R, robot[:3], px, xyz, px, patch, np.zeros((17, 17), dtype=np.uint8))
        ref = patch[3:20, 3:20]                                   # matchPatch[i][j] <- initPatch.at(i + 3, j + 3)
        # index 16 maps to 19 + rounding noise: when the noise is positive, ceil gives 20 and the reference's bound
        # `right < 2*HP_INIT` (1889) leaves the pixel untouched — only rows / columns 0..15 are certain to be written
        assert np.abs(out[:16, :16].astype(int) - ref[:16, :16].astype(int)).max() <= 1
        edge = np.abs(out.astype(int) - ref.astype(int)) > 1
        assert np.all(out[edge] == 0)


@pytest.mark.parametrize("N,F,kw", [(1, 3, {}), (8, 4, {}), (20, 3, {}), (8, 3, {"weight_type": 1}), (8, 3, {"weight_type": 2})])
def test_matched_cpu_baseline_equals_oracle(oracle, synth, N, F, kw):
    """oracle/srukf_matched.c (the algorithm-matched OpenMP baseline bench.py times beside the GPU) against the
    reference-structured oracle in batched mode: same trajectory, same X, same P."""
    p = synth.scene_params(); p.update(kw)
    sc = synth.make_scene(N, F, seed=30 + N, p=p)
    o = oracle.Oracle(N, p); o.set_state(sc["X0"], sc["S0"])
    to = o.run_frames(sc["odo"], sc["z"], sc["matched"], mode=oracle.Oracle.BATCHED)
    m = oracle.Matched(N, p, threads=2); m.set_state(sc["X0"], sc["S0"])
    tm = m.run_frames(sc["odo"], sc["z"], sc["matched"])
    Xo, So = o.get_state(); Xm, Sm = m.get_state()
    tol = 1e-6 if kw.get("weight_type") == 1 else 1e-11          # type 1 cancels six digits in every weighted mean
    np.testing.assert_allclose(tm, to, rtol=0, atol=tol)
    np.testing.assert_allclose(Xm, Xo, rtol=0, atol=tol)
    np.testing.assert_allclose(Sm.T @ Sm, So.T @ So, rtol=0, atol=tol)
    assert np.all(np.tril(Sm, -1) == 0.0) and m.clamp_fallbacks() == 0


@pytest.mark.parametrize("N,F", [(24, 4), (50, 3)])
def test_matched_rank_aware_equals_oracle(oracle, synth, N, F):
    """The CPU port in the GPU path's rank-aware form (mt_set_rank_aware: the 3 (N - 1) structurally null pivots skipped, K <= r in S^T S - U U^T and in the
    cross covariance, NullSkip in the projection) against the reference-structured oracle: a third of the landmarks unmatched in every frame."""
    p = synth.scene_params()
    sc = synth.make_scene(N, F, seed=70 + N, p=p)
    matched = sc["matched"].copy()
    for t in range(F):
        matched[t, (np.arange(N) + t) % 3 == 0] = 0
    o = oracle.Oracle(N, p); o.set_state(sc["X0"], sc["S0"])
    to = o.run_frames(sc["odo"], sc["z"], matched, mode=oracle.Oracle.BATCHED)
    Xo, So = o.get_state()
    for threads in (1, 3):
        m = oracle.Matched(N, p, threads=threads); m.set_state(sc["X0"], sc["S0"])
        assert m.set_rank_aware(True) == 3 * (N - 1) == m.null_directions()
        tm = m.run_frames(sc["odo"], sc["z"], matched)
        Xm, Sm = m.get_state()
        assert m.rank_fallbacks() == 0 and m.clamp_fallbacks() == 0
        np.testing.assert_allclose(tm, to, rtol=0, atol=1e-10)
        np.testing.assert_allclose(Xm, Xo, rtol=0, atol=1e-10)
        np.testing.assert_allclose(Sm.T @ Sm, So.T @ So, rtol=0, atol=1e-11)
        assert np.all(np.tril(Sm, -1) == 0.0)
        # the skipped rows are exactly what the reference's clamp leaves: sqrt(EPSILON) e_k
        null = [k for k in range(6 * N) if np.count_nonzero(Sm[k]) == 1 and Sm[k, k] == np.sqrt(p["epsilon"])]
        assert len(null) == 3 * (N - 1)


def test_matched_rank_aware_falls_back_on_theta_clamp(oracle, synth):
    """Shipped a1..a4 = 8 at N = 24: the theta clamp fires; the rank-aware form notices BEFORE it writes anything, repeats the refactorisation on the full-rank
    path (which in turn takes the exact orc_gmw) and re-derives the null set."""
    p = synth.default_params()
    N, F = 24, 3
    sc = synth.make_scene(N, F, seed=1, p=p)
    o = oracle.Oracle(N, p); o.set_state(sc["X0"], sc["S0"])
    to = o.run_frames(sc["odo"], sc["z"], sc["matched"], mode=oracle.Oracle.BATCHED)
    m = oracle.Matched(N, p, threads=2); m.set_state(sc["X0"], sc["S0"]); m.set_rank_aware(True)
    tm = m.run_frames(sc["odo"], sc["z"], sc["matched"])
    assert o.clamp_stats()["theta"] > 0 and m.rank_fallbacks() >= 1 and m.clamp_fallbacks() >= 1
    # (with these constants S^T S - U U^T is indefinite and the filter diverges — test_default_gain_structure_... — so the 1e-13 by which the canonical
    #  null rows differ from the zero rows of S0 grows by two orders per frame: 2e-11, 2e-9, 1e-6 relative; the full-rank port: 1e-12, 6e-11, 1e-10)
    rel = (np.abs(tm - to) / np.maximum(1.0, np.abs(to))).max(axis=1)
    assert rel[0] <= 1e-9 and rel[1] <= 1e-7 and rel[2] <= 1e-4, rel


def test_matched_cpu_baseline_theta_clamp_fallback(oracle, synth):
    """With the shipped a1..a4 = 8 the theta clamp fires in frame 3: the blocked factorisation notices afterwards and
    repeats that refactor with the exact orc_gmw, like the device path."""
    p = synth.default_params()
    sc = synth.make_scene(8, 3, seed=1, p=p)
    o = oracle.Oracle(8, p); o.set_state(sc["X0"], sc["S0"])
    to = o.run_frames(sc["odo"], sc["z"], sc["matched"], mode=oracle.Oracle.BATCHED)
    m = oracle.Matched(8, p, threads=2); m.set_state(sc["X0"], sc["S0"])
    tm = m.run_frames(sc["odo"], sc["z"], sc["matched"])
    assert m.clamp_fallbacks() >= 1
    assert np.all(np.abs(tm - to) <= 1e-9 * np.maximum(1.0, np.abs(to)))


@pytest.mark.parametrize("name", ["g7_sequential_n50", "g7_sequential_n200"])
def test_g7_batched_equals_sequential_at_benchmark_sizes(oracle, golden, synth, name):
    """The structural choice the headline rests on, at the sizes the benchmark runs ("must be re-verified at
    N = 20/50/200"): ONE batched refactor per frame against the reference's 2M per-column refactors (SLAM.cpp:2066-2095,
    2116-2154).  The fixture is the SEQUENTIAL oracle (N = 200: 3 minutes of CPU, generated once); the BATCHED oracle runs here."""
    g = golden[name]
    N, F = int(g["N"]), int(g["F"])
    p = synth.scene_params()
    sc = synth.make_scene(N, F, seed=int(g["seed"]), p=p)
    o = oracle.Oracle(N, p); o.set_state(sc["X0"], sc["S0"])
    traj = o.run_frames(sc["odo"], sc["z"], sc["matched"], oracle.Oracle.BATCHED)
    X, S = o.get_state()
    np.testing.assert_allclose(traj[:, :4], g["traj"][:, :4], rtol=0, atol=1e-11)
    np.testing.assert_allclose(traj[:, 4:], g["traj"][:, 4:], rtol=0, atol=1e-14)
    _g7_check(g, X, S.T @ S, 1e-11, 1e-13)
    assert g["clamps"][1] == 0                                  # the theta clamp never fired in the sequential run either


@pytest.mark.parametrize("rank_aware", [False, True])
def test_g8_matched_port_over_40_frames(oracle, golden, synth, rank_aware):
    """The CPU port bench.py times (both forms) against the g8 fixture: 40 frames at N = 200 of the oracle in BATCHED mode, partial / empty / single-match
    frames included (tests/golden/make_golden.py g8) — the fixture the device's default replay is held to in tests/test_gpu_parity_r4.py."""
    from g7_check import g7_check
    g = golden["g8_batched_n200"]
    N, F = int(g["N"]), int(g["F"])
    p = synth.scene_params()
    sc = synth.make_scene(N, F, seed=int(g["seed"]), p=p)
    matched = g["matched"].astype(np.int32)
    assert matched[7].sum() == 0 and matched[19].sum() == 1 and all(matched[t].sum() == N - N // 3 for t in range(F) if t not in (7, 19))
    m = oracle.Matched(N, p, threads=4); m.set_state(sc["X0"], sc["S0"])
    if rank_aware:
        assert m.set_rank_aware(True) == 3 * (N - 1)
    traj = m.run_frames(sc["odo"], sc["z"], matched)
    assert m.clamp_fallbacks() == 0 and m.rank_fallbacks() == 0
    np.testing.assert_allclose(traj[:, :4], g["traj"][:, :4], rtol=0, atol=1e-9)
    np.testing.assert_allclose(traj[:, 4:], g["traj"][:, 4:], rtol=0, atol=1e-12)
    X, S = m.get_state()
    g7_check(g, X, S.T @ S, 1e-9, 1e-11)
