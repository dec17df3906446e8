"""GPU parity tests added in round 3 (-m gpu): the BATCHED device frame held to the reference-structured (SEQUENTIAL, per-column
refactor, SLAM.cpp:2066-2095) oracle fixture g7 at the benchmark sizes N = 50 and N = 200, through the step-wise API and the staged
replay the benchmark times, with the rank-aware refactorisation on and off; map changes keep the per-context switches."""
import numpy as np
import pytest

from g7_check import g7_check

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["g7_sequential_n50", "g7_sequential_n200"])
@pytest.mark.parametrize("rank_aware", [True, False])
def test_g7_device_batched_against_sequential_fixture(srukf, golden, synth, name, rank_aware):
    g = golden[name]
    N, F = int(g["N"]), int(g["F"])
    p = synth.scene_params()
    sc = synth.make_scene(N, F, seed=int(g["seed"]), p=p)
    # step-wise API (predictMotion -> predictMeasurement -> KalmanUpdate, host in between)
    f = srukf.Filter(N, p); f.set_rank_aware(rank_aware); f.set_state(sc["X0"], sc["S0"])
    assert (f.null_directions() > 0) == (rank_aware and N >= 50)
    for t in range(F):
        f.predict_motion(sc["odo"][t], sc["odo"][t + 1]); f.predict_measurement()
        f.update(sc["z"][t], sc["matched"][t], mode=srukf.UPDATE_BATCHED)
    X, S = f.get_state()
    g7_check(g, X, S.T @ S, 1e-9, 1e-11)
    # staged replay (what bench.py times): one captured graph per block of frames
    r = srukf.Filter(N, p); r.set_rank_aware(rank_aware); r.set_state(sc["X0"], sc["S0"]); r.stage_sequence(sc["odo"], sc["z"], sc["matched"])
    traj = r.run_frames(0, F)
    np.testing.assert_allclose(traj[:, :4], g["traj"][:, :4], rtol=0, atol=1e-9)
    np.testing.assert_allclose(traj[:, 4:], g["traj"][:, 4:], rtol=0, atol=1e-12)
    X, S = r.get_state()
    g7_check(g, X, S.T @ S, 1e-9, 1e-11)


def test_map_changes_keep_the_context_switches(srukf, synth):
    """srukf_set_rank_aware(ctx, 0) — the documented way to the reference-faithful full-rank refactorisation — must survive
    srukf_add_landmarks / srukf_delete_landmark, which rebuild the context behind the handle (round-2 advisor finding)."""
    p = synth.scene_params()
    N = 24
    sc = synth.make_scene(N, 1, seed=5, p=p)
    f = srukf.Filter(N, p); f.set_state(sc["X0"], sc["S0"])
    assert f.null_directions() > 0
    f.set_rank_aware(False)
    assert f.null_directions() == 0
    f.delete_landmark(3)
    assert f.null_directions() == 0
    f.add_landmarks(np.array([[300.0, 200.0], [340.0, 260.0]]))
    assert f.null_directions() == 0
    f.set_rank_aware(True)
    assert f.null_directions() > 0
    f.delete_landmark(0)
    assert f.null_directions() > 0


def test_bench_force_dist_runs_the_rccl_collectives_with_one_rank():
    """`bench.py --gpus 1 --force-dist`: torch.distributed.run -> one rank -> init_process_group("nccl") -> broadcast of the map ->
    barrier -> max all-reduce -> all-gather of the trajectories.  On a 1-GPU box this is the only way the RCCL calls of the
    N-rank path execute at all.  Fresh child processes; nothing is re-exec'ed."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--force-dist", "--steps", "6", "--warmup", "2",
                        "--profile-frames", "4", "--no-cpu-baseline", "--sequences-per-gpu", "0", "--no-configs4"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 1 and d["collectives"] == "nccl" and d["steps"] == 6
    assert d["value"] > 100 and d["pose_rmse_vs_truth_m"] < 1e-3
    # the keys a scaling run is checked against (round 4): world size the process group saw, one rate per rank, the map broadcast's size and time
    assert d["rccl_world_size"] == 1 and len(d["per_rank_frames_per_s"]) == 1 and d["slowest_rank"] == 0
    assert d["map_broadcast"]["bytes"] == 8 * (1204 + 1204 * 1204) and d["map_broadcast"]["ms"] > 0


@pytest.mark.parametrize("N", [20, 200])
def test_replay_motion_modes_agree(srukf, synth, N):
    """The three forms of the replay's motion step — its own launch (k_motion, as in the step-wise API), inside the projection launch
    (k_project_motion), and "table" mode (the previous frame's tail prepares the robot part of every sigma point, k_project_table
    reduces it; rank-aware replay at N = 200, falls back to the second form at N = 20) — are the same arithmetic in different
    orders: trajectories and states agree to rounding, over a run that is split into two calls."""
    p = synth.scene_params()
    F = 12
    sc = synth.make_scene(N, F, seed=4, p=p)
    res = []
    for mode in (0, 1, 2):
        f = srukf.Filter(N, p); f.debug_set("fused_motion", mode); f.set_state(sc["X0"], sc["S0"]); f.stage_sequence(sc["odo"], sc["z"], sc["matched"])
        traj = np.vstack([f.run_frames(0, 5), f.run_frames(5, F - 5)])
        X, S = f.get_state()
        res.append((traj, X, S.T @ S))
    for traj, X, P in res[1:]:
        np.testing.assert_allclose(traj[:, :4], res[0][0][:, :4], rtol=0, atol=1e-11)
        np.testing.assert_allclose(traj[:, 4:], res[0][0][:, 4:], rtol=0, atol=1e-14)
        np.testing.assert_allclose(X, res[0][1], rtol=0, atol=2e-10)      # weakly observed directions carry the rounding differences of twelve frames
        np.testing.assert_allclose(P, res[0][2], rtol=0, atol=1e-12)


def test_head_fold_is_the_same_arithmetic(srukf, synth):
    """Exclusive rank-aware replay at N = 200: the head tiles of S^T S - U U^T, the pending state update and the dropped diagonal as
    helper workgroups of the persistent factorisation launch (head fold, default) against the k_syrk launch in front of it: the
    same tile routine in the same summation order, so the trajectories and the states are bit-identical."""
    p = synth.scene_params()
    N, F = 200, 10
    sc = synth.make_scene(N, F, seed=6, p=p)
    res = []
    for fold in (1, 0):
        f = srukf.Filter(N, p); f.debug_set("head_fold", fold); f.set_state(sc["X0"], sc["S0"]); f.stage_sequence(sc["odo"], sc["z"], sc["matched"])
        traj = f.run_frames(0, F)
        X, S = f.get_state()
        res.append((traj, X, S))
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1]) and np.array_equal(res[0][2], res[1][2])


def test_fused_tail_agrees_with_table_mode(srukf, synth):
    """ "Fused tail" mode (default at N = 200: k_rank_expand also projects the next frame's sigma points, the frame's motion reduction rides on
    k_pxy2, k_gain re-centres the robot rows of the cross covariances; a frame is four launches) against "table" mode (k_project_table in front of
    every frame).  The projection is the same arithmetic on the same values — Z and DZ of the frame that follows are bit-identical —; the robot
    rows of Pxy are summed around the centre point instead of the mean and re-centred afterwards, so trajectories and states agree to rounding.
    Over a run split into three calls, with graphs and with eager launches."""
    p = synth.scene_params()
    N, F = 200, 14
    sc = synth.make_scene(N, F, seed=9, p=p)
    n = 6 * N + 4; L = 2 * (n + 5) + 1; mp = ((2 * N + 63) // 64) * 64; npad = ((n + 63) // 64) * 64
    res = []
    for fuse, graph in ((1, 1), (0, 1), (1, 0)):
        f = srukf.Filter(N, p); f.debug_set("tail_fuse", fuse); f.debug_set("use_graph", graph)
        f.set_state(sc["X0"], sc["S0"]); f.stage_sequence(sc["odo"], sc["z"], sc["matched"])
        f.run_frames(0, 1)
        X1, S1 = f.get_state()
        # after ONE frame both modes have the same state bit for bit?  no: the robot rows differ in rounding -> compare to rounding, and the
        # projection of frame 1 (fused: already in the work buffers; table: after frame 1 has run) on bit-identical inputs below
        traj = np.vstack([f.run_frames(1, 9), f.run_frames(10, F - 10)])
        X, S = f.get_state()
        assert f.debug_get("gmw_aborts") == 0 and f.debug_get("clamp_rows") == 0
        res.append((traj, X, S.T @ S, X1, S1))
    for r in res[1:]:
        np.testing.assert_allclose(r[3], res[0][3], rtol=0, atol=1e-12)
        np.testing.assert_allclose(r[0][:, :4], res[0][0][:, :4], rtol=0, atol=1e-11)
        np.testing.assert_allclose(r[0][:, 4:], res[0][0][:, 4:], rtol=0, atol=1e-14)
        np.testing.assert_allclose(r[1], res[0][1], rtol=0, atol=2e-10)
        np.testing.assert_allclose(r[2], res[0][2], rtol=0, atol=1e-12)
    # graphs against eager launches of the same mode: the same launches, bit-identical
    assert np.array_equal(res[0][0], res[2][0]) and np.array_equal(res[0][1], res[2][1])
    # the projection itself: start both modes from the SAME state, run one frame; the fused tail has then projected frame 1, table mode does it when frame 1 runs
    # (two frames: the structurally null rows of the joint-initialised S0 are zero rows, not sqrt(EPSILON) e_k, so a run's FIRST frame takes the launch sequence that
    #  reads them as they are — k_project_motion, k_pxy — and the fused tail starts with the second; round 4, srukf_api.hip null_canonical)
    g = srukf.Filter(N, p); g.debug_set("tail_fuse", 1); g.set_state(sc["X0"], sc["S0"]); g.stage_sequence(sc["odo"], sc["z"], sc["matched"]); g.run_frames(0, 2)
    Xg, Sg = g.get_state()
    Zf, DZf = g.debug_copy("Z", L * mp), g.debug_copy("DZ", npad * mp)
    t = srukf.Filter(N, p); t.debug_set("tail_fuse", 0); t.set_state(Xg, Sg); t.stage_sequence(sc["odo"], sc["z"], sc["matched"]); t.run_frames(2, 1)
    Zt, DZt = t.debug_copy("Z", L * mp), t.debug_copy("DZ", npad * mp)
    # (compared where the buffers are DEFINED: a structurally null direction is projected for its own landmark only — NullSkip —, the rest of its rows is
    #  never written and never read, and g's buffers still hold there what its first frame's launch sequence, which projects everything, left)
    Na = n + 5
    Zf, Zt = Zf.reshape(L, mp), Zt.reshape(L, mp); DZf, DZt = DZf.reshape(npad, mp), DZt.reshape(npad, mp)
    null = [i for i in range(n - 4) if np.count_nonzero(Sg[i]) == 1 and Sg[i, i] == np.sqrt(p["epsilon"]) and i >= 2]
    assert len(null) >= 3 * (N - 1) - 2
    full = np.ones(Na, dtype=bool); full[null] = False
    rows = np.concatenate([[0], 1 + np.flatnonzero(full), 1 + Na + np.flatnonzero(full)])
    assert np.array_equal(Zf[rows, :2 * N], Zt[rows, :2 * N])
    for i in null:
        k = i // 6
        for r in (1 + i, 1 + Na + i):
            assert np.array_equal(Zf[r, 2 * k:2 * k + 2], Zt[r, 2 * k:2 * k + 2])
    # DZ rows are in permuted order (kept directions first): the kept rows in full
    nk = int(full[:n].sum())
    assert np.array_equal(DZf[:nk, :2 * N], DZt[:nk, :2 * N])


@pytest.mark.parametrize("N,storage", [(100, "f64"), (300, "f64"), (200, "f32"), (500, "f32")])
def test_fused_tail_on_the_other_paths(srukf, synth, N, storage):
    """ "Fused tail" mode away from the headline configuration: where the owners do not fold (k_syrk over the kept rows: N = 100, 300, 500 — srukf_debug_set
    "table_perm") and with fp32 storage of the state (configs[4]: N = 500; the tail and the state update round what they write — "f32_fuse" — instead of
    k_quantize / k_rank_round / k_traj launches behind the tail).  Against the launch sequence of round 2 for those cases (k_project_motion, k_pxy, ...):
    the same rounding points, agreement at fp64 rounding level, and with fp32 storage the stored state is float-representable."""
    p = synth.scene_params()
    F = 8
    sc = synth.make_scene(N, F, seed=12, p=p)
    res = []
    for on in (1, 0):
        f = srukf.Filter(N, p)
        if storage == "f32": f.set_storage(srukf.STORAGE_F32); f.debug_set("f32_fuse", on)
        else: f.debug_set("table_perm", on)
        f.set_state(sc["X0"], sc["S0"]); f.stage_sequence(sc["odo"], sc["z"], sc["matched"])
        traj = np.vstack([f.run_frames(0, 3), f.run_frames(3, F - 3)])
        X, S = f.get_state()
        assert f.debug_get("gmw_aborts") == 0 and f.debug_get("clamp_rows") == 0
        if storage == "f32":
            X32, S32 = f.get_state_f32()
            assert np.array_equal(X, X32.astype(np.float64)) and np.array_equal(np.triu(S), np.triu(S32).astype(np.float64)) and np.all(np.tril(S, -1) == 0.0)
        res.append((traj, X, S.T @ S))
    np.testing.assert_allclose(res[0][0][:, :4], res[1][0][:, :4], rtol=0, atol=1e-11)
    np.testing.assert_allclose(res[0][0][:, 4:], res[1][0][:, 4:], rtol=0, atol=1e-14)
    np.testing.assert_allclose(res[0][1], res[1][1], rtol=0, atol=2e-10)
    # (fp32 storage: an fp64 value that sits on a float rounding boundary may round the other way — one float ulp of one entry of S, seen in two of 1.4 million entries of P)
    np.testing.assert_allclose(res[0][2], res[1][2], rtol=0, atol=2e-11 if storage == "f64" else 2e-9)


def test_fused_tail_with_partial_and_empty_matches(srukf, synth):
    """The default replay at N = 200 ("fused tail": four launches per frame) on a sequence where a third of the landmarks is unmatched in every
    frame, one frame has no match at all and one has a single match, against the replay with the motion step and the projection as launches of
    their own (srukf_debug_set "fused_motion" 0: the step-wise API's kernels): same filter to rounding."""
    p = synth.scene_params()
    N, F = 200, 10
    sc = synth.make_scene(N, F, seed=13, p=p)
    rng = np.random.default_rng(5)
    m = np.array(sc["matched"], copy=True)
    m[rng.random(m.shape) < 0.33] = 0
    m[4, :] = 0
    m[7, :] = 0; m[7, 17] = 1
    res = []
    for mode in (2, 0):
        f = srukf.Filter(N, p); f.debug_set("fused_motion", mode)
        f.set_state(sc["X0"], sc["S0"]); f.stage_sequence(sc["odo"], sc["z"], m)
        traj = np.vstack([f.run_frames(0, 4), f.run_frames(4, F - 4)])
        X, S = f.get_state()
        assert f.debug_get("gmw_aborts") == 0 and f.debug_get("clamp_rows") == 0
        res.append((traj, X, S.T @ S))
    np.testing.assert_allclose(res[0][0][:, :4], res[1][0][:, :4], rtol=0, atol=1e-11)
    np.testing.assert_allclose(res[0][0][:, 4:], res[1][0][:, 4:], rtol=0, atol=1e-14)
    np.testing.assert_allclose(res[0][1], res[1][1], rtol=0, atol=2e-10)
    np.testing.assert_allclose(res[0][2], res[1][2], rtol=0, atol=1e-12)
