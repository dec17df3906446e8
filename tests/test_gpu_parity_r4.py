"""GPU parity tests added in round 4 (-m gpu).

g8 (tests/golden/g8_batched_n200.npz, make_golden.py g8): 40 consecutive frames at N = 200 through the ORACLE (BATCHED mode, itself held to the reference's
SEQUENTIAL structure by g7), a third of the landmarks unmatched in every frame, one frame without any match, one with a single match.  The path bench.py times —
the default staged replay: fused tail, head fold, graphs — and the launch sequences around it are held to it, not to each other."""
import numpy as np
import pytest

from g7_check import g7_check

pytestmark = pytest.mark.gpu


def _g8_scene(golden, synth):
    g = golden["g8_batched_n200"]
    N, F = int(g["N"]), int(g["F"])
    p = synth.scene_params()
    sc = synth.make_scene(N, F, seed=int(g["seed"]), p=p)
    return g, N, F, p, sc, g["matched"].astype(np.int32)


def _hold_to_g8(g, traj, X, S):
    np.testing.assert_allclose(traj[:, :4], g["traj"][:, :4], rtol=0, atol=1e-9)
    np.testing.assert_allclose(traj[:, 4:], g["traj"][:, 4:], rtol=0, atol=1e-12)
    g7_check(g, X, S.T @ S, 1e-9, 1e-11)


@pytest.mark.parametrize("variant", ["default", "eager", "shared", "table", "split_calls"])
def test_g8_default_replay_against_the_oracle_over_40_frames(srukf, golden, synth, variant):
    """default: one call, captured graphs (what bench.py times); eager: the same launches without graphs; shared: SRUKF_GPU_SHARED (persistent launches of half the
    CUs behind the gate, k_syrk head launch instead of the head fold); table: k_project_table in front of every frame instead of the fused tail; split_calls: the
    run cut into blocks of 1 + 6 + 13 + 20 frames (every call starts with the projection launch its predecessor's tail would have made unnecessary)."""
    g, N, F, p, sc, matched = _g8_scene(golden, synth)
    f = srukf.Filter(N, p)
    if variant == "eager":
        f.debug_set("use_graph", 0)
    if variant == "shared":
        f.set_exclusive(srukf.GPU_SHARED)
    if variant == "table":
        f.debug_set("tail_fuse", 0)
    f.set_state(sc["X0"], sc["S0"]); f.stage_sequence(sc["odo"], sc["z"], matched)
    assert f.null_directions() == 3 * (N - 1)
    if variant == "split_calls":
        traj = np.vstack([f.run_frames(0, 1), f.run_frames(1, 6), f.run_frames(7, 13), f.run_frames(20, F - 20)])
    else:
        traj = f.run_frames(0, F)
    assert f.debug_get("gmw_aborts") == 0 and f.debug_get("clamp_rows") == 0
    X, S = f.get_state()
    _hold_to_g8(g, traj, X, S)


SWITCHES = [("fused_motion", 0, False), ("fused_motion", 1, False), ("pxy2", 0, False), ("nullskip", 0, False), ("head_fold", 0, False), ("gain_fold", 1, False),
            ("gmw_persist", 0, True), ("gmw_fused", 0, True), ("rank_fused", 0, True), ("rank_fold", 0, True), ("rank_aware", 0, True)]


@pytest.mark.parametrize("key,value,process_wide", SWITCHES)
def test_g8_every_measurement_switch_against_the_oracle(srukf, golden, synth, key, value, process_wide):
    """Every launch-sequence switch of srukf_debug_set that changes what the staged replay at N = 200 launches (include/srukf.h lists them; "tail_fuse" and
    "use_graph" are the `table` and `eager` variants above), flipped away from the default, against the same 40 oracle frames: the alternatives are held to the
    ORACLE, not only to the default path.  fused_motion 0 / 1: k_motion + k_project / k_project_motion instead of "table" mode; pxy2 0: k_pxy; nullskip 0: every
    direction projected for every landmark; head_fold 0: k_syrk launch in front of the persistent launch; gmw_persist 0: one launch per 64-row panel; gmw_fused /
    rank_fused / rank_fold 0: the owners never form their tiles (k_syrk does, then the permutation pass / the kept rows only); rank_aware 0: every pivot factored;
    gain_fold 1: k_gain's work in the tile epilogue of k_pxy2_fold (three launches per frame; round 6's experiment, off by default because it is slower)."""
    g, N, F, p, sc, matched = _g8_scene(golden, synth)
    if process_wide:
        srukf.debug_set_global(key, value)
    try:
        f = srukf.Filter(N, p)
        if not process_wide:
            f.debug_set(key, value)
        f.set_state(sc["X0"], sc["S0"]); f.stage_sequence(sc["odo"], sc["z"], matched)
        assert f.null_directions() == (0 if key == "rank_aware" else 3 * (N - 1))
        traj = f.run_frames(0, F)
        assert f.debug_get("gmw_aborts") == 0 and f.debug_get("clamp_rows") == 0
        assert (f.debug_get("fold_seqs") > 0) == (key == "gain_fold")
        X, S = f.get_state()
        f.close()
    finally:
        if process_wide:
            srukf.debug_set_global(key, 1)
    _hold_to_g8(g, traj, X, S)


def test_g8_step_api_against_the_oracle_over_40_frames(srukf, golden, synth):
    """The same 40 frames through the step-wise API (predictMotion -> predictMeasurement -> host -> KalmanUpdate), rank-aware form on."""
    g, N, F, p, sc, matched = _g8_scene(golden, synth)
    f = srukf.Filter(N, p); f.set_state(sc["X0"], sc["S0"])
    traj = np.zeros((F, 8))
    for t in range(F):
        f.predict_motion(sc["odo"][t], sc["odo"][t + 1]); f.predict_measurement()
        f.update(sc["z"][t], matched[t], mode=srukf.UPDATE_BATCHED)
        pose, P4 = f.get_robot()
        traj[t, :4] = pose; traj[t, 4:] = np.asarray(P4).reshape(4, 4)[:2, :2].ravel()
    X, S = f.get_state()
    _hold_to_g8(g, traj, X, S)


ORACLE_JOBS = [("batched_filter", dict(b=b, N=200, F=12, Fo=2)) for b in range(8)]


@pytest.mark.parametrize("B,wide", [(2, 1), (4, 1), (8, 1), (4, 0)])
def test_batched_filters_against_the_oracle_and_against_solo_runs(srukf, oracle_pool, synth, B, wide):
    """srukf_run_frames_batch with B filters at N = 200 (Monte-Carlo runs: one map, own measurement noise).  wide = 1 (default): ONE launch per stage for all
    filters on one stream — k_pxy2_b, k_gain_b, k_syrk_b, k_syrk_own_b, one k_gmw_step64_b per panel, k_rank_expand_b — behind each filter's first frame, which a
    fresh state runs alone.  wide = 0: round 3's form, one stream per filter and one tenant per filter (B persistent launches of 256 / B CUs admitted at a time,
    two register tiles per worker at B = 4).  Every filter's first two frames against the ORACLE (BATCHED), and the whole block against the same sequence
    replayed ALONE (exclusive mode, owners' fold, head fold) — bit for bit, trajectory and state."""
    N, F, Fo = 200, 12, 2
    srukf.debug_set_global("batch_wide", wide)
    p = synth.scene_params()
    scs = [synth.make_scene(N, F, seed=0, p=p, obs_seed=7000 + b) for b in range(B)]
    fs = []
    for sc in scs:
        f = srukf.Filter(N, p); f.set_state(sc["X0"], sc["S0"]); f.stage_sequence(sc["odo"], sc["z"], sc["matched"]); fs.append(f)
    t2 = srukf.run_frames_batch(fs, 0, Fo)
    mid = [f.get_state() for f in fs]
    t10 = srukf.run_frames_batch(fs, Fo, F - Fo)
    for b, sc in enumerate(scs):
        assert fs[b].debug_get("gmw_aborts") == 0 and fs[b].debug_get("clamp_rows") == 0 and fs[b].debug_get("gate_timeouts") == 0
        assert fs[b].debug_get("gmw_shared") == (0 if wide else 1)
        r = oracle_pool.get("batched_filter", b=b, N=N, F=F, Fo=Fo)             # (filter b's oracle frames are the same in every parametrisation: tests/oracle_jobs.py)
        to, Xo, So = r["to"], r["Xo"], r["So"]
        np.testing.assert_allclose(t2[b][:, :4], to[:, :4], rtol=0, atol=1e-9)
        np.testing.assert_allclose(t2[b][:, 4:], to[:, 4:], rtol=0, atol=1e-12)
        np.testing.assert_allclose(mid[b][0], Xo, rtol=0, atol=1e-9)
        np.testing.assert_allclose(mid[b][1].T @ mid[b][1], So.T @ So, rtol=0, atol=1e-11)
        g = srukf.Filter(N, p); g.set_state(sc["X0"], sc["S0"]); g.stage_sequence(sc["odo"], sc["z"], sc["matched"])
        ts = np.vstack([g.run_frames(0, Fo), g.run_frames(Fo, F - Fo)])
        Xs, Ss = g.get_state(); Xb, Sb = fs[b].get_state()
        assert np.array_equal(np.vstack([t2[b], t10[b]]), ts) and np.array_equal(Xs, Xb) and np.array_equal(Ss, Sb)
        g.close()
    for f in fs:
        f.close()
    srukf.debug_set_global("batch_wide", 1)


def test_batched_filters_recover_from_flagged_frames(srukf, oracle, synth):
    """The batched replay with the shipped a1..a4 = 8 at N = 200: S^T S - U U^T turns indefinite within a few frames and the reference's theta clamp becomes active
    (test_replay_recovers_from_theta_clamp_frame at N = 8).  The frame tail of the batched launches flags the frame per filter, the rest of that filter's block is
    void; srukf_run_frames_batch rewinds the filter to the state before the block and reruns it alone through srukf_run_frames (good frames replayed, the flagged
    frame on the exact column path).  Result: what the same filter computes alone, bit for bit, and the oracle's frames to the tolerance a diverging filter allows."""
    p = synth.default_params()
    N, F, B = 200, 4, 2
    scs = [synth.make_scene(N, F, seed=1, p=p, obs_seed=300 + b) for b in range(B)]
    fs = []
    for sc in scs:
        f = srukf.Filter(N, p); f.set_state(sc["X0"], sc["S0"]); f.stage_sequence(sc["odo"], sc["z"], sc["matched"]); fs.append(f)
    traj = srukf.run_frames_batch(fs, 0, F)
    for b, sc in enumerate(scs):
        g = srukf.Filter(N, p); g.set_state(sc["X0"], sc["S0"]); g.stage_sequence(sc["odo"], sc["z"], sc["matched"])
        ts = g.run_frames(0, F)
        Xs, Ss = g.get_state(); Xb, Sb = fs[b].get_state()
        assert np.array_equal(traj[b], ts) and np.array_equal(Xs, Xb) and np.array_equal(Ss, Sb)
        g.close()
    o = oracle.Oracle(N, p); o.set_state(scs[0]["X0"], scs[0]["S0"])
    to = o.run_frames(scs[0]["odo"], scs[0]["z"], scs[0]["matched"], mode=oracle.Oracle.BATCHED)
    assert o.clamp_stats()["theta"] > 0                                  # (the scenario does need the clamp)
    rel = np.abs(traj[0] - to) / np.maximum(1.0, np.abs(to))
    assert rel[:2].max() <= 1e-8 and rel.max() <= 1e-5, rel.max(axis=1)
    for f in fs:
        f.close()


@pytest.mark.parametrize("N,storage,rank_aware", [(400, "f64", 1), (400, "f64", 0), (500, "f32", 1)])
def test_split_form_of_the_persistent_factorisation(srukf, synth, N, storage, rank_aware):
    """Sizes beyond two register tiles per worker (N >= 400 in the rank-aware form): the factorisation is k_gmw_pivslab_persist + k_gmw_tiles_persist side by side (the split form), captured in the
    frame graphs as two branches.  Same arithmetic as the memory-tile instance of k_gmw_persist it replaces (srukf_debug_set "mem_split" 0), which the oracle tests of
    round 2 hold (test_oracle_frame_n500 now runs the split form): bit for bit over a staged run, no abandoned launch."""
    p = synth.scene_params()
    F = 14
    sc = synth.make_scene(N, F, seed=21, p=p)
    res = []
    for split in (1, 0):
        srukf.debug_set_global("mem_split", split)
        try:
            f = srukf.Filter(N, p)
            if not rank_aware: f.set_rank_aware(0)
            if storage == "f32": f.set_storage(srukf.STORAGE_F32)
            f.set_state(sc["X0"], sc["S0"]); f.stage_sequence(sc["odo"], sc["z"], sc["matched"])
            tr = [f.run_frames(0, 2), f.run_frames(2, F - 2)]  # (graph replay: blocks of eight frames and single frames)
            assert f.debug_get("split_form") == split          # (the plan in use: rank-aware once the null directions are known)
            X, S = f.get_state()
            assert f.debug_get("gmw_aborts") == 0 and f.debug_get("gmw_shared") == 0
            res.append((np.vstack(tr), X, S))
            f.close()
        finally:
            srukf.debug_set_global("mem_split", 1)
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1]) and np.array_equal(res[0][2], res[1][2])
    assert np.abs(res[0][0][-1, :2] - sc["odo"][F, :2]).max() < 5e-3      # and it tracks


@pytest.mark.parametrize("N,storage", [(400, "f64"), (500, "f32")])
def test_split_fold_forms_the_same_tiles_as_k_syrk(srukf, synth, N, storage):
    """Round 6: in the rank-aware replay of the split form the tile launch forms the tiles of S^T S - U U^T itself (k_gmw_tiles_fold: forming jobs in k_syrk's summation
    order, row by row in front of the row's tile workgroups) while the pivot chain is already running; srukf_debug_set "split_fold" 0 is the k_syrk launch over the kept
    rows in front of the pair.  Same state and trajectory bit for bit, nothing abandoned, and the switch took effect ("split_fold_seqs")."""
    p = synth.scene_params()
    F = 12
    sc = synth.make_scene(N, F, seed=23, p=p)
    res = []
    for fold in (1, 0):
        f = srukf.Filter(N, p)
        if storage == "f32": f.set_storage(srukf.STORAGE_F32)
        f.debug_set("split_fold", fold)
        f.set_state(sc["X0"], sc["S0"]); f.stage_sequence(sc["odo"], sc["z"], sc["matched"])
        tr = [f.run_frames(0, 3), f.run_frames(3, F - 3)]
        if not f.debug_get("split_form"):
            f.close(); pytest.skip("this plan does not run the split form here")
        assert (f.debug_get("split_fold_seqs") > 0) == (fold == 1)
        assert f.debug_get("gmw_aborts") == 0 and f.debug_get("clamp_rows") == 0 and f.debug_get("exact_frames") == 0
        X, S = f.get_state()
        res.append((np.vstack(tr), X, S))
        f.close()
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1]) and np.array_equal(res[0][2], res[1][2])


def test_split_form_without_its_tile_launch_falls_back(srukf, synth):
    """The split form whose tile launch never arrives (srukf_debug_starve_workers: as if somebody else held the GPU) must not hang: the bounded waits of the pivot and of
    the slab workgroups expire, the frame is flagged and repeated on the exact path, the filter goes on with one launch per panel — and ends where an undisturbed filter ends
    (P to 1e-11: the repeated frame's exact column path rounds differently)."""
    p = synth.scene_params()
    N, F = 400, 5
    sc = synth.make_scene(N, F, seed=22, p=p)
    res = []
    for starve in (True, False):
        f = srukf.Filter(N, p)
        f.set_state(sc["X0"], sc["S0"]); f.stage_sequence(sc["odo"], sc["z"], sc["matched"])
        tr = [f.run_frames(0, 1)]
        assert f.debug_get("split_form") == 1
        if starve: f.debug_starve_workers(True)
        tr.append(f.run_frames(1, 2))
        if starve:
            assert f.debug_get("gmw_shared") == 2              # fell back to per-panel launches
            f.debug_starve_workers(False)
        tr.append(f.run_frames(3, F - 3))
        X, S = f.get_state()
        res.append((np.vstack(tr), X, S.T @ S))
        f.close()
    np.testing.assert_allclose(res[0][0][:, :4], res[1][0][:, :4], rtol=0, atol=1e-9)
    np.testing.assert_allclose(res[0][1], res[1][1], rtol=0, atol=1e-9)
    np.testing.assert_allclose(res[0][2], res[1][2], rtol=0, atol=1e-11)


def test_batched_filters_at_n400_against_the_split_form_run_alone(srukf, synth):
    """Beyond two register tiles per worker a filter run ALONE factors with the split form (k_gmw_pivslab_persist + k_gmw_tiles_persist) behind a split-K k_syrk over the
    kept rows; the batched replay forms S^T S - U U^T in the owners' summation order (k_syrk_b + k_syrk_own_b: what makes it bit-identical to solo runs where the solo
    path's owners fold, N <= 260) and factors with its per-panel launches.  Different summation order of the contraction, same factorisation arithmetic: the same
    filter to rounding (X 1e-9, P 1e-11: the per-frame parity bounds), nothing abandoned — several split-form contexts alive in one process included."""
    N, F, B = 400, 6, 2
    p = synth.scene_params()
    scs = [synth.make_scene(N, F, seed=0, p=p, obs_seed=7100 + b) for b in range(B)]
    fs = []
    for sc in scs:
        f = srukf.Filter(N, p); f.set_state(sc["X0"], sc["S0"]); f.stage_sequence(sc["odo"], sc["z"], sc["matched"]); fs.append(f)
    tb = srukf.run_frames_batch(fs, 0, F)
    for b, sc in enumerate(scs):
        assert fs[b].debug_get("gmw_aborts") == 0 and fs[b].debug_get("clamp_rows") == 0 and fs[b].debug_get("gmw_shared") == 0
        g = srukf.Filter(N, p); g.set_state(sc["X0"], sc["S0"]); g.stage_sequence(sc["odo"], sc["z"], sc["matched"])
        ts = g.run_frames(0, F)
        assert g.debug_get("split_form") == 1 and g.debug_get("gmw_shared") == 0
        Xs, Ss = g.get_state(); Xb, Sb = fs[b].get_state()
        np.testing.assert_allclose(tb[b][:, :4], ts[:, :4], rtol=0, atol=1e-9)
        np.testing.assert_allclose(tb[b][:, 4:], ts[:, 4:], rtol=0, atol=1e-12)
        np.testing.assert_allclose(Xb, Xs, rtol=0, atol=1e-9)
        np.testing.assert_allclose(Sb.T @ Sb, Ss.T @ Ss, rtol=0, atol=1e-11)
        g.close()
    for f in fs:
        f.close()


@pytest.mark.parametrize("n", [40, 300])
def test_exact_path_forms_under_active_theta_clamps(srukf, oracle, n):
    """The exact path (what a flagged frame is repeated on) in its three forms — left-looking k_gmw_col (srukf_debug_set "exact_rl" 0), right-looking with one pivot per launch
    (2) and with 8 pivots per launch (default, round 6) — on a matrix whose tiny diagonal entries under O(1) off-diagonals make the reference's theta clamp
    (SLAM.cpp:2279-2285) win at dozens of pivots: the same clamp count as the oracle's modifiedCholeskyDecomposition, P = S^T S within 1e-13 of the matrix' scale, and the two
    right-looking forms bit for bit (same operations on every element in the same order)."""
    rng = np.random.default_rng(5)
    A = rng.standard_normal((n + 10, n)); G = A.T @ A
    k = n // 3
    for kk in (k, k + 7, 2 * k):
        G[kk, kk] = 1e-7
    G[:, n - 5:] *= 1e-4; G[n - 5:, :] *= 1e-4
    So, Do, _, ce, ct = oracle.gmw(G)
    assert ct > 10                                                       # the theta clamp is active at many pivots
    Po, out = So.T @ So, {}
    try:
        for name, v in (("left", 0), ("right_1", 2), ("right_8", 1)):
            srukf.debug_set_global("exact_rl", v)
            out[name] = srukf.gmw(G, force_slow=True)
    finally:
        srukf.debug_set_global("exact_rl", 1)
    scale = np.abs(Po).max()
    for name, (S, D, hit) in out.items():
        assert hit == ct, (name, hit, ct)
        assert np.abs(S.T @ S - Po).max() <= 1e-13 * scale and np.abs(D - Do).max() <= 1e-12 * np.abs(Do).max(), name
    assert np.array_equal(out["right_1"][0], out["right_8"][0]) and np.array_equal(out["right_1"][1], out["right_8"][1])
