"""CPU tests of the host side: the C-ABI library loads, exports every symbol include/srukf.h
declares, refuses to run without a GPU (no fallback), and the scene generator is deterministic."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions():
    txt = open(os.path.join(ROOT, "include", "srukf.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(srukf_[a-z_0-9]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol(pkg):
    lib = pkg.srukf.load_library()
    names = _header_functions()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), f"libsrukf_hip.so does not export {n}"
    assert sorted(pkg.srukf.EXPORTS) == names
    assert lib.srukf_abi_version() == 6


def test_default_params_match_reference_constants(pkg):
    d = pkg.srukf.default_params()
    ref = pkg.synth.default_params()
    for k, v in ref.items():
        assert d[k] == pytest.approx(v), k
    assert d["a1"] == 8 and d["epsilon"] == 1e-13 and d["newton_iters"] == 100     # SLAM.cpp:195, 52, 3186
    assert d["cam_f"] / d["cam_dx"] == pytest.approx(776.25, abs=0.01)             # f1 = f/dx, SLAM.cpp:336


def test_params_struct_layout_matches_oracle_binding(pkg, oracle):
    assert C.sizeof(pkg.srukf.Params) == C.sizeof(oracle.Params) == 23 * 8 + 4 * 4
    assert [f[0] for f in pkg.srukf.Params._fields_] == [f[0] for f in oracle.Params._fields_]


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="a GPU is present")
def test_no_gpu_fails_loudly(pkg):
    """The product path has no CPU fallback: without a device, create() reports NO_DEVICE."""
    with pytest.raises(pkg.srukf.SrukfError) as e:
        pkg.srukf.Filter(8, pkg.synth.scene_params())
    assert e.value.rc == -4
    with pytest.raises(pkg.srukf.SrukfError):
        pkg.srukf.gmw(np.eye(4))


def test_bad_arguments(pkg):
    lib = pkg.srukf.load_library()
    assert lib.srukf_default_params(None) == -1
    h = C.c_void_p()
    p = pkg.srukf.Params.from_dict(pkg.synth.scene_params())
    assert lib.srukf_create(C.byref(h), -1, C.byref(p), 0, None) == -1      # N = 0 is legal: the robot block only
    p.noise_type = 1
    assert lib.srukf_create(C.byref(h), 4, C.byref(p), 0, None) == -6            # random-noise models unsupported
    assert lib.srukf_destroy(None) == 0
    assert lib.srukf_update(None, None, None, 1, 1) == -1


def test_scene_is_deterministic_and_inside_the_image(synth):
    a = synth.make_scene(20, 30, seed=7)
    b = synth.make_scene(20, 30, seed=7)
    for k in ("X0", "S0", "odo", "z"):
        assert np.array_equal(a[k], b[k])
    c = synth.make_scene(20, 30, seed=7, obs_seed=3)
    assert np.array_equal(a["X0"], c["X0"]) and not np.array_equal(a["z"], c["z"])   # shared map, own noise
    # measurements stay clear of the 10-px zeroing border (SLAM.cpp:3341) and 20-px deletion border (2443-2446)
    zx, zy = a["z"][:, 0::2], a["z"][:, 1::2]
    assert zx.min() > 40 and zx.max() < 600 and zy.min() > 40 and zy.max() < 440
    # heading never crosses +-pi (no angle wrap in SLAM.cpp:1448)
    assert np.abs(a["odo"][:, 2]).max() < 2.5
    # S0 upper triangular, rank-deficient
    assert np.allclose(np.tril(a["S0"], -1), 0)
    assert np.linalg.matrix_rank(a["S0"].T @ a["S0"], tol=1e-12) == 4 + 3 * 20


def test_figure8_closes(synth):
    odo = synth.figure8_odometry(600)
    assert np.abs(odo[600, :2] - odo[0, :2]).max() < 0.05
    assert np.ptp(odo[:, 0]) < 0.12 and np.ptp(odo[:, 1]) < 0.3


def test_device_sincos_header_is_correctly_rounded(tmp_path):
    """srukf_crtrig.h (what k_warp_patch uses for the heading's cos / sin) against binary128 values rounded once — the
    definition orc_warp_patch uses.  Same source, +, -, *, fma only, so the device computes the same bits.  The host
    libm (glibc, < 0.55 ulp) is NOT correctly rounded: about 0.15 % of arguments differ, which is why neither side
    of the byte-exact wrapPatch comparison takes its cos / sin from a libm."""
    import ctypes as C
    import subprocess
    so = str(tmp_path / "crtrig_host.so")
    subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-mfma", "-shared", "-fPIC", "-o", so,
                           os.path.join(ROOT, "tests", "crtrig_host.c"), "-lquadmath", "-lm"])
    L = C.CDLL(so)
    L.crt_sweep.argtypes = [C.c_long, C.c_double, C.c_uint64, C.POINTER(C.c_long)]
    for n, rng in [(400000, 3.2), (100000, 50.0), (100000, 1e-3), (100000, 1e5)]:
        out = (C.c_long * 4)()
        L.crt_sweep(n, rng, 12345, out)
        assert out[0] == 0 and out[1] == 0, (rng, list(out))
    s, c = C.c_double(), C.c_double()
    L.crt_sincos_host.argtypes = [C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    for x in [0.0, -0.0, np.pi / 4, -np.pi / 2, np.pi, 1e-300, 2.0 ** 20]:
        L.crt_sincos_host(x, C.byref(s), C.byref(c))
        assert abs(s.value - np.sin(x)) <= 2 * np.spacing(abs(np.sin(x))) and abs(c.value - np.cos(x)) <= 2 * np.spacing(abs(np.cos(x)))


def test_pxy2_tile_list_covers_every_tile_once(pkg):
    """Host-side tile list of k_pxy2 (srukf_pxy2_build_tiles): every (measurement tile, state block, K half) exactly once, for matrices up to
    200 block columns (N = 2 130; round 3 kept the (block, half) pairs in a fixed array of 128 and silently dropped the lowest blocks from N = 693)."""
    lib = pkg.srukf.load_library()
    fn = lib.srukf_pxy2_build_tiles
    fn.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p]
    split = lib.srukf_pxy2_split_groups()
    for T in (1, 2, 5, 19, 47, 64, 65, 66, 67, 100, 128, 200):
        np_, mp = 64 * T, 64 * max(1, T // 3)
        for kr in sorted({16, np_ // 2 // 16 * 16 or 16, np_}):
            cnt = fn(mp, np_, kr, None)
            out = np.zeros(4 * cnt, dtype=np.int32)
            assert fn(mp, np_, kr, out.ctypes.data) == cnt and cnt % 8 == 0
            tl = out.reshape(-1, 4)
            live = tl[tl[:, 0] >= 0]
            seen = {}
            for mt, bt, h, halves in live:
                ng = min(4 * (bt + 1), (kr + 63) // 64 * 4)
                assert halves == (2 if ng >= split else 1) and 0 <= h < halves and 0 <= mt < mp // 64 and 0 <= bt < T
                seen[(mt, bt, h)] = seen.get((mt, bt, h), 0) + 1
            want = sum((2 if min(4 * (bt + 1), (kr + 63) // 64 * 4) >= split else 1) for bt in range(T)) * (mp // 64)
            assert len(seen) == want == len(live) and set(seen.values()) == {1}, (T, kr)
            # all tiles that contract the same slab of A (one (bt, half) pair) sit on ONE XCD (slot index % 8)
            xcd = {}
            for slot, (mt, bt, h, halves) in enumerate(tl):
                if mt >= 0:
                    assert xcd.setdefault((bt, h), slot % 8) == slot % 8


def test_split_fold_grid_forms_every_tile_before_its_owner_waits_for_it(pkg):
    """Grid of k_gmw_tiles_fold (srukf_gmw_build_fold_list, round 6): for every plan shape — every 32 x 32 tile of the pivoted block rows >= 1 is formed exactly once, by a job of
    the grid or by the k_syrk launch in front (srukf_gmw_fold_head_tile); every tile workgroup of the split form's list appears exactly once and BEHIND all the jobs that form its
    quarters (dispatch is in grid order: what it waits for is never behind it); segments are multiples of 8 entries and the jobs of one column pair sit on one XCD per segment."""
    lib = pkg.srukf.load_library()
    build, tiles = lib.srukf_gmw_build_fold_list, lib.srukf_gmw_build_tiles
    build.argtypes = [C.c_int, C.c_int, C.c_void_p]; tiles.argtypes = [C.c_int, C.c_int, C.c_void_p]; head_t = lib.srukf_gmw_fold_head_tile; head_t.argtypes = [C.c_int, C.c_int, C.c_int]
    rows = lib.srukf_gmw_fold_head_rows; rows.argtypes = [C.c_int]
    for T, Tp in [(19, 10), (38, 19), (47, 24), (47, 47), (57, 29), (76, 38), (8, 3), (5, 5)]:
        hr = rows(Tp)
        assert 1 <= hr <= max(1, Tp - 4)
        head = lambda tr, tc: head_t(tr, tc, hr)
        cnt = build(T, Tp, None)
        out = np.zeros(4 * max(cnt, 1), dtype=np.int16)
        assert build(T, Tp, out.ctypes.data) == cnt and cnt % 8 == 0
        ent = out.reshape(-1, 4)[:cnt]
        nt = tiles(T, Tp, None)
        tk = np.zeros(4 * max(nt, 1), dtype=np.int16); tiles(T, Tp, tk.ctypes.data)
        nreal = nt - (T - Tp if Tp < T else 0)
        want_tiles = {(int(a), int(b)): int(c) for a, b, c, _ in tk.reshape(-1, 4)[:nreal]}
        Ilast = Tp - 1 if Tp < T else T - 1
        want_jobs = {(tr, tc) for I in range(1, Ilast + 1) for tr in (2 * I, 2 * I + 1) for tc in range(tr, 2 * T) if not head(tr, tc)}
        jobs, owners = {}, {}
        for pos, (a, b, c, _) in enumerate(ent):
            if c == -1:
                assert (int(a), int(b)) not in jobs
                jobs[(int(a), int(b))] = pos
            elif c >= 0:
                assert (int(a), int(b)) not in owners and want_tiles.get((int(a), int(b))) == int(c)
                owners[(int(a), int(b))] = pos
            else:
                assert c == -2
        assert set(jobs) == want_jobs and set(owners) == set(want_tiles), (T, Tp)
        for (I, J), pos in owners.items():
            for tr in (2 * I, 2 * I + 1):
                for tc in (2 * J, 2 * J + 1):
                    if tc >= tr and not head(tr, tc):
                        assert jobs[(tr, tc)] < pos, (T, Tp, I, J)
        # block row 0 and tile (1, 1) stay with k_syrk: what the pivot and the slab workgroups read without a version
        assert all(head(tr, tc) for tr in (0, 1) for tc in range(tr, 2 * T)) and head(2, 2) and head(2, 3) and head(3, 3) and (hr > 1 or not head(2, 4))


@pytest.mark.parametrize("exe", ["cslam_replay.bin", "cslam_step_bench.bin", "cslam_replay_multi.bin"])
def test_cpp_hosts_are_built_and_print_their_usage(exe):
    """The C++ hosts of cv-monoslam_amd/host (the reference's language) link against the in-tree libraries and start without a GPU: no arguments -> usage, exit code 2."""
    import subprocess
    path = os.path.join(ROOT, "cv-monoslam_amd", exe)
    assert os.path.exists(path), "run __graft_entry__.build() first"
    r = subprocess.run([path], capture_output=True, text=True, timeout=60)
    assert r.returncode == 2 and "usage" in r.stderr
