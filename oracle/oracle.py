"""ctypes binding of the CPU oracle (oracle/libsrukf_oracle.so).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product (cv-monoslam_amd/) never imports this.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libsrukf_oracle.so")

_DBL_FIELDS = ["cam_dx", "cam_dy", "cam_cx", "cam_cy", "cam_k1", "cam_k2", "cam_f", "image_w", "image_h",
               "a1", "a2", "a3", "a4", "sigma_measure", "rho0", "sigma_rho", "sigma_x", "sigma_y", "sigma_z",
               "sigma_theta", "epsilon", "ut_alpha", "ut_beta"]
_INT_FIELDS = ["weight_type", "noise_type", "newton_iters", "reserved_"]


class Params(C.Structure):
    """struct srukf_params (include/srukf.h)."""
    _fields_ = [(k, C.c_double) for k in _DBL_FIELDS] + [(k, C.c_int) for k in _INT_FIELDS]

    @classmethod
    def from_dict(cls, d):
        p = cls()
        for k in _DBL_FIELDS:
            setattr(p, k, float(d[k]))
        for k in _INT_FIELDS:
            setattr(p, k, int(d.get(k, 0)))
        return p


def build(force=False):
    srcs = [os.path.join(_HERE, f) for f in ("srukf_oracle.c", "srukf_matched.c")]
    libs = [_LIB_PATH, _MATCHED_PATH]
    if force or any(not os.path.exists(l) or os.path.getmtime(l) < max(os.path.getmtime(s) for s in srcs) for l in libs):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB_PATH


_MATCHED_PATH = os.path.join(_HERE, "libsrukf_matched.so")
_lib = None
_mlib = None
_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)


def _d(a):
    return a.ctypes.data_as(_dp) if a is not None else None


def _i(a):
    return a.ctypes.data_as(_ip) if a is not None else None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        L.orc_create.restype = C.c_void_p
        L.orc_create.argtypes = [C.c_int, C.POINTER(Params)]
        L.orc_destroy.argtypes = [C.c_void_p]
        L.orc_set_state.argtypes = [C.c_void_p, _dp, _dp]
        L.orc_get_state.argtypes = [C.c_void_p, _dp, _dp]
        L.orc_set_newton_early_exit.argtypes = [C.c_void_p, C.c_int]
        L.orc_set_frozen_center.argtypes = [C.c_void_p, C.c_int]
        L.orc_get_clamp_stats.argtypes = [C.c_void_p, C.POINTER(C.c_longlong)]
        L.orc_predict_motion.argtypes = [C.c_void_p, _dp, _dp]
        L.orc_predict_measurement.argtypes = [C.c_void_p, _dp, _dp, _ip]
        L.orc_update.argtypes = [C.c_void_p, _dp, _ip, C.c_int, C.c_int, C.c_int]
        L.orc_run_frames.argtypes = [C.c_void_p, C.c_int, _dp, _dp, _ip, C.c_int, _dp]
        L.orc_time_refactor_columns.argtypes = [C.c_void_p, C.c_int]
        L.orc_qr_r.argtypes = [_dp, C.c_int, C.c_int, _dp]
        L.orc_gmw.argtypes = [_dp, C.c_int, C.c_double, _dp, _dp, _dp, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]
        L.orc_sts.argtypes = [_dp, C.c_int, _dp]
        L.orc_project.argtypes = [C.POINTER(Params), C.c_int, _dp, _dp, _dp, _dp, _dp, C.c_int]
        L.orc_sample_parameter.argtypes = [C.c_int, C.c_int, C.c_double, C.c_double, _dp]
        L.orc_joint_init.argtypes = [C.POINTER(Params), C.c_int, _dp, _dp, C.c_int, _dp, _dp, _dp]
        L.orc_delete_feature.argtypes = [C.POINTER(Params), C.c_int, _dp, _dp, C.c_int, _dp, _dp]
        _bp = C.POINTER(C.c_ubyte)
        L.orc_warp_patch.argtypes = [C.POINTER(Params), _dp, _dp, _dp, _dp, _dp, _dp, _bp, _bp]
        L.orc_warp_patch.restype = None
        L.orc_associate_one.argtypes = [C.POINTER(Params), _bp, _dp, _dp, _bp, _dp, _dp]
        L.orc_sigma_ptr.restype = _dp
        L.orc_sigma_ptr.argtypes = [C.c_void_p]
        L.orc_Z_ptr.restype = _dp
        L.orc_Z_ptr.argtypes = [C.c_void_p]
        _lib = L
    return _lib


def _c(a, dtype=np.float64):
    return np.ascontiguousarray(a, dtype=dtype)


class Oracle:
    """One CPU filter (CSLAM numeric state) for N landmarks."""

    SEQUENTIAL, BATCHED = 0, 1
    NEED_REORDER, NEEDNOT_REORDER = 0, 1

    def __init__(self, N, params):
        self.N, self.n = N, 6 * N + 4
        self.params = Params.from_dict(params)
        self._h = lib().orc_create(N, C.byref(self.params))

    def close(self):
        if self._h:
            lib().orc_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_state(self, X, S):
        X, S = _c(X), _c(S)
        assert X.shape == (self.n,) and S.shape == (self.n, self.n)
        lib().orc_set_state(self._h, _d(X), _d(S))

    def get_state(self):
        X = np.empty(self.n)
        S = np.empty((self.n, self.n))
        lib().orc_get_state(self._h, _d(X), _d(S))
        return X, S

    def set_newton_early_exit(self, on):
        lib().orc_set_newton_early_exit(self._h, int(on))

    def set_frozen_center(self, on):
        """Test knob (not the reference): centre the cross covariance on the state at the start of the update."""
        lib().orc_set_frozen_center(self._h, int(on))

    def clamp_stats(self):
        out = (C.c_longlong * 3)()
        lib().orc_get_clamp_stats(self._h, out)
        return {"eps": out[0], "theta": out[1], "pivots": out[2]}

    def predict_motion(self, odo_prev, odo_cur):
        a, b = _c(odo_prev), _c(odo_cur)
        rc = lib().orc_predict_motion(self._h, _d(a), _d(b))
        assert rc == 0, rc

    def predict_measurement(self):
        h = np.empty(2 * self.N)
        Si = np.empty((self.N, 2, 2))
        vis = np.empty(self.N, dtype=np.int32)
        rc = lib().orc_predict_measurement(self._h, _d(h), _d(Si), _i(vis))
        assert rc == 0, rc
        return h, Si, vis

    def update(self, z, matched, reorder=1, k_new=0, mode=0):
        z, m = _c(z), _c(matched, np.int32)
        rc = lib().orc_update(self._h, _d(z), _i(m), reorder, k_new, mode)
        assert rc == 0, rc

    def run_frames(self, odo, z, matched, mode=0):
        odo, z, m = _c(odo), _c(z), _c(matched, np.int32)
        F = z.shape[0]
        traj = np.empty((F, 8))
        rc = lib().orc_run_frames(self._h, F, _d(odo), _d(z), _i(m), mode, _d(traj))
        assert rc == 0, rc
        return traj

    def time_refactor_columns(self, cols):
        lib().orc_time_refactor_columns(self._h, cols)

    def sigma(self):
        Na = self.n + 5
        L = 2 * Na + 1
        return np.ctypeslib.as_array(lib().orc_sigma_ptr(self._h), shape=(Na, L)).copy()

    def Z(self):
        Na = self.n + 5
        L = 2 * Na + 1
        return np.ctypeslib.as_array(lib().orc_Z_ptr(self._h), shape=(2 * self.N, L)).copy()


def qr_r(A):
    A = _c(A)
    m, k = A.shape
    R = np.zeros((k, k))
    lib().orc_qr_r(_d(A), m, k, _d(R))
    return R


def gmw(G, eps=1e-13):
    G = _c(G)
    n = G.shape[0]
    S, D, L = np.zeros((n, n)), np.zeros(n), np.zeros((n, n))
    ce, ct = C.c_longlong(0), C.c_longlong(0)
    lib().orc_gmw(_d(G), n, eps, _d(S), _d(D), _d(L), C.byref(ce), C.byref(ct))
    return S, D, L, ce.value, ct.value


def sts(S):
    S = _c(S)
    P = np.zeros_like(S)
    lib().orc_sts(_d(S), S.shape[0], _d(P))
    return P


def project(params, feat6, pos3, psi, err2, early_exit=1):
    p = Params.from_dict(params)
    feat6, pos3, psi, err2 = _c(feat6).reshape(-1, 6), _c(pos3).reshape(-1, 3), _c(psi).reshape(-1), _c(err2).reshape(-1, 2)
    out = np.zeros((feat6.shape[0], 2))
    lib().orc_project(C.byref(p), feat6.shape[0], _d(feat6), _d(pos3), _d(psi), _d(err2), _d(out), early_exit)
    return out


def sample_parameter(Na, weight_type=0, alpha=1e-3, beta=2.0):
    out = np.zeros(7)
    lib().orc_sample_parameter(Na, weight_type, alpha, beta, _d(out))
    return dict(zip(["wm0", "wc0", "wi", "wi_sr", "gamma", "wm0_sr", "wc0_sr"], out))


def warp_patch(params, robot, initR, initT, initPixel, xyz, predict, initPatch, matchPatch):
    """wrapPatch for one landmark; returns the new 17x17 matchPatch (the input one is not modified)."""
    p = Params.from_dict(params)
    ip = np.ascontiguousarray(initPatch, dtype=np.uint8); assert ip.shape == (21, 21)
    mp = np.array(matchPatch, dtype=np.uint8, copy=True, order="C"); assert mp.shape == (17, 17)
    bp = C.POINTER(C.c_ubyte)
    lib().orc_warp_patch(C.byref(p), _d(_c(robot)), _d(_c(initR).reshape(9)), _d(_c(initT)), _d(_c(initPixel)), _d(_c(xyz)), _d(_c(predict)),
                         ip.ctypes.data_as(bp), mp.ctypes.data_as(bp))
    return mp


def associate_one(params, image, predict, Si, matchPatch):
    """dataAssociation for one visible landmark; returns (matched, best correlation, match location[2])."""
    p = Params.from_dict(params)
    img = np.ascontiguousarray(image, dtype=np.uint8); mp = np.ascontiguousarray(matchPatch, dtype=np.uint8)
    bp = C.POINTER(C.c_ubyte)
    best, loc = np.zeros(1), np.zeros(2)
    ok = lib().orc_associate_one(C.byref(p), img.ctypes.data_as(bp), _d(_c(predict)), _d(_c(Si).reshape(4)), mp.ctypes.data_as(bp), _d(best), _d(loc))
    return bool(ok), float(best[0]), loc


def delete_feature(params, X, S, idx):
    """deleteOneFeature, numeric part: (X_new, S_new) without landmark idx (0-based state order)."""
    p = Params.from_dict(params)
    X, S = _c(X), _c(S)
    dim = X.shape[0]
    Xn, Sn = np.zeros(dim - 6), np.zeros((dim - 6, dim - 6))
    rc = lib().orc_delete_feature(C.byref(p), dim, _d(X), _d(S), int(idx), _d(Xn), _d(Sn))
    assert rc == 0
    return Xn, Sn


def joint_init(params, X, S, uv):
    p = Params.from_dict(params)
    X, S, uv = _c(X), _c(S), _c(uv).reshape(-1, 2)
    dim, K = X.shape[0], uv.shape[0]
    Xn, Sn = np.zeros(dim + 6 * K), np.zeros((dim + 6 * K, dim + 6 * K))
    rc = lib().orc_joint_init(C.byref(p), dim, _d(X), _d(S), K, _d(uv), _d(Xn), _d(Sn))
    assert rc == 0
    return Xn, Sn


def matched_lib():
    """libsrukf_matched.so: the algorithm-matched, OpenMP CPU baseline (srukf_matched.c)."""
    global _mlib
    if _mlib is None:
        build()
        L = C.CDLL(_MATCHED_PATH)
        L.mt_create.restype = C.c_void_p
        L.mt_create.argtypes = [C.c_int, C.POINTER(Params), C.c_int]
        L.mt_destroy.argtypes = [C.c_void_p]
        L.mt_set_state.argtypes = [C.c_void_p, _dp, _dp]
        L.mt_get_state.argtypes = [C.c_void_p, _dp, _dp]
        L.mt_frame.argtypes = [C.c_void_p, _dp, _dp, _dp, _ip]
        L.mt_run_frames.argtypes = [C.c_void_p, C.c_int, _dp, _dp, _ip, _dp]
        L.mt_clamp_fallbacks.restype = C.c_longlong
        L.mt_clamp_fallbacks.argtypes = [C.c_void_p]
        L.mt_phase_times.argtypes = [C.c_void_p, _dp]
        L.mt_isa_name.restype = C.c_char_p
        L.mt_set_rank_aware.argtypes = [C.c_void_p, C.c_int]
        L.mt_null_directions.argtypes = [C.c_void_p]
        L.mt_rank_fallbacks.restype = C.c_longlong
        L.mt_rank_fallbacks.argtypes = [C.c_void_p]
        _mlib = L
    return _mlib


class Matched:
    """The B-matched CPU baseline: same formulation as the GPU path (batched refactor, structured motion update),
    OpenMP on `threads` cores (0 = the OpenMP default, i.e. all)."""

    def __init__(self, N, params, threads=0):
        self.N, self.n = N, 6 * N + 4
        self._p = Params.from_dict(params)
        self._h = matched_lib().mt_create(N, C.byref(self._p), int(threads))
        self.threads = matched_lib().mt_threads()
        self.isa = matched_lib().mt_isa_name().decode()

    def close(self):
        if getattr(self, "_h", None):
            matched_lib().mt_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_state(self, X, S):
        X, S = _c(X), _c(S)
        matched_lib().mt_set_state(self._h, _d(X), _d(S))

    def get_state(self):
        X, S = np.empty(self.n), np.empty((self.n, self.n))
        matched_lib().mt_get_state(self._h, _d(X), _d(S))
        return X, S

    def run_frames(self, odo, z, matched):
        odo, z, m = _c(odo), _c(z), _c(matched, np.int32)
        F = z.shape[0]
        traj = np.empty((F, 8))
        rc = matched_lib().mt_run_frames(self._h, F, _d(odo), _d(z), _i(m), _d(traj))
        if rc != 0:
            raise RuntimeError(f"mt_run_frames rc={rc}")
        return traj

    def clamp_fallbacks(self):
        return int(matched_lib().mt_clamp_fallbacks(self._h))

    def set_rank_aware(self, on=True):
        """The GPU path's rank-aware refactorisation (structurally null pivots skipped, K <= r in the contractions, NullSkip in the projection) for the
        CPU port as well: takes the null set from the CURRENT state (call after set_state).  Returns the number of skipped directions."""
        return int(matched_lib().mt_set_rank_aware(self._h, 1 if on else 0))

    def null_directions(self):
        return int(matched_lib().mt_null_directions(self._h))

    def rank_fallbacks(self):
        """Frames of the rank-aware form that were repeated on the full-rank path (a skipped direction not null, or the theta clamp)."""
        return int(matched_lib().mt_rank_fallbacks(self._h))

    def phase_times(self):
        """Seconds spent so far in (motion, measurement, gains, S^T S - U U^T, factorisation, exact fallback)."""
        t = np.zeros(6)
        matched_lib().mt_phase_times(self._h, _d(t))
        return dict(zip(("motion", "measure", "gain", "syrk", "gmw", "fallback"), t.tolist()))
