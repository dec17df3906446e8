"""ctypes binding of the C-ABI in include/srukf.h (libsrukf_hip.so, gfx950 kernels).

This is host plumbing only: every numeric result comes from the HIP kernels.  There is no CPU
fallback — if the library or a gfx950 device is missing, construction raises.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libsrukf_hip.so")

_DBL_FIELDS = ["cam_dx", "cam_dy", "cam_cx", "cam_cy", "cam_k1", "cam_k2", "cam_f", "image_w", "image_h",
               "a1", "a2", "a3", "a4", "sigma_measure", "rho0", "sigma_rho", "sigma_x", "sigma_y", "sigma_z",
               "sigma_theta", "epsilon", "ut_alpha", "ut_beta"]
_INT_FIELDS = ["weight_type", "noise_type", "newton_iters", "reserved_"]

# names of every entry point declared in include/srukf.h (checked by the CPU test-suite)
EXPORTS = [
    "srukf_abi_version", "srukf_default_params", "srukf_create", "srukf_destroy", "srukf_reset", "srukf_last_error",
    "srukf_set_state", "srukf_get_state", "srukf_set_state_device", "srukf_get_state_device", "srukf_get_robot",
    "srukf_get_landmark_block", "srukf_get_landmarks_cartesian", "srukf_get_frame_view", "srukf_get_covariance", "srukf_predict_motion", "srukf_predict_motion_next", "srukf_predict_measurement",
    "srukf_update", "srukf_set_new_landmarks", "srukf_add_landmarks", "srukf_delete_landmark", "srukf_set_storage", "srukf_set_exclusive", "srukf_set_rank_aware", "srukf_null_directions", "srukf_run_frames_batch", "srukf_prepare_frames", "srukf_debug_poke_state", "srukf_get_state_f32", "srukf_set_landmark_appearance", "srukf_associate", "srukf_get_match_patch", "srukf_stage_sequence", "srukf_run_frames_async", "srukf_run_frames", "srukf_synchronize", "srukf_set_profiling",
    "srukf_clamp_info", "srukf_debug_set", "srukf_debug_get", "srukf_debug_copy", "srukf_debug_upload", "srukf_debug_split_replay", "srukf_debug_gmw_stamps", "srukf_debug_starve_workers", "srukf_debug_allow_mixed", "srukf_profile_count", "srukf_profile_get", "srukf_profile_reset", "srukf_dims", "srukf_gmw_host",
    "srukf_project_host",
]

STATUS = {0: "SRUKF_OK", -1: "SRUKF_ERR_BAD_ARG", -2: "SRUKF_ERR_DIM_MISMATCH", -3: "SRUKF_ERR_HIP",
          -4: "SRUKF_ERR_NO_DEVICE", -5: "SRUKF_ERR_SEQUENCE", -6: "SRUKF_ERR_UNSUPPORTED",
          -7: "SRUKF_ERR_CLAMP_PENDING", -8: "SRUKF_ERR_NOMEM"}

STORAGE_F64, STORAGE_F32, STORAGE_F32_MIXED = 0, 1, 2
GPU_SHARED, GPU_EXCLUSIVE, GPU_SHARED_PER_PANEL = 0, 1, 2          # srukf_set_exclusive
UPDATE_SEQUENTIAL, UPDATE_BATCHED = 0, 1
NEED_REORDER, NEEDNOT_REORDER = 0, 1


class Params(C.Structure):
    """struct srukf_params."""
    _fields_ = [(k, C.c_double) for k in _DBL_FIELDS] + [(k, C.c_int) for k in _INT_FIELDS]

    @classmethod
    def from_dict(cls, d):
        p = cls()
        for k in _DBL_FIELDS:
            setattr(p, k, float(d[k]))
        for k in _INT_FIELDS:
            setattr(p, k, int(d.get(k, 0)))
        return p

    def to_dict(self):
        return {k: getattr(self, k) for k in _DBL_FIELDS + _INT_FIELDS}


class SrukfError(RuntimeError):
    def __init__(self, rc, msg=""):
        self.rc = rc
        super().__init__(f"{STATUS.get(rc, rc)}: {msg}")


_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)
_lib = None


def load_library(path=None):
    """dlopen the in-tree libsrukf_hip.so; raises if it has not been built (no fallback).
    path: measurement scripts only (bench.py --lib, scripts/ab_bench.sh) — an A/B build of the same library, named on the FIRST call; the product reads no
    environment variable and loads nothing else."""
    global _lib, LIB_PATH
    if _lib is not None:
        if path and os.path.abspath(path) != os.path.abspath(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is already loaded; an A/B library must be named on the first load_library call")
        return _lib
    if path:
        LIB_PATH = path
    # PyTorch-ROCm bundles its own libamdhip64.so; if ours (linked against /opt/rocm) initialises
    # HIP first, a later `import torch` in the same process finds no GPUs.  Loading torch first
    # makes both share one runtime.  torch is plumbing here (streams / RCCL), never compute.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
    L = C.CDLL(LIB_PATH)
    L.srukf_last_error.restype = C.c_char_p
    L.srukf_last_error.argtypes = [C.c_void_p]
    L.srukf_default_params.argtypes = [C.POINTER(Params)]
    L.srukf_create.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.POINTER(Params), C.c_int, C.c_void_p]
    L.srukf_destroy.argtypes = [C.c_void_p]
    L.srukf_reset.argtypes = [C.c_void_p]
    L.srukf_set_state.argtypes = [C.c_void_p, _dp, _dp]
    L.srukf_get_state.argtypes = [C.c_void_p, _dp, _dp]
    L.srukf_set_state_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    L.srukf_get_state_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    L.srukf_get_robot.argtypes = [C.c_void_p, _dp, _dp]
    L.srukf_get_landmark_block.argtypes = [C.c_void_p, C.c_int, _dp, _dp]
    L.srukf_get_covariance.argtypes = [C.c_void_p, _dp]
    L.srukf_get_landmarks_cartesian.argtypes = [C.c_void_p, _dp, _dp]
    L.srukf_get_frame_view.argtypes = [C.c_void_p, _dp, _dp, _dp, _dp, _dp]
    L.srukf_predict_motion.argtypes = [C.c_void_p, _dp, _dp]
    L.srukf_predict_motion_next.argtypes = [C.c_void_p, _dp, _dp]
    L.srukf_predict_measurement.argtypes = [C.c_void_p, _dp, _dp, _ip]
    L.srukf_update.argtypes = [C.c_void_p, _dp, _ip, C.c_int, C.c_int]
    L.srukf_set_new_landmarks.argtypes = [C.c_void_p, C.c_int]
    L.srukf_add_landmarks.argtypes = [C.c_void_p, C.c_int, _dp]
    L.srukf_delete_landmark.argtypes = [C.c_void_p, C.c_int]
    _bp = C.POINTER(C.c_ubyte)
    L.srukf_set_landmark_appearance.argtypes = [C.c_void_p, C.c_int, _bp, _dp, _dp, _dp]
    L.srukf_associate.argtypes = [C.c_void_p, _bp, _dp, _ip, _dp]
    L.srukf_get_match_patch.argtypes = [C.c_void_p, C.c_int, _bp]
    L.srukf_set_storage.argtypes = [C.c_void_p, C.c_int]
    L.srukf_set_exclusive.argtypes = [C.c_void_p, C.c_int]
    L.srukf_set_rank_aware.argtypes = [C.c_void_p, C.c_int]
    L.srukf_prepare_frames.argtypes = [C.c_void_p, C.c_int]
    L.srukf_debug_poke_state.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_double]
    L.srukf_run_frames_batch.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int)]
    L.srukf_null_directions.argtypes = [C.c_void_p]
    L.srukf_get_state_f32.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.srukf_stage_sequence.argtypes = [C.c_void_p, C.c_int, _dp, _dp, _ip]
    L.srukf_run_frames_async.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
    L.srukf_run_frames.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, _dp]
    L.srukf_synchronize.argtypes = [C.c_void_p]
    L.srukf_clamp_info.argtypes = [C.c_void_p, _ip, _ip]
    L.srukf_debug_starve_workers.argtypes = [C.c_void_p, C.c_int]
    L.srukf_debug_set.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
    L.srukf_debug_get.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_longlong)]
    L.srukf_debug_copy.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_double), C.c_longlong]
    L.srukf_debug_upload.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_double), C.c_longlong]
    L.srukf_debug_split_replay.argtypes = [C.c_void_p, C.c_int, C.c_int]
    L.srukf_debug_gmw_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
    L.srukf_debug_allow_mixed.argtypes = [C.c_void_p, C.c_int]
    L.srukf_set_profiling.argtypes = [C.c_void_p, C.c_int]
    L.srukf_profile_count.argtypes = [C.c_void_p]
    L.srukf_profile_get.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_char_p), _dp, C.POINTER(C.c_longlong), _dp, _dp]
    L.srukf_profile_reset.argtypes = [C.c_void_p]
    L.srukf_dims.argtypes = [C.c_void_p, _ip, _ip, _ip, _ip]
    L.srukf_gmw_host.argtypes = [C.c_int, C.c_int, _dp, _dp, _dp, C.c_double, C.c_int, _ip]
    L.srukf_project_host.argtypes = [C.c_int, C.POINTER(Params), C.c_int, _dp, _dp, _dp, _dp, _dp]
    _lib = L
    return L


def _d(a):
    return a.ctypes.data_as(_dp) if a is not None else None


def _i(a):
    return a.ctypes.data_as(_ip) if a is not None else None


def _c(a, dtype=np.float64):
    return np.ascontiguousarray(a, dtype=dtype)


def default_params():
    p = Params()
    load_library().srukf_default_params(C.byref(p))
    return p.to_dict()


class Filter:
    """One SRUKF on one MI355X (mirror of the numeric side of CSLAM, SLAM.h:118-398)."""

    def __init__(self, n_landmarks, params, device=0, stream=None):
        self._lib = load_library()
        self._h = C.c_void_p()
        self.params = Params.from_dict(params)
        rc = self._lib.srukf_create(C.byref(self._h), n_landmarks, C.byref(self.params), device,
                                    C.c_void_p(stream) if stream else None)
        if rc != 0:
            msg = self._lib.srukf_last_error(None)
            self._h = None
            raise SrukfError(rc, msg.decode() if msg else "")
        self._refresh_dims()

    def _refresh_dims(self):
        N, n, Na, L = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        self._lib.srukf_dims(self._h, C.byref(N), C.byref(n), C.byref(Na), C.byref(L))
        self.N, self.n, self.Na, self.L = N.value, n.value, Na.value, L.value

    def _chk(self, rc):
        if rc != 0:
            msg = self._lib.srukf_last_error(self._h)
            raise SrukfError(rc, msg.decode() if msg else "")

    def close(self):
        if getattr(self, "_h", None):
            self._lib.srukf_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reset(self):
        self._chk(self._lib.srukf_reset(self._h))

    def set_state(self, X, S):
        X, S = _c(X), _c(S)
        assert X.shape == (self.n,) and S.shape == (self.n, self.n)
        self._chk(self._lib.srukf_set_state(self._h, _d(X), _d(S)))

    def get_state(self):
        X, S = np.empty(self.n), np.empty((self.n, self.n))
        self._chk(self._lib.srukf_get_state(self._h, _d(X), _d(S)))
        return X, S

    def set_state_device(self, dX_ptr, dS_ptr, ld):
        self._chk(self._lib.srukf_set_state_device(self._h, C.c_void_p(dX_ptr), C.c_void_p(dS_ptr), ld))

    def get_state_device(self, dX_ptr, dS_ptr, ld):
        self._chk(self._lib.srukf_get_state_device(self._h, C.c_void_p(dX_ptr), C.c_void_p(dS_ptr), ld))

    def get_robot(self):
        pose, P4 = np.empty(4), np.empty((4, 4))
        self._chk(self._lib.srukf_get_robot(self._h, _d(pose), _d(P4)))
        return pose, P4

    def get_landmark_block(self, k):
        x6, P = np.empty(6), np.empty((6, 6))
        self._chk(self._lib.srukf_get_landmark_block(self._h, k, _d(x6), _d(P)))
        return x6, P

    def get_covariance(self):
        P = np.empty((self.n, self.n))
        self._chk(self._lib.srukf_get_covariance(self._h, _d(P)))
        return P

    def predict_motion(self, odo_prev, odo_cur):
        a, b = _c(odo_prev), _c(odo_cur)
        self._chk(self._lib.srukf_predict_motion(self._h, _d(a), _d(b)))

    def predict_motion_next(self, odo_prev, odo_cur):
        """Look-ahead: the odometry pair of the NEXT frame (srukf_predict_motion_next)."""
        a, b = _c(odo_prev), _c(odo_cur)
        self._chk(self._lib.srukf_predict_motion_next(self._h, _d(a), _d(b)))

    def predict_measurement(self):
        h, Si, vis = np.empty(2 * self.N), np.empty((self.N, 2, 2)), np.empty(self.N, dtype=np.int32)
        self._chk(self._lib.srukf_predict_measurement(self._h, _d(h), _d(Si), _i(vis)))
        return h, Si, vis

    def update(self, z, matched, reorder=NEEDNOT_REORDER, mode=UPDATE_BATCHED):
        z, m = _c(z), _c(matched, np.int32)
        assert z.shape == (2 * self.N,) and m.shape == (self.N,)
        self._chk(self._lib.srukf_update(self._h, _d(z), _i(m), reorder, mode))

    def set_new_landmarks(self, K_new):
        """m_nFilters: the last K_new landmarks of the map were just added (NEED_REORDER updates use it)."""
        self._chk(self._lib.srukf_set_new_landmarks(self._h, int(K_new)))

    def add_landmarks(self, uv):
        """Joint initialisation of K new landmarks at distorted pixels uv[K][2]; the filter grows to N + K."""
        uv = _c(uv).reshape(-1, 2)
        self._chk(self._lib.srukf_add_landmarks(self._h, uv.shape[0], _d(uv)))
        self._refresh_dims()

    def get_landmarks_cartesian(self):
        """(xyz[N,3], cov[N,3,3]) of every landmark: getFeatureCartesianInformation batched on the device."""
        xyz, cov = np.zeros((self.N, 3)), np.zeros((self.N, 3, 3))
        self._chk(self._lib.srukf_get_landmarks_cartesian(self._h, _d(xyz), _d(cov)))
        return xyz, cov

    def get_frame_view(self):
        """(X, xyz[N,3], cov[N,3,3], pose[4], P4[4,4]) in one device round trip (srukf_get_frame_view)."""
        X, xyz, cov, pose, P4 = np.zeros(self.n), np.zeros((self.N, 3)), np.zeros((self.N, 3, 3)), np.zeros(4), np.zeros((4, 4))
        self._chk(self._lib.srukf_get_frame_view(self._h, _d(X), _d(xyz), _d(cov), _d(pose), _d(P4)))
        return X, xyz, cov, pose, P4

    def set_landmark_appearance(self, k, patch, R, t, px):
        """PointsMap::initPatch (21x21 uint8), initRotation (3x3), initTrans (3), initPixel (2) of landmark k."""
        patch = np.ascontiguousarray(patch, dtype=np.uint8); assert patch.shape == (21, 21)
        self._chk(self._lib.srukf_set_landmark_appearance(self._h, int(k), patch.ctypes.data_as(C.POINTER(C.c_ubyte)),
                                                          _d(_c(R).reshape(9)), _d(_c(t).reshape(3)), _d(_c(px).reshape(2))))

    def associate(self, gray):
        """wrapPatch + dataAssociation on the device; returns (z[2N], matched[N], corr[N])."""
        gray = np.ascontiguousarray(gray, dtype=np.uint8)
        z, m, cr = np.zeros(2 * self.N), np.zeros(self.N, dtype=np.int32), np.zeros(self.N)
        self._chk(self._lib.srukf_associate(self._h, gray.ctypes.data_as(C.POINTER(C.c_ubyte)), _d(z), _i(m), _d(cr)))
        return z, m, cr

    def get_match_patch(self, k):
        out = np.zeros((17, 17), dtype=np.uint8)
        self._chk(self._lib.srukf_get_match_patch(self._h, int(k), out.ctypes.data_as(C.POINTER(C.c_ubyte))))
        return out

    def set_exclusive(self, exclusive):
        """True / GPU_EXCLUSIVE (default): the filter has the GPU to itself (one persistent refactorisation launch per frame that may
        use every CU); False / GPU_SHARED: several filters replay concurrently on this GPU (persistent launches of half the CUs,
        at most two admitted at a time); GPU_SHARED_PER_PANEL: one launch per 64-row panel."""
        self._chk(self._lib.srukf_set_exclusive(self._h, int(exclusive)))

    def debug_poke_S(self, row, col, value):
        """Test hook: one entry of S on the device, untracked."""
        self._chk(self._lib.srukf_debug_poke_state(self._h, int(row), int(col), float(value)))

    def set_rank_aware(self, on):
        """Rank-aware refactorisation (default on): structurally null pivots are not factored."""
        self._chk(self._lib.srukf_set_rank_aware(self._h, 1 if on else 0))

    def null_directions(self):
        return self._lib.srukf_null_directions(self._h)

    def set_storage(self, storage):
        """STORAGE_F64 (default) or STORAGE_F32: precision of the state kept between frames."""
        self._chk(self._lib.srukf_set_storage(self._h, int(storage)))

    def get_state_f32(self):
        X, S = np.zeros(self.n, dtype=np.float32), np.zeros((self.n, self.n), dtype=np.float32)
        fp = C.POINTER(C.c_float)
        self._chk(self._lib.srukf_get_state_f32(self._h, X.ctypes.data_as(fp), S.ctypes.data_as(fp)))
        return X, S

    def delete_landmark(self, idx):
        """deleteOneFeature: landmark idx (0-based state order) leaves the map; the filter shrinks to N - 1."""
        self._chk(self._lib.srukf_delete_landmark(self._h, int(idx)))
        self._refresh_dims()

    def stage_sequence(self, odo, z, matched):
        odo, z, m = _c(odo), _c(z), _c(matched, np.int32)
        F = z.shape[0]
        assert odo.shape == (F + 1, 3) and z.shape == (F, 2 * self.N) and m.shape == (F, self.N)
        self._chk(self._lib.srukf_stage_sequence(self._h, F, _d(odo), _d(z), _i(m)))
        self.F = F

    def prepare_frames(self, count):
        """Capture `count` staged frames as one graph for the following run_frames_async(*, count) calls (nothing runs)."""
        self._chk(self._lib.srukf_prepare_frames(self._h, int(count)))

    def run_frames_async(self, first, count, mode=UPDATE_BATCHED, d_traj_ptr=None):
        self._chk(self._lib.srukf_run_frames_async(self._h, first, count, mode,
                                                   C.c_void_p(d_traj_ptr) if d_traj_ptr else None))

    def run_frames(self, first, count, mode=UPDATE_BATCHED):
        traj = np.empty((count, 8))
        self._chk(self._lib.srukf_run_frames(self._h, first, count, mode, _d(traj)))
        return traj

    def synchronize(self):
        self._chk(self._lib.srukf_synchronize(self._h))

    def clamp_info(self):
        """(frame, row) of the last SRUKF_ERR_CLAMP_PENDING; (-1, -1) if none."""
        fr, row = C.c_int(), C.c_int()
        self._chk(self._lib.srukf_clamp_info(self._h, C.byref(fr), C.byref(row)))
        return fr.value, row.value

    def debug_allow_mixed(self, on):
        """Study hook: accept STORAGE_F32_MIXED below its epsilon floor."""
        self._chk(self._lib.srukf_debug_allow_mixed(self._h, int(on)))

    def debug_set(self, key, value):
        """Measurement / test switch of this filter (or a process-wide one): see srukf_debug_set in include/srukf.h."""
        self._chk(self._lib.srukf_debug_set(self._h, key.encode(), int(value)))

    def debug_get(self, key):
        v = C.c_longlong()
        self._chk(self._lib.srukf_debug_get(self._h, key.encode(), C.byref(v)))
        return v.value

    def debug_copy(self, key, count):
        """Diagnostic copy of the first `count` doubles of a device work buffer ("Z", "DZ", "sigR", "Cmat", "Xr1", "Utp", "P1", "h", "Si")."""
        out = np.empty(int(count), dtype=np.float64)
        self._chk(self._lib.srukf_debug_copy(self._h, key.encode(), out.ctypes.data_as(C.POINTER(C.c_double)), int(count)))
        return out

    def debug_upload(self, key, arr):
        """The other direction of debug_copy (measurement scripts)."""
        a = np.ascontiguousarray(arr, dtype=np.float64).ravel()
        self._chk(self._lib.srukf_debug_upload(self._h, key.encode(), a.ctypes.data_as(C.POINTER(C.c_double)), a.size))

    def debug_split_replay(self, which, reps=1):
        """One launch of the split form's pair alone against recorded buffers (scripts/split_replay.py)."""
        self._chk(self._lib.srukf_debug_split_replay(self._h, int(which), int(reps)))

    def debug_gmw_stamps(self):
        """Diagnostic builds only: the time stamps the persistent launch left since the last call (and arms the next launches)."""
        buf = (C.c_ulonglong * 4096)()
        self._chk(self._lib.srukf_debug_gmw_stamps(self._h, buf))
        return np.ctypeslib.as_array(buf).copy()

    def debug_starve_workers(self, on):
        """Test hook: persistent factorisation launches start without their workers."""
        self._chk(self._lib.srukf_debug_starve_workers(self._h, int(on)))

    def set_profiling(self, on):
        self._chk(self._lib.srukf_set_profiling(self._h, int(on)))

    def profile_reset(self):
        self._chk(self._lib.srukf_profile_reset(self._h))

    def profile(self):
        out = {}
        for i in range(self._lib.srukf_profile_count(self._h)):
            name, ms, cnt, fl, by = C.c_char_p(), C.c_double(), C.c_longlong(), C.c_double(), C.c_double()
            self._chk(self._lib.srukf_profile_get(self._h, i, C.byref(name), C.byref(ms), C.byref(cnt), C.byref(fl), C.byref(by)))
            out[name.value.decode()] = {"ms": ms.value, "launches": cnt.value, "alg_flops": fl.value, "alg_bytes": by.value}
        return out


def debug_set_global(key, value):
    """Process-wide measurement / test switch (srukf_debug_set with a NULL context)."""
    rc = load_library().srukf_debug_set(None, key.encode(), int(value))
    if rc != 0:
        raise SrukfError(rc, f"srukf_debug_set({key})")


def run_frames_batch(filters, first, count, mode=UPDATE_BATCHED):
    """B filters through the same block of staged frames concurrently on one GPU (srukf_run_frames_batch) -> traj[B, count, 8]."""
    B = len(filters)
    hs = (C.c_void_p * B)(*[f._h.value for f in filters])
    traj = np.empty((B, count, 8))
    st = (C.c_int * B)()
    rc = load_library().srukf_run_frames_batch(hs, B, first, count, mode, _d(traj), st)
    if rc != 0:
        bad = [b for b in range(B) if st[b] != 0]
        filters[bad[0] if bad else 0]._chk(rc)
    return traj


def gmw(G, eps=1e-13, force_slow=False, device=0):
    """Device modified Cholesky of a symmetric matrix (SLAM.cpp:2197-2327) -> (S, D, clamp_hit)."""
    G = _c(G)
    n = G.shape[0]
    S, D, hit = np.zeros((n, n)), np.zeros(n), C.c_int(0)
    rc = load_library().srukf_gmw_host(device, n, _d(G), _d(S), _d(D), eps, int(force_slow), C.byref(hit))
    if rc != 0:
        raise SrukfError(rc, "srukf_gmw_host")
    return S, D, hit.value


def project(params, feat6, pos3, psi, err2, device=0):
    """Device camera projection of individual points (SLAM.cpp:1662-1670)."""
    p = Params.from_dict(params)
    feat6, pos3, psi, err2 = _c(feat6).reshape(-1, 6), _c(pos3).reshape(-1, 3), _c(psi).reshape(-1), _c(err2).reshape(-1, 2)
    out = np.zeros((feat6.shape[0], 2))
    rc = load_library().srukf_project_host(device, C.byref(p), feat6.shape[0], _d(feat6), _d(pos3), _d(psi), _d(err2), _d(out))
    if rc != 0:
        raise SrukfError(rc, "srukf_project_host")
    return out
