"""Seeded synthetic 640x480 ceiling-SLAM scene generator (numpy, host side).

The reference ships no image sequence or odometry file (its defaults point at ``E:\\SLAM\\...``,
SLAM.cpp:205-209), so benchmark and parity inputs are synthetic.  This module
produces, for a landmark count N, a seed and a frame count F:

* ``X0``/``S0``: the filter state after the reference's *joint initialisation* of N landmarks
  seen in the first frame (integrateFeaturesInformation numeric part, SLAM.cpp:826-871,
  1177-1334) — a faithful rank-deficient sqrt-covariance, ``rank(S0^T S0) = 4 + 3N``;
* ``odo[(F+1), 3]``: odometry poses (x, y, theta) on a small closed figure-8 (heading stays
  inside (-pi, pi), see figure8_odometry);
* ``z[F, 2N]``: pixel measurements = the reference's own projection model (quirks included,
  SLAM.cpp:1634-1674, 3177-3347) of the true landmarks from the true pose + N(0, 0.5^2) noise;
* ``matched[F, N]``: all ones (every landmark visible and matched, M = N).

It is input generation only — it is never on the measured path.  It is also an independent
(vectorised numpy, LAPACK QR) restatement of the camera model and of the joint initialisation,
which the tests cross-check against the C oracle in ``oracle/``.
"""
from __future__ import annotations

import numpy as np

# field order == struct srukf_params in include/srukf.h
PARAM_FIELDS = [
    ("cam_dx", 0.0028), ("cam_dy", 0.0028), ("cam_cx", 310.1129), ("cam_cy", 236.7526),
    ("cam_k1", 0.0001), ("cam_k2", 0.0), ("cam_f", 2.1735), ("image_w", 640.0), ("image_h", 480.0),
    ("a1", 8.0), ("a2", 8.0), ("a3", 8.0), ("a4", 8.0), ("sigma_measure", 3.0),
    ("rho0", 1.0 / 3.0), ("sigma_rho", (1.0 / 3.0) / 2.0),
    ("sigma_x", 0.02), ("sigma_y", 0.02), ("sigma_z", 0.005), ("sigma_theta", 0.02),
    ("epsilon", 1e-13), ("ut_alpha", 1e-3), ("ut_beta", 2.0),
]
PARAM_INT_FIELDS = [("weight_type", 0), ("noise_type", 0), ("newton_iters", 100), ("reserved_", 0)]


def default_params() -> dict:
    """Reference debug-model defaults (SLAM.cpp:172-198, 221-224, 238-242, 263-264, 329-337)."""
    d = {k: v for k, v in PARAM_FIELDS}
    d.update({k: v for k, v in PARAM_INT_FIELDS})
    return d


def scene_params() -> dict:
    """Parameters of the synthetic benchmark scene: the reference defaults with the odometry
    noise constants a1..a4 set to the reference's own alternative values (commented at
    SLAM.cpp:191-194: 0.0004, 0.0004, 0.0006, 0.0006).  With the shipped a1..a4 = 8 the
    reference update (independent per-landmark gains whose covariance reductions are summed,
    SLAM.cpp:2066-2095) over-subtracts the shared robot process noise and diverges as soon as
    >= 8 landmarks are matched per frame; with the small constants it is stable at
    N = 8..200 ("Synthetic scene")."""
    d = default_params()
    d.update(a1=0.0004, a2=0.0004, a3=0.0006, a4=0.0006)
    return d


def ut_weights(Na: int, weight_type: int = 0, alpha: float = 1e-3, beta: float = 2.0):
    """calculateSampleParameter, SLAM.cpp:1050-1103 -> (wm0, wc0, wi, wi_sr, gamma)."""
    if weight_type == 0:
        wm0 = 1.0 - Na / 3.0
        wc0 = wm0
        wi = (1.0 - wc0) / (2 * Na)
        gamma = np.sqrt(Na / (1.0 - wm0))
    elif weight_type == 1:
        lam = alpha ** 2 * Na - Na
        gamma = np.sqrt(Na + lam)
        wm0 = lam / (Na + lam)
        wc0 = wm0 + (1 - alpha ** 2 + beta)
        wi = 1.0 / (2 * (Na + lam))
    else:
        gamma = np.sqrt(3.0 * Na / 2.0)
        wm0 = wc0 = 1.0 / 3.0
        wi = 1.0 / (3.0 * Na)
    return wm0, wc0, wi, np.sqrt(abs(wi)), gamma


def project(feat, pos, psi, err, p=None, iters=None):
    """Vectorised camera projection (SLAM.cpp:1662-1670 with helpers 3250-3347, 3177-3213).

    feat[..., 6] = (xi yi zi theta phi rho), pos[..., 3], psi[...], err[..., 2] -> uv[..., 2]
    """
    p = p or default_params()
    feat = np.asarray(feat, dtype=np.float64)
    pos = np.asarray(pos, dtype=np.float64)
    psi = np.asarray(psi, dtype=np.float64)
    err = np.asarray(err, dtype=np.float64)
    xi, yi, zi, th, ph, rho = [feat[..., i] for i in range(6)]
    hx = xi + 1 / rho * np.cos(ph) * np.sin(th) - pos[..., 0]
    hy = yi - 1 / rho * np.sin(ph) - pos[..., 1]
    hz = zi + 1 / rho * np.cos(ph) * np.cos(th) - pos[..., 2]
    c, s = np.cos(psi), np.sin(psi)
    det = c * c + s * s                      # Rwc.inv() closed form (cofactors / det)
    rx = (c * hx + s * hy) / det
    ry = (-s * hx + c * hy) / det
    rz = hz * (det / det)
    f1, f2 = p["cam_f"] / p["cam_dx"], p["cam_f"] / p["cam_dy"]
    W, H = p["image_w"], p["image_h"]
    with np.errstate(divide="ignore", invalid="ignore"):
        uy = p["cam_cx"] + f1 * rx / rz + err[..., 0]      # x/y swap, SLAM.cpp:3338-3339
        ux = p["cam_cy"] + f2 * ry / rz + err[..., 1]
    bad = (rz == 0) | (ux < 10) | (ux > W - 10) | (uy < 10) | (uy > H - 10)
    ux = np.where(bad, 0.0, ux)
    uy = np.where(bad, 0.0, uy)
    # distortOnePointRW
    k1, k2 = p["cam_k1"], p["cam_k2"]
    xu = (ux - p["cam_cx"]) * p["cam_dx"]
    yu = (uy - p["cam_cy"]) * p["cam_dy"]
    ru = np.sqrt(xu * xu + yu * yu)
    rd = ru / (1 + k1 * ru * ru + k2 * ru ** 4)
    for _ in range(iters if iters is not None else p["newton_iters"]):
        f = rd + k1 * rd ** 3 + k2 * rd ** 5 - ru
        ff = 1.0 + 3.0 * k1 * rd * rd + 5.0 * k2 * rd ** 4
        rd = rd - f / ff
    d = 1 + k1 * rd * rd + k2 * rd ** 4
    d = np.where(d == 0, p["epsilon"], d)
    vx = p["cam_cx"] + (xu / d) / p["cam_dx"]
    vy = p["cam_cy"] + (yu / d) / p["cam_dy"]
    vis = (vx >= 0) & (vx <= W) & (vy >= 0) & (vy <= H)
    return np.stack([np.where(vis, vx, 0.0), np.where(vis, vy, 0.0)], axis=-1)


def pixel_to_angles(uvd, psi, p=None):
    """undistort -> image2camera -> camera2world -> world2state (SLAM.cpp:3224-3236,
    3358-3363, 3382-3387, 3401-3420): returns (theta, phi) of the ray through pixel uvd."""
    p = p or default_params()
    uvd = np.asarray(uvd, dtype=np.float64)
    xd = (uvd[..., 0] - p["cam_cx"]) * p["cam_dx"]
    yd = (uvd[..., 1] - p["cam_cy"]) * p["cam_dy"]
    rd = np.sqrt(xd * xd + yd * yd)
    d = 1 + p["cam_k1"] * rd ** 2 + p["cam_k2"] * rd ** 4
    ux = p["cam_cx"] + xd * d / p["cam_dx"]
    uy = p["cam_cy"] + yd * d / p["cam_dy"]
    f1, f2 = p["cam_f"] / p["cam_dx"], p["cam_f"] / p["cam_dy"]
    hx, hy, hz = (uy - p["cam_cx"]) / f1, (ux - p["cam_cy"]) / f2, np.ones_like(ux)
    c, s = np.cos(psi), np.sin(psi)
    wx, wy, wz = c * hx - s * hy, s * hx + c * hy, hz
    return np.arctan2(wx, wz), np.arctan2(-wy, np.sqrt(wx * wx + wz * wz))


def joint_init(X, S, uv, p=None):
    """Augment (X, S) with K landmarks first seen at pixels uv[K, 2]; returns (X_new, S_new)
    in normal order.  numpy restatement of SLAM.cpp:826-871, 1177-1250, 1260-1334."""
    p = p or default_params()
    X = np.asarray(X, dtype=np.float64)
    S = np.asarray(S, dtype=np.float64)
    uv = np.asarray(uv, dtype=np.float64).reshape(-1, 2)
    dim, K = X.shape[0], uv.shape[0]
    Na = dim + 3 * K
    wm0, wc0, wi, wi_sr, gamma = ut_weights(Na, p["weight_type"], p["ut_alpha"], p["ut_beta"])
    mu = np.concatenate([X, np.column_stack([uv, np.full(K, p["rho0"])]).ravel()])
    sr = np.zeros((Na, Na))
    sr[:dim, :dim] = S
    sd = np.tile([p["sigma_measure"], p["sigma_measure"], p["sigma_rho"]], K)
    sr[np.arange(dim, Na), np.arange(dim, Na)] = sd
    sig = np.concatenate([mu[:, None], mu[:, None] + gamma * sr.T, mu[:, None] - gamma * sr.T], axis=1)  # Na x L
    L = 2 * Na + 1
    pos = sig[dim - 4:dim - 1, :]                       # 3 x L
    psi = sig[dim - 1, :]                               # L
    new = sig[dim:, :].reshape(K, 3, L)                 # K x (u, v, rho) x L
    th, ph = pixel_to_angles(np.stack([new[:, 0, :], new[:, 1, :]], axis=-1), psi[None, :], p)
    ang = np.stack([th, ph, new[:, 2, :]], axis=1)      # K x 3 x L
    out = np.concatenate([sig[:dim, :], ang.reshape(3 * K, L), np.tile(pos, (K, 1))], axis=0)   # disordered
    w = np.full(L, wi)
    w[0] = wm0
    mu_angle = ang.reshape(3 * K, L) @ w
    xdis = np.concatenate([X, mu_angle, np.tile(X[dim - 4:dim - 1], K)])
    A = wi_sr * (out[:, 1:] - out[:, :1]).T             # 2Na x dimn
    Sdis = np.linalg.qr(A, mode="r")
    dimn = dim + 6 * K
    if Sdis.shape[0] < dimn:
        Sdis = np.vstack([Sdis, np.zeros((dimn - Sdis.shape[0], dimn))])
    perm = permutation(dimn, K)
    Xn = xdis[perm]
    Sn = np.linalg.qr(Sdis[np.ix_(perm, perm)], mode="r")
    return Xn, np.triu(Sn)


def permutation(dim, K):
    """getPermutationMatrix (SLAM.cpp:1303-1334) as an index vector: X_normal = X_dis[perm]."""
    dim_old = dim - 6 * K
    perm = np.zeros(dim, dtype=np.int64)
    perm[:dim_old - 4] = np.arange(dim_old - 4)
    perm[dim - 4:] = dim_old - 4 + np.arange(4)
    for i in range(K):
        perm[dim_old - 4 + 6 * i + np.arange(3)] = dim_old + 3 * K + 3 * i + np.arange(3)
        perm[dim_old - 4 + 6 * i + 3 + np.arange(3)] = dim_old + 3 * i + np.arange(3)
    return perm


def figure8_odometry(F, period=60, step=0.01, amplitude=2.404825557695773):
    """(F+1) odometry poses (x, y, theta) of a robot driving forward `step` metres per frame with
    heading theta_t = A sin(2 pi t / period).  A = 2.4048 (first zero of J0) makes the path a
    closed figure-8 (0.09 m x 0.23 m at period 60), and keeps |theta| < pi: the reference's
    control extraction rot1 = atan2(dy, dx) - theta_prev has no angle wrap (SLAM.cpp:1448), so a
    heading that crosses +-pi injects a 2*pi "rotation" and its process noise."""
    th = amplitude * np.sin(2 * np.pi * np.arange(F + 1) / period)
    odo = np.zeros((F + 1, 3))
    for t in range(F):
        d = th[t + 1] - th[t]
        x, y, h = odo[t]
        h1 = h + d / 2                       # rot1 = rot2 = d/2
        odo[t + 1] = (x + step * np.cos(h1), y + step * np.sin(h1), h1 + d / 2)
    return odo


def make_scene(N, F, seed=0, p=None, meas_sigma=0.5, disc_radius=80.0, ceiling=3.0, init="joint",
               obs_seed=None):
    """Build one synthetic sequence.  See module docstring.

    init = "joint": reference joint initialisation (rank-deficient S0).
    init = "fullrank": S0 = triu(N(0, 0.01^2)) + diag(U(0.02, 0.1)) for tolerance sweeps.
    obs_seed: seed of the measurement noise (defaults to seed) so several Monte-Carlo runs can
    share one map (same `seed`) with independent noise.
    """
    p = p or scene_params()
    rng = np.random.default_rng(seed)
    # detections uniform in a disc around the principal point as the reference's swapped model
    # sees it: pt.x centred on cam_cy, pt.y on cam_cx (SLAM.cpp:3338-3339).
    r = disc_radius * np.sqrt(rng.uniform(0, 1, N))
    a = rng.uniform(0, 2 * np.pi, N)
    uv = np.column_stack([p["cam_cy"] + r * np.cos(a), p["cam_cx"] + r * np.sin(a)])
    X4 = np.zeros(4)
    S4 = np.diag([p["sigma_x"], p["sigma_y"], p["sigma_z"], p["sigma_theta"]])
    th, ph = pixel_to_angles(uv, 0.0, p)
    rho_true = np.cos(ph) * np.cos(th) / ceiling
    truth = np.column_stack([np.zeros((N, 3)), th, ph, rho_true])
    if init == "joint":
        X0, S0 = joint_init(X4, S4, uv, p)
    else:
        n = 6 * N + 4
        X0 = np.concatenate([np.column_stack([np.zeros((N, 3)), th, ph, np.full(N, p["rho0"])]).ravel(), X4])
        S0 = np.triu(rng.normal(0, 0.01, (n, n)), 1) + np.diag(rng.uniform(0.02, 0.1, n))
    odo = figure8_odometry(F)
    orng = np.random.default_rng(seed if obs_seed is None else obs_seed + 7919)
    z = np.zeros((F, 2 * N))
    for t in range(F):
        x, y, psi = odo[t + 1]
        uvp = project(truth, np.broadcast_to([x, y, 0.0], (N, 3)), np.full(N, psi), np.zeros((N, 2)), p, iters=12)
        z[t] = (uvp + orng.normal(0, meas_sigma, (N, 2))).ravel()
    return {"N": N, "F": F, "X0": X0, "S0": S0, "odo": odo, "z": z,
            "matched": np.ones((F, N), dtype=np.int32), "truth": truth, "uv0": uv, "params": p}
