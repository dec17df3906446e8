"""Independent numpy restatement of the SRUKF frame (LAPACK QR, vectorised projection).

Used by the tests as a second, structurally different implementation to pin the C oracle
(oracle/srukf_oracle.c) on convention-independent quantities (X, P = S^T S, h, |Si|), and as a
fast explorer of scenario stability at large N.  Follows SLAM.cpp 1430-1555 (motion),
1615-1795 (measurement), 2020-2155 + 2197-2327 (update / modified Cholesky).
"""
from __future__ import annotations

import numpy as np


def gmw(G, eps=1e-13):
    """modifiedCholeskyDecomposition (SLAM.cpp:2197-2327), vectorised by column.
    Returns S (upper, P+E = S^T S), D, number of eps clamps, number of theta clamps,
    most negative pivot seen."""
    G = np.asarray(G, dtype=np.float64)
    n = G.shape[0]
    gamma = np.max(np.diag(G))
    off = G - np.diag(np.diag(G))
    xi = np.max(off)
    nu = max(1.0, np.sqrt(n * n - 1.0))
    beta2 = max(gamma, xi / nu, 1e-15)
    C = np.zeros((n, n))           # C[i, j], i >= j (lower storage)
    D = np.zeros(n)
    Lm = np.zeros((n, n))
    cd = np.diag(G).copy()         # running diagonal C_jj
    n_eps = n_theta = 0
    min_piv = np.inf
    for j in range(n):
        if j > 0:
            Lm[j, :j] = C[j, :j] / D[:j]
            C[j + 1:, j] = G[j + 1:, j] - C[j + 1:, :j] @ Lm[j, :j]
        else:
            C[1:, 0] = G[1:, 0]
        theta = np.max(np.abs(C[j + 1:, j])) if j < n - 1 else 0.0
        cand = (eps, abs(cd[j]), theta * theta / beta2)
        which = int(np.argmax(cand))
        D[j] = cand[which]
        min_piv = min(min_piv, cd[j])
        n_eps += which == 0
        n_theta += which == 2
        cd[j + 1:] -= C[j + 1:, j] ** 2 / D[j]
    Lm[np.arange(n), np.arange(n)] = 1.0
    S = np.sqrt(D)[:, None] * Lm.T
    return np.triu(S), D, n_eps, n_theta, min_piv


class NpFilter:
    def __init__(self, N, params, synth):
        self.N, self.n, self.p, self.synth = N, 6 * N + 4, params, synth
        self.X = np.zeros(self.n)
        self.S = np.zeros((self.n, self.n))
        self.stats = {"eps": 0, "theta": 0, "min_pivot": np.inf}

    def set_state(self, X, S):
        self.X, self.S = np.array(X, dtype=np.float64), np.triu(np.array(S, dtype=np.float64))

    def predict_motion(self, odo_prev, odo_cur):
        p, n = self.p, self.n
        Na = n + 5
        self.Na, self.L = Na, 2 * Na + 1
        self.wm0, self.wc0, self.wi, self.wi_sr, self.gamma = self.synth.ut_weights(Na, p["weight_type"], p["ut_alpha"], p["ut_beta"])
        dx, dy = odo_cur[0] - odo_prev[0], odo_cur[1] - odo_prev[1]
        rot1 = np.arctan2(dy, dx) - odo_prev[2]
        trans = np.sqrt(dy * dy + dx * dx)
        rot2 = odo_cur[2] - odo_prev[2] - rot1
        Mt = np.array([p["a1"] * rot1 ** 2 + p["a2"] * trans ** 2,
                       p["a3"] * trans ** 2 + p["a4"] * rot1 ** 2 + p["a4"] * rot2 ** 2,
                       p["a1"] * rot2 ** 2 + p["a2"] * trans ** 2])
        sr = np.zeros((Na, Na))
        sr[:n, :n] = self.S
        sr[np.arange(n, n + 3), np.arange(n, n + 3)] = Mt
        sr[np.arange(n + 3, Na), np.arange(n + 3, Na)] = p["sigma_measure"]
        mu = np.concatenate([self.X, np.zeros(5)])
        sig = np.concatenate([mu[:, None], mu[:, None] + self.gamma * sr.T, mu[:, None] - self.gamma * sr.T], axis=1)
        r1, tr, r2 = rot1 - sig[n], trans - sig[n + 1], rot2 - sig[n + 2]
        th = sig[n - 1].copy()
        sig[n - 4] += tr * np.cos(th + r1)
        sig[n - 3] += tr * np.sin(th + r1)
        sig[n - 1] += r1 + r2
        w = np.full(self.L, self.wi)
        w[0] = self.wm0
        self.w = w
        self.X[n - 4:] = sig[n - 4:n] @ w
        A = self.wi_sr * (sig[:n, 1:] - sig[:n, :1]).T
        self.S = np.triu(np.linalg.qr(A, mode="r"))
        self.sig = sig

    def predict_measurement(self):
        n, N, L, sig = self.n, self.N, self.L, self.sig
        feat = sig[:6 * N].reshape(N, 6, L).transpose(0, 2, 1)          # N x L x 6
        pos = np.broadcast_to(sig[n - 4:n - 1].T[None], (N, L, 3))
        psi = np.broadcast_to(sig[n - 1][None], (N, L))
        err = np.broadcast_to(sig[n + 3:n + 5].T[None], (N, L, 2))
        uv = self.synth.project(feat, pos, psi, err, self.p, iters=20)   # N x L x 2
        self.Z = uv.transpose(0, 2, 1).reshape(2 * N, L)
        self.h = self.Z @ self.w
        Si = np.zeros((N, 2, 2))
        for k in range(N):
            A = self.wi_sr * (self.Z[2 * k:2 * k + 2, 1:] - self.Z[2 * k:2 * k + 2, :1]).T
            Si[k] = np.triu(np.linalg.qr(A, mode="r"))
        self.Si = Si
        vis = (self.h[0::2] != 0) & (self.h[1::2] != 0)
        return self.h.copy(), Si.copy(), vis

    def update(self, z, matched, mode=1):
        n, N = self.n, self.N
        wc = self.w.copy()
        wc[0] = self.wc0
        cols = []
        for k in range(N):
            if not matched[k]:
                continue
            hk = self.h[2 * k:2 * k + 2]
            Pxy = ((self.sig[:n] - self.X[:, None]) * wc) @ (self.Z[2 * k:2 * k + 2] - hk[:, None]).T
            si = self.Si[k]
            sii = np.linalg.inv(si)
            K = Pxy @ sii @ sii.T
            self.X = self.X + K @ (z[2 * k:2 * k + 2] - hk)
            U = K @ si.T
            if mode == 0:
                for c in range(2):
                    G = self.S.T @ self.S - np.outer(U[:, c], U[:, c])
                    self._refactor(G)
            else:
                cols.append(U)
        if mode == 1 and cols:
            Uall = np.concatenate(cols, axis=1)
            self._refactor(self.S.T @ self.S - Uall @ Uall.T)

    def _refactor(self, G):
        S, D, ne, nt, mp = gmw(G, self.p["epsilon"])
        self.S = S
        self.stats["eps"] += ne
        self.stats["theta"] += nt
        self.stats["min_pivot"] = min(self.stats["min_pivot"], mp)

    def run(self, sc, mode=1, frames=None):
        F = sc["z"].shape[0] if frames is None else frames
        traj = np.zeros((F, 8))
        n = self.n
        for f in range(F):
            self.predict_motion(sc["odo"][f], sc["odo"][f + 1])
            self.predict_measurement()
            self.update(sc["z"][f], sc["matched"][f], mode)
            traj[f, :4] = self.X[n - 4:]
            Pr = self.S[:, n - 4:n - 2].T @ self.S[:, n - 4:n - 2]
            traj[f, 4:] = Pr.ravel()
        return traj
