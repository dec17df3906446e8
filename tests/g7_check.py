"""Shared by the CPU and the GPU parity tests: holds a state to a g7 fixture (tests/golden/make_golden.py::g7_sequential)."""
import numpy as np


def g7_check(g, X, P, tol_x, tol_p):
    """Holds a state (X, P = S^T S) to a g7 fixture: the state, diag P, the robot columns of P, every landmark's 6 x 6 block and the
    16 probe products P V (tests/golden/make_golden.py::g7_sequential)."""
    N = int(g["N"]); n = 6 * N + 4
    np.testing.assert_allclose(X, g["X"], rtol=0, atol=tol_x)
    np.testing.assert_allclose(np.diag(P), g["P_diag"], rtol=0, atol=tol_p)
    np.testing.assert_allclose(P[:, n - 4:], g["P_robot_cols"], rtol=0, atol=tol_p)
    blocks = np.stack([P[6 * k:6 * k + 6, 6 * k:6 * k + 6] for k in range(N)])
    np.testing.assert_allclose(blocks, g["P_blocks"], rtol=0, atol=tol_p)
    # |(P - P_ref) V| <= n |dP|max: an entry off by more than tol_p shows with probability 1 - 2^-16 per row
    np.testing.assert_allclose(P @ g["V"], g["PV"], rtol=0, atol=tol_p * np.sqrt(n) * 4)
