"""Host-side helper of the reference package that callers of the hot path use."""
import numpy as np


def angleaxis_to_rotation_matrix(aa):
    """3-vector angle axis -> 3x3 rotation (same convention as reference helpers.py:37-57: angle = |aa|,
    identity below 1e-6, Rodrigues otherwise)."""
    aa = np.asarray(aa, dtype=np.float64).reshape(3)
    angle = float(np.linalg.norm(aa))
    if angle <= 1e-6:
        return np.eye(3)
    u = aa / angle
    K = np.array([[0.0, -u[2], u[1]], [u[2], 0.0, -u[0]], [-u[1], u[0], 0.0]])
    return np.cos(angle) * np.eye(3) + np.sin(angle) * K + (1.0 - np.cos(angle)) * np.outer(u, u)
