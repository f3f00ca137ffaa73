"""nerfstudio.cameras.camera_utils quaternion helpers (numpy, real part first) — the classic `transformations`
routines nerfstudio re-exports, restated from their published definitions."""
import math
import numpy as np

_EPS = np.finfo(float).eps * 4.0


def quaternion_from_matrix(matrix, isprecise=False):
    M = np.array(matrix, dtype=np.float64)[:4, :4]
    m00, m01, m02 = M[0, 0], M[0, 1], M[0, 2]
    m10, m11, m12 = M[1, 0], M[1, 1], M[1, 2]
    m20, m21, m22 = M[2, 0], M[2, 1], M[2, 2]
    K = np.array([[m00 - m11 - m22, 0.0, 0.0, 0.0],
                  [m01 + m10, m11 - m00 - m22, 0.0, 0.0],
                  [m02 + m20, m12 + m21, m22 - m00 - m11, 0.0],
                  [m21 - m12, m02 - m20, m10 - m01, m00 + m11 + m22]]) / 3.0
    w, V = np.linalg.eigh(K)                      # eigenvector of the largest eigenvalue
    q = V[[3, 0, 1, 2], np.argmax(w)]
    if q[0] < 0.0:
        q = -q
    return q


def quaternion_matrix(quaternion):
    q = np.array(quaternion, dtype=np.float64, copy=True)
    n = np.dot(q, q)
    if n < _EPS:
        return np.identity(4)
    q *= math.sqrt(2.0 / n)
    q = np.outer(q, q)
    return np.array([[1.0 - q[2, 2] - q[3, 3], q[1, 2] - q[3, 0], q[1, 3] + q[2, 0], 0.0],
                     [q[1, 2] + q[3, 0], 1.0 - q[1, 1] - q[3, 3], q[2, 3] - q[1, 0], 0.0],
                     [q[1, 3] - q[2, 0], q[2, 3] + q[1, 0], 1.0 - q[1, 1] - q[2, 2], 0.0],
                     [0.0, 0.0, 0.0, 1.0]])


def quaternion_slerp(quat0, quat1, fraction, spin=0, shortestpath=True):
    q0 = np.array(quat0[:4], dtype=np.float64); q0 = q0 / np.linalg.norm(q0)
    q1 = np.array(quat1[:4], dtype=np.float64); q1 = q1 / np.linalg.norm(q1)
    if fraction == 0.0:
        return q0
    if fraction == 1.0:
        return q1
    d = np.dot(q0, q1)
    if abs(abs(d) - 1.0) < _EPS:
        return q0
    if shortestpath and d < 0.0:
        d, q1 = -d, -q1
    angle = math.acos(d) + spin * math.pi
    if abs(angle) < _EPS:
        return q0
    isin = 1.0 / math.sin(angle)
    return q0 * (math.sin((1.0 - fraction) * angle) * isin) + q1 * (math.sin(fraction * angle) * isin)
