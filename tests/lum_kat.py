"""Known-answer check of the LUM linearisation (SURVEY App. A.6 asks for it: "re-derive from the Borrmann 6-D
linearisation and pin with a KAT"), shared by the oracle's and the product's tests.

Borrmann et al. linearise a pose X = (t, theta) with R(theta) = Rx(tx) Ry(ty) Rz(tz):
    d (R(theta) p + t) / dX  =  M(p') H(X),     p' = R p + t,
where M (3x6) is the matrix whose normal equations LUM::computeEdge accumulates and H (6x6) is
LUM::incidenceCorrection.  Both sides are independent of each other here: the left one is a central-difference
Jacobian of the pose map, M is read back from the SUMS computeEdge returns (MM = sum M^T M, MZ = sum M^T d), H from
the implementation under test.  A wrong sign or a swapped sin/cos in any of M's 6 or H's 13 off-diagonal entries
breaks the identity at a non-zero pose and a long lever arm (|p| ~ 917 mm, the data's)."""
import numpy as np


def rot(axis, a):
    c, s = np.cos(a), np.sin(a)
    return {0: np.array([[1, 0, 0], [0, c, -s], [0, s, c]]), 1: np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]]),
            2: np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]])}[axis]


def pose_map(X, p):
    return rot(0, X[3]) @ rot(1, X[4]) @ rot(2, X[5]) @ p + X[:3]


def numeric_jacobian(X, p, h=1e-6):
    J = np.zeros((3, 6))
    for k in range(6):
        d = np.zeros(6); d[k] = h
        J[:, k] = (pose_map(X + d, p) - pose_map(X - d, p)) / (2 * h)
    return J


def M_from_edge_sums(lum_edge, p):
    """M(p) (3x6) recovered from computeEdge's sums for ONE point p: with the poses at zero, source = p + d/2 and
    target = p - d/2 have average p and difference d, so that MZ = M^T d; three independent d read M off row by row.
    lum_edge(src (n,3), tgt (n,3)) -> (MM, MZ) must need >= 3 correspondences: the point is repeated."""
    M = np.zeros((3, 6))
    for r in range(3):
        d = np.zeros(3); d[r] = 1.0
        src, tgt = np.tile(p + d / 2, (4, 1)), np.tile(p - d / 2, (4, 1))
        MM, MZ = lum_edge(src, tgt)
        M[r] = MZ / 4.0
    return M


def M_full(p):
    """the derived form: [I | ex x p, ez x p, ey x p] (rotational unknowns in the order x, z, y)"""
    x, y, z = p
    return np.array([[1, 0, 0, 0, -y, z], [0, 1, 0, -z, x, 0], [0, 0, 1, y, 0, -x]], float)


def check(lum_edge, incidence, rng, trials=5):
    worst = 0.0
    for _ in range(trials):
        X = np.concatenate([rng.normal(0, 5, 3), rng.normal(0, 0.3, 3)])          # a NON-zero pose: mm, radians
        p = np.array([-13.0, 50.0, 917.0]) + rng.normal(0, 40, 3)                # the data's lever arm
        pp = pose_map(X, p)
        M = M_from_edge_sums(lum_edge, pp)
        assert np.allclose(M, M_full(pp), atol=1e-5), (M, M_full(pp))
        J = numeric_jacobian(X, p)
        A = M @ incidence(X)
        worst = max(worst, np.abs(J - A).max())
        assert np.abs(J - A).max() < 1e-5 * max(1.0, np.abs(J).max()), (X, p, J, A)
    return worst
