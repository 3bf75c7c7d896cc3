"""CPU: the oracle's restatement of getTransformFromMatchesG2O (transformation_estimation.cpp:37-170) -- a two-view
bundle adjustment by Gauss-Newton with the point blocks eliminated.  g2o is not in the reference tree, so there is no
pin; what can be checked is that the restatement minimises the reference's cost: against scipy's least-squares solver
on the same residuals (u, v, depth per edge, information diag(1, 1, 1/depth_cov)), and that it recovers a known pose."""
import ctypes as C

import numpy as np
import pytest
from scipy.optimize import least_squares
from scipy.spatial.transform import Rotation

from oracle import pyoracle as po

K = (521.0, 521.0, 319.5, 239.5)  # transformation_estimation.cpp:56


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def g2o_refine(qxyz, txyz, qkp, tkp, mq, mt, sel, T, iterations, depth_cov=1e-4):
    L = po.lib()
    L.orc_g2o_refine.restype = C.c_int
    L.orc_g2o_refine.argtypes = [C.c_void_p] * 7 + [C.c_int, C.c_void_p, C.c_int, C.c_double]
    Tc = np.ascontiguousarray(T.T, np.float32).copy()      # column-major storage
    ok = L.orc_g2o_refine(_p(qxyz), _p(txyz), _p(qkp), _p(tkp), _p(mq), _p(mt), _p(sel), len(sel), _p(Tc), iterations, depth_cov)
    return ok, Tc.reshape(4, 4).T.copy()


def make_scene(rng, n=120, pix_noise=0.3, z_noise=0.004):
    # points in the newer camera's frame (camera 2 = world), T maps newer -> older (the RANSAC convention)
    X2 = np.stack([rng.uniform(-1.2, 1.2, n), rng.uniform(-0.9, 0.9, n), rng.uniform(1.0, 3.5, n)], 1)
    Rt = Rotation.from_euler("xyz", rng.uniform(-4, 4, 3), degrees=True).as_matrix()
    tt = rng.uniform(-0.08, 0.08, 3)
    T = np.eye(4)
    T[:3, :3], T[:3, 3] = Rt, tt
    X1 = X2 @ Rt.T + tt

    def observe(X):
        u = K[0] * X[:, 0] / X[:, 2] + K[2] + rng.normal(0, pix_noise, len(X))
        v = K[1] * X[:, 1] / X[:, 2] + K[3] + rng.normal(0, pix_noise, len(X))
        z = X[:, 2] + rng.normal(0, z_noise, len(X))
        kp = np.stack([u, v], 1).astype(np.float32)
        xyz = np.stack([(kp[:, 0] - K[2]) * z / K[0], (kp[:, 1] - K[3]) * z / K[1], z, np.ones(len(X))], 1).astype(np.float32)
        return kp, xyz
    qkp, qxyz = observe(X2)
    tkp, txyz = observe(X1)
    return T, qkp, qxyz, tkp, txyz


def cost_and_residuals(params, qkp, qxyz, tkp, txyz, wz):
    """The reference's cost: camera 1 pose P (camera-to-world) as rotvec + t, points X (world = camera 2 frame)."""
    n = len(qkp)
    R1 = Rotation.from_rotvec(params[:3]).as_matrix()
    t1 = params[3:6]
    X = params[6:].reshape(n, 3)
    Y1 = (X - t1) @ R1                     # R1^T (X - t1)
    sw = np.sqrt(wz)
    r = []
    for Y, kp, xyz in ((X, qkp, qxyz), (Y1, tkp, txyz)):
        r.append(K[0] * Y[:, 0] / Y[:, 2] + K[2] - kp[:, 0])
        r.append(K[1] * Y[:, 1] / Y[:, 2] + K[3] - kp[:, 1])
        r.append(sw * (Y[:, 2] - xyz[:, 2]))
    return np.concatenate(r)


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_gauss_newton_reaches_the_minimum_of_the_reference_cost(seed):
    rng = np.random.default_rng(seed)
    T, qkp, qxyz, tkp, txyz = make_scene(rng)
    n = len(qkp)
    mq = np.arange(n, dtype=np.int32)
    mt = np.arange(n, dtype=np.int32)
    sel = np.arange(n, dtype=np.int32)
    dc = 1e-4
    # start from a perturbed estimate, as the RANSAC result would be
    Tp = T.copy()
    Tp[:3, :3] = Tp[:3, :3] @ Rotation.from_euler("xyz", [0.4, -0.3, 0.2], degrees=True).as_matrix()
    Tp[:3, 3] += [0.01, -0.008, 0.012]
    ok, Tg = g2o_refine(qxyz, txyz, qkp, tkp, mq, mt, sel, Tp.astype(np.float32), 10, dc)
    assert ok == 1
    # scipy on the same cost, started at the same camera pose (camera 1 estimate is initialised with T itself, :86-89,
    # and the result is its inverse, :169 -- so the optimum is P = T^-1)
    P0 = np.linalg.inv(Tp)
    x0 = np.concatenate([Rotation.from_matrix(P0[:3, :3]).as_rotvec(), P0[:3, 3], qxyz[:, :3].astype(np.float64).ravel()])
    sol = least_squares(cost_and_residuals, x0, args=(qkp, qxyz, tkp, txyz, 1.0 / dc), method="trf", xtol=1e-14, ftol=1e-14,
                        gtol=1e-14)
    Ps = np.eye(4)
    Ps[:3, :3] = Rotation.from_rotvec(sol.x[:3]).as_matrix()
    Ps[:3, 3] = sol.x[3:6]
    Ts = np.linalg.inv(Ps)
    assert np.abs(Tg - Ts).max() < 2e-5, np.abs(Tg - Ts).max()
    # ... and both are closer to the truth than the start
    assert np.abs(Tg - T).max() < 0.5 * np.abs(Tp - T).max()


def test_fixed_point_and_iteration_count():
    rng = np.random.default_rng(5)
    T, qkp, qxyz, tkp, txyz = make_scene(rng, n=80, pix_noise=0.0, z_noise=0.0)
    n = len(qkp)
    ids = np.arange(n, dtype=np.int32)
    # note the reference's convention: camera 1 is INITIALISED with T but the answer is read as its inverse, so the
    # optimiser starts at the inverse of the optimum; with noise-free data it must still arrive at T
    ok, T1 = g2o_refine(qxyz, txyz, qkp, tkp, ids, ids, ids, T.astype(np.float32), 1)
    ok, T8 = g2o_refine(qxyz, txyz, qkp, tkp, ids, ids, ids, T.astype(np.float32), 8)
    assert ok == 1 and np.abs(T8 - T).max() < 1e-5
    assert np.abs(T1 - T).max() > np.abs(T8 - T).max()
    ok, T0 = g2o_refine(qxyz, txyz, qkp, tkp, ids, ids, ids, T.astype(np.float32), 0)   # no iteration: inverse of the start
    assert np.abs(T0 - np.linalg.inv(T)).max() < 1e-6
    # a subset of the matches, in any order of `sel`
    sel = np.array([5, 3, 60, 7, 9, 11, 40, 41, 42, 43, 44, 45], np.int32)
    ok, Ts = g2o_refine(qxyz, txyz, qkp, tkp, ids, ids, sel, T.astype(np.float32), 8)
    assert ok == 1 and np.abs(Ts - T).max() < 1e-5
