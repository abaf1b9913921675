"""numpy front-end of the C oracle (oracle/pzn_oracle.c) — TEST INFRASTRUCTURE ONLY.

Every function mirrors one reference function and cites it; the arithmetic
lives in the C file.  Arrays in, arrays out (fp32 / int64, C-contiguous).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libpzn_oracle.so")
_lib = None

_f = ctypes.POINTER(ctypes.c_float)
_d = ctypes.POINTER(ctypes.c_double)
_i64 = ctypes.POINTER(ctypes.c_int64)
_i32 = ctypes.POINTER(ctypes.c_int32)


def build(force=False):
    """Compile the C oracle with gcc (oracle/Makefile)."""
    src = os.path.join(_HERE, "pzn_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
    return _lib


def _p(a, t):
    return a.ctypes.data_as(t) if a is not None else None


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _idx(a):
    return np.ascontiguousarray(a, dtype=np.int64)


def square_distance(src, dst):
    """pointnet_util.py:22-36"""
    src, dst = _f32(src), _f32(dst)
    B, S, _ = src.shape
    N = dst.shape[1]
    out = np.empty((B, S, N), np.float32)
    lib().orc_square_distance_f32(_p(src, _f), _p(dst, _f), B, S, N, _p(out, _f))
    return out


def farthest_point_sample(xyz, npoint, start_idx):
    """pointnet_util.py:53-73 (start_idx = the randint draw at :65)"""
    xyz, start_idx = _f32(xyz), _idx(start_idx)
    B, N, _ = xyz.shape
    out = np.empty((B, npoint), np.int64)
    lib().orc_fps_f32(_p(xyz, _f), B, N, npoint, _p(start_idx, _i64), _p(out, _i64))
    return out


def knn(xyz, new_xyz, K):
    """pointnet_util.py:118-119"""
    xyz, new_xyz = _f32(xyz), _f32(new_xyz)
    B, N, _ = xyz.shape
    S = new_xyz.shape[1]
    out = np.empty((B, S, K), np.int64)
    lib().orc_knn_f32(_p(xyz, _f), _p(new_xyz, _f), B, N, S, K, _p(out, _i64))
    return out


def query_ball_point(radius, nsample, xyz, new_xyz):
    """pointnet_util.py:76-96"""
    xyz, new_xyz = _f32(xyz), _f32(new_xyz)
    B, N, _ = xyz.shape
    S = new_xyz.shape[1]
    out = np.empty((B, S, nsample), np.int64)
    r2 = ctypes.c_float(np.float32(radius ** 2))
    lib().orc_ball_query_f32(r2, nsample, _p(xyz, _f), _p(new_xyz, _f), B, N, S, _p(out, _i64))
    return out


def index_points(points, idx):
    """pointnet_util.py:39-50"""
    points, idx = _f32(points), _idx(idx)
    B, N, C = points.shape
    flat = idx.reshape(B, -1)
    M = flat.shape[1]
    out = np.empty((B, M, C), np.float32)
    lib().orc_gather_fwd_f32(_p(points, _f), _p(flat, _i64), B, N, M, C, _p(out, _f))
    return out.reshape(*idx.shape, C)


def index_points_grad(grad_out, idx, N):
    grad_out, idx = _f32(grad_out), _idx(idx)
    B = idx.shape[0]
    C = grad_out.shape[-1]
    flat = idx.reshape(B, -1)
    M = flat.shape[1]
    g = np.empty((B, N, C), np.float32)
    lib().orc_gather_bwd_f32(_p(grad_out.reshape(B, M, C), _f), _p(flat, _i64), B, N, M, C, _p(g, _f))
    return g


def group(xyz, feat, new_xyz, idx, want_grouped_xyz=False):
    """pointnet_util.py:123-132"""
    xyz, new_xyz, idx = _f32(xyz), _f32(new_xyz), _idx(idx)
    B, N, _ = xyz.shape
    _, S, K = idx.shape
    D = 0 if feat is None else feat.shape[-1]
    feat = None if feat is None else _f32(feat)
    out = np.empty((B, S, K, 3 + D), np.float32)
    gx = np.empty((B, S, K, 3), np.float32) if want_grouped_xyz else None
    lib().orc_group_fwd_f32(_p(xyz, _f), _p(feat, _f), _p(new_xyz, _f), _p(idx, _i64),
                            B, N, S, K, D, _p(out, _f), _p(gx, _f))
    return (out, gx) if want_grouped_xyz else out


def group_grad(grad_out, idx, N):
    grad_out, idx = _f32(grad_out), _idx(idx)
    B, S, K = idx.shape
    D = grad_out.shape[-1] - 3
    gxyz = np.empty((B, N, 3), np.float32)
    gfeat = np.empty((B, N, D), np.float32) if D else None
    gnew = np.empty((B, S, 3), np.float32)
    lib().orc_group_bwd_f32(_p(grad_out, _f), _p(idx, _i64), B, N, S, K, D,
                            _p(gxyz, _f), _p(gfeat, _f), _p(gnew, _f))
    return gxyz, gfeat, gnew


def sample_and_group(npoint, radius, nsample, xyz, points, start_idx, returnfps=False, knn_mode=False):
    """pointnet_util.py:99-136 composed from the pieces above."""
    xyz = _f32(xyz)
    fps_idx = farthest_point_sample(xyz, npoint, start_idx)
    new_xyz = index_points(xyz, fps_idx)
    if knn_mode:
        idx = knn(xyz, new_xyz, nsample)
    else:
        idx = query_ball_point(radius, nsample, xyz, new_xyz)
    new_points, grouped_xyz = group(xyz, points, new_xyz, idx, want_grouped_xyz=True)
    if returnfps:
        return new_xyz, new_points, grouped_xyz, fps_idx
    return new_xyz, new_points


# --- EMD (PyTorchEMD/cuda/emd_kernel.cu) --------------------------------------

def _emd_t(dtype):
    if np.dtype(dtype) == np.float64:
        return np.float64, _d, "f64"
    return np.float32, _f, "f32"


def emd_approxmatch(xyz1, xyz2):
    """emd_kernel.cu:25-158 / :171-193 -> match[B,m,n]"""
    dt, pt, suf = _emd_t(xyz1.dtype)
    xyz1 = np.ascontiguousarray(xyz1, dt)
    xyz2 = np.ascontiguousarray(xyz2, dt)
    B, n, _ = xyz1.shape
    m = xyz2.shape[1]
    match = np.empty((B, m, n), dt)
    getattr(lib(), "orc_emd_approxmatch_" + suf)(_p(xyz1, pt), _p(xyz2, pt), B, n, m, _p(match, pt))
    return match


def emd_matchcost(xyz1, xyz2, match):
    """emd_kernel.cu:200-243 / :257-279 -> cost[B]"""
    dt, pt, suf = _emd_t(xyz1.dtype)
    xyz1, xyz2, match = (np.ascontiguousarray(a, dt) for a in (xyz1, xyz2, match))
    B, n, _ = xyz1.shape
    m = xyz2.shape[1]
    cost = np.empty((B,), dt)
    getattr(lib(), "orc_emd_matchcost_" + suf)(_p(xyz1, pt), _p(xyz2, pt), _p(match, pt), B, n, m, _p(cost, pt))
    return cost


def emd_matchcost_grad(grad_cost, xyz1, xyz2, match):
    """emd_kernel.cu:286-355 / :373-398 -> grad1[B,n,3], grad2[B,m,3]"""
    dt, pt, suf = _emd_t(xyz1.dtype)
    grad_cost, xyz1, xyz2, match = (np.ascontiguousarray(a, dt) for a in (grad_cost, xyz1, xyz2, match))
    B, n, _ = xyz1.shape
    m = xyz2.shape[1]
    g1 = np.empty((B, n, 3), dt)
    g2 = np.empty((B, m, 3), dt)
    getattr(lib(), "orc_emd_matchcost_grad_" + suf)(_p(grad_cost, pt), _p(xyz1, pt), _p(xyz2, pt), _p(match, pt),
                                                     B, n, m, _p(g1, pt), _p(g2, pt))
    return g1, g2


def earth_mover_distance(xyz1, xyz2):
    """PyTorchEMD/emd.py:5-14 forward: cost[B] (and the match it used)."""
    match = emd_approxmatch(xyz1, xyz2)
    return emd_matchcost(xyz1, xyz2, match), match


def chamfer(a, b):
    """model5_b.py:1495-1505 -> (min over a [B,m], min over b [B,n]) + arg-mins"""
    a, b = _f32(a), _f32(b)
    B, n, _ = a.shape
    m = b.shape[1]
    moa = np.empty((B, m), np.float32)
    aoa = np.empty((B, m), np.int32)
    mob = np.empty((B, n), np.float32)
    aob = np.empty((B, n), np.int32)
    lib().orc_chamfer_fwd_f32(_p(a, _f), _p(b, _f), B, n, m, _p(moa, _f), _p(aoa, _i32), _p(mob, _f), _p(aob, _i32))
    return moa, aoa, mob, aob
