"""Drop-in for the reference's ``pointnet_util`` hot-path functions
(pointnet_util.py:22-136): same names, argument order, shapes and dtypes, but
each call is one or two hand-written gfx950 kernels behind the C ABI.

    import puzzlenet_amd.pointnet_util as pu      # instead of `import pointnet_util as pu`

Differences a caller can observe:
  * tensors must be on the GPU (the reference's code is device-agnostic torch);
  * kNN ties (exactly equal distances) are ordered by ascending index, where
    the reference's unstable ``argsort`` leaves them implementation-defined;
  * no ``torch.cuda.empty_cache()`` calls (pointnet_util.py:114-126): nothing
    large is allocated in between.
"""
import torch

from . import _lib, ops

__all__ = ["square_distance", "index_points", "farthest_point_sample", "query_ball_point", "sample_and_group"]


def square_distance(src, dst):
    """[B,N,3], [B,M,3] -> [B,N,M]  (pointnet_util.py:22-36)."""
    return ops.square_distance(src, dst)


def index_points(points, idx):
    """points[B,N,C], idx int64 [B,S] or [B,S,K] -> [B,S(,K),C]  (pointnet_util.py:39-50)."""
    return ops.index_points(points, idx)


def farthest_point_sample(xyz, npoint, start_idx=None):
    """xyz[B,N,3] -> int64 [B,npoint]  (pointnet_util.py:53-73).

    The first centroid is drawn exactly as the reference does
    (``torch.randint(0, N, (B,))`` from the global CPU generator, line 65) so a
    seeded run picks the same points; pass ``start_idx`` to fix it explicitly.
    """
    B, N, _ = xyz.shape
    if start_idx is None:
        start_idx = torch.randint(0, N, (B,), dtype=torch.long)
    return ops.farthest_point_sample(xyz, npoint, start_idx.to(xyz.device))


def query_ball_point(radius, nsample, xyz, new_xyz):
    """-> int64 [B,S,nsample]  (pointnet_util.py:76-96)."""
    return ops.ball_query(radius, nsample, xyz, new_xyz)


def sample_and_group(npoint, radius, nsample, xyz, points, returnfps=False, knn=False):
    """(pointnet_util.py:99-136) -> new_xyz[B,S,3], new_points[B,S,K,3+D]
    (+ grouped_xyz[B,S,K,3], fps_idx[B,S] when returnfps)."""
    fps_idx = farthest_point_sample(xyz, npoint)                      # :113
    new_xyz = ops.index_points(xyz, fps_idx)                          # :115
    if knn and ops.knn_group_supported(xyz, points, nsample):
        # :117-132 in ONE launch (pzn_knn_group_f32): search and reference-layout group write by the same wavefront
        try:
            new_points, grouped_xyz, _ = ops.knn_group(xyz, points, new_xyz, want_grouped_xyz=returnfps)
            return (new_xyz, new_points, grouped_xyz, fps_idx) if returnfps else (new_xyz, new_points)
        except _lib.PznUnsupported:      # e.g. rows too wide for the LDS pieces: the two single launches below
            pass
    if knn:
        idx = ops.knn(xyz, new_xyz.detach(), nsample)                 # :118-119
    else:
        idx = ops.ball_query(radius, nsample, xyz, new_xyz.detach())  # :121
    if returnfps:
        new_points, grouped_xyz = ops.group(xyz, points, new_xyz, idx, want_grouped_xyz=True)   # :123-132
        return new_xyz, new_points, grouped_xyz, fps_idx
    return new_xyz, ops.group(xyz, points, new_xyz, idx)
