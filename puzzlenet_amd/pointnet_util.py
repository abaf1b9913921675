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

from . import ops

__all__ = ["square_distance", "index_points", "farthest_point_sample", "query_ball_point", "sample_and_group",
           "StartIndexFeed", "set_start_index_feed"]


class StartIndexFeed:
    """Stages the FPS start indices through static device buffers.

    The reference draws the first centroid with ``torch.randint(0, N, (B,))`` on the CPU generator
    at every call (pointnet_util.py:65).  A host draw + host-to-device copy cannot live inside a
    HIP graph, so when a whole training step is captured the draws are made by ``refill()`` *before*
    each replay — same generator, same order, same values as the eager path — into buffers the
    captured FPS kernels read.
    """

    def __init__(self):
        self.slots = []          # (device tensor, N) in call order
        self.cursor = 0
        self.recording = True    # first pass (capture): create slots; afterwards: slots are fixed

    def next(self, B, N, device):
        if self.recording:
            t = torch.randint(0, N, (B,), dtype=torch.long).to(device)
            self.slots.append((t, N))
            return t
        t, n = self.slots[self.cursor]
        assert n == N and t.shape[0] == B, "FPS call sequence changed since capture"
        self.cursor += 1
        return t

    def freeze(self):
        self.recording = False
        self.cursor = 0

    def refill(self):
        """Draw fresh start indices for every recorded FPS call (call before each graph replay)."""
        self.cursor = 0
        for t, n in self.slots:
            t.copy_(torch.randint(0, n, (t.shape[0],), dtype=torch.long), non_blocking=False)


_FEED = None


def set_start_index_feed(feed):
    """Install (or clear with None) a StartIndexFeed used by farthest_point_sample."""
    global _FEED
    _FEED = feed


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
        if _FEED is not None:
            return ops.farthest_point_sample(xyz, npoint, _FEED.next(B, N, xyz.device))
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
    if knn:
        idx = ops.knn(xyz, new_xyz.detach(), nsample)                 # :118-119
    else:
        idx = ops.ball_query(radius, nsample, xyz, new_xyz.detach())  # :121
    if returnfps:
        new_points, grouped_xyz = ops.group(xyz, points, new_xyz, idx, want_grouped_xyz=True)   # :123-132
        return new_xyz, new_points, grouped_xyz, fps_idx
    return new_xyz, ops.group(xyz, points, new_xyz, idx)
