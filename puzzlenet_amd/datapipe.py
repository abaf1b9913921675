"""Training-pair construction on the GPU: the single-cut path of the reference's `CADDataset`
(dataset.py:1165-1190) + `MovedCADDataset2.__getitem__` (:98-105), batched, on the kernels of the hot path.

    raw cloud [M,3]  --plane cut (dataset.py:761-775)-->  up / down  --numpy FPS to N (:1147-1163)-->  N-point pieces
        --get_boundary (:1357-1367: chamfer both ways, the 128 points nearest to the other piece + 0/1 masks)-->
        --RandomTransformSE3 (se_math/transforms.py:151-197: unit twist * mag, exp, apply to `up`)-->
    (down, moved_up, igt, up, down_boundary, up_boundary, down_mask, up_mask)  == the 8-tuple the model consumes.

The reference does this per sample in 64 CPU workers (train.py:101: numpy FPS loops over ~5-12 k points); here a batch
is a handful of launches: the FPS of both pieces is `pzn_fps_f32` on the compacted pieces (padded with copies of their
first point: a copy never beats the original under the first-maximum rule, so the selection sequence is the
reference's), the boundary is `pzn_chamfer_fwd_f32` + top-k, the motion `pzn_se3_exp_fwd_f32`.

Randomness stays OUTSIDE: the caller passes the draws (plane normal / offset, FPS start indices, unit twists) — e.g.
`draws_like_reference` replays numpy / torch generators in the reference's order — so results can be compared draw for
draw with the reference's functions (tests/golden/make_golden_data.py -> data.npz).  The mesh-based cuts (sphere,
cylinder, cone: dataset.py:716-758) go through open3d ray casting and are out of scope (no open3d here).
"""
import numpy as np
import torch

from . import _lib, ops, se3


def draws_like_reference(raw, n=1024, mag=0.8, max_tries=100):
    """The random draws of one sample in the reference's order, from numpy's and torch's GLOBAL generators (seed them as
    the reference run would): plane normal `np.random.rand(3,1)` and offset `np.random.rand(1)/3` (dataset.py:767-769),
    re-drawn while a piece holds fewer than n points (:1176-1180); FPS starts `np.random.randint(0, n_piece)` for up
    then down (:1153, called at :1181-1182); the twist `randn(1,6)` normalised to `mag` (transforms.py:163-168).
    raw: [M,3] float32 numpy array (host side: the piece sizes decide the range of the start indices).
    -> dict(normal[3] f64, z[1] f64, s_up, s_down, twist[6] f32)"""
    raw = np.asarray(raw, dtype=np.float32)
    for _ in range(max_tries):
        normal = np.random.rand(3, 1)
        z = np.random.rand(1) / 3
        dis = np.dot(raw, normal) + z
        n_up, n_down = int((dis >= 0).sum()), int((dis < 0).sum())
        if n_up >= n and n_down >= n:
            break
    else:
        raise _lib.PznError(f"draws_like_reference: no plane left {n} points on both sides in {max_tries} tries")
    s_up = int(np.random.randint(0, n_up))
    s_down = int(np.random.randint(0, n_down))
    x = torch.randn(1, 6)
    x = x / x.norm(p=2, dim=1, keepdim=True) * mag
    return dict(normal=normal.reshape(3), z=z.reshape(1), s_up=s_up, s_down=s_down, twist=x.reshape(6).numpy())


def plane_cut_mask(raw, normal, z):
    """dataset.py:767-772: `dis = points . normal + z`, up = dis >= 0.  The reference multiplies float32 points with
    float64 draws, i.e. evaluates in float64: so does this (one fused pass over B*M*3 values)."""
    dis = (raw.to(torch.float64) * normal.to(torch.float64).unsqueeze(1)).sum(-1)
    # numpy's dot of a row with the 3-vector sums in index order; so does the line above after the product
    dis = dis + z.to(torch.float64).reshape(-1, 1)
    return dis >= 0


def _compact(raw, mask, cap):
    """Rows of `raw` where mask, in their original order, padded to `cap` rows with copies of the first kept row.
    -> (packed [B,cap,3], count [B])"""
    B, M, _ = raw.shape
    count = mask.sum(1)
    # stable partition: kept rows first, original order inside each part
    order = torch.sort((~mask).to(torch.int8), dim=1, stable=True)[1]
    packed = torch.gather(raw, 1, order[:, :cap].unsqueeze(-1).expand(-1, -1, 3))
    pos = torch.arange(cap, device=raw.device).unsqueeze(0)
    first = packed[:, :1, :]
    return torch.where((pos < count.unsqueeze(1)).unsqueeze(-1), packed, first.expand(-1, cap, -1)).contiguous(), count


def fps_to_n(piece, count, start, n):
    """dataset.py:1147-1163 on every piece of the batch: farthest point sampling from `start`, points returned in
    selection order.  piece [B,cap,3] from _compact (padding = copies of row 0)."""
    if int(piece.shape[1]) > 32768:
        raise _lib.PznUnsupported(f"fps_to_n: pieces of up to {piece.shape[1]} points (the FPS kernel holds <= 32768)")
    idx = ops.farthest_point_sample(piece, n, start.to(torch.int64))
    return torch.gather(piece, 1, idx.unsqueeze(-1).expand(-1, -1, 3))


def boundary(down, up, k=128):
    """dataset.py:1357-1367 `get_boundary(self.down, self.up)`: the k points of each piece that lie nearest to the
    other piece, and their 0/1 masks.  -> (down_boundary [B,k,3], up_boundary [B,k,3], down_mask [B,N], up_mask [B,N])"""
    cd_over_up, cd_over_down = ops.chamfer(down, up)          # min over `down` per up-point, min over `up` per down-point
    top_up = torch.topk(-cd_over_up, k, dim=1)[1]
    top_down = torch.topk(-cd_over_down, k, dim=1)[1]
    upb = torch.gather(up, 1, top_up.unsqueeze(-1).expand(-1, -1, 3))
    downb = torch.gather(down, 1, top_down.unsqueeze(-1).expand(-1, -1, 3))
    down_mask = torch.zeros(down.shape[:2], dtype=torch.float32, device=down.device).scatter_(1, top_down, 1.0)
    up_mask = torch.zeros(up.shape[:2], dtype=torch.float32, device=up.device).scatter_(1, top_up, 1.0)
    return downb, upb, down_mask, up_mask


def move(up, twist):
    """transforms.py:176-186: g = exp(x), p1 = g . p0, igt = g."""
    g = se3.exp(twist.to(torch.float32))
    moved = se3.transform(g, up.permute(0, 2, 1)).permute(0, 2, 1).contiguous()
    return moved, g


def make_pairs(raw, normal, z, start_up, start_down, twist, n=1024, k=128, cap=None):
    """raw [B,M,3] fp32 on the GPU + the draws -> the 8-tuple (down, moved_up, igt, up, down_boundary, up_boundary,
    down_mask, up_mask) and `ok` [B] (both pieces of the cut hold >= n points: the reference re-draws the cut
    otherwise, dataset.py:1176-1180 — the caller re-draws for the rows where ok is False)."""
    if not raw.is_cuda:
        raise _lib.PznError("datapipe.make_pairs runs on the GPU (puzzlenet_amd has no CPU fallback)")
    raw = raw.to(torch.float32).contiguous()
    B, M, _ = raw.shape
    cap = M if cap is None else int(cap)
    mask = plane_cut_mask(raw, normal, z)
    up_piece, n_up = _compact(raw, mask, cap)
    down_piece, n_down = _compact(raw, ~mask, cap)
    ok = (n_up >= n) & (n_down >= n)
    # one FPS launch for both pieces of every sample (a workgroup per piece: 2B workgroups instead of 2 x B)
    both = fps_to_n(torch.cat([up_piece, down_piece], 0), torch.cat([n_up, n_down], 0),
                    torch.cat([start_up.reshape(-1), start_down.reshape(-1)], 0), n)
    up, down = both[:B].contiguous(), both[B:].contiguous()
    downb, upb, down_mask, up_mask = boundary(down, up, k)
    moved, igt = move(up, twist)
    return (down, moved, igt, up, downb, upb, down_mask, up_mask), ok
