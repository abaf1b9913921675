"""Training-pair construction on the GPU: the reference's `CADDataset` — single cut
(dataset.py:1165-1190) and the double-cut variants of `__getitem__` (:1203-1355) — and `BuildingDataset` (:1370-1429),
each followed by `MovedCADDataset2.__getitem__` (:98-105), batched, on the kernels of the hot path.

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
draw with the reference's functions (tests/golden/make_golden_data.py -> data.npz).  The double-cut variants are the
same machinery on REGIONS of two planes (`make_pairs_regions`; `plan_double_cut_like_reference` replays the
reference's branch decisions and draws); `building_pairs` is the item contract of `BuildingDataset` (two given pieces).
The solid cuts (sphere, cylinder, cone: dataset.py:716-758) are `solid_cut_mask` + `make_pairs_solid`: the reference
asks open3d (0.15.2, README.md:25) for the signed distance to a tessellated mesh and keeps `distance < 0`.  open3d is not
in this image, so its published mesh construction is restated (oracle/solids.py: vertex formulas of create_sphere /
create_cylinder / create_cone at resolution 50) and `solid_cut_mask` evaluates membership of exactly those convex
polyhedra in closed form; tests/test_datapipe_cpu.py holds it to the oracle's brute-force face-plane test point for
point.  Parity is pinned to that restatement, not to the library itself (no fixture can be produced here).
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


def rotation_from_axis_angle(w):
    """open3d.geometry.get_rotation_matrix_from_axis_angle(w): Rodrigues rotation by |w| about w / |w| (the reference
    feeds np.random.rand(3,1), dataset.py:735, :751).  w [B,3] float64 -> R [B,3,3] float64."""
    w = w.to(torch.float64)
    th = w.norm(dim=1, keepdim=True).clamp_min(1e-300)
    k = w / th
    K = torch.zeros(w.shape[0], 3, 3, dtype=torch.float64, device=w.device)
    K[:, 0, 1], K[:, 0, 2], K[:, 1, 0] = -k[:, 2], k[:, 1], k[:, 2]
    K[:, 1, 2], K[:, 2, 0], K[:, 2, 1] = -k[:, 0], -k[:, 1], k[:, 0]
    th = th.unsqueeze(-1)
    eye = torch.eye(3, dtype=torch.float64, device=w.device).expand_as(K)
    return eye + torch.sin(th) * K + (1 - torch.cos(th)) * (K @ K)


_MESH_RES = 50        # resolution of the reference's three meshes (dataset.py:717, :733, :750)


def _sphere_poly_inside(q, radius, res):
    """Strictly inside the polyhedron of open3d's create_sphere(radius, res) centred at the origin (oracle/solids.py has
    the vertex formulas): rings at polar angles alpha_i = i pi / res, 2 res meridians.  The polyhedron is convex and
    contains the origin, so q is inside iff it is on the inner side of the face its ray from the origin pierces; that
    face lies in q's own meridian sector (meridian edges project to meridians) and in its own latitude band or a
    neighbouring one (ring chords project to great-circle arcs that bulge across the latitude circle by less than a
    band): three plane tests.  q [...,3] float64."""
    step = torch.pi / res
    rho = q.norm(dim=-1).clamp_min(1e-300)
    alpha = torch.acos((q[..., 2] / rho).clamp(-1.0, 1.0))
    theta = torch.atan2(q[..., 1], q[..., 0])
    theta = torch.where(theta < 0, theta + 2 * torch.pi, theta)
    j = torch.clamp((theta / step).floor(), 0, 2 * res - 1)
    i0 = torch.clamp((alpha / step).floor(), 0, res - 1)
    t0, t1 = j * step, (j + 1) * step

    def ring(i, t):          # vertex of ring i (0 = north pole, res = south pole) on meridian t
        a = i * step
        return torch.stack([radius * torch.sin(a) * torch.cos(t), radius * torch.sin(a) * torch.sin(t), radius * torch.cos(a)], -1)

    inside = torch.ones(q.shape[:-1], dtype=torch.bool, device=q.device)
    for di in (-1, 0, 1):
        i = torch.clamp(i0 + di, 0, res - 1)
        # three non-collinear corners of the face (band i, sector j): a triangle at the poles, a planar trapezoid elsewhere
        a_ = ring(i, t0)
        b_ = torch.where((i == 0).unsqueeze(-1), ring(i + 1, t0), ring(i, t1))
        c_ = ring(i + 1, t1)
        n = torch.cross(b_ - a_, c_ - a_, dim=-1)
        n = n * torch.sign((n * a_).sum(-1, keepdim=True) + (n * c_).sum(-1, keepdim=True))      # outward (origin inside)
        n = n / n.norm(dim=-1, keepdim=True)
        inside &= (n * q).sum(-1) < (n * c_).sum(-1)
    return inside


def _ngon_inside(x, y, radius, res):
    """Strictly inside the regular res-gon with vertices radius (cos(2 pi j / res), sin(2 pi j / res))."""
    step = 2 * torch.pi / res
    theta = torch.atan2(y, x)
    theta = torch.where(theta < 0, theta + 2 * torch.pi, theta)
    j = torch.clamp((theta / step).floor(), 0, res - 1)
    phi = (j + 0.5) * step                                   # outward normal of the edge (V_j, V_j+1)
    return x * torch.cos(phi) + y * torch.sin(phi) < radius * torch.cos(torch.tensor(step / 2, dtype=x.dtype, device=x.device))


def solid_cut_mask(raw, kind, rot=None, shift=None, exact_solid=False):
    """up-mask (signed distance < 0 = strictly INSIDE the closed mesh) of the reference's mesh cuts:
      "sphere"   (dataset.py:716-730): create_sphere(0.5, 50), centre = shift (np.random.rand(3,1)/3)
      "cylinder" (:732-747): create_cylinder(0.6, 1, 50) about z, rotated by the axis-angle vector rot
                 (np.random.rand(3,1)) about the origin, then translated by shift (np.random.rand(3,1)/3)
      "cone"     (:749-763): create_cone(1, 2, 50) translated by (0,0,-1) (base at z = -1, apex at z = +1), rotated by
                 rot about the origin
    evaluated on the POLYHEDRA open3d 0.15.2 builds (resolution 50: a 4902-vertex UV sphere, a 50-gon prism, a 50-gon
    pyramid; vertex formulas restated in oracle/solids.py) in closed form: the meshes are convex, so membership is a
    handful of plane tests per point.  exact_solid=True: the smooth solids instead (round 3's form; differs in a band of
    <= 0.1 % of the radius under the surface).  raw [B,M,3] fp32, rot / shift [B,3] float64 (the draws) -> bool [B,M];
    float64 like the plane cut."""
    p = raw.to(torch.float64)
    res = _MESH_RES
    if kind == "sphere":
        d = p - shift.to(torch.float64).unsqueeze(1)
        if exact_solid:
            return (d * d).sum(-1) < 0.25
        return _sphere_poly_inside(d, 0.5, res)
    R = rotation_from_axis_angle(rot)                     # mesh point = R x (+ shift): x = R^T (p - shift)
    if kind == "cylinder":
        q = torch.einsum("bji,bmj->bmi", R, p - shift.to(torch.float64).unsqueeze(1))
        if exact_solid:
            return (q[..., 0] ** 2 + q[..., 1] ** 2 < 0.36) & (q[..., 2].abs() < 0.5)
        return _ngon_inside(q[..., 0], q[..., 1], 0.6, res) & (q[..., 2].abs() < 0.5)
    if kind == "cone":
        q = torch.einsum("bji,bmj->bmi", R, p)
        h = q[..., 2] + 1.0                               # height above the base plane, apex at h = 2
        if exact_solid:
            rad = (q[..., 0] ** 2 + q[..., 1] ** 2).sqrt()
            return (h > 0) & (h < 2) & (rad < 1.0 - h / 2)
        # side face of sector j: the plane through the apex (0,0,2) and the base edge (V_j, V_j+1); its outward normal is
        # (cos(phi) 2, sin(phi) 2, cos(step / 2)) up to scale, phi = (j + 1/2) step: 2 (x cos phi + y sin phi) + c h < 2 c
        step = 2 * torch.pi / res
        theta = torch.atan2(q[..., 1], q[..., 0])
        theta = torch.where(theta < 0, theta + 2 * torch.pi, theta)
        j = torch.clamp((theta / step).floor(), 0, res - 1)
        phi = (j + 0.5) * step
        c = torch.cos(torch.tensor(step / 2, dtype=torch.float64, device=q.device))
        side = 2.0 * (q[..., 0] * torch.cos(phi) + q[..., 1] * torch.sin(phi)) + c * h < 2.0 * c
        return (h > 0) & side
    raise _lib.PznError(f"solid_cut_mask: unknown solid {kind!r}")


def make_pairs_mask(raw, mask, start_up, start_down, twist, n=1024, k=128, cap=None):
    """The pair construction of make_pairs from a given up-mask [B,M] (whatever cut produced it)."""
    if not raw.is_cuda:
        raise _lib.PznError("datapipe.make_pairs_mask runs on the GPU (puzzlenet_amd has no CPU fallback)")
    raw = raw.to(torch.float32).contiguous()
    B, M, _ = raw.shape
    cap = M if cap is None else int(cap)
    up_piece, n_up = _compact(raw, mask, cap)
    down_piece, n_down = _compact(raw, ~mask, cap)
    ok = (n_up >= n) & (n_down >= n) & (n_up <= cap) & (n_down <= cap)
    both = fps_to_n(torch.cat([up_piece, down_piece], 0), torch.cat([n_up, n_down], 0),
                    torch.cat([start_up.reshape(-1), start_down.reshape(-1)], 0), n)
    up, down = both[:B].contiguous(), both[B:].contiguous()
    downb, upb, down_mask, up_mask = boundary(down, up, k)
    moved, igt = move(up, twist)
    return (down, moved, igt, up, downb, upb, down_mask, up_mask), ok


def make_pairs_solid(raw, kind, rot, shift, start_up, start_down, twist, n=1024, k=128, cap=None):
    """CADDataset with slice = sphere_split / cylinder_split / cone_split (dataset.py:1463-1546 keys *_sphere, *_cyl,
    *_cone; BASELINE configs[0] "bed_sphere"): the mesh's inside is `up` (parity: see the module header)."""
    return make_pairs_mask(raw, solid_cut_mask(raw, kind, rot, shift), start_up, start_down, twist, n, k, cap)


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


def fps_to_n(piece, count, start, n, background=False):
    """dataset.py:1147-1163 on every piece of the batch: farthest point sampling from `start`, points returned in
    selection order.  piece [B,cap,3] from _compact (padding = copies of row 0)."""
    if int(piece.shape[1]) > 32768:
        raise _lib.PznUnsupported(f"fps_to_n: pieces of up to {piece.shape[1]} points (the FPS kernel holds <= 32768)")
    idx = ops.farthest_point_sample(piece, n, start.to(torch.int64), background=background, counts=count if background else None)
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


def make_pairs(raw, normal, z, start_up, start_down, twist, n=1024, k=128, cap=None, background=False):
    """raw [B,M,3] fp32 on the GPU + the draws -> the 8-tuple (down, moved_up, igt, up, down_boundary, up_boundary,
    down_mask, up_mask) and `ok` [B] (both pieces of the cut hold >= n points: the reference re-draws the cut
    otherwise, dataset.py:1176-1180 — the caller re-draws for the rows where ok is False).  background=True: the sampling as
    the small-footprint launch that skips the padding (for a side stream beside a training step: PairFeeder)."""
    if not raw.is_cuda:
        raise _lib.PznError("datapipe.make_pairs runs on the GPU (puzzlenet_amd has no CPU fallback)")
    raw = raw.to(torch.float32).contiguous()
    B, M, _ = raw.shape
    cap = M if cap is None else int(cap)
    mask = plane_cut_mask(raw, normal, z)
    up_piece, n_up = _compact(raw, mask, cap)
    down_piece, n_down = _compact(raw, ~mask, cap)
    ok = (n_up >= n) & (n_down >= n) & (n_up <= cap) & (n_down <= cap)      # (a piece larger than `cap` would be truncated)
    # one FPS launch for both pieces of every sample (a workgroup per piece: 2B workgroups instead of 2 x B)
    both = fps_to_n(torch.cat([up_piece, down_piece], 0), torch.cat([n_up, n_down], 0),
                    torch.cat([start_up.reshape(-1), start_down.reshape(-1)], 0), n, background=background)
    up, down = both[:B].contiguous(), both[B:].contiguous()
    downb, upb, down_mask, up_mask = boundary(down, up, k)
    moved, igt = move(up, twist)
    return (down, moved, igt, up, downb, upb, down_mask, up_mask), ok


def cut_pairs(raw, normals, zs, u, twist, n=1024, k=128, cap=None):
    """make_pairs for a loader that runs BESIDE the training step (PairFeeder): the cut with its re-draw (K candidate planes per
    sample, the first valid one taken on the device), both pieces compacted and padded, the FPS start indices - one launch
    (csrc/datapipe.hip, ops.cut_compact); sampling by the small-footprint FPS that skips the padding; boundary picks by
    ops.topk_rows; masks by one launch.  raw [B,M,3]; normals [B,K,3], zs [B,K], u [B,2] float64 draws; twist [B,6].
    -> ((down, moved_up, igt, up, down_boundary, up_boundary, down_mask, up_mask), ok [B], plane [B,4])"""
    if not raw.is_cuda:
        raise _lib.PznError("datapipe.cut_pairs runs on the GPU (puzzlenet_amd has no CPU fallback)")
    B, M, _ = raw.shape
    cap = M if cap is None else int(cap)
    if cap > 32768:
        raise _lib.PznUnsupported(f"cut_pairs: pieces of up to {cap} points (the FPS kernel holds <= 32768)")
    pieces, counts, start, plane, ok = ops.cut_compact(raw, normals, zs, u, n, cap)
    # (a valid cut leaves >= n points on either side, so a piece holds <= M - n; rows of samples without a valid cut are re-drawn)
    idx = ops.farthest_point_sample(pieces, n, start, background=True, counts=counts, max_count=max(M - n, n))      # dataset.py:1147-1163
    both = ops.index_points(pieces, idx)
    up, down = both[:B], both[B:]
    cd_over_up, cd_over_down = ops.chamfer(down, up)                                             # dataset.py:1357-1367
    top = ops.topk_rows(torch.cat([cd_over_up, cd_over_down], 0).neg_(), k)                      # [2B,k]: up picks, down picks
    bnd = ops.index_points(both, top)
    masks = ops.pick_mask(top, n)
    g = se3.exp(twist.to(torch.float32))                                                         # transforms.py:176-186
    moved = se3.transform_points(g, up)
    return (down, moved, g, up, bnd[B:], bnd[:B], masks[B:], masks[:B]), ok & (counts[:B] >= n) & (counts[B:] >= n), plane


class PairBatch(list):
    """The 8-tuple of a training batch + `ready`: the event behind which its tensors exist (they were produced on the
    feeder's stream), + `ok` [B] (a valid plane was among the candidates)."""
    ready = None
    ok = None
    plane = None      # (normal [B,3], z [B]) float64: the plane each sample was cut with


class PairFeeder:
    """What the reference's DataLoader(num_workers=64) over CADDataset + MovedCADDataset2 does for the trainer (train.py:101-104,
    dataset.py:1165-1190, 98-105): a FRESH batch of pairs per step from resident raw clouds - plane cut with re-draw, FPS of
    both pieces to n points, boundary labels, random rigid motion - built by cut_pairs on a background stream, so that batch
    k + 1 is cut and sampled while step k trains.  The host part of a batch is a handful of draws (K candidate planes, two
    uniform numbers for the FPS start points, a twist) from PRIVATE generators and one pinned, asynchronous upload: nothing
    in next_batch() waits for the device.  engine.TrainStep.step(next_batch=feeder.next_batch()) orders its streams behind
    `ready` and keeps the tensors alive across the streams that read them."""

    def __init__(self, raw, device, n=1024, k=128, mag=0.8, candidates=16, seed=0):
        raw = torch.as_tensor(raw, dtype=torch.float32)
        if raw.dim() != 3 or raw.shape[2] != 3:
            raise _lib.PznError("PairFeeder: raw clouds as [B, M, 3]")
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.PznError("PairFeeder runs on the GPU (puzzlenet_amd has no CPU fallback)")
        self.raw = raw.to(self.device).contiguous()
        self.n, self.k, self.mag, self.K = int(n), int(k), float(mag), int(candidates)
        self.rng = np.random.RandomState(seed)
        self.gen = torch.Generator().manual_seed(seed)
        self.stream = torch.cuda.Stream(device=self.device)
        B = self.raw.shape[0]
        # one pinned staging block per batch in flight (two: the upload of batch k + 1 may still be queued when k + 2 is drawn)
        self._width = self.K * 3 + self.K + 2 + 6
        self._stage = [torch.empty((B, self._width), dtype=torch.float64, pin_memory=True) for _ in range(3)]
        self._turn = 0
        self._busy = [None] * 3

    def next_batch(self):
        B, K, n = self.raw.shape[0], self.K, self.n
        st = self._stage[self._turn]
        if self._busy[self._turn] is not None:
            self._busy[self._turn].synchronize()          # (three batches back: long done)
        h = st.numpy()
        h[:, :3 * K] = self.rng.rand(B, 3 * K)                                  # plane normals, dataset.py:767
        h[:, 3 * K:4 * K] = self.rng.rand(B, K) / 3                             # plane offsets, :769
        h[:, 4 * K:4 * K + 2] = self.rng.rand(B, 2)                             # FPS start points as fractions of the piece sizes, :1153
        x = torch.randn(B, 6, generator=self.gen, dtype=torch.float64)          # transforms.py:163-168
        h[:, 4 * K + 2:] = (x / x.norm(p=2, dim=1, keepdim=True) * self.mag).numpy()
        with torch.cuda.stream(self.stream):
            d = st.to(self.device, non_blocking=True)
            up_ev = torch.cuda.Event()
            up_ev.record(self.stream)
            self._busy[self._turn] = up_ev
            tensors, ok, plane = cut_pairs(self.raw, d[:, :3 * K].reshape(B, K, 3), d[:, 3 * K:4 * K], d[:, 4 * K:4 * K + 2],
                                           d[:, 4 * K + 2:], n=n, k=self.k)
            ready = torch.cuda.Event()
            ready.record(self.stream)
        self._turn = (self._turn + 1) % 3
        out = PairBatch(tensors)
        out.ready, out.ok, out.plane = ready, ok, (plane[:, :3], plane[:, 3])
        return out

    def close(self):
        self.stream.synchronize()
        self._busy = [None] * 3


# ---------------------------------------------------------------------------------------------------------------------
# Double-cut variants (dataset.py:1203-1355) and BuildingDataset (:1370-1429)
#
# Every pair the reference's `CADDataset.__getitem__` can return is (U, D) = two point sets defined by the sides of at
# most two planes: plane 1 cuts the cloud into up / down (:1208), plane 2 cuts `up` (slice_seed 1) or `down`
# (slice_seed 2) into uppc / downpc (:1222, :1296).  A REGION is a set of (side of plane 1, side of plane 2) cells,
# written as a 4-bit table with bit 2*s1 + s2 (s = 1: dis >= 0); a piece is an ordered list of up to two regions
# (np.vstack((other, down)) at :1237 keeps `other` first: the order decides which point a start index names).
UP, DOWN = 0b1100, 0b0011                      # sides of plane 1, either side of plane 2
UP_UPPC, UP_DOWNPC = 0b1000, 0b0100            # plane 2 inside `up`
DOWN_UPPC, DOWN_DOWNPC = 0b0010, 0b0001        # plane 2 inside `down`


def _compact_segments(raw, seg, cap):
    """Rows with seg 0, then rows with seg 1 (original order inside each), padded to `cap` with copies of the first
    kept row; seg 2 = left out.  -> (packed [B,cap,3], count [B])"""
    count = (seg < 2).sum(1)
    order = torch.sort(seg.to(torch.int8), dim=1, stable=True)[1]
    packed = torch.gather(raw, 1, order[:, :cap].unsqueeze(-1).expand(-1, -1, 3))
    pos = torch.arange(cap, device=raw.device).unsqueeze(0)
    first = packed[:, :1, :]
    return torch.where((pos < count.unsqueeze(1)).unsqueeze(-1), packed, first.expand(-1, cap, -1)).contiguous(), count


def _segments(code, tab):
    """code [B,M] in 0..3, tab [B,2] region tables -> segment id 0 / 1 / 2 (left out) per point."""
    in0 = (tab[:, 0:1] >> code) & 1
    in1 = (tab[:, 1:2] >> code) & 1
    return torch.where(in0 == 1, torch.zeros_like(code), torch.where(in1 == 1, torch.ones_like(code), torch.full_like(code, 2)))


def pairs_from_pieces(U, D, twist, k=128):
    """(U, D) N-point pieces -> the 8-tuple of MovedCADDataset2.__getitem__ (:98-105) on top of get_boundary(D, U)
    (:1357-1367): (D, moved U, igt, U, D boundary, U boundary, D mask, U mask)."""
    Db, Ub, Dm, Um = boundary(D, U, k)
    moved, igt = move(U, twist)
    return D, moved, igt, U, Db, Ub, Dm, Um


def make_pairs_regions(raw, normal1, z1, normal2, z2, u_tab, d_tab, start_u, start_d, twist, n=1024, k=128, cap=None):
    """The general form of make_pairs: raw [B,M,3] on the GPU, two planes per sample, and for each of the two pieces an
    ordered pair of region tables (u_tab, d_tab: int64 [B,2]; a second table of 0 = single region).  U = FPS(n) of the
    points of u_tab from start_u, D likewise; -> (8-tuple, ok [B]).  The single cut is u_tab = (UP, 0), d_tab = (DOWN, 0)."""
    if not raw.is_cuda:
        raise _lib.PznError("datapipe.make_pairs_regions runs on the GPU (puzzlenet_amd has no CPU fallback)")
    raw = raw.to(torch.float32).contiguous()
    B, M, _ = raw.shape
    cap = M if cap is None else int(cap)
    s1 = plane_cut_mask(raw, normal1, z1)
    s2 = plane_cut_mask(raw, normal2, z2)
    code = 2 * s1.to(torch.int64) + s2.to(torch.int64)
    u_piece, n_u = _compact_segments(raw, _segments(code, u_tab.to(raw.device)), cap)
    d_piece, n_d = _compact_segments(raw, _segments(code, d_tab.to(raw.device)), cap)
    ok = (n_u >= n) & (n_d >= n) & (n_u <= cap) & (n_d <= cap)
    both = fps_to_n(torch.cat([u_piece, d_piece], 0), torch.cat([n_u, n_d], 0),
                    torch.cat([start_u.reshape(-1), start_d.reshape(-1)], 0), n)
    U, D = both[:B].contiguous(), both[B:].contiguous()
    return pairs_from_pieces(U, D, twist, k), ok


def building_pairs(fpcs, rpcs, twist, k=128):
    """BuildingDataset.__getitem__ (:1423-1429) for a batch: the item is (rpc, fpc, get_boundary(fpc, rpc)), i.e. U = the
    roof cloud, D = the facade cloud, both given (no cut, no sampling); -> the 8-tuple of MovedCADDataset2 on top."""
    if not (fpcs.is_cuda and rpcs.is_cuda):
        raise _lib.PznError("datapipe.building_pairs runs on the GPU (puzzlenet_amd has no CPU fallback)")
    return pairs_from_pieces(rpcs.to(torch.float32).contiguous(), fpcs.to(torch.float32).contiguous(), twist, k)


def _plane_draw():
    return np.random.rand(3, 1), np.random.rand(1) / 3          # dataset.py:767-769 (z=None)


def _side(pts, normal, z):
    return (np.dot(pts, normal) + z >= 0).reshape(-1)            # :770-771


def plan_double_cut_like_reference(raw, accept, n=1024, mag=0.8):
    """The branch decisions and random draws of ONE `CADDataset.__getitem__` call with split_twice=True
    (dataset.py:1203-1355) followed by MovedCADDataset2 (:98-105), made from numpy's and torch's GLOBAL generators in the
    reference's order (seed them as the reference run would), as a recipe for make_pairs_regions.
      raw     [M,3] float32 numpy array (host side: piece sizes steer the branches and the start-index ranges)
      accept  callable(recipe without motion) -> float: the chamfer distance of the two boundaries of the candidate pair —
              the one decision of the reference that needs the pieces themselves (:1250-1253, :1322-1325: the pair is
              kept when it is <= 0.015).  Called at most once.
    -> dict(kind, normal1, z1, normal2, z2, u_tab, d_tab, s_u, s_d, twist)"""
    raw = np.asarray(raw, dtype=np.float32)
    none = (np.zeros((3, 1)), np.zeros(1))

    def recipe(kind, planes, u_tab, d_tab, s_u, s_d):
        (n1, z1), (n2, z2) = planes
        return dict(kind=kind, normal1=n1.reshape(3), z1=z1.reshape(1), normal2=n2.reshape(3), z2=z2.reshape(1),
                    u_tab=np.array(u_tab, np.int64), d_tab=np.array(d_tab, np.int64), s_u=int(s_u), s_d=int(s_d))

    def single(p1, s1):                                          # self.slice(pc, None, up, down), :1192-1201
        while int(s1.sum()) < n or int((~s1).sum()) < n:
            p1 = _plane_draw()
            s1 = _side(raw, *p1)
        s_u = np.random.randint(0, int(s1.sum()))
        s_d = np.random.randint(0, int((~s1).sum()))
        return recipe("single", (p1, none), (UP, 0), (DOWN, 0), s_u, s_d)

    def item():
        slice_seed = int(torch.randint(0, 3, (1,)))              # :1206-1207
        p1 = _plane_draw()                                       # :1208
        s1 = _side(raw, *p1)
        n_up, n_down = int(s1.sum()), int((~s1).sum())
        if slice_seed == 1 and n_up < 3000:                      # :1211-1214
            slice_seed = 2
        if slice_seed == 2 and n_down < 3000:
            slice_seed = 1
        if slice_seed == 0:
            return single(p1, s1)
        inner = s1 if slice_seed == 1 else ~s1                   # the piece that is cut again
        n_other = n_down if slice_seed == 1 else n_up            # the piece that is not
        sub = raw[inner]
        p2 = _plane_draw()                                       # :1222 / :1296
        s2 = _side(sub, *p2)
        tries = 0
        while tries <= 5 and (int(s2.sum()) < n or int((~s2).sum()) < n):
            p2 = _plane_draw()
            s2 = _side(sub, *p2)
            tries += 1
        n_a, n_b = int(s2.sum()), int((~s2).sum())               # uppc, downpc
        if n_a < n or n_b < n:                                   # :1226-1227
            return single(p1, s1)
        A, Bt = (UP_UPPC, UP_DOWNPC) if slice_seed == 1 else (DOWN_UPPC, DOWN_DOWNPC)
        OTHER = DOWN if slice_seed == 1 else UP
        se = int(torch.randint(0, 3, (1,)))                      # :1229 / :1302
        if se == 0 or n_other < n:                               # U = one half, D = the other half + the other piece
            choice = int(torch.randint(0, 2, (1,)))
            first, second = (A, Bt) if choice == 0 else (Bt, A)
            n_first, n_second = (n_a, n_b) if choice == 0 else (n_b, n_a)
            s_u = np.random.randint(0, n_first)
            s_d = np.random.randint(0, n_second + n_other)
            return recipe("half_vs_rest", (p1, p2), (first, 0), (second, OTHER), s_u, s_d)
        if se == 1:                                              # U = one half, D = the other PIECE, kept if they touch
            choice = int(torch.randint(0, 2, (1,)))
            first = A if choice == 0 else Bt
            s_u = np.random.randint(0, n_a if choice == 0 else n_b)
            s_d = np.random.randint(0, n_other)
            cand = recipe("half_vs_other", (p1, p2), (first, 0), (OTHER, 0), s_u, s_d)
            if float(accept(cand)) > 0.015:                      # :1250-1253 / :1322-1325
                return single(p1, s1)
            return cand
        s_u = np.random.randint(0, n_a)                          # se == 2: the two halves, :1255-1256 / :1326-1327
        s_d = np.random.randint(0, n_b)
        re_now = np.random.rand(1)                               # :1260 / :1330
        if not (re_now > (0.7 if slice_seed == 1 else 0.6)) and n_other >= 1200:
            np.random.randint(0, n_other + n)                    # :1281 / :1350: two more FPS runs whose results the
            np.random.randint(0, n)                              # reference overwrites (:1282-1283, :1351-1352)
        return recipe("halves", (p1, p2), (A, 0), (Bt, 0), s_u, s_d)

    rec = item()
    x = torch.randn(1, 6)                                        # MovedCADDataset2: rigid_transform(up), transforms.py:163-168
    x = x / x.norm(p=2, dim=1, keepdim=True) * mag
    torch.randn(1, 6)                                            # rigid_transform(upb), dataset.py:101: drawn, result unused
    rec["twist"] = x.reshape(6).numpy()
    return rec
