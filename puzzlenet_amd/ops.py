"""Host-side operators over the C ABI (include/pzn.h).

torch is used for device memory, the current HIP stream and autograd plumbing
only: every computation below is a call into libpzn.so with raw device
pointers.  Inputs must live on a HIP device; there is no CPU path.
"""
import ctypes
import os
import weakref

import torch

from . import _lib


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_cur_device = getattr(torch._C, "_cuda_getDevice", None)


def _stream():
    """Raw handle of torch's current stream on the current device.  (torch.cuda.current_stream() builds a Stream object:
    9 us a call, 77 calls per training step; the raw getter is what torch's own C++ extensions use.)"""
    if _raw_stream is not None and _cur_device is not None:
        return _raw_stream(_cur_device())
    return torch.cuda.current_stream().cuda_stream


class _NoCtx:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


_NO_CTX = _NoCtx()


def _on(dev):
    """`with _on(dev):` = torch.cuda.device(dev), skipped when dev is already the current device (one process per GPU:
    always; the context manager cost 10 us per use, 70 uses per step)."""
    if _cur_device is not None and dev.index is not None and dev.index == _cur_device():
        return _NO_CTX
    return torch.cuda.device(dev)


class KernelTimer:
    """Optional per-entry-point device timing with HIP events recorded on the stream the
    kernels are launched on (torch's current stream).  Off by default; bench.py turns it on
    for the timed region to price individual kernels (roofline.achieved)."""
    enabled = False
    records = {}
    flops = {}
    shapes = None   # set to {} before start() to also collect {(entry point, int args): [ms, ...]} (tools/)
    variants = {}   # {(entry point, variant tag): (launches, ms, flops)} for calls that pass variant= (one kernel instantiation each)

    @classmethod
    def start(cls):
        cls.records = {}
        cls.enabled = True

    @classmethod
    def stop(cls):
        """-> {entry point: (launches, total milliseconds)}; synchronises the device."""
        cls.enabled = False
        torch.cuda.synchronize()
        out = {k: (len(v), sum(a.elapsed_time(b) for a, b, _ in v)) for k, v in cls.records.items()}
        cls.variants = {}
        for k, v in cls.records.items():
            for a, b, f in v:
                if isinstance(f, tuple) and isinstance(f[-1], str):
                    n_, ms_, fl_ = cls.variants.get((k, f[-1]), (0, 0.0, 0))
                    cls.variants[(k, f[-1])] = (n_ + 1, ms_ + a.elapsed_time(b), fl_ + f[0])
            v[:] = [(a, b, f[0] if (isinstance(f, tuple) and isinstance(f[-1], str)) else f) for a, b, f in v]
        if cls.shapes is not None:
            for k, v in cls.records.items():
                for a, b, f in v:
                    if isinstance(f, tuple):
                        cls.shapes.setdefault((k,) + f[1], []).append(a.elapsed_time(b))
            for v in cls.records.values():
                v[:] = [(a, b, f[0] if isinstance(f, tuple) else f) for a, b, f in v]
        cls.flops = {k: sum(f for _, _, f in v) for k, v in cls.records.items()}
        cls.records = {}
        return out


def ktimer_start():
    """Per-KERNEL device time from inside the library (pzn_ktimer_*, csrc/core.hip): from here on every kernel libpzn.so
    launches is bracketed by its own HIP event pair on its own stream.  Measurement only (bench.py)."""
    _lib.check(_lib.load().pzn_ktimer_collect() < 0, "pzn_ktimer_collect")      # (drop spans of an earlier window)
    _lib.check(_lib.load().pzn_ktimer_enable(1), "pzn_ktimer_enable")


def ktimer_stop():
    """-> {kernel name as in a rocprofv3 trace (template arguments included): (launches, summed milliseconds)}."""
    lib = _lib.load()
    _lib.check(lib.pzn_ktimer_enable(0), "pzn_ktimer_enable")
    n = lib.pzn_ktimer_collect()
    if n < 0:
        _lib.check(n, "pzn_ktimer_collect")
    out = {}
    buf = ctypes.create_string_buffer(512)
    cnt, ms = ctypes.c_int(0), ctypes.c_double(0.0)
    for i in range(n):
        _lib.check(lib.pzn_ktimer_row(i, buf, 512, ctypes.byref(cnt), ctypes.byref(ms)), "pzn_ktimer_row")
        out[buf.value.decode()] = (cnt.value, ms.value)
    return out


# tests set this to a list: every max-pool of the model path appends (kind, address of the layer's weight or None, arg-max
# tensor) - "sa": [B,S,C2] neighbour slots of a set-abstraction level (weight = its second layer's), "gmax": [B,Nout] rows of
# the out projection's max over the points (weight = the projection's), "maxpts": [B,C] rows of max_over_points
WINNER_CAPTURE = None


def _call(name, *args, flops=0, variant=None):
    """Invoke a C-ABI entry point; `flops` = algorithmic 2*M*N*K of the dense entry points (bench accounting); `variant` =
    the template arguments of the kernel this call launches, where one entry point has several instantiations."""
    if KernelTimer.enabled:
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        _lib.call(name, *args)
        b.record()
        if variant is not None and KernelTimer.shapes is None:
            flops = (flops, variant)
        if KernelTimer.shapes is not None:
            flops = (flops, tuple(x for x in args if isinstance(x, int) and 0 <= x < (1 << 31)))
        KernelTimer.records.setdefault(name, []).append((a, b, flops))
    else:
        _lib.call(name, *args)


def _req(t, dtype, name):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise _lib.PznError(f"{name} must be a tensor on the GPU (got {type(t).__name__}"
                            f"{'' if not isinstance(t, torch.Tensor) else ' on ' + str(t.device)}); "
                            "puzzlenet_amd has no CPU fallback")
    if t.dtype != dtype:
        t = t.to(dtype)
    return t.contiguous()


def _f32(t, name):
    return _req(t, torch.float32, name)


def _i64(t, name):
    return _req(t, torch.int64, name)


def _p(t):
    return None if t is None else t.data_ptr()


# --------------------------------------------------------------------------- point ops

def square_distance(src, dst):
    src, dst = _f32(src, "src"), _f32(dst, "dst")
    B, S, C = src.shape
    if C != 3 or dst.shape[-1] != 3 or dst.shape[0] != B:
        raise _lib.PznError("square_distance: expected src[B,S,3], dst[B,N,3]")
    N = dst.shape[1]
    out = torch.empty((B, S, N), dtype=torch.float32, device=src.device)
    with _on(src.device):
        _call("pzn_square_distance_f32", _p(src), _p(dst), B, S, N, _p(out), _stream())
    return out


def farthest_point_sample(xyz, npoint, start_idx, background=False, counts=None, max_count=0):
    """background=True: the same picks from the small-footprint launch (no LDS image of the cloud; pzn_fps_background_f32):
    for sampling that runs on a side stream beside a training step (datapipe.PairFeeder); counts [B] int64: rows >= counts[b]
    of cloud b are padding copies of its row 0 and are skipped; max_count: the caller's promise counts <= max_count (0: none)."""
    xyz = _f32(xyz, "xyz")
    B, N, C = xyz.shape
    if C != 3:
        raise _lib.PznError("farthest_point_sample: expected xyz[B,N,3]")
    start_idx = _i64(start_idx, "start_idx")
    out = torch.empty((B, npoint), dtype=torch.int64, device=xyz.device)
    with _on(xyz.device):
        if background:
            counts = None if counts is None else _i64(counts, "counts")
            _call("pzn_fps_background_f32", _p(xyz), B, N, int(npoint), _p(start_idx), _p(out), _p(counts), int(max_count), _stream())
        else:
            _call("pzn_fps_f32", _p(xyz), B, N, int(npoint), _p(start_idx), _p(out), _stream())
    return out


def knn(xyz, new_xyz, K):
    xyz, new_xyz = _f32(xyz, "xyz"), _f32(new_xyz, "new_xyz")
    B, N, _ = xyz.shape
    S = new_xyz.shape[1]
    out = torch.empty((B, S, K), dtype=torch.int64, device=xyz.device)
    with _on(xyz.device):
        _call("pzn_knn_f32", _p(xyz), _p(new_xyz), B, N, S, int(K), _p(out), _stream())
    return out


def ball_query(radius, nsample, xyz, new_xyz):
    xyz, new_xyz = _f32(xyz, "xyz"), _f32(new_xyz, "new_xyz")
    B, N, _ = xyz.shape
    S = new_xyz.shape[1]
    out = torch.empty((B, S, nsample), dtype=torch.int64, device=xyz.device)
    # `sqrdists > radius ** 2` is an fp32 comparison in the reference (pointnet_util.py:91)
    r2 = float(torch.tensor(float(radius) ** 2, dtype=torch.float32))
    with _on(xyz.device):
        _call("pzn_ball_query_f32", r2, int(nsample), _p(xyz), _p(new_xyz), B, N, S, _p(out), _stream())
    return out


class _Gather(torch.autograd.Function):
    """index_points (pointnet_util.py:39-50): forward gather, backward scatter-add."""

    @staticmethod
    def forward(ctx, points, idx):
        points = _f32(points, "points")
        idx = _i64(idx, "idx")
        B, N, C = points.shape
        flat = idx.reshape(B, -1)
        M = flat.shape[1]
        out = torch.empty((B, M, C), dtype=torch.float32, device=points.device)
        with _on(points.device):
            _call("pzn_gather_fwd_f32", _p(points), _p(flat), B, N, M, C, _p(out), _stream())
        ctx.save_for_backward(flat)
        ctx.dims = (B, N, M, C)
        return out.reshape(*idx.shape, C)

    @staticmethod
    def backward(ctx, grad_out):
        (flat,) = ctx.saved_tensors
        B, N, M, C = ctx.dims
        grad_out = _f32(grad_out, "grad_out")
        g = torch.zeros((B, N, C), dtype=torch.float32, device=grad_out.device)
        with _on(grad_out.device):
            _call("pzn_gather_bwd_f32", _p(grad_out), _p(flat), B, N, M, C, _p(g), _stream())
        return g, None


def index_points(points, idx):
    return _Gather.apply(points, idx)


class _MaxOverPoints(torch.autograd.Function):
    """torch.max(x, dim=1)[0] for x[B, L, C] (model5_b.py:475, :741): one pass forward (value + arg-max row), one
    streaming pass backward that writes every element of dx (no zero fill + scatter)."""

    @staticmethod
    def forward(ctx, x):
        x = _f32(x, "x")
        B, L, C = x.shape
        out = torch.empty((B, C), dtype=torch.float32, device=x.device)
        idx = torch.empty((B, C), dtype=torch.int32, device=x.device)
        with _on(x.device):
            _call("pzn_maxpool_points_fwd_f32", _p(x), B, L, C, _p(out), _p(idx), _stream())
        if WINNER_CAPTURE is not None:
            WINNER_CAPTURE.append(("maxpts", None, idx))
        ctx.save_for_backward(idx)
        ctx.dims = (B, L, C)
        return out

    @staticmethod
    def backward(ctx, dout):
        (idx,) = ctx.saved_tensors
        B, L, C = ctx.dims
        dout = _f32(dout, "dout")
        dx = torch.empty((B, L, C), dtype=torch.float32, device=dout.device)
        with _on(dout.device):
            _call("pzn_maxpool_points_bwd_f32", _p(dout), _p(idx), B, L, C, _p(dx), _stream())
        return dx


class _Se3Exp(torch.autograd.Function):
    """se3.exp (se_math/se3.py:57-80): twist [B,6] -> [B,4,4], one launch each way."""

    @staticmethod
    def forward(ctx, x):
        x = _f32(x, "twist")
        B = x.shape[0]
        g = torch.empty((B, 4, 4), dtype=torch.float32, device=x.device)
        with _on(x.device):
            _call("pzn_se3_exp_fwd_f32", _p(x), B, _p(g), _stream())
        ctx.save_for_backward(x)
        return g

    @staticmethod
    def backward(ctx, dg):
        (x,) = ctx.saved_tensors
        dg = _f32(dg, "dg")
        dx = torch.empty_like(x)
        with _on(x.device):
            _call("pzn_se3_exp_bwd_f32", _p(x), _p(dg), x.shape[0], _p(dx), _stream())
        return dx


class _Se3Transform(torch.autograd.Function):
    """se3.transform (se_math/se3.py:110-120) on points: out[b,n,:] = R_b p[b,n,:] + t_b, one launch each way."""

    @staticmethod
    def forward(ctx, g, p):
        g, p = _f32(g, "g"), _f32(p, "points")
        B, N, _ = p.shape
        out = torch.empty_like(p)
        with _on(p.device):
            _call("pzn_se3_transform_fwd_f32", _p(g), _p(p), B, N, _p(out), _stream())
        ctx.save_for_backward(g, p)
        return out

    @staticmethod
    def backward(ctx, dout):
        g, p = ctx.saved_tensors
        B, N, _ = p.shape
        dout = _f32(dout, "dout")
        dg = torch.empty_like(g) if ctx.needs_input_grad[0] else None
        dp = torch.empty_like(p) if ctx.needs_input_grad[1] else None
        if dg is None and dp is None:
            return None, None
        with _on(p.device):
            _call("pzn_se3_transform_bwd_f32", _p(g), _p(p), _p(dout), B, N, _p(dp), _p(dg), _stream())
        return dg, dp


def se3_transform_points(g, p):
    """g [B,4,4], p [B,N,3] on the GPU -> [B,N,3]"""
    if g.dim() != 3 or p.dim() != 3 or g.shape[0] != p.shape[0] or p.shape[2] != 3 or tuple(g.shape[1:]) != (4, 4):
        raise _lib.PznError(f"se3_transform_points expects g[B,4,4], p[B,N,3]; got {tuple(g.shape)}, {tuple(p.shape)}")
    return _Se3Transform.apply(g, p)


class _CompLoss(torch.autograd.Function):
    """TouchedRegraster.comp (model5_b.py:1512-1519): 16 * mean((g igt - I)^2), one launch each way."""

    @staticmethod
    def forward(ctx, g, igt):
        g, igt = _f32(g, "g"), _f32(igt, "igt")
        loss = torch.empty((1,), dtype=torch.float32, device=g.device)
        with _on(g.device):
            _call("pzn_comp_fwd_f32", _p(g), _p(igt), g.shape[0], _p(loss), _stream())
        ctx.save_for_backward(g, igt)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, dloss):
        g, igt = ctx.saved_tensors
        dl = _f32(dloss, "dloss").reshape(1)
        dg = torch.empty_like(g)
        with _on(g.device):
            _call("pzn_comp_bwd_f32", _p(g), _p(igt), _p(dl), g.shape[0], _p(dg), _stream())
        return dg, None


def comp_loss(g, igt):
    return _CompLoss.apply(g, igt)


class _BoundaryCE(torch.autograd.Function):
    """F.cross_entropy(logits[B,2,N], labels[B,N]) (model5_b.py:1063-1064) together with softmax(logits, 1)[:, 1, :]
    (:1085-1090) -> (loss, prob1); prob1 only feeds the (non-differentiable) top-128 selection."""

    @staticmethod
    def forward(ctx, logits, labels):
        labels = _f32(labels, "labels")
        if not isinstance(logits, torch.Tensor) or logits.dim() != 3:
            raise _lib.PznError("boundary_ce expects logits[B,2,N]")
        B, C, N = logits.shape
        # the heads' [B,N,2] output seen through permute(0,2,1) (model5_b.py:751-754) is read through its strides: no transposed
        # copy here, and the gradient goes back in the same layout (contiguous for the heads' backward)
        pm = logits.is_cuda and logits.dtype == torch.float32 and C == 2 and logits.stride() == (2 * N, 1, 2)
        if not pm:
            logits = _f32(logits, "logits")
        if C != 2 or tuple(labels.shape) != (B, N):
            raise _lib.PznError(f"boundary_ce expects logits[B,2,N], labels[B,N]; got {tuple(logits.shape)}, {tuple(labels.shape)}")
        prob1 = torch.empty((B, N), dtype=torch.float32, device=logits.device)
        loss = torch.empty((513,), dtype=torch.float32, device=logits.device)   # PZN_BOUNDARY_CE_LOSS_FLOATS: value + partials
        with _on(logits.device):
            _call("pzn_boundary_ce_fwd_f32", _p(logits), _p(labels), B, N, int(pm), _p(prob1), _p(loss), _stream())
        ctx.save_for_backward(logits, labels)
        ctx.points_major = pm
        ctx.mark_non_differentiable(prob1)
        ctx.set_materialize_grads(False)      # (prob1 carries no gradient: no zero-filled [B,N] stand-in for it)
        return loss[0], prob1

    @staticmethod
    def backward(ctx, dloss, _dprob):
        logits, labels = ctx.saved_tensors
        if dloss is None:
            return None, None
        B, _, N = logits.shape
        dl = _f32(dloss, "dloss").reshape(1)
        pm = ctx.points_major
        dlogits = torch.empty((B, N, 2), dtype=torch.float32, device=logits.device).permute(0, 2, 1) if pm else torch.empty_like(logits)
        with _on(logits.device):
            _call("pzn_boundary_ce_bwd_f32", _p(logits), _p(labels), _p(dl), B, N, int(pm), _p(dlogits), _stream())
        return dlogits, None


def boundary_ce(logits, labels):
    """-> (cross-entropy loss, class-1 probability [B,N])"""
    return _BoundaryCE.apply(logits, labels)


def topk_rows(x, k):
    """torch.topk(x, k, dim=1)[1] for x[R,N] on the GPU (value descending, ties by ascending index)."""
    x = _f32(x.detach(), "x")
    R, N = x.shape
    idx = torch.empty((R, int(k)), dtype=torch.int64, device=x.device)
    with _on(x.device):
        _call("pzn_topk_rows_f32", _p(x), R, N, int(k), _p(idx), _stream())
    return idx


def cut_compact(raw, normals, zs, u, n_min, cap):
    """dataset.py:761-775 + 1176-1180 for a batch in one launch (pzn_cut_compact_f32): raw [B,M,3] f32, K candidate planes per
    sample (normals [B,K,3], zs [B,K], float64), start fractions u [B,2] float64
    -> (pieces [2B,cap,3]: up pieces then down pieces, counts [2B] int64, start [2B] int64, plane [B,4] float64, ok [B] bool)"""
    raw = _f32(raw, "raw")
    normals, zs, u = (_req(t, torch.float64, n_) for t, n_ in ((normals, "normals"), (zs, "zs"), (u, "u")))
    B, M, _ = raw.shape
    K = normals.shape[1]
    dev = raw.device
    pieces = torch.empty((2 * B, int(cap), 3), dtype=torch.float32, device=dev)
    counts = torch.empty((2 * B,), dtype=torch.int64, device=dev)
    start = torch.empty((2 * B,), dtype=torch.int64, device=dev)
    plane = torch.empty((B, 4), dtype=torch.float64, device=dev)
    ok = torch.empty((B,), dtype=torch.uint8, device=dev)
    with _on(dev):
        _call("pzn_cut_compact_f32", _p(raw), _p(normals), _p(zs), _p(u), B, M, K, int(n_min), int(cap), _p(pieces), _p(counts),
              _p(start), _p(plane), _p(ok), _stream())
    return pieces, counts, start, plane, ok.to(torch.bool)


def pick_mask(idx, N):
    """0/1 float masks [R,N] with ones at idx [R,k] (dataset.py:1363-1366), one launch."""
    idx = _i64(idx, "idx")
    R, k = idx.shape
    mask = torch.empty((R, int(N)), dtype=torch.float32, device=idx.device)
    with _on(idx.device):
        _call("pzn_pick_mask_f32", _p(idx), R, k, int(N), _p(mask), _stream())
    return mask


class _Avg4(torch.autograd.Function):
    """(((a + b) + c) + d) / 4 (model5_b.py:468-469) in one launch."""

    @staticmethod
    def forward(ctx, a, b, c, d):
        a, b, c, d = (_f32(t, "map") for t in (a, b, c, d))
        out = torch.empty_like(a)
        with _on(a.device):
            _call("pzn_avg4_f32", _p(a), _p(b), _p(c), _p(d), a.numel(), _p(out), _stream())
        return out

    @staticmethod
    def backward(ctx, dout):
        g = dout / 4
        return g, g, g, g


def avg4(a, b, c, d):
    return _Avg4.apply(a, b, c, d)


def colmean_argmax(a):
    """a[B,R,C] on the GPU -> (a.mean(dim=1) [B,C], index of its largest entry [B] int64); no gradient (it feeds an index)."""
    a = _f32(a.detach(), "a")
    B, R, C = a.shape
    mean = torch.empty((B, C), dtype=torch.float32, device=a.device)
    arg = torch.empty((B,), dtype=torch.int64, device=a.device)
    ws = torch.empty((_lib.load().pzn_colmean_workspace_bytes(B, C) + 3) // 4, dtype=torch.float32, device=a.device)
    with _on(a.device):
        _call("pzn_colmean_argmax_f32", _p(a), B, R, C, _p(mean), _p(arg), _p(ws), _stream())
    return mean, arg


def se3_exp(x):
    """[B,6] on the GPU -> [B,4,4]"""
    return _Se3Exp.apply(x)


def max_over_points(x):
    """[B, L, C] -> [B, C]"""
    if x.shape[-1] % 4 != 0:
        raise _lib.PznError(f"max_over_points: channel count must be a multiple of 4, got {x.shape[-1]}")
    return _MaxOverPoints.apply(x)


class _Group(torch.autograd.Function):
    """pointnet_util.py:123-132: cat(xyz[idx] - new_xyz, feat[idx]) in one kernel.  idx=None: the K = 32 neighbour
    search of pointnet_util.py:118-119 runs in the same launch (pzn_knn_group_f32) and idx becomes an output."""

    @staticmethod
    def forward(ctx, xyz, feat, new_xyz, idx, want_grouped_xyz):
        xyz, new_xyz = _f32(xyz, "xyz"), _f32(new_xyz, "new_xyz")
        B, N, _ = xyz.shape
        S = new_xyz.shape[1]
        fuse_knn = idx is None
        if fuse_knn:
            K = 32
            idx = torch.empty((B, S, K), dtype=torch.int64, device=xyz.device)
        else:
            idx = _i64(idx, "idx")
            K = idx.shape[2]
        D = 0 if feat is None else feat.shape[-1]
        feat_c = None if feat is None else _f32(feat, "points")
        out = torch.empty((B, S, K, 3 + D), dtype=torch.float32, device=xyz.device)
        gx = torch.empty((B, S, K, 3), dtype=torch.float32, device=xyz.device) if want_grouped_xyz else None
        with _on(xyz.device):
            if fuse_knn:
                _call("pzn_knn_group_f32", _p(xyz), _p(feat_c), _p(new_xyz), B, N, S, D, _p(idx), _p(out), _p(gx), _stream())
            else:
                _call("pzn_group_fwd_f32", _p(xyz), _p(feat_c), _p(new_xyz), _p(idx), B, N, S, K, D,
                      _p(out), _p(gx), _stream())
        ctx.save_for_backward(idx)
        ctx.dims = (B, N, S, K, D)
        ctx.has_feat = feat is not None
        ctx.mark_non_differentiable(idx)
        if gx is not None:
            ctx.mark_non_differentiable(gx)
        return out, gx, idx

    @staticmethod
    def backward(ctx, grad_out, _grad_gx, _grad_idx):
        (idx,) = ctx.saved_tensors
        B, N, S, K, D = ctx.dims
        need_xyz, need_feat, need_new = ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.needs_input_grad[2]
        need_feat = need_feat and ctx.has_feat
        if not (need_xyz or need_feat or need_new):
            return None, None, None, None, None
        grad_out = _f32(grad_out, "grad_out")
        dev = grad_out.device
        gxyz = torch.zeros((B, N, 3), dtype=torch.float32, device=dev) if need_xyz else None
        gfeat = torch.zeros((B, N, D), dtype=torch.float32, device=dev) if need_feat else None
        gnew = torch.empty((B, S, 3), dtype=torch.float32, device=dev) if need_new else None
        with _on(dev):
            _call("pzn_group_bwd_f32", _p(grad_out), _p(idx), B, N, S, K, D, _p(gxyz), _p(gfeat), _p(gnew),
                      _stream())
        return gxyz, gfeat, gnew, None, None


def group(xyz, feat, new_xyz, idx, want_grouped_xyz=False):
    out, gx, _ = _Group.apply(xyz, feat, new_xyz, idx, want_grouped_xyz)
    return (out, gx) if want_grouped_xyz else out


def knn_group_supported(xyz, feat, nsample):
    """Shapes pzn_knn_group_f32 takes (include/pzn.h): K = 32, 64 <= N <= 8192, a feature table with D % 4 == 0."""
    return nsample == 32 and feat is not None and feat.shape[-1] > 0 and feat.shape[-1] % 4 == 0 and \
        64 <= xyz.shape[1] <= 8192


def knn_group(xyz, feat, new_xyz, want_grouped_xyz=False):
    """Neighbour search (K = 32) + grouping in one launch -> (new_points[B,S,32,3+D], grouped_xyz or None, idx)."""
    return _Group.apply(xyz, feat, new_xyz, None, want_grouped_xyz)


# --------------------------------------------------------------------------- EMD

def _emd_ws(B, n, m, device):
    nbytes = _lib.load().pzn_emd_workspace_bytes(B, n, m)
    return torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=device)


def _emd_shapes(xyz1, xyz2):
    if xyz1.dim() != 3 or xyz2.dim() != 3 or xyz1.shape[2] != 3 or xyz2.shape[2] != 3 or xyz1.shape[0] != xyz2.shape[0]:
        # emd_kernel.cu:178-180 CHECK_EQ
        raise _lib.PznError(f"EMD expects xyz1[B,n,3], xyz2[B,m,3]; got {tuple(xyz1.shape)}, {tuple(xyz2.shape)}")
    return xyz1.shape[0], xyz1.shape[1], xyz2.shape[1]


def _emd_suffix(*ts):
    """emd_kernel.cu:187,273,391 dispatch on the floating type: fp32 (csrc/emd.hip) or double (csrc/emd64.hip)."""
    if all(t.dtype == torch.float64 for t in ts):
        return torch.float64, "f64"
    return torch.float32, "f32"


def emd_approxmatch(xyz1, xyz2):
    """emd_cuda.approxmatch_forward (emd.cpp:24, emd_kernel.cu:171-193) -> match[B,m,n]"""
    dt, suf = _emd_suffix(xyz1, xyz2)
    xyz1, xyz2 = _req(xyz1, dt, "xyz1"), _req(xyz2, dt, "xyz2")
    B, n, m = _emd_shapes(xyz1, xyz2)
    match = torch.empty((B, m, n), dtype=dt, device=xyz1.device)
    if suf == "f64":
        ws = torch.empty(_lib.load().pzn_emd_workspace_bytes_f64(B, n, m) // 8, dtype=torch.float64, device=xyz1.device)
    else:
        ws = _emd_ws(B, n, m, xyz1.device)
    with _on(xyz1.device):
        _call("pzn_emd_approxmatch_" + suf, _p(xyz1), _p(xyz2), B, n, m, _p(match), _p(ws), _stream())
    return match


def emd_matchcost(xyz1, xyz2, match):
    """emd_cuda.matchcost_forward (emd.cpp:25, emd_kernel.cu:257-279) -> cost[B]"""
    dt, suf = _emd_suffix(xyz1, xyz2, match)
    xyz1, xyz2, match = _req(xyz1, dt, "xyz1"), _req(xyz2, dt, "xyz2"), _req(match, dt, "match")
    B, n, m = _emd_shapes(xyz1, xyz2)
    cost = torch.empty((B,), dtype=dt, device=xyz1.device)
    with _on(xyz1.device):
        _call("pzn_emd_matchcost_" + suf, _p(xyz1), _p(xyz2), _p(match), B, n, m, _p(cost), _stream())
    return cost


def emd_matchcost_grad(grad_cost, xyz1, xyz2, match):
    """emd_cuda.matchcost_backward (emd.cpp:26, emd_kernel.cu:373-398) -> [grad1, grad2]"""
    dt, suf = _emd_suffix(grad_cost, xyz1, xyz2, match)
    grad_cost = _req(grad_cost, dt, "grad_cost")
    xyz1, xyz2, match = _req(xyz1, dt, "xyz1"), _req(xyz2, dt, "xyz2"), _req(match, dt, "match")
    B, n, m = _emd_shapes(xyz1, xyz2)
    g1 = torch.empty((B, n, 3), dtype=dt, device=xyz1.device)
    g2 = torch.empty((B, m, 3), dtype=dt, device=xyz1.device)
    with _on(xyz1.device):
        _call("pzn_emd_matchcost_grad_" + suf, _p(grad_cost), _p(xyz1), _p(xyz2), _p(match), B, n, m,
              _p(g1), _p(g2), _stream())
    return [g1, g2]


EMD_WALK_STATS = None      # bench.py sets this to a list: every fused call appends (walk counters int64 view or None, B, n, m)


class _EmdFused(torch.autograd.Function):
    """EarthMoverDistanceFunction (PyTorchEMD/emd.py:5-21) without the match tensor."""

    @staticmethod
    def forward(ctx, xyz1, xyz2):
        xyz1, xyz2 = _f32(xyz1, "xyz1"), _f32(xyz2, "xyz2")
        B, n, m = _emd_shapes(xyz1, xyz2)
        dev = xyz1.device
        cost = torch.empty((B,), dtype=torch.float32, device=dev)
        g1 = torch.empty((B, n, 3), dtype=torch.float32, device=dev)
        g2 = torch.empty((B, m, 3), dtype=torch.float32, device=dev)
        ws = _emd_ws(B, n, m, dev)
        with _on(dev):
            _call("pzn_emd_fused_f32", _p(xyz1), _p(xyz2), B, n, m, _p(cost), _p(g1), _p(g2), _p(ws), _stream())
        if EMD_WALK_STATS is not None:       # measurement only (bench.py): lengths of the active lists the passes walked
            off = _lib.load().pzn_emd_walk_counter_offset(B, n, m)
            ctr = None if off == 2 ** 64 - 1 else ws.view(torch.int64)[off // 8: off // 8 + _lib.load().pzn_emd_walk_counter_count()]
            EMD_WALK_STATS.append((ctr, B, n, m))
        ctx.save_for_backward(g1, g2)
        return cost

    @staticmethod
    def backward(ctx, grad_cost):
        g1, g2 = ctx.saved_tensors
        gc = grad_cost.contiguous().view(-1, 1, 1)
        return g1 * gc, g2 * gc


def emd_fused(xyz1, xyz2):
    return _EmdFused.apply(xyz1, xyz2)


EMD_SMALL_MAX = 256      # csrc/emd.hip: n, m up to here run as one workgroup per pair


class _EmdFusedSmallMulti(torch.autograd.Function):
    """Several small EarthMoverDistanceFunction calls (PyTorchEMD/emd.py:5-21; n, m <= 256 each) in one launch:
    apply(a1, b1, a2, b2, ...) -> (cost1, cost2, ...), each what _EmdFused gives for its pair, bit for bit."""

    @staticmethod
    def forward(ctx, *clouds):
        k = len(clouds) // 2
        a = [_f32(t, "xyz1") for t in clouds[0::2]]
        b = [_f32(t, "xyz2") for t in clouds[1::2]]
        shapes = [_emd_shapes(x, y) for x, y in zip(a, b)]
        dev = a[0].device
        cost = [torch.empty((B,), dtype=torch.float32, device=dev) for B, _, _ in shapes]
        g1 = [torch.empty((B, n, 3), dtype=torch.float32, device=dev) for B, n, _ in shapes]
        g2 = [torch.empty((B, m, 3), dtype=torch.float32, device=dev) for B, _, m in shapes]
        ints = lambda j: (ctypes.c_int * k)(*[int(sh[j]) for sh in shapes])
        with _on(dev):
            _call("pzn_emd_fused_small_multi_f32", k, _ptrs(a), _ptrs(b), ints(0), ints(1), ints(2), _ptrs(cost), _ptrs(g1),
                  _ptrs(g2), _stream())
        ctx.save_for_backward(*g1, *g2)
        ctx.k = k
        return tuple(cost)

    @staticmethod
    def backward(ctx, *grad_cost):
        k = ctx.k
        g1, g2 = ctx.saved_tensors[:k], ctx.saved_tensors[k:]
        out = []
        for i in range(k):
            if grad_cost[i] is None:
                out += [None, None]
            else:
                gc = grad_cost[i].contiguous().view(-1, 1, 1)
                out += [g1[i] * gc, g2[i] * gc]
        return tuple(out)


def emd_fused_small_multi(pairs):
    """[(xyz1, xyz2), ...] (1..4 pairs, every cloud of at most 256 points) -> [cost[B], ...] from one launch."""
    flat = [t for pr in pairs for t in pr]
    return list(_EmdFusedSmallMulti.apply(*flat))


# --------------------------------------------------------------------------- chamfer

class _Chamfer(torch.autograd.Function):
    """TouchedRegraster.chamfer_loss (model5_b.py:1495-1505) without P[B,n,m]."""

    @staticmethod
    def forward(ctx, a, b):
        a, b = _f32(a, "a"), _f32(b, "b")
        if a.dim() != 3 or b.dim() != 3 or a.shape[2] != 3 or b.shape[2] != 3 or a.shape[0] != b.shape[0]:
            raise _lib.PznError(f"chamfer expects a[B,n,3], b[B,m,3]; got {tuple(a.shape)}, {tuple(b.shape)}")
        B, n, m = a.shape[0], a.shape[1], b.shape[1]
        dev = a.device
        moa = torch.empty((B, m), dtype=torch.float32, device=dev)
        mob = torch.empty((B, n), dtype=torch.float32, device=dev)
        aoa = torch.empty((B, m), dtype=torch.int32, device=dev)
        aob = torch.empty((B, n), dtype=torch.int32, device=dev)
        ws = torch.empty((_lib.load().pzn_chamfer_workspace_bytes(B, n, m) + 3) // 4, dtype=torch.float32, device=dev)
        with _on(dev):
            _call("pzn_chamfer_fwd_f32", _p(a), _p(b), B, n, m, _p(moa), _p(aoa), _p(mob), _p(aob), _p(ws), _stream())
        ctx.save_for_backward(a, b, aoa, aob)
        ctx.mark_non_differentiable(aoa, aob)
        ctx.set_materialize_grads(False)      # (the arg-min outputs carry no gradient: no zero-filled stand-ins for them)
        return moa, mob, aoa, aob

    @staticmethod
    def backward(ctx, g_moa, g_mob, _ga, _gb):
        a, b, aoa, aob = ctx.saved_tensors
        B, n, m = a.shape[0], a.shape[1], b.shape[1]
        g_moa = None if g_moa is None else _f32(g_moa, "g_over_a")
        g_mob = None if g_mob is None else _f32(g_mob, "g_over_b")
        if g_moa is None and g_mob is None:
            return None, None
        both = torch.zeros((B * (n + m) * 3,), dtype=torch.float32, device=a.device)      # one fill for the two atomic targets
        ga, gb = both[:B * n * 3].view(B, n, 3), both[B * n * 3:].view(B, m, 3)
        with _on(a.device):
            _call("pzn_chamfer_bwd_f32", _p(a), _p(b), B, n, m, _p(g_moa), _p(aoa), _p(g_mob), _p(aob),
                  _p(ga), _p(gb), _stream())
        return ga, gb


def chamfer(a, b):
    """-> (min over a per b-point [B,m], min over b per a-point [B,n])  == torch.min(P,1)[0], torch.min(P,2)[0]."""
    moa, mob, _, _ = _Chamfer.apply(a, b)
    return moa, mob


# --------------------------------------------------------------------------- dense (MFMA fp32)

# Gradient sinks: parameter storage address -> its (pre-zeroed) gradient tensor.  When a parameter has a
# sink, the weight-gradient kernels ADD straight into it and the autograd Function returns None for that
# input: no zero-fill launch, no temporary dW, no separate AccumulateGrad "grad += dW" kernel
# (~280 small launches per training step).  distributed.FlatGradAllReduce registers the views of its flat
# bucket; without sinks the Functions return ordinary gradient tensors.
_GRAD_SINKS = {}      # parameter address -> (weak reference to the parameter, its gradient buffer)


def register_grad_sinks(params):
    for p_ in params:
        if p_.grad is not None:
            _GRAD_SINKS[p_.data_ptr()] = (weakref.ref(p_), p_.grad)


def clear_grad_sinks():
    _GRAD_SINKS.clear()


def _sink(t, needed):
    """The registered gradient buffer of parameter tensor `t`, or None.  An entry only counts while its parameter is
    alive, still sits at that address and has this shape (a freed parameter's address is re-used by the allocator:
    a stale entry must never capture the gradient of an unrelated tensor)."""
    if not needed or t is None:
        return None
    ent = _GRAD_SINKS.get(t.data_ptr())
    if ent is None:
        return None
    owner, grad = ent
    o = owner()
    if o is None or o.data_ptr() != t.data_ptr() or o.shape != t.shape or o.grad is not grad:
        _GRAD_SINKS.pop(t.data_ptr(), None)
        return None
    return grad

class _Linear(torch.autograd.Function):
    """nn.Linear (+ReLU) on the fp32 matrix-core engine: y = act(x W^T + b)."""

    @staticmethod
    def forward(ctx, x, weight, bias, relu):
        x = _f32(x, "x")
        weight = _f32(weight, "weight")
        bias_c = None if bias is None else _f32(bias, "bias")
        Kin = x.shape[-1]
        Nout = weight.shape[0]
        if weight.shape[1] != Kin:
            raise _lib.PznError(f"linear: x[..., {Kin}] vs weight{tuple(weight.shape)}")
        x2 = x.reshape(-1, Kin)
        M = x2.shape[0]
        y = torch.empty((M, Nout), dtype=torch.float32, device=x.device)
        with _on(x.device):
            _call("pzn_linear_fwd_f32", _p(x2), _p(weight), _p(bias_c), M, Kin, Nout, int(bool(relu)), _p(y), _stream(),
                  flops=2 * M * Kin * Nout)
        ctx.save_for_backward(x2, weight, y if relu else None)
        ctx.has_bias = bias is not None
        ctx.bias_ref = bias_c
        ctx.in_shape = x.shape
        return y.reshape(*x.shape[:-1], Nout)

    @staticmethod
    def backward(ctx, dy):
        x2, weight, y = ctx.saved_tensors
        M, Kin = x2.shape
        Nout = weight.shape[0]
        dy = _f32(dy, "dy").reshape(M, Nout)
        dev = dy.device
        dx = dW = db = None
        with _on(dev):
            if ctx.needs_input_grad[0]:
                dx = torch.empty((M, Kin), dtype=torch.float32, device=dev)
                _call("pzn_linear_dgrad_f32", _p(dy), _p(y), _p(weight), M, Kin, Nout, None, _p(dx), _stream(),
                      flops=2 * M * Kin * Nout)
                dx = dx.reshape(ctx.in_shape)
            if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
                sw = _sink(weight, ctx.needs_input_grad[1])
                sb = _sink(ctx.bias_ref, ctx.has_bias and ctx.needs_input_grad[2])
                if sw is not None and (sb is not None or not ctx.has_bias):
                    _call("pzn_linear_wgrad_f32", _p(dy), _p(y), _p(x2), M, Kin, Nout, _p(sw), _p(sb), 1, _stream(),
                          flops=2 * M * Kin * Nout)
                else:
                    dW = torch.empty((Nout, Kin), dtype=torch.float32, device=dev)
                    db = torch.empty((Nout,), dtype=torch.float32, device=dev) if ctx.has_bias else None
                    _call("pzn_linear_wgrad_f32", _p(dy), _p(y), _p(x2), M, Kin, Nout, _p(dW), _p(db), 0, _stream(),
                          flops=2 * M * Kin * Nout)
        return dx, dW, db, None


def linear(x, weight, bias=None, relu=False):
    return _Linear.apply(x, weight, bias, relu)


class _CatGlobalLinearRelu(torch.autograd.Function):
    """relu(Linear(cat([g.repeat(1, N, 1), x], -1))) without the concatenation (model5_b.py:745-752, first layer of the
    boundary heads): the product splits into x W[:, Cg:]^T per point and g W[:, :Cg]^T + b per CLOUD; the per-cloud part
    is a [B, Cout] bias added (with the ReLU) by pzn_cloud_bias_relu_f32.  Backward: the gated gradient goes through the
    point part's two products (the kernels take the ReLU mask), its per-cloud column sums (pzn_cloud_gated_colsum_f32)
    through the global part's.  x[B,N,C], g[B,Cg] (or [B,1,Cg]), weight[Cout, Cg + C], bias[Cout]."""

    @staticmethod
    def forward(ctx, x, g, weight, bias):
        x, g, weight, bias = _f32(x, "x"), _f32(g, "g"), _f32(weight, "weight"), _f32(bias, "bias")
        B, N, C = x.shape
        g2 = g.reshape(B, -1)
        Cg = g2.shape[1]
        Co = weight.shape[0]
        if weight.shape[1] != Cg + C:
            raise _lib.PznError(f"cat_global_linear_relu: weight{tuple(weight.shape)} vs {Cg} + {C} input columns")
        dev = x.device
        w_g, w_x = weight[:, :Cg].contiguous(), weight[:, Cg:].contiguous()
        x2 = x.reshape(B * N, C)
        y = torch.empty((B * N, Co), dtype=torch.float32, device=dev)
        cb = torch.empty((B, Co), dtype=torch.float32, device=dev)
        with _on(dev):
            st = _stream()
            _call("pzn_linear_fwd_f32", _p(g2), _p(w_g), _p(bias), B, Cg, Co, 0, _p(cb), st, flops=2 * B * Cg * Co)
            _call("pzn_linear_fwd_f32", _p(x2), _p(w_x), None, B * N, C, Co, 0, _p(y), st, flops=2 * B * N * C * Co)
            _call("pzn_cloud_bias_relu_f32", _p(y), _p(cb), B, N, Co, st)
        ctx.save_for_backward(x2, g2, w_g, w_x, y)
        ctx.dims = (B, N, C, Cg, Co)
        ctx.g_shape = g.shape
        return y.view(B, N, Co)

    @staticmethod
    def backward(ctx, dy):
        x2, g2, w_g, w_x, y = ctx.saved_tensors
        B, N, C, Cg, Co = ctx.dims
        dev = y.device
        dy = _f32(dy, "dy").reshape(B * N, Co)
        mk = lambda *sh: torch.empty(sh, dtype=torch.float32, device=dev)
        dx = dg = dW = db = None
        with _on(dev):
            st = _stream()
            dcb = mk(B, Co)
            _call("pzn_cloud_gated_colsum_f32", _p(dy), _p(y), B, N, Co, _p(dcb), st)
            if ctx.needs_input_grad[0]:
                dx = mk(B * N, C)
                _call("pzn_linear_dgrad_f32", _p(dy), _p(y), _p(w_x), B * N, C, Co, None, _p(dx), st, flops=2 * B * N * C * Co)
                dx = dx.view(B, N, C)
            if ctx.needs_input_grad[1]:
                dg = mk(B, Cg)
                _call("pzn_linear_dgrad_f32", _p(dcb), None, _p(w_g), B, Cg, Co, None, _p(dg), st, flops=2 * B * Cg * Co)
                dg = dg.view(ctx.g_shape)
            if ctx.needs_input_grad[2] or ctx.needs_input_grad[3]:
                dWx, dWg, db = mk(Co, C), mk(Co, Cg), mk(Co)
                _call("pzn_linear_wgrad_f32", _p(dy), _p(y), _p(x2), B * N, C, Co, _p(dWx), _p(db), 0, st,
                      flops=2 * B * N * C * Co)                                   # db = column sums of the gated gradient
                _call("pzn_linear_wgrad_f32", _p(dcb), None, _p(g2), B, Cg, Co, _p(dWg), None, 0, st, flops=2 * B * Cg * Co)
                dW = torch.cat([dWg, dWx], dim=1)
        return dx, dg, dW, db


def cat_global_linear_relu(x, g, weight, bias):
    return _CatGlobalLinearRelu.apply(x, g, weight, bias)


class _PointMlp3(torch.autograd.Function):
    """Three Linear layers with ReLU after the first two on [B, N, 64] rows, one launch each way (csrc/pointmlp.hip):
    nn.Sequential(Linear, ReLU, Linear, ReLU, Linear) of the boundary heads (model5_b.py:571-592, 738-739, 751-754).
    g is None: 64 -> 64 -> C2 -> C3 as it stands.  g[B, Cg] (or [B, 1, Cg]): the first layer acts on
    cat([g.repeat(1, N, 1), x], -1) (model5_b.py:745-749) — its global half becomes a per-cloud bias g W1[:, :Cg]^T + b1,
    whose gradient (the per-cloud column sums of the first gated gradient) the backward pass returns with the rest."""

    @staticmethod
    def forward(ctx, x, g, w1, b1, w2, b2, w3, b3):
        x = _f32(x, "x")
        w1, b1, w2, b2, w3, b3 = (_f32(t, n) for t, n in ((w1, "w1"), (b1, "b1"), (w2, "w2"), (b2, "b2"), (w3, "w3"), (b3, "b3")))
        B, N, C = x.shape
        M, C2, C3 = B * N, w2.shape[0], w3.shape[0]
        dev = x.device
        mk = lambda *sh: torch.empty(sh, dtype=torch.float32, device=dev)
        x2 = x.reshape(M, C)
        h1, h2, y = mk(M, 64), mk(M, C2), mk(M, C3)
        Cg = 0
        g2 = w_g = None
        with _on(dev):
            st = _stream()
            if g is not None:
                g2 = _f32(g, "g").reshape(B, -1)
                Cg = g2.shape[1]
                if w1.shape[1] != Cg + C:
                    raise _lib.PznError(f"point_mlp3: w1{tuple(w1.shape)} vs {Cg} + {C} input columns")
                bias1 = mk(B, 64)      # g W1[:, :Cg]^T + b1, the columns read in place (rows of Cg + C floats)
                _call("pzn_linear_slice_fwd_f32", _p(g2), _p(w1), Cg + C, _p(b1), B, Cg, 64, 0, _p(bias1), st, flops=2 * B * Cg * 64)
            else:
                bias1 = b1
            _call("pzn_point_mlp3_fwd_f32", _p(x2), M, N, _p(w1) + 4 * Cg, w1.shape[1], _p(bias1), 1 if g is not None else 0,
                  _p(w2), _p(b2), _p(w3), _p(b3), C2, C3, _p(h1), _p(h2), _p(y), st,
                  flops=2 * M * (64 * 64 + 64 * C2 + C2 * C3))
        ctx.save_for_backward(x2, g2, w_g, w1, w2, w3, h1, h2)
        ctx.bias_refs = (b1, b2, b3)      # (for their registered gradient buffers; parameters, alive anyway)
        ctx.dims = (B, N, C2, C3, Cg)
        ctx.g_shape = None if g is None else g.shape
        return y.view(B, N, C3)

    @staticmethod
    def backward(ctx, dy):
        x2, g2, w_g, w1, w2, w3, h1, h2 = ctx.saved_tensors
        B, N, C2, C3, Cg = ctx.dims
        M = B * N
        dev = x2.device
        dy = _f32(dy, "dy").reshape(M, C3)
        mk = lambda *sh: torch.empty(sh, dtype=torch.float32, device=dev)
        per_cloud = 1 if g2 is not None else 0
        nws = _lib.load().pzn_point_mlp3_bwd_workspace_bytes(M, N, per_cloud, C2, C3)
        ws = torch.empty((max(16, nws),), dtype=torch.uint8, device=dev)
        # the six parameter gradients straight into their registered buffers (ops.register_grad_sinks) when all six have one:
        # no six AccumulateGrad adds per head behind the node
        b1p, b2p, b3p = ctx.bias_refs
        sinks = [_sink(t, ctx.needs_input_grad[2 + i]) for i, t in enumerate((w1, b1p, w2, b2p, w3, b3p))]
        direct = all(s_ is not None for s_ in sinks)
        dx = mk(M, 64)
        if direct:
            dW1, db1_out, dW2, db2, dW3, db3 = sinks
        else:
            dW1, dW2, dW3, db2, db3, db1_out = mk(64, Cg + 64), mk(C2, 64), mk(C3, C2), mk(C2), mk(C3), mk(64)
        db1 = mk(B, 64) if per_cloud else db1_out
        dg = None
        with _on(dev):
            st = _stream()
            _call("pzn_point_mlp3_bwd_f32", _p(dy), _p(x2), _p(h1), _p(h2), M, N, _p(w1) + 4 * Cg, Cg + 64, per_cloud, _p(w2),
                  _p(w3), C2, C3, _p(dx), _p(dW1) + 4 * Cg, _p(db1), _p(dW2), _p(db2), _p(dW3), _p(db3), int(direct), _p(ws), st,
                  flops=4 * M * (64 * 64 + 64 * C2 + C2 * C3))
            if per_cloud:       # the global half of the first layer: [B, Cg] products on the per-cloud bias gradient
                dcb = db1
                if direct:      # += into the first Cg columns of the registered dW1 and into db1
                    _call("pzn_linear_slice_wgrad_f32", _p(dcb), _p(g2), B, Cg, 64, _p(dW1), Cg + 64, _p(db1_out), st,
                          flops=2 * B * Cg * 64)
                else:
                    dWg = mk(64, Cg)
                    _call("pzn_linear_wgrad_f32", _p(dcb), None, _p(g2), B, Cg, 64, _p(dWg), _p(db1_out), 0, st, flops=2 * B * Cg * 64)
                    dW1[:, :Cg] = dWg
                if ctx.needs_input_grad[1]:
                    dg = mk(B, Cg)
                    _call("pzn_linear_slice_dgrad_f32", _p(dcb), _p(w1), Cg + 64, B, Cg, 64, None, _p(dg), st, flops=2 * B * Cg * 64)
                    dg = dg.view(ctx.g_shape)
        if direct:
            return dx.view(B, N, 64), dg, None, None, None, None, None, None
        return dx.view(B, N, 64), dg, dW1, db1_out, dW2, db2, dW3, db3


def point_mlp3_available(C0, C1, C2, C3):
    return bool(_lib.load().pzn_point_mlp3_supported(int(C0), int(C1), int(C2), int(C3)))


def point_mlp3(x, w1, b1, w2, b2, w3, b3, g=None):
    """(relu(relu([g |] x) W1^T + b1) W2^T + b2) W3^T + b3 on [B, N, 64] rows; see _PointMlp3."""
    return _PointMlp3.apply(x, g, w1, b1, w2, b2, w3, b3)


class _SharedMlpMax(torch.autograd.Function):
    """relu(x W1^T + b1) -> relu(. W2^T + b2) -> max over the K=32 axis (model5_b.py:452-454, 459-461)."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2):
        x = _f32(x, "x")
        w1, b1, w2, b2 = _f32(w1, "w1"), _f32(b1, "b1"), _f32(w2, "w2"), _f32(b2, "b2")
        Bq, S, K, C0 = x.shape
        if K != 32:
            raise _lib.PznError(f"shared_mlp_max: the fused epilogue pools over K=32 neighbours, got K={K}")
        C1, C2 = w1.shape[0], w2.shape[0]
        R = Bq * S
        dev = x.device
        h = torch.empty((R * 32, C1), dtype=torch.float32, device=dev)
        out = torch.empty((R, C2), dtype=torch.float32, device=dev)
        arg = torch.empty((R, C2), dtype=torch.int32, device=dev)
        with _on(dev):
            _call("pzn_sharedmlp_max_fwd_f32", _p(x), _p(w1), _p(b1), _p(w2), _p(b2), R, C0, C1, C2,
                  _p(h), _p(out), _p(arg), _stream())
        ctx.save_for_backward(x, w1, w2, h, out, arg)
        ctx.dims = (Bq, S, R, C0, C1, C2)
        return out.reshape(Bq, S, C2)

    @staticmethod
    def backward(ctx, dout):
        x, w1, w2, h, out, arg = ctx.saved_tensors
        Bq, S, R, C0, C1, C2 = ctx.dims
        dout = _f32(dout, "dout").reshape(R, C2)
        dev = dout.device
        dh = torch.empty((R * 32, C1), dtype=torch.float32, device=dev)
        dx = torch.empty((R * 32, C0), dtype=torch.float32, device=dev) if ctx.needs_input_grad[0] else None
        dW1 = torch.empty_like(w1)
        db1 = torch.empty((C1,), dtype=torch.float32, device=dev)
        dW2 = torch.empty_like(w2)
        db2 = torch.empty((C2,), dtype=torch.float32, device=dev)
        with _on(dev):
            _call("pzn_sharedmlp_max_bwd_f32", _p(x), _p(w1), _p(w2), _p(h), _p(out), _p(arg), _p(dout),
                  R, C0, C1, C2, _p(dh), _p(dx), _p(dW1), _p(db1), _p(dW2), _p(db2), 0, _stream())
        return (None if dx is None else dx.reshape(Bq, S, 32, C0)), dW1, db1, dW2, db2


def shared_mlp_max(x, w1, b1, w2, b2):
    return _SharedMlpMax.apply(x, w1, b1, w2, b2)


class _Attention(torch.autograd.Function):
    """scaled_dot_production (model5_b.py:67-75) -> (values, attention)."""

    @staticmethod
    def forward(ctx, q, k, v):
        q, k, v = _f32(q, "q"), _f32(k, "k"), _f32(v, "v")
        B, L, dk = q.shape
        dv = v.shape[-1]
        dev = q.device
        attn = torch.empty((B, L, L), dtype=torch.float32, device=dev)
        out = torch.empty((B, L, dv), dtype=torch.float32, device=dev)
        with _on(dev):
            _call("pzn_attn_fwd_f32", _p(q), _p(k), _p(v), B, L, dk, dv, _p(attn), _p(out), _stream(),
                  flops=2 * B * L * L * (dk + dv))
        ctx.save_for_backward(q, k, v, attn)
        ctx.set_materialize_grads(False)
        return out, attn

    @staticmethod
    def backward(ctx, d_out, d_attn):
        q, k, v, attn = ctx.saved_tensors
        B, L, dk = q.shape
        dv = v.shape[-1]
        dev = q.device
        d_out = torch.zeros((B, L, dv), dtype=torch.float32, device=dev) if d_out is None else _f32(d_out, "d_out")
        d_attn = None if d_attn is None else _f32(d_attn, "d_attn")
        dq, dk_, dv_ = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        ws = torch.empty((B, L, L), dtype=torch.float32, device=dev)
        with _on(dev):
            _call("pzn_attn_bwd_f32", _p(q), _p(k), _p(v), _p(attn), _p(d_out), _p(d_attn), B, L, dk, dv,
                  _p(dq), _p(dk_), _p(dv_), _p(ws), _stream(), flops=2 * B * L * L * (2 * dk + 2 * dv))
        return dq, dk_, dv_


def attention(q, k, v):
    return _Attention.apply(q, k, v)


class _AttentionBlock(torch.autograd.Function):
    """layerAttention as one unit (model5_b.py:83-101) -> (r, attention): the residual / offset lines ride in GEMM
    epilogues forward, and the five contributions to dx are summed by accumulate epilogues backward."""

    @staticmethod
    def forward(ctx, x, wq, bq, wk, bk, wv, bv, wo, bo):
        x = _f32(x, "x")
        ws_ = [_f32(t, "param") for t in (wq, bq, wk, bk, wv, bv, wo, bo)]
        wq, bq, wk, bk, wv, bv, wo, bo = ws_
        B, L, E = x.shape
        dk = wq.shape[0]
        dev = x.device
        M = B * L
        mk = lambda *sh: torch.empty(sh, dtype=torch.float32, device=dev)
        q, k, v = mk(M, dk), mk(M, dk), mk(M, E)
        attn, r, yo, out = mk(B, L, L), mk(M, E), mk(M, E), mk(M, E)
        with _on(dev):
            _call("pzn_attn_block_fwd_f32", _p(x), _p(wq), _p(bq), _p(wk), _p(bk), _p(wv), _p(bv), _p(wo), _p(bo),
                  B, L, E, dk, _p(q), _p(k), _p(v), _p(attn), _p(r), _p(yo), _p(out), _stream(),
                  flops=2 * M * E * (2 * dk + 2 * E) + 2 * B * L * L * (dk + E))
        ctx.save_for_backward(x, wq, wk, wv, wo, q, k, v, attn, r, yo)
        ctx.set_materialize_grads(False)       # an unused attention map must not cost a zero-filled [B,L,L] gradient
        ctx.dims = (B, L, E, dk)
        ctx.param_refs = (wq, bq, wk, bk, wv, bv, wo, bo)
        return out.view(B, L, E), attn

    @staticmethod
    def backward(ctx, dout, dattn):
        x, wq, wk, wv, wo, q, k, v, attn, r, yo = ctx.saved_tensors
        B, L, E, dk = ctx.dims
        dev = x.device
        M = B * L
        dout = torch.zeros((M, E), dtype=torch.float32, device=dev) if dout is None else _f32(dout, "dout").reshape(M, E)
        dattn = None if dattn is None else _f32(dattn, "dattn")
        sinks = [_sink(t, ctx.needs_input_grad[1 + i]) for i, t in enumerate(ctx.param_refs)]
        direct = all(s_ is not None for s_ in sinks)
        if direct:
            grads = sinks
        else:
            grads = [torch.empty_like(t) for t in ctx.param_refs]
        dx = torch.empty((M, E), dtype=torch.float32, device=dev)
        nbytes = _lib.load().pzn_attn_block_bwd_workspace_bytes(B, L, E, dk)
        ws = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=dev)
        gq, gbq, gk, gbk, gv, gbv, go, gbo = grads
        with _on(dev):
            _call("pzn_attn_block_bwd_f32", _p(x), _p(wq), _p(wk), _p(wv), _p(wo), _p(q), _p(k), _p(v), _p(attn), _p(r),
                  _p(yo), _p(dout), _p(dattn), B, L, E, dk, _p(ws), _p(dx), _p(gq), _p(gbq), _p(gk), _p(gbk), _p(gv),
                  _p(gbv), _p(go), _p(gbo), int(direct), _stream(),
                  flops=2 * (2 * M * E * (2 * dk + 2 * E)) + 2 * B * L * L * (2 * dk + 2 * E))
        if direct:
            return (dx.view(B, L, E),) + (None,) * 8
        return (dx.view(B, L, E),) + tuple(grads)


def attention_block_supported(x, dk):
    """Shapes the fused block takes (the weight-stationary kernel's domain): else compose it from linear + attention."""
    return x.is_cuda and x.dim() == 3 and x.shape[0] * x.shape[1] >= 4096 and x.shape[2] % 32 == 0 and dk % 32 == 0 \
        and x.shape[1] % 4 == 0 and _lib.load().pzn_gemm_get_precision() != 0


def attention_block(x, wq, bq, wk, bk, wv, bv, wo, bo):
    return _AttentionBlock.apply(x, wq, bq, wk, bk, wv, bv, wo, bo)


def _ptrs(ts):
    """Array of device pointers, one per problem, for the nprob entry points."""
    return (ctypes.c_void_p * len(ts))(*[None if t is None else t.data_ptr() for t in ts])


def attention_chain_fused_available():
    """The chained kernels run on the bf16 matrix pipe: three planes per operand (fp32 results, default) or, in the opt-in
    bf16 attention mode (pzn_attn_set_precision(1)), one plane (single bf16 MFMAs, fp32 accumulation and softmax) - not with
    the exact-fp32 engine selected."""
    return _lib.load().pzn_gemm_get_precision() != 0


def attention_tile_image_rows(img, F):
    """A register-order tile image the chained kernels pass to each other (u, dq), as rows [M, F] - for checks only:
    [16-row tile][feature tile][g][c][r], feature = 16 ft + 4 g + r (csrc/pzn_mfma16.h, store_tiles16)."""
    M = img.numel() // F
    return img.view(M // 16, F // 16, 4, 16, 4).permute(0, 3, 1, 2, 4).reshape(M, F)


def attention_chain_fused_supported(x, dk, w_out):
    """The chained-kernel path (csrc/attnfused.hip) takes the model's shape only: L = 256, E = 256, dk = 64."""
    return (x.is_cuda and x.dim() == 3 and w_out.shape[1] == 5 * x.shape[2]
            and bool(_lib.load().pzn_attn_fused_supported(x.shape[1], x.shape[2], dk))
            and attention_chain_fused_available())


class _AttnChainFused(torch.autograd.Function):
    """model5_b.py:462-475 for nprob = 1 or 2 encoders in the same launches: the four layerAttention blocks as chained
    matrix-core kernels (pzn_attn_fused_*: no q / k / v / score tensors in memory, one projection + one block launch
    per layer forward, two launches + the weight gradients per layer backward), the running mean of the four maps
    written by the block kernel, then — per encoder — the out projection over the five slices and the max over the
    points exactly as _AttnChainOut does them.
    inputs: nprob, flags (bit 0 clear: `out` is not materialised, an empty placeholder stands in the tuple; bit 1: the map as
    column sums per strip of 16 rows, [B,L/16,L], whose mean over dim 1 IS the mean attention's mean over its rows), then per
    problem x[B,L,E], 4 x (wq,bq,wk,bk,wv,bv,wo,bo), w_out, b_out
    -> per problem (out[B,L,Nout], attention[B,L,L], f_global[B,Nout])"""

    @staticmethod
    def forward(ctx, nprob, flags, *args):
        need_out, strips = bool(flags & 1), bool(flags & 2)
        per = 35
        xs = [_f32(args[per * i], "x") for i in range(nprob)]
        pss = [[_f32(t, "param") for t in args[per * i + 1: per * i + 35]] for i in range(nprob)]
        B, L, E = xs[0].shape
        dk = pss[0][0].shape[0]
        Nout = pss[0][32].shape[0]
        dev = xs[0].device
        M = B * L
        lib = _lib.load()
        wbytes, qkb, vb = lib.pzn_attn_fused_weight_bytes(), lib.pzn_attn_fused_qk_image_bytes(B), lib.pzn_attn_fused_v_image_bytes(B)
        mk = lambda *sh: torch.empty(sh, dtype=torch.float32, device=dev)
        raw = lambda n: torch.empty(n, dtype=torch.uint8, device=dev)
        R = range(nprob)
        with _on(dev):
            st = _stream()
            W = [[raw(wbytes) for _ in range(4)] for _ in R]
            for p in R:      # the four layers' weight planes in one launch per encoder
                _call("pzn_attn_fused_prep_weights_n", 4, *[_ptrs([pss[p][8 * i + j] for i in range(4)]) for j in (0, 2, 4, 6)],
                      _ptrs(W[p]), st)
            maps = [mk(B, L // 16, L) if strips else mk(B, L, L) for _ in R]
            cur = [x.reshape(M, E) for x in xs]
            saved = [[] for _ in R]
            for i in range(4):
                img = [[raw(qkb), raw(qkb), raw(vb)] for _ in R]   # Rp images of q, k, v
                _call("pzn_attn_fused_proj", nprob, _ptrs(cur), _ptrs([W[p][i] for p in R]),
                      _ptrs([pss[p][8 * i + 1] for p in R]), _ptrs([pss[p][8 * i + 3] for p in R]),
                      _ptrs([pss[p][8 * i + 5] for p in R]), B, *[_ptrs([img[p][j] for p in R]) for j in range(3)], st,
                      flops=nprob * 2 * M * E * (2 * dk + E))
                r = [mk(M, E) for _ in R]
                t = [mk(M, E) for _ in R]
                mask = [torch.empty((M, 8), dtype=torch.int32, device=dev) for _ in R]
                lse = [mk(M) for _ in R]
                _call("pzn_attn_fused_fwd", nprob, _ptrs(cur), _ptrs([img[p][0] for p in R]), _ptrs([img[p][1] for p in R]),
                      _ptrs([img[p][2] for p in R]), _ptrs([W[p][i] for p in R]), _ptrs([pss[p][8 * i + 7] for p in R]), B,
                      _ptrs(r), _ptrs(t), _ptrs(mask), _ptrs(maps), _ptrs(lse), int(i > 0) | (2 if strips else 0),
                      0.25 / 16 if strips else 0.25, st,
                      flops=nprob * (2 * M * E * E + 2 * B * L * L * (dk + E)))
                for p in R:
                    saved[p].append((cur[p], t[p], mask[p], lse[p], img[p][0], img[p][1], img[p][2], W[p][i]))
                cur = r
            outs, tosave, empties = [], [], []
            for p in R:
                w_out, b_out = pss[p][32], pss[p][33]
                xsl = [saved[p][1][0], saved[p][2][0], saved[p][3][0], cur[p], xs[p].reshape(M, E)]   # att1..att4, f2f (:466)
                f_global = mk(B, Nout)
                arg = torch.empty((B, Nout), dtype=torch.int32, device=dev)
                y = mk(M, Nout) if need_out else None
                ws_bytes = lib.pzn_outproj_maxpts_workspace_bytes(L, E, 5, Nout)
                fused = False
                if ws_bytes:
                    # :466-475 in one launch (csrc/outproj.hip); `out` only when the caller wants it
                    try:
                        _call("pzn_outproj_maxpts_fwd_f32", _ptrs(xsl), 5, _p(w_out), _p(b_out), B, L, E, Nout, _p(y), _p(f_global),
                              _p(arg), _p(raw(ws_bytes)), st, flops=2 * M * 5 * E * Nout)
                        fused = True
                    except _lib.PznUnsupported:
                        pass
                if not fused:
                    y = mk(M, Nout) if y is None else y
                    for i, xi in enumerate(xsl):
                        _call("pzn_linear_slice_fwd_f32", _p(xi), w_out.data_ptr() + 4 * E * i, 5 * E, _p(b_out), M, E, Nout,
                              int(i > 0), _p(y), st, flops=2 * M * E * Nout)
                    _call("pzn_maxpool_points_fwd_f32", _p(y), B, L, Nout, _p(f_global), _p(arg), st)
                if y is None:         # (a placeholder keeps the output tuple's shape; it carries no data and no gradient)
                    y = f_global.new_empty((0,))
                    outs += [y, maps[p], f_global]
                    empties.append(y)
                else:
                    outs += [y.view(B, L, Nout), maps[p], f_global]
                tosave += [t_ for blk in saved[p] for t_ in blk] + [cur[p]] + pss[p] + [arg]
                if WINNER_CAPTURE is not None:
                    WINNER_CAPTURE.append(("gmax", args[per * p + 33].data_ptr(), arg))
        ctx.save_for_backward(*tosave)
        ctx.nprob = nprob
        ctx.dims = (B, L, E, dk, Nout)
        # the kernels pick the one- or three-plane instantiation from the process-wide precision mode at every launch, and the
        # single-plane projection leaves planes 1 and 2 of its images unwritten: the backward must run in the forward's mode
        ctx.attn_mode = _lib.load().pzn_attn_get_precision()
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(*maps, *empties)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gout):
        if _lib.load().pzn_attn_get_precision() != ctx.attn_mode:
            raise _lib.PznError("pzn_attn_set_precision() changed between the forward and the backward of an attention chain "
                                f"(forward: mode {ctx.attn_mode}): its bf16-plane images were written for the forward's mode")
        B, L, E, dk, Nout = ctx.dims
        nprob = ctx.nprob
        R = range(nprob)
        per_saved = 32 + 1 + 34 + 1
        T = ctx.saved_tensors
        M = B * L
        dev = T[0].device
        mk = lambda *sh: torch.empty(sh, dtype=torch.float32, device=dev)
        raw = lambda n: torch.empty(n, dtype=torch.uint8, device=dev)
        lib = _lib.load()
        vb = lib.pzn_attn_fused_v_image_bytes(B)
        grads = [None] * (2 + 35 * nprob)
        saved, ps, g, Gs, direct_blk = [], [], [], [], []
        with _on(dev):
            st = _stream()
            for p in R:
                t = T[per_saved * p: per_saved * (p + 1)]
                saved.append([t[8 * i: 8 * i + 8] for i in range(4)])
                att4 = t[32]
                ps.append(t[33:67])
                arg = t[67]
                dy, dfg = gout[3 * p], gout[3 * p + 2]
                w_out, b_out = ps[p][32], ps[p][33]
                base = 2 + 35 * p
                sparse = dy is None and dfg is not None and (5 * E) % 64 == 0 and L <= 600
                if sparse:
                    dfg = _f32(dfg, "df_global")
                else:
                    if dfg is not None:
                        dmax = mk(B, L, Nout)
                        _call("pzn_maxpool_points_bwd_f32", _p(_f32(dfg, "df_global")), _p(arg), B, L, Nout, _p(dmax), st)
                        dy = dmax if dy is None else dy + dmax
                    if dy is None:
                        dy = torch.zeros((M, Nout), dtype=torch.float32, device=dev)
                    dy = _f32(dy, "dy").reshape(M, Nout)
                xsl = [saved[p][1][0], saved[p][2][0], saved[p][3][0], att4, saved[p][0][0]]
                sink_w = _sink(w_out, ctx.needs_input_grad[base + 33])
                sink_b = _sink(b_out, ctx.needs_input_grad[base + 34])
                direct_out = sink_w is not None and sink_b is not None
                dW_out = sink_w if direct_out else torch.zeros_like(w_out)
                db_out = sink_b if direct_out else torch.zeros_like(b_out)
                G = mk(M, 5 * E)
                if sparse:
                    segs = (ctypes.c_void_p * 5)(*[_p(xi) for xi in xsl])
                    _call("pzn_linear_maxpts_wgrad_f32", _p(dfg), _p(arg), segs, 5, E, B, L, Nout, _p(dW_out), _p(db_out), st)
                    mws = torch.empty((lib.pzn_linear_maxpts_workspace_bytes(B, Nout) + 3) // 4, dtype=torch.int32, device=dev)
                    _call("pzn_linear_maxpts_dgrad_f32", _p(dfg), _p(arg), _p(w_out), B, L, 5 * E, Nout, _p(mws), _p(G), st)
                else:
                    for i, xi in enumerate(xsl):
                        _call("pzn_linear_slice_wgrad_f32", _p(dy), _p(xi), M, E, Nout, dW_out.data_ptr() + 4 * E * i, 5 * E,
                              _p(db_out) if i == 0 else None, st, flops=2 * M * E * Nout)
                    _call("pzn_linear_dgrad_f32", _p(dy), None, _p(w_out), M, 5 * E, Nout, None, _p(G), st,
                          flops=2 * M * 5 * E * Nout)
                if not direct_out:
                    grads[base + 33], grads[base + 34] = dW_out, db_out
                Gs.append(G)
                g.append(G[:, 3 * E: 4 * E])          # gradient of att4: its slice of the projection only (read in place)
            # scratch of the block backward, shared by the four layers
            dz, u, dx = [mk(M, E) for _ in R], [mk(M, E) for _ in R], [mk(M, E) for _ in R]
            dq, dkk, dvv = [mk(M, dk) for _ in R], [mk(M, dk) for _ in R], [mk(M, E) for _ in R]
            dqt = [mk(M, dk) for _ in R]          # dq once more in the kernels' tile order (query side -> key side)
            darp, delta = [raw(vb) for _ in R], [mk(M) for _ in R]
            g2 = None
            for i in (3, 2, 1, 0):
                blk = [saved[p][i] for p in R]        # (x, t, mask, lse, qrp, krp, vrp, W)
                col = lambda j: _ptrs([b[j] for b in blk])
                # the block's output gradient = its slice of the projection's input gradient (+ what the next block passed
                # back): both read in place by the kernel, no copy and no tensor add
                _call("pzn_attn_fused_bwd_q", nprob, _ptrs(g), 5 * E, _ptrs(g2) if g2 is not None else None, E,
                      col(2), col(4), col(5), col(6), col(7), B, _ptrs(dz),
                      _ptrs(u), _ptrs(dq), _ptrs(dqt), _ptrs(darp), _ptrs(delta), st,
                      flops=nprob * (2 * M * E * (E + dk) + 2 * B * L * L * (2 * dk + E)))
                _call("pzn_attn_fused_bwd_k", nprob, col(4), col(5), col(6), _ptrs(darp), col(7), col(3),
                      _ptrs(delta), _ptrs(u), _ptrs(dqt), B, _ptrs(dkk), _ptrs(dvv), _ptrs(dx), st,
                      flops=nprob * (2 * M * E * (E + dk) + 2 * B * L * L * (2 * dk + 2 * E)))
                for p in R:
                    base = 2 + 35 * p
                    prm = ps[p][8 * i: 8 * i + 8]
                    sinks = [_sink(p_, ctx.needs_input_grad[base + 1 + 8 * i + j]) for j, p_ in enumerate(prm)]
                    direct = all(s_ is not None for s_ in sinks)
                    gp = sinks if direct else [torch.empty_like(p_) for p_ in prm]
                    gq, gbq, gk, gbk, gv, gbv, go, gbo = gp
                    _call("pzn_attn_fused_wgrads", _p(dz[p]), _p(blk[p][1]), _p(dq[p]), _p(dkk[p]), _p(dvv[p]), _p(blk[p][0]),
                          M, E, dk, _p(gq), _p(gbq), _p(gk), _p(gbk), _p(gv), _p(gbv), _p(go), _p(gbo), int(direct), st,
                          flops=2 * M * E * (2 * dk + 2 * E))
                    if not direct:
                        grads[base + 1 + 8 * i: base + 9 + 8 * i] = gp
                sl = i - 1 if i > 0 else 4            # att_i sits in slice i-1 of the concatenation, f2f in slice 4
                g = [Gs[p][:, sl * E: (sl + 1) * E] for p in R]
                g2, dx = dx, [mk(M, E) for _ in R] if i > 0 else dx      # (dx of this block is the next one's second addend)
            for p in R:
                grads[2 + 35 * p] = torch.add(g[p], g2[p]).view(B, L, E)
        return tuple(grads)


class _AttnChainOne(torch.autograd.Function):
    """model5_b.py:462-475 of ONE encoder behind one C entry point each way (csrc/attnchain.hip: the kernels of
    _AttnChainFused, enqueued by the library on one caller-owned buffer per direction) for the case predict5 creates: `out` is
    not wanted and only f_global = max over the points carries a gradient.  inputs: strips flag, x[B,L,E], the 34 parameters
    -> (mean map [B,L,L] or its strip column sums [B,L/16,L], f_global[B,Nout])."""

    @staticmethod
    def forward(ctx, strips, x, *params):
        x = _f32(x, "x")
        ps = [_f32(t, "param") for t in params]
        B, L, E = x.shape
        Nout = ps[32].shape[0]
        dev = x.device
        lib = _lib.load()
        amap = torch.empty((B, L // 16, L) if strips else (B, L, L), dtype=torch.float32, device=dev)
        f_global = torch.empty((B, Nout), dtype=torch.float32, device=dev)
        arg = torch.empty((B, Nout), dtype=torch.int32, device=dev)
        saved = torch.empty((lib.pzn_attn_chain_saved_bytes(B),), dtype=torch.uint8, device=dev)
        with _on(dev):
            _call("pzn_attn_chain_fwd_f32", _p(x), _ptrs(ps), B, int(bool(strips)), _p(amap), None, _p(f_global), _p(arg),
                  _p(saved), _stream(),
                  flops=4 * (2 * B * L * E * (2 * (E // 4) + E) + 2 * B * L * E * E + 2 * B * L * L * (E // 4 + E)) + 2 * B * L * 5 * E * Nout)
        if WINNER_CAPTURE is not None:
            WINNER_CAPTURE.append(("gmax", params[32].data_ptr(), arg))
        ctx.save_for_backward(x, saved, arg, *ps)
        ctx.attn_mode = lib.pzn_attn_get_precision()
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(amap)
        return amap, f_global

    @staticmethod
    def backward(ctx, _dmap, dfg):
        lib = _lib.load()
        if lib.pzn_attn_get_precision() != ctx.attn_mode:
            raise _lib.PznError("pzn_attn_set_precision() changed between the forward and the backward of an attention chain "
                                f"(forward: mode {ctx.attn_mode}): its bf16-plane images were written for the forward's mode")
        x, saved, arg = ctx.saved_tensors[:3]
        ps = list(ctx.saved_tensors[3:])
        if dfg is None:
            return (None,) * 36
        B, L, E = x.shape
        dev = x.device
        dfg = _f32(dfg, "df_global")
        sinks = [_sink(p_, ctx.needs_input_grad[2 + j]) for j, p_ in enumerate(ps)]
        direct = all(s_ is not None for s_ in sinks)
        if direct:
            gp = sinks
        else:      # (the out projection's sparse backward ADDS: zero-initialised; the blocks' gradients are overwritten)
            gp = [torch.empty_like(p_) for p_ in ps[:32]] + [torch.zeros_like(ps[32]), torch.zeros_like(ps[33])]
        dx = torch.empty((B, L, E), dtype=torch.float32, device=dev)
        scratch = torch.empty((lib.pzn_attn_chain_scratch_bytes(B),), dtype=torch.uint8, device=dev)
        with _on(dev):
            _call("pzn_attn_chain_bwd_f32", _p(x), _ptrs(ps), _p(saved), _p(arg), _p(dfg), B, _ptrs(gp), int(direct), _p(dx),
                  _p(scratch), _stream())
        if direct:
            return (None, dx) + (None,) * 34
        return (None, dx) + tuple(gp)


def attention_chain_one_supported(x, dk, w_out):
    """The shape the one-call chain takes: the encoder's (256 points, E = 256, dk = 64, a 1280 -> 1024 out projection) on the
    split-precision matrix-core path, 16-byte aligned."""
    return (attention_chain_fused_supported(x, dk, w_out) and tuple(w_out.shape) == (1024, 1280) and
            _lib.load().pzn_outproj_maxpts_workspace_bytes(256, 256, 5, 1024) > 0 and _lib.load().pzn_gemm_get_precision() != 0)


def attention_chain_fused(xs, blocks_list, w_outs, b_outs, need_out=True, map_strips=False):
    """xs: list of nprob inputs [B,L,E]; blocks_list[p]: four 8-tuples; -> list of (out, mean map, f_global) per problem.
    need_out=False: only f_global = max over the points of the out projection is wanted (predict5, model5_b.py:723);
    `out` is then None and its 67 MB per encoder at B = 64 are never written.
    map_strips=True: the caller only takes the mean of the mean map over its rows (training_step, model5_b.py:937-942): the
    second result is [B,L/16,L], each row the scaled column sums of a strip of 16 query rows, with
    result.mean(dim=1) == mean_map.mean(dim=1); the [B,L,L] map (16.7 MB read + written per block and encoder) never exists."""
    if not need_out and not KernelTimer.enabled and all(
            attention_chain_one_supported(x, blocks[0][0].shape[0], w) for x, blocks, w in zip(xs, blocks_list, w_outs)):
        # what predict5 asks for (only the maximum of the projection): one C call per encoder and direction.  (With the
        # entry-point timer on - bench.py's `stages` - the chain is composed here, so that every kernel's entry point is timed.)
        res = []
        for x, blocks, w, b in zip(xs, blocks_list, w_outs, b_outs):
            amap, fg = _AttnChainOne.apply(bool(map_strips), x, *[p_ for blk in blocks for p_ in blk], w, b)
            res.append((None, amap, fg))
        return res
    flat = []
    for x, blocks, w, b in zip(xs, blocks_list, w_outs, b_outs):
        flat += [x] + [p_ for blk in blocks for p_ in blk] + [w, b]
    res = _AttnChainFused.apply(len(xs), int(bool(need_out)) | (2 if map_strips else 0), *flat)
    return [(res[3 * i] if res[3 * i].numel() else None, res[3 * i + 1], res[3 * i + 2]) for i in range(len(xs))]


class _BnPointsRelu(torch.autograd.Function):
    """relu(BatchNorm1d(num_points)(x)) for x[B, N, C] (model5_b.py:424, :447-448: the BN channel axis is the point
    index) as one launch each way (csrc/bnpoints.hip) instead of BN + clamp and their two backward kernels."""

    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, training, momentum, eps):
        x = _f32(x, "x")
        B, N, C = x.shape
        dev = x.device
        y = torch.empty_like(x)
        mean = torch.empty((N,), dtype=torch.float32, device=dev)
        invstd = torch.empty((N,), dtype=torch.float32, device=dev)
        with _on(dev):
            _call("pzn_bn_points_relu_fwd_f32", _p(x), _p(weight), _p(bias), _p(running_mean), _p(running_var),
                  int(bool(training)), float(momentum), float(eps), B, N, C, _p(y), _p(mean), _p(invstd), _stream())
        ctx.save_for_backward(x, weight, bias, mean, invstd)
        ctx.training = bool(training)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, bias, mean, invstd = ctx.saved_tensors
        B, N, C = x.shape
        dy = _f32(dy, "dy")
        dev = dy.device
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        need_w = weight is not None and ctx.needs_input_grad[1]
        need_b = bias is not None and ctx.needs_input_grad[2]
        sw, sb = _sink(weight, need_w), _sink(bias, need_b)
        direct = (sw is not None or not need_w) and (sb is not None or not need_b)
        if direct:
            dw, db = sw, sb
        else:
            dw = torch.zeros((N,), dtype=torch.float32, device=dev) if need_w else None
            db = torch.zeros((N,), dtype=torch.float32, device=dev) if need_b else None
        with _on(dev):
            _call("pzn_bn_points_relu_bwd_f32", _p(x), _p(dy), _p(weight), _p(bias), _p(mean), _p(invstd),
                  int(ctx.training), B, N, C, _p(dx), _p(dw), _p(db), _stream())
        if direct:
            return dx, None, None, None, None, None, None, None
        return dx, dw, db, None, None, None, None, None


def bn_points_relu(x, bn):
    """relu(bn(x)) for an nn.BatchNorm1d(num_points) applied to x[B, N, C] on the GPU: batch statistics and running
    buffers exactly as the module keeps them (momentum must be a number, as in the reference)."""
    if x.dim() != 3 or x.shape[1] != bn.num_features or bn.momentum is None:
        raise _lib.PznError(f"bn_points_relu: x{tuple(x.shape)} vs BatchNorm1d({bn.num_features}), momentum={bn.momentum}")
    use_batch = bn.training or not bn.track_running_stats
    rm = bn.running_mean if bn.track_running_stats else None
    rv = bn.running_var if bn.track_running_stats else None
    if bn.training and bn.track_running_stats and bn.num_batches_tracked is not None:
        bn.num_batches_tracked.add_(1)
    return _BnPointsRelu.apply(x, bn.weight, bn.bias, rm, rv, use_batch, bn.momentum, bn.eps)


class _StemFused(torch.autograd.Function):
    """model5_b.py:447-448 in one launch each way (csrc/stem.hip): relu(bn2(mlp2(relu(bn1(mlp1(xyz)))))) with both
    BatchNorm1d(num_points) over the point axis; the backward recomputes every activation from xyz and the saved statistics."""

    @staticmethod
    def forward(ctx, xyz, w1, b1, g1, o1, w2, b2, g2, o2, rm1, rv1, rm2, rv2, use_batch, mom1, eps1, mom2, eps2, two=False):
        xyz = _f32(xyz, "xyz")
        w1, b1, w2, b2 = _f32(w1, "w1"), _f32(b1, "b1"), _f32(w2, "w2"), _f32(b2, "b2")
        B, N, _ = xyz.shape
        dev = xyz.device
        out = torch.empty((B, N, 64), dtype=torch.float32, device=dev)
        stats = torch.empty((4, N), dtype=torch.float32, device=dev)
        with _on(dev):
            _call("pzn_stem_fwd_f32", _p(xyz), _p(w1), _p(b1), _p(g1), _p(o1), _p(rm1), _p(rv1), float(mom1), float(eps1),
                  _p(w2), _p(b2), _p(g2), _p(o2), _p(rm2), _p(rv2), float(mom2), float(eps2), int(bool(use_batch)), B, N,
                  _p(out), _p(stats[0]), _p(stats[1]), _p(stats[2]), _p(stats[3]), _stream(),
                  flops=2 * B * N * 64 * (3 + 64))
        ctx.save_for_backward(xyz, w1, b1, g1, o1, w2, b2, g2, o2, stats)
        ctx.param_refs = (w1, b1, g1, o1, w2, b2, g2, o2)
        ctx.use_batch = bool(use_batch)
        ctx.set_materialize_grads(False)
        if two:      # the same features under two names: each consumer's gradient arrives on its own and the kernel adds them
            return out, out.view_as(out)
        return out

    @staticmethod
    def backward(ctx, dout, dout2=None):
        xyz, w1, b1, g1, o1, w2, b2, g2, o2, stats = ctx.saved_tensors
        if ctx.needs_input_grad[0]:
            raise _lib.PznError("stem: gradients w.r.t. point coordinates are not provided on the fused stem; "
                                "use ops.linear + ops.bn_points_relu")
        if dout is None:
            dout, dout2 = dout2, None
        if dout is None:
            return (None,) * 19
        B, N, _ = xyz.shape
        dout = _f32(dout, "dout")
        dout2 = None if dout2 is None else _f32(dout2, "dout2")
        dev = dout.device
        params = (w1, b1, g1, o1, w2, b2, g2, o2)
        sinks = [None if t is None else _sink(t, ctx.needs_input_grad[1 + i]) for i, t in enumerate(params)]
        direct = all(s_ is not None or t is None for s_, t in zip(sinks, params))
        if direct:
            grads = sinks
        else:
            grads = [None if t is None else torch.zeros_like(t) for t in params]      # the kernel adds into these
        dw1, db1, dg1, do1, dw2, db2, dg2, do2 = grads
        ws = torch.empty((_lib.load().pzn_stem_bwd_workspace_bytes(N) + 3) // 4, dtype=torch.float32, device=dev)
        with _on(dev):
            _call("pzn_stem_bwd_f32", _p(xyz), _p(dout), _p(dout2), _p(w1), _p(b1), _p(w2), _p(b2), _p(g1), _p(o1), _p(g2), _p(o2),
                  _p(stats[0]), _p(stats[1]), _p(stats[2]), _p(stats[3]), int(ctx.use_batch), B, N, _p(dw1), _p(db1),
                  _p(dw2), _p(db2), _p(dg1), _p(do1), _p(dg2), _p(do2), _p(ws), _stream(), flops=2 * B * N * 64 * (3 + 3 * 64))
        if direct:
            return (None,) * 19
        return (None, dw1, db1, dg1, do1, dw2, db2, dg2, do2) + (None,) * 10


def stem_supported(xyz, lin1, bn1, lin2, bn2):
    return (xyz.is_cuda and xyz.dim() == 3 and xyz.shape[2] == 3 and xyz.shape[0] <= 64 and not xyz.requires_grad and
            tuple(lin1.weight.shape) == (64, 3) and tuple(lin2.weight.shape) == (64, 64) and lin1.bias is not None and
            lin2.bias is not None and bn1.num_features == xyz.shape[1] == bn2.num_features and bn1.momentum is not None and
            bn2.momentum is not None and bn1.training == bn2.training and bn1.track_running_stats == bn2.track_running_stats)


def stem(xyz, lin1, bn1, lin2, bn2, two=False):
    """relu(bn2(lin2(relu(bn1(lin1(xyz)))))) for the encoder's shapes (stem_supported); the modules' running buffers are kept
    exactly as the modules keep them.  two=True: -> (features, the same features under a second name): hand each of two consumers
    its own name and their gradients are added inside the backward launch instead of by autograd's pass over two 33.5 MB
    tensors."""
    use_batch = bn1.training or not bn1.track_running_stats
    track = bn1.track_running_stats
    for bn in (bn1, bn2):
        if bn.training and track and bn.num_batches_tracked is not None:
            bn.num_batches_tracked.add_(1)
    return _StemFused.apply(xyz, lin1.weight, lin1.bias, bn1.weight, bn1.bias, lin2.weight, lin2.bias, bn2.weight, bn2.bias,
                            bn1.running_mean if track else None, bn1.running_var if track else None,
                            bn2.running_mean if track else None, bn2.running_var if track else None,
                            use_batch, bn1.momentum, bn1.eps, bn2.momentum, bn2.eps, bool(two))


class _SaLevelFused(torch.autograd.Function):
    """The set-abstraction level (model5_b.py:449-454 / :456-461) with the first layer per point and its rows never in
    memory: W1[:,0:3] (xyz[j] - centre) is split into a per-point and a per-group part, so a grouped row is
    relu(Pp[idx] + Q[group]) with Pp = feat W1[:,3:]^T + W1[:,0:3] xyz and Q = b1 - W1[:,0:3] centre; the rows are generated
    inside the matrix-core kernel's operand loader forward and inside the backward passes (weight gradients of the pooled
    layer, walk by point).  Same result as group + shared_mlp_max up to the order of the fp32 sum; no h tensor (537 MB per
    level and cloud at B = 64).  One C entry point each way (csrc/sachain.hip) on one buffer per direction; with the
    entry-point timer on the same launches are made one by one from here."""

    @staticmethod
    def forward(ctx, xyz, feat, new_xyz, idx, w1, b1, w2, b2):
        xyz, feat, new_xyz = _f32(xyz, "xyz"), _f32(feat, "points"), _f32(new_xyz, "new_xyz")
        idx = None if idx is None else _i64(idx, "idx")
        w1, b1, w2, b2 = _f32(w1, "w1"), _f32(b1, "b1"), _f32(w2, "w2"), _f32(b2, "b2")
        B, N, _ = xyz.shape
        S = new_xyz.shape[1]
        D = feat.shape[-1]
        C1, C2 = w1.shape[0], w2.shape[0]
        dev = xyz.device
        R = B * S
        lib = _lib.load()
        out = torch.empty((R, C2), dtype=torch.float32, device=dev)
        arg = torch.empty((R, C2), dtype=torch.int32, device=dev)
        ctx.stepwise = KernelTimer.enabled
        if ctx.stepwise:
            saved = _SaLevelFused._forward_stepwise(xyz, feat, new_xyz, idx, w1, b1, w2, b2, out, arg)
        else:
            saved = (torch.empty((lib.pzn_sa_level_chain_saved_bytes(B, N, S, D, C1, C2),), dtype=torch.uint8, device=dev),)
            with _on(dev):
                _call("pzn_sa_level_chain_fwd_f32", _p(xyz), _p(feat), _p(new_xyz), _p(idx), _p(w1), _p(b1), _p(w2), _p(b2),
                      B, N, S, D, C1, C2, _p(out), _p(arg), _p(saved[0]), _stream())
        if WINNER_CAPTURE is not None:
            WINNER_CAPTURE.append(("sa", w2.data_ptr(), arg.view(B, S, C2)))
        ctx.have_idx = idx is not None
        ctx.save_for_backward(xyz, feat, new_xyz, idx if idx is not None else xyz.new_empty(0), w1, w2, out, arg, *saved)
        ctx.dims = (B, N, S, D, R, C1, C2)
        ctx.param_refs = (w1, b1, w2, b2)
        return out.reshape(B, S, C2)

    @staticmethod
    def _forward_stepwise(xyz, feat, new_xyz, idx, w1, b1, w2, b2, out, arg):
        """The forward's launches one entry point at a time (bench.py's `stages`) -> (idx, w_f, P, Q) for the backward."""
        B, N, _ = xyz.shape
        S, D, C1, C2 = new_xyz.shape[1], feat.shape[-1], w1.shape[0], w2.shape[0]
        dev, R = xyz.device, B * new_xyz.shape[1]
        w_f = w1[:, 3:].contiguous()
        P = torch.empty((B * N, C1), dtype=torch.float32, device=dev)
        Q = torch.empty((R, C1), dtype=torch.float32, device=dev)
        with _on(dev):
            _call("pzn_linear_fwd_f32", _p(feat), _p(w_f), None, B * N, D, C1, 0, _p(P), _stream(), flops=2 * B * N * D * C1)
            if idx is None:
                idx = torch.empty((B, S, 32), dtype=torch.int64, device=dev)
                _call("pzn_knn_f32", _p(xyz), _p(new_xyz), B, N, S, 32, _p(idx), _stream())
            _call("pzn_sa_prep_f32", _p(xyz), _p(new_xyz), _p(w1), _p(b1), B, N, S, D, C1, _p(P), _p(Q), _stream())
            ws_bytes = _lib.load().pzn_sa_level_fwd_workspace_bytes(C1, C2)
            ws = torch.empty((max(ws_bytes, 16),), dtype=torch.uint8, device=dev)
            done = False
            if ws_bytes:      # streamed-weights kernel: the weight split and the level as two entry points
                try:
                    _call("pzn_sa_level_prep_weights_f32", _p(w2), C1, C2, _p(ws), _stream())
                    _call("pzn_sa_level_fwd_packed_f32", _p(P), _p(Q), _p(idx), _p(b2), B, N, S, C1, C2, _p(out), _p(arg),
                          _p(ws), _stream(), flops=2 * R * 32 * C1 * C2, variant=f"<{C1}, {C2 // 32}>")
                    done = True
                except _lib.PznUnsupported:
                    pass
            if not done:
                _call("pzn_sa_level_fwd_ws_f32", _p(P), _p(Q), _p(idx), _p(w2), _p(b2), B, N, S, C1, C2, _p(out), _p(arg),
                      _p(ws), _stream(), flops=2 * R * 32 * C1 * C2)
        return idx, w_f, P, Q

    @staticmethod
    def backward(ctx, dout):
        xyz, feat, new_xyz, idx, w1, w2, out, arg = ctx.saved_tensors[:8]
        B, N, S, D, R, C1, C2 = ctx.dims
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[2]:
            raise _lib.PznError("sa_mlp_max: gradients w.r.t. point coordinates are not provided on the fused "
                                "encoder path; use pointnet_util.sample_and_group + ops.shared_mlp_max")
        dout = _f32(dout, "dout").reshape(R, C2)
        dev = dout.device
        need_feat = ctx.needs_input_grad[1]
        sinks = [_sink(t, ctx.needs_input_grad[4 + i]) for i, t in enumerate(ctx.param_refs)]
        direct = all(s_ is not None for s_ in sinks)
        if direct:
            dW1, db1, dW2, db2 = sinks
        else:      # (the stepwise kernels ADD into dW1 / db1: zero-filled here; the one-call form fills them itself)
            mk1 = torch.zeros if ctx.stepwise else torch.empty
            dW1 = mk1((C1, 3 + D), dtype=torch.float32, device=dev)
            db1 = mk1((C1,), dtype=torch.float32, device=dev)
            dW2 = torch.empty_like(w2)
            db2 = torch.empty((C2,), dtype=torch.float32, device=dev)
        dfeat = torch.empty((B, N, D), dtype=torch.float32, device=dev) if need_feat else None
        lib = _lib.load()
        if not ctx.stepwise:
            saved = ctx.saved_tensors[8]
            scratch = torch.empty((lib.pzn_sa_level_chain_scratch_bytes(B, N, S, C1, C2),), dtype=torch.uint8, device=dev)
            with _on(dev):
                _call("pzn_sa_level_chain_bwd_f32", _p(dout), _p(arg), _p(out), _p(xyz), _p(feat), _p(new_xyz),
                      _p(idx) if ctx.have_idx else None, _p(w1), _p(w2), _p(saved), B, N, S, D, C1, C2, _p(dfeat), _p(dW1),
                      _p(db1), _p(dW2), _p(db2), int(direct), _p(scratch), _stream())
        else:
            idx_s, w_f, P, Q = ctx.saved_tensors[8:12]
            off = torch.empty((B * (N + 1),), dtype=torch.int32, device=dev)
            rows = torch.empty((B * S * 32,), dtype=torch.int32, device=dev)
            pts = torch.empty((B * S * 32,), dtype=torch.int32, device=dev)
            dP = torch.empty((B * N, C1), dtype=torch.float32, device=dev)
            with _on(dev):
                _call("pzn_knn_inverse_lists", _p(idx_s), B, N, S, 32, _p(off), _p(rows), _p(pts), _stream())
                ws = torch.empty((lib.pzn_sa_level_bwd_pt_workspace_bytes(B, S, C2) + 3) // 4, dtype=torch.float32, device=dev)
                _call("pzn_sa_level_bwd_pt_f32", _p(dout), _p(arg), _p(out), _p(w2), _p(P), _p(Q), _p(idx_s), _p(xyz), _p(new_xyz),
                      _p(off), _p(rows), _p(pts), B, N, S, D, C1, C2, _p(dP), _p(dW2), _p(db2), _p(dW1), _p(db1), int(direct),
                      _p(ws), _stream(), flops=2 * R * (2 * C1 * C2))
                if need_feat:
                    _call("pzn_linear_dgrad_f32", _p(dP), None, _p(w_f), B * N, D, C1, None, _p(dfeat), _stream(),
                          flops=2 * B * N * D * C1)
                _call("pzn_linear_slice_wgrad_f32", _p(dP), _p(feat), B * N, D, C1, dW1.data_ptr() + 12, 3 + D, None, _stream(),
                      flops=2 * B * N * D * C1)
        if direct:
            return None, dfeat, None, None, None, None, None, None
        return None, dfeat, None, None, dW1, db1, dW2, db2


def sa_level_fused_supported(feat, idx, w1, w2):
    """The shapes the per-point set-abstraction kernels take, decided BEFORE anything is launched (forward: pzn_sa_prep_f32 +
    pzn_sa_level_fwd_*; backward: pzn_sa_level_bwd_pt_f32): 32 neighbours, a first layer of 128 or 256 channels over
    [xyz | features], a second layer of 64 / 128 / 256 channels, the split-precision matrix-core path."""
    K = 32 if idx is None else idx.shape[2]
    return (K == 32 and w1.shape[0] in (128, 256) and w1.shape[1] == 3 + feat.shape[-1] and w2.shape[0] in (64, 128, 256)
            and w2.shape[1] == w1.shape[0] and _lib.load().pzn_gemm_get_precision() != 0)


def sa_mlp_max(xyz, feat, new_xyz, idx, w1, b1, w2, b2):
    """sample_and_group(.., knn=True)'s grouping (pointnet_util.py:117-132) + relu(lin_a) + relu(lin_b) + max over the
    neighbours (model5_b.py:449-454 / :456-461) for given centroids.  The encoder's shapes run per point with no grouped tensor
    in memory (_SaLevelFused); anything else is composed as the reference composes it: the [B,S,K,3+D] group tensor (search
    fused with the group write when idx is None) and the shared MLP + max on its rows."""
    if sa_level_fused_supported(feat, idx, w1, w2):
        try:
            return _SaLevelFused.apply(xyz, feat, new_xyz, idx, w1, b1, w2, b2)
        except _lib.PznUnsupported:
            pass        # a table beyond the 32-bit buffer offsets: the composed form below
    if idx is None and knn_group_supported(xyz, feat, 32):
        grouped = knn_group(xyz, feat, new_xyz)[0]
    else:
        grouped = group(xyz, feat, new_xyz, knn(xyz, new_xyz, 32) if idx is None else idx)
    return shared_mlp_max(grouped, w1, b1, w2, b2)
