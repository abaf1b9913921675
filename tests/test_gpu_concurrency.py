"""Kernels that share the chip with another stream's kernels must give the results they give alone.

The training step runs the two encoders on two HIP streams, so workgroups of different kernels sit on the same CU.  This
caught a real fault: the packed-fp32 form of the set-abstraction prep kernel (v_pk_fma_f32 with op_sel modifiers) returned
sums with one term missing in lanes 48-63 while the general matrix-core engine ran beside it (csrc/sapoint.hip).  Every
deterministic forward kernel family is run alone (reference, checked to be bit-reproducible) and then 40 times while an
aggressor kernel is kept busy on a second stream; outputs must be bit-identical."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
REPS = 40


@pytest.fixture(scope="module")
def rig():
    from puzzlenet_amd import ops
    from puzzlenet_amd.ops import _call, _p
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(3)
    B, N, S, D, C1, C2 = 8, 2048, 512, 64, 128, 128
    xyz = torch.rand(B, N, 3, generator=g).to(dev)
    feat = torch.randn(B, N, D, generator=g).to(dev)
    new_xyz = ops.index_points(xyz, ops.farthest_point_sample(xyz, S, torch.zeros(B, dtype=torch.long, device=dev)))
    w1 = (torch.randn(C1, 3 + D, generator=g) / 8).to(dev)
    b1 = torch.randn(C1, generator=g).to(dev)
    w2 = (torch.randn(C2, C1, generator=g) / 11).to(dev)
    b2 = torch.randn(C2, generator=g).to(dev)
    a_pts, b_pts = torch.rand(16, 512, 3, generator=g).to(dev), torch.rand(16, 512, 3, generator=g).to(dev)
    xa = (0.5 * torch.randn(16, 256, 256, generator=g)).to(dev)
    aw = [(torch.randn(*s, generator=g) / (16 if len(s) == 2 else 4)).to(dev)
          for s in [(64, 256), (64,), (64, 256), (64,), (256, 256), (256,), (256, 256), (256,)]]
    xl = torch.randn(32768, 64, generator=g).to(dev)
    wl, bl = (torch.randn(64, 64, generator=g) / 8).to(dev), torch.randn(64, generator=g).to(dev)
    xg = torch.randn(4096, 1280, generator=g).to(dev)
    wg = (torch.randn(1024, 1280, generator=g) / 30).to(dev)
    yg = torch.empty(4096, 1024, device=dev)
    side = torch.cuda.Stream()
    bg = torch.zeros(1024, device=dev)
    fps_counts = torch.full((B,), N - 100, dtype=torch.int64, device=dev)

    def chain():      # the four blocks of an encoder as chained kernels (csrc/attnfused.hip: AGPR accumulators) + out projection
        return ops.attention_chain_fused([xa], [[tuple(aw)] * 4], [wg], [bg])[0]
    # a boundary-head chain forward + backward (csrc/pointmlp.hip: LDS-DMA staging tiles, AGPR accumulators, partial sums)
    pm_w = [(torch.randn(o, i, generator=g) / 8).to(dev) for o, i in ((64, 128), (64, 64), (32, 64), (2, 32))][1:]
    pm_w1 = (torch.randn(64, 128, generator=g) / 8).to(dev)
    pm_b = [torch.randn(o, generator=g).to(dev) for o in (64, 32, 2)]
    pm_x, pm_g = torch.randn(16, 2048, 64, generator=g).to(dev), torch.randn(16, 1, 64, generator=g).to(dev)
    pm_go = torch.randn(16, 2048, 2, generator=g).to(dev)

    def point_mlp():
        with torch.enable_grad():
            leaves = [t.detach().requires_grad_(True) for t in (pm_x, pm_g, pm_w1, pm_b[0], pm_w[1], pm_b[1], pm_w[2], pm_b[2])]
            y = ops.point_mlp3(leaves[0], leaves[2], leaves[3], leaves[4], leaves[5], leaves[6], leaves[7], g=leaves[1])
            grads = torch.autograd.grad(y, leaves, pm_go)
        return torch.cat([y.detach().reshape(-1)] + [t.reshape(-1) for t in grads])
    def cat_global():      # the boundary heads' first layer on the layer-by-layer path (csrc/losstail.hip: per-cloud bias, gated
        with torch.enable_grad():      # column sums in a fixed order), forward + backward
            leaves = [t.detach().requires_grad_(True) for t in (pm_x, pm_g)]      # (input gradients: the weight gradients of
            y = ops.cat_global_linear_relu(leaves[0], leaves[1], pm_w1, pm_b[0])   #  this path end in float atomics)
            grads = torch.autograd.grad(y, leaves, torch.ones_like(y))
        return torch.cat([y.detach().reshape(-1)] + [t.reshape(-1) for t in grads])
    def sa_backward_victim():      # the level's backward by point (csrc/sapool.hip, poolbwd.hip): the order-fixed gradients (dP -> feat,
        with torch.enable_grad():   # dW2, db2; the first layer's weight gradients end in float atomics and are left out)
            leaves = [t.detach().requires_grad_(True) for t in (feat, w1, b1, w2, b2)]
            y = ops.sa_mlp_max(xyz, leaves[0], new_xyz, None, *leaves[1:])
            gf, _, _, gw2, gb2 = torch.autograd.grad(y, leaves, torch.ones_like(y))
        return torch.cat([gf.reshape(-1), gw2.reshape(-1), gb2.reshape(-1)])
    victims = {
        "point_mlp3": point_mlp,
        "cat_global_linear_relu": cat_global,
        "knn": lambda: ops.knn(xyz, new_xyz, 32),
        "knn_group": lambda: ops.knn_group(xyz, feat, new_xyz)[0],
        "fps": lambda: ops.farthest_point_sample(xyz, 256, torch.zeros(B, dtype=torch.long, device=dev)),
        # the data pipeline's form (csrc/fps.hip: packed fp32 distance update - the instruction family of the fault this file is
        # about -, two clouds per workgroup): what runs beside a training step when it is fed from raw clouds
        "fps_background": lambda: ops.farthest_point_sample(xyz, 256, torch.zeros(B, dtype=torch.long, device=dev), background=True,
                                                            counts=fps_counts, max_count=N - 100),
        "ball_query": lambda: ops.ball_query(0.2, 32, xyz, new_xyz),
        "sa_level": lambda: ops.sa_mlp_max(xyz, feat, new_xyz, None, w1, b1, w2, b2),
        "sa_level_backward": sa_backward_victim,
        "chamfer": lambda: torch.cat([t.reshape(-1).float() for t in ops.chamfer(a_pts, b_pts)]),
        "attention_block": lambda: ops.attention_block(xa, *aw)[0],
        "attention_chain_fused": lambda: torch.cat([t.reshape(-1) for t in chain()]),
        "linear_weight_stationary": lambda: ops.linear(xl, wl, bl, relu=True),
        "linear_general_engine": lambda: ops.linear(xg, wg, None),
        "max_over_points": lambda: ops.max_over_points(xa),
    }

    def agg_general():
        _call("pzn_linear_fwd_f32", _p(xg), _p(wg), None, 4096, 1280, 1024, 0, _p(yg), side.cuda_stream)

    def agg_level():
        with torch.cuda.stream(side):
            ops.sa_mlp_max(xyz, feat, new_xyz, None, w1, b1, w2, b2)

    def agg_attention():
        with torch.cuda.stream(side):
            ops.attention_block(xa, *aw)

    def agg_chain():
        with torch.cuda.stream(side):
            chain()

    def agg_point_mlp():
        with torch.cuda.stream(side):
            point_mlp()

    return dev, side, victims, {"general_engine": agg_general, "sa_level": agg_level, "attention_block": agg_attention,
                                "attention_fused": agg_chain, "point_mlp3": agg_point_mlp}


def _trace(line):
    path = os.environ.get("PZN_TEST_TRACE")
    if path:
        with open(path, "a") as f:
            f.write(line + "\n")


@pytest.mark.parametrize("aggressor", ["general_engine", "sa_level", "attention_block", "attention_fused", "point_mlp3"])
def test_results_do_not_depend_on_the_other_stream(rig, aggressor):
    dev, side, victims, aggressors = rig
    ag = aggressors[aggressor]
    report = {}
    with torch.no_grad():
        for name, fn in victims.items():
            _trace(f"[{aggressor}] {name}")      # (PZN_TEST_TRACE=<file>: which pair was running if the GPU faults)
            ref, again = fn(), fn()
            torch.cuda.synchronize()
            assert torch.equal(ref, again), f"{name} is not reproducible even alone"
            bad = torch.zeros((), dtype=torch.int64, device=dev)
            side.wait_stream(torch.cuda.current_stream())
            for _ in range(REPS):
                ag()
                ag()
                bad += (fn() != ref).sum()
            torch.cuda.synchronize()
            if int(bad):
                report[name] = int(bad)
    assert not report, f"wrong elements over {REPS} launches beside {aggressor}: {report}"
