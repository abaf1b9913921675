#!/usr/bin/env python3
"""bench.py — point-cloud pairs / second, forward + backward + optimizer step.

A "step" = one pass of the hot path over one batch of synthetic pairs:
predict5(training=True) -> every loss term of training_step (loss_mode 1: recovery
chamfer + comp + EMD(NxN) + CE x2 + boundary chamfer x2, EMD x4 computed as the
reference does) -> backward -> gradient all-reduce (N>1) -> Adam + StepLR.
Workload = BASELINE.json configs[1]: N=2048 points, 64 pairs per GPU, fp32.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0 (contract in the task statement) carrying `roofline`
(kNN + group stage, HBM-bound; algorithmic bytes from SURVEY §8(d)) and `cpu_baseline`
(the torch-CPU + C restatement in oracle/, kind "port", timed on the host cores).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 achievable
MFMA_F32_PEAK_TFLOPS = 157.3  # dense fp32 matrix rate (MI355X_MICROARCH.md)


class Cfg:
    dataset = "synthetic"
    loss_mode = 1
    loss_sum = False
    use_emd2 = False
    use_cd2 = False
    use_emd3 = False
    pretrain_epochs = 0
    lr = 0.9e-3
    m = "bench"
    output_path = "TRG"
    num_points = 2048


def knn_group_bytes(N, S, K, D):
    """bytes(N,S,K,D) = 12N + 4ND + 12S + 8SK + 4SK(D+3) per cloud per call."""
    return 12 * N + 4 * N * D + 12 * S + 8 * S * K + 4 * S * K * (D + 3)


def time_knn_group_api(batch, dev, reps):
    """The drop-in grouping stage (pointnet_util.sample_and_group on the encoder's two levels: neighbour search +
    materialised grouped tensor, one launch each) timed on its own on the step's clouds: the model's fused path no
    longer materialises the grouped tensor (csrc/sapoint.hip), the API does.  Returns KernelTimer records for
    `reps` step-equivalents (2 encoders x 2 levels each)."""
    from puzzlenet_amd import ops
    g = torch.Generator(device="cpu").manual_seed(7)
    clouds = [batch[0], batch[1]]
    work = []
    for xyz in clouds:
        xyz = xyz.contiguous()
        Bc, Nc, _ = xyz.shape
        for (n, s, d) in ((Nc, 512, 64), (512, 256, 128)):
            x = xyz[:, :n].contiguous()
            feat = torch.randn(Bc, n, d, generator=g).to(dev)
            new_xyz = x[:, :s].contiguous()
            idx = torch.empty((Bc, s, 32), dtype=torch.int64, device=dev)
            xg = torch.empty((Bc * s * 32, 4 + d), dtype=torch.float32, device=dev)
            work.append((x, feat, new_xyz, Bc, n, s, d, idx, xg))
    from puzzlenet_amd import _lib

    def run():
        for (x, feat, new_xyz, Bc, n, s, d, idx, xg) in work:
            try:
                ops._call("pzn_knn_group_pad_f32", ops._p(x), ops._p(feat), ops._p(new_xyz), Bc, n, s, d, ops._p(idx),
                          ops._p(xg), ops._stream())
            except _lib.PznUnsupported:      # N > 4096: search and padded group write as two launches
                ops._call("pzn_knn_f32", ops._p(x), ops._p(new_xyz), Bc, n, s, 32, ops._p(idx), ops._stream())
                ops._call("pzn_group_pad_fwd_f32", ops._p(x), ops._p(feat), ops._p(new_xyz), ops._p(idx), Bc, n, s, 32, d,
                          ops._p(xg), ops._stream())
    run()
    torch.cuda.synchronize()
    ops.KernelTimer.start()
    for _ in range(reps):
        run()
    return ops.KernelTimer.stop()


def cpu_baseline(N, pairs, iters):
    """The reference's algorithm on the host cores: oracle/model_ref.py (torch CPU ops in the
    reference's own sequence) + the C EMD restatement.  Bounded sample of the same workload."""
    from oracle import model_ref as mr
    # the GPU box gives one job a 16-core share: more threads than that only spin
    ncpu = min(len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1), 16)
    torch.set_num_threads(max(1, ncpu))
    cfg = mr.Cfg(num_points=N, loss_mode=1)
    model = mr.RefModel(cfg)
    opt = torch.optim.Adam(model.parameters(), lr=cfg.lr)
    g = torch.Generator().manual_seed(1234)
    fpc, rpc = torch.rand(pairs, N, 3, generator=g), torch.rand(pairs, N, 3, generator=g)
    x = torch.randn(pairs, 6, generator=g)
    igt = mr.se3_exp(0.8 * x / x.norm(dim=1, keepdim=True))
    mrpc = mr.se3_transform(igt, rpc.permute(0, 2, 1)).permute(0, 2, 1).contiguous()
    cd1, cd2 = mr.chamfer_loss(fpc, rpc)
    r_top, f_top = torch.topk(-cd1, 128, dim=1)[1], torch.topk(-cd2, 128, dim=1)[1]
    rpcb = torch.gather(rpc, 1, r_top.unsqueeze(-1).repeat(1, 1, 3))
    fpcb = torch.gather(fpc, 1, f_top.unsqueeze(-1).repeat(1, 1, 3))
    fi = torch.zeros(pairs, N).scatter_(1, f_top, 1.0)
    ri = torch.zeros(pairs, N).scatter_(1, r_top, 1.0)
    batch = [fpc, mrpc, igt, rpc, fpcb, rpcb, fi, ri]

    def step():
        opt.zero_grad()
        loss = model.training_step(batch)
        loss.backward()
        opt.step()

    t0 = time.perf_counter()
    step()                                   # warm-up
    print(f"[bench] cpu_baseline warm-up step {time.perf_counter() - t0:.1f}s", file=sys.stderr, flush=True)
    t0 = time.perf_counter()
    for _ in range(iters):
        step()
    dt = time.perf_counter() - t0
    return {
        "value": pairs * iters / dt, "unit": "pairs/s", "cores": torch.get_num_threads(), "kind": "port",
        "sample": f"{pairs} pairs, N={N}, 1 warm-up + {iters} timed fwd+bwd+Adam steps of oracle/model_ref.py "
                  f"(torch CPU ops on {torch.get_num_threads()} threads; EMD = single-thread C restatement)",
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=64, help="pairs per GPU")
    ap.add_argument("--points", type=int, default=2048)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--graph", action="store_true",
                    help="EXPERIMENTAL: replay the step as one HIP graph (see DESIGN.md: small memset nodes misreplay on this ROCm)")
    ap.add_argument("--cpu-pairs", type=int, default=4)
    ap.add_argument("--cpu-iters", type=int, default=2)
    args = ap.parse_args()

    from puzzlenet_amd import distributed as pdist
    rank, world, local = pdist.init_from_env()
    if world != args.gpus and rank == 0:
        print(f"[bench] warning: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
    local = local % max(1, torch.cuda.device_count())     # (rehearsals with more ranks than GPUs share a device)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    from puzzlenet_amd import _lib, engine, model5_b, ops, synthetic
    _lib.check(_lib.load().pzn_device_check(), "pzn_device_check")      # fail loudly off-gfx950

    cfg = Cfg()
    cfg.num_points = args.points
    torch.manual_seed(0)
    model = model5_b.TouchedRegraster(cfg).to(dev)
    pdist.broadcast_parameters(model)
    B, N = args.batch, args.points
    batch = synthetic.make_batch(B, N, dev, seed=1234 + rank)           # inputs resident in HBM before timing
    torch.manual_seed(1000 + rank)                                        # FPS start indices (pointnet_util.py:65)

    use_graph = bool(args.graph)
    runner = engine.TrainStep(model, batch, cfg.lr, world=world, use_graph=use_graph, warmup=max(1, args.warmup))
    if rank == 0:
        print(f"[bench] runner ready (hip graph: {use_graph})", file=sys.stderr, flush=True)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        runner.step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = runner.step()
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    loss_val = float(loss.item())
    runner.close()
    if rank == 0:
        print(f"[bench] gpu: {dt / args.steps * 1e3:.2f} ms/step, {world * args.batch * args.steps / dt:.1f} pairs/s",
              file=sys.stderr, flush=True)
    # Per-kernel pricing (roofline.achieved, stages): the same step, same shapes, same stream, with a HIP
    # event pair around every C-ABI launch.  Creating ~500 event pairs per step costs host time, so this
    # instrumented pass runs right after the timed region instead of inside it (3 steps).
    kern, prof_steps = {}, 3
    if rank == 0:
        model.two_streams = False      # price kernels one at a time (the timed loop overlaps the two encoders)
        eager = engine.TrainStep(model, batch, cfg.lr, world=1, use_graph=False)
        eager.step()
        torch.cuda.synchronize()
        ops.KernelTimer.start()
        for _ in range(prof_steps):
            eager.step()
        kern = ops.KernelTimer.stop()
        kern_flops = dict(ops.KernelTimer.flops)
        kern_api = time_knn_group_api(batch, dev, prof_steps)

    if rank == 0:
        # (1) the dominant kernel of the step: the matrix-core tile engine (csrc/gemm.hip), MFMA-bound.
        #     achieved = algorithmic 2*M*N*K of every dense entry point / their summed launch durations.
        dense_names = ("pzn_linear_fwd_f32", "pzn_linear_dgrad_f32", "pzn_linear_wgrad_f32",
                       "pzn_linear_maxpool_fwd_f32", "pzn_pooled_layer_bwd_f32", "pzn_sa_pooled_layer_bwd_f32",
                       "pzn_sharedmlp_max_fwd_f32", "pzn_sa_mlp_max_bwd_f32", "pzn_sa_mlp_max_bwd_scatter_f32",
                       "pzn_attn_fwd_f32", "pzn_attn_bwd_f32", "pzn_attn_block_fwd_f32", "pzn_attn_block_bwd_f32")
        d_ms = sum(kern.get(k, (0, 0.0))[1] for k in dense_names)
        d_fl = sum(kern_flops.get(k, 0) for k in dense_names)
        d_n = sum(kern.get(k, (0, 0.0))[0] for k in dense_names)
        mfma_achieved = d_fl / (d_ms * 1e-3) / 1e12 if d_ms > 0 else 0.0
        mfma_traffic = None      # HBM bytes per step of these kernels from the committed PMC passes (tools/pmc_summary.py)
        tpath0 = os.path.join(ROOT, "profiles", "r1_pmc_traffic.json")
        if os.path.exists(tpath0) and (B, N) == (64, 2048):
            mfma_traffic = json.load(open(tpath0)).get("mfma_family_bytes_per_step")
        roofline = {
            "bound": "mfma",
            "kernel": "the bf16x3 split-precision matrix-core kernels (fp32 result): ws_gemm_kernel (weight-stationary, forward / "
                      "input gradients of the skinny layers), df_wgrad_kernel (direct-fragment weight gradients), gemm_kernel "
                      "(general tile engine: wide layers, attention products) behind pzn_linear_* / pzn_linear_maxpool_fwd / "
                      "pzn_(sa_)pooled_layer_bwd / pzn_attn_*; the time also contains the sparse max-pool backward kernels of "
                      "pzn_(sa_)pooled_layer_bwd (vector-ALU passes, 1.6 ms per step), their flops are the 2*R*2*C1*C2 they execute; the first set-abstraction layer is counted as the per-point "
                      "product it now is (B*N rows, csrc/sapoint.hip), not as the B*S*32-row product of the reference",
            "achieved": mfma_achieved, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": mfma_achieved / MFMA_F32_PEAK_TFLOPS, "traffic": mfma_traffic,
            "algorithmic_flops_per_step": d_fl / max(1, prof_steps),
            "ms_per_step": d_ms / max(1, prof_steps), "launches_per_step": d_n / max(1, prof_steps),
            "note": "algorithmic fp32 flops / summed launch time; peak = dense fp32 MFMA rate (157.3 TFLOP/s); the skinny "
                    "K, N = 64..256 products stream 10^5..10^6 rows and sit near the HBM ridge (forward of 1M x 128 x 128 "
                    "moves 1.07 GB: 0.30 ms at 3.6 TB/s); see roofline_knn_group for the HBM-bound stage",
        }
        # (2) the stage the north star names: kNN + group (HBM-bound), SURVEY 8(d) bytes.
        per_pair = 2 * (knn_group_bytes(N, 512, 32, 64) + knn_group_bytes(512, 256, 32, 128))
        if "pzn_knn_group_pad_f32" in kern:       # kNN + group fused into one launch (PZN_SA_POINT=0: the grouped-row model path)
            n_st, ms_st = kern["pzn_knn_group_pad_f32"]
            stage_names = "knn_group_pad_kernel (pzn_knn_group_pad_f32: neighbour search + group write in one launch, 4 launches/step)"
            avg_launch = {"knn_group_pad_kernel": ms_st / max(1, n_st)}
        elif "pzn_knn_group_pad_f32" in kern_api:  # the drop-in stage, timed on its own (the model path gathers per-point rows instead)
            n_st, ms_st = kern_api["pzn_knn_group_pad_f32"]
            stage_names = ("knn_group_pad_kernel (pzn_knn_group_pad_f32: neighbour search + group write in one launch), the "
                           "sample_and_group drop-in stage timed ON ITS OWN on the step's clouds (4 launches per step-equivalent); "
                           "the training step itself no longer materialises the grouped tensor, see roofline_sa_gather")
            avg_launch = {"knn_group_pad_kernel": ms_st / max(1, n_st)}
        else:
            n_knn, ms_knn = kern_api.get("pzn_knn_f32", kern.get("pzn_knn_f32", (0, 0.0)))
            n_grp, ms_grp = kern_api.get("pzn_group_pad_fwd_f32", kern.get("pzn_group_pad_fwd_f32", kern.get("pzn_group_fwd_f32", (0, 0.0))))
            ms_st = ms_knn + ms_grp
            stage_names = "knn32_reg_kernel (pzn_knn_f32) + group_pad_direct_kernel (pzn_group_pad_fwd_f32)"
            avg_launch = {"knn32_reg_kernel": ms_knn / max(1, n_knn), "group_pad_direct_kernel": ms_grp / max(1, n_grp)}
        stage_ms_per_step = ms_st / max(1, prof_steps)
        achieved = per_pair * B / (stage_ms_per_step * 1e-3) / 1e9 if stage_ms_per_step > 0 else 0.0
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "r1_pmc_traffic.json")
        if os.path.exists(tpath) and (B, N) == (64, 2048):
            traffic = json.load(open(tpath)).get("knn_group_stage_bytes_per_step")
        roofline_knn_group = {
            "bound": "hbm", "kernel": stage_names,
            "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "traffic": traffic,
            "algorithmic_bytes_per_step": per_pair * B,
            "avg_launch_ms": avg_launch,
        }
        # (3) the HBM-bound stage of the model path: first set-abstraction layer as a gather of per-point rows.
        #     bytes per level and cloud: forward h write + P read + idx; backward dh read + dP write + inverse lists.
        def gather_bytes(n, s, c1):
            rows = s * 32
            fwd = 4 * rows * c1 + 4 * n * c1 + 8 * rows + 12 * (n + s)
            bwd = 4 * rows * c1 + 4 * n * c1 + 4 * rows + 4 * (n + 1) + 12 * (n + s)
            return fwd, bwd
        gf1, gb1 = gather_bytes(N, 512, 128)
        gf2, gb2 = gather_bytes(512, 256, 256)
        g_bytes = 2 * B * (gf1 + gb1 + gf2 + gb2)
        n_gf, ms_gf = kern.get("pzn_sa_point_l1_fwd_f32", (0, 0.0))
        n_gb, ms_gb = kern.get("pzn_sa_point_l1_bwd_f32", (0, 0.0))
        g_ms = (ms_gf + ms_gb) / max(1, prof_steps)
        g_ach = g_bytes / (g_ms * 1e-3) / 1e9 if g_ms > 0 else 0.0
        g_traffic = None
        if os.path.exists(tpath) and (B, N) == (64, 2048):
            g_traffic = json.load(open(tpath)).get("sa_gather_stage_bytes_per_step")
        roofline_sa_gather = {
            "bound": "hbm", "kernel": "sa_point_l1_fwd_kernel / sa_point_l1_bwd_kernel (pzn_sa_point_l1_{fwd,bwd}_f32: first "
                                      "set-abstraction layer as a gather of per-point rows / a sum over inverse neighbour lists)",
            "achieved": g_ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": g_ach / HBM_PEAK_GBS, "traffic": g_traffic,
            "algorithmic_bytes_per_step": g_bytes,
            "avg_launch_ms": {"sa_point_l1_fwd_kernel": ms_gf / max(1, n_gf), "sa_point_l1_bwd_kernel": ms_gb / max(1, n_gb)},
        }
        stages = {k: {"launches_per_step": n / prof_steps, "ms_per_step": ms / prof_steps} for k, (n, ms) in sorted(kern.items())}
        out = {
            "metric": "point-cloud pairs/sec (fwd+bwd) at N=2048, B=64; FPS/kNN idx bit-exact",
            "value": world * B * args.steps / dt, "unit": "pairs/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"BASELINE configs[1] shape: N={N} points, {B} pairs/GPU, fp32 train step "
                                   f"(predict5 + loss_mode 1 losses incl. 4x EMD + backward + Adam)",
                       "global_batch": world * B, "points": N, "parallelism": f"dp{world}", "hip_graph": use_graph, "encoder_streams": 2},
            "roofline": roofline,
            "roofline_knn_group": roofline_knn_group,
            "roofline_sa_gather": roofline_sa_gather,
            "stages": stages,
            "loss": loss_val,
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(N, args.cpu_pairs, args.cpu_iters)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
