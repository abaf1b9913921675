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

Prints ONE JSON line on rank 0 (contract in the task statement).  Everything a reader of the record needs is a SCALAR inside
`roofline`, `config` and `cpu_baseline` (nested tables follow as extra top-level keys):

  roofline       the DOMINANT kernel = the kernel with the largest EXCLUSIVE device time per step among those with a byte / flop
                 model (kernel_work below), timed live by the library's own per-kernel event pairs (pzn_ktimer_*, csrc/core.hip)
                 in a pass that runs the encoders one after the other, so that a launch's duration is the kernel's and not its
                 wait for the other stream's kernels.  `frac` = algorithmic work per launch / that average / peak;
                 `frac_from_profile_avg` = the same from the committed `rocprofv3 --kernel-trace --stats` summary of
                 `bench.py --one-stream` (profiles/<round>_kernel_stats_one_stream.csv); `two_stream_frac` = the same kernel
                 as the timed loop (two streams) sees it.  north_star_* = the stage BASELINE.json prices (kNN + group, HBM);
                 attention_* / emd_* / pool_bwd_* / sa_level_* = the other stages' fractions.
  config         the workload + host enqueue time + launches per step + BASELINE configs[3] / [4] at their per-GPU share
                 (n4096_b64_*, n8192_b32_*, n8192_b32_bf16_*) + the step fed from raw clouds (from_raw_*).
  cpu_baseline   oracle/model_ref.py on the host cores, with per-stage milliseconds.
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

ROUND = "r6"
HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 achievable
VALU_LANE_SLOTS_PER_S = 39.3e12   # 256 CUs x 4 SIMDs x 16 lanes x 2.4 GHz: one wave64 vector instruction per 4 cycles and SIMD
MFMA_F32_PEAK_TFLOPS = 157.3  # dense fp32-input matrix rate (v_mfma_f32_32x32x2_f32; MI355X_MICROARCH.md)
MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16 matrix rate (v_mfma_f32_32x32x16_bf16)
# Every hot matrix-core kernel issues SIX bf16 MFMAs per fp32 product (bf16x3 split precision: fp32-GEMM accuracy),
# so the ceiling of its fp32-equivalent rate is the bf16 pipe / 6:
MFMA_X3_PEAK_TFLOPS = MFMA_BF16_PEAK_TFLOPS / 6.0


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


def kernel_work(name, B, N):
    """Algorithmic work of ONE launch of a kernel that can be the dominant one, by the name a rocprofv3 trace shows:
    -> (bound, work, unit of work, peak, unit of peak, what is counted) or None (no model: latency-bound chains, the EMD walks
    priced as a stage of their own, ...).  Matrix-core kernels: 2 M N K of the products the reference's layer defines (the
    recomputed scores of the attention backward are NOT counted) against the bf16 pipe / 6 (bf16x3: six MFMAs per fp32
    product; the `<1>` instantiations of the attention kernels issue one: bf16 pipe).  HBM kernels: the bytes the kernel must
    move once (rows re-read from L2 between its own workgroups are not counted); a kernel launched once per level with
    different sizes is priced at the mean of its launches."""
    M, G1, G2 = B * 256, B * 512, B * 256
    E, dk, L = 256, 64, 256
    x3 = MFMA_BF16_PEAK_TFLOPS if name.endswith("<1>") and name.startswith("attn_") else MFMA_X3_PEAK_TFLOPS
    mf = lambda fl, what: ("mfma", float(fl), "flop", x3, "TFLOP/s", what)
    hb = lambda by, what: ("hbm", float(by), "B", HBM_PEAK_GBS, "GB/s", what)
    lv1, lv2 = (G1, 128, 128, N), (G2, 256, 256, 512)                    # (groups, C1, C2, points per cloud) of the two levels
    if name.startswith("sa_level_stream_kernel<256"):
        return mf(2 * G2 * 32 * 256 * 256, "2 x (B 256 groups x 32 rows) x 256 x 256: 2nd shared-MLP layer of level 2 + max over 32")
    if name.startswith("sa_level_stream_kernel<128"):
        return mf(2 * G1 * 32 * 128 * 128, "2 x (B 512 groups x 32 rows) x 128 x 128: 2nd shared-MLP layer of level 1 + max over 32")
    if name.startswith("outproj_maxpts_kernel"):
        return mf(2 * M * 1280 * 1024, "2 x (B 256) x 1280 x 1024: out projection of the five slices + max over the points")
    if name.startswith("attn_proj_kernel"):
        return mf(2 * M * E * (2 * dk + E), "q, k, v projections of one block, one encoder")
    if name.startswith("attn_fwd_kernel"):
        return mf(2 * M * E * E + 2 * B * L * L * (dk + E), "scores, P V, out projection of one block")
    if name.startswith("attn_bwd_q_kernel"):
        return mf(2 * M * E * (E + dk) + 2 * B * L * L * (2 * dk + E), "query side of one block's backward")
    if name.startswith("attn_bwd_k_kernel"):
        return mf(2 * M * E * (E + dk) + 2 * B * L * L * (2 * dk + 2 * E), "key side of one block's backward")
    if name.startswith("attn_wgrad_kernel"):
        return mf(2 * M * E * (2 * dk + 2 * E), "the four weight gradients of one block")
    if name.startswith("attn_wgrad_reduce_kernel"):
        rows = -(-(M // 64) // 24) * 64                                   # csrc/attnwgrad.hip: 24 row ranges of whole 64-row units
        return hb(4.0 * (640 * 256 + 640) * (-(-M // rows) + 2), "partial tiles of the row ranges read, the gradients read and written")
    if name.startswith("pool_point_kernel"):       # both levels through one instantiation: mean of the two launch sizes
        by = sum(8.0 * G * C2 + 4.0 * B * n_ * C1 for G, C1, C2, n_ in (lv1, lv2)) / 2
        return hb(by, "hit records read + per-point gradient rows written, once (mean of level 1 / 2 launches)")
    if name.startswith("pool_wgrad_kernel"):
        lv = lv1 if name.startswith("pool_wgrad_kernel<8") else lv2
        return hb(12.0 * lv[0] * lv[2] + 4.0 * lv[1] * lv[2], "pooled tensors (arg-max, out, dout) read + dW2 written, once")
    if name.startswith("stem_bwd_kernel"):
        return hb(B * N * (12 + 2 * 256), "xyz + the two output gradients read once")
    if name.startswith("stem_fwd_kernel"):
        return hb(B * N * (12 + 256), "xyz read, features written")
    if name.startswith("point_mlp3_bwd_kernel<64, 64>"):
        return hb(4.0 * B * N * (64 + 64 + 64 + 64 + 64), "head chain 64-64-64-64 backward: rows in, gradients out")
    if name.startswith("point_mlp3_fwd_kernel<64, 64>"):
        return hb(4.0 * B * N * (64 + 64 + 64 + 64), "head chain 64-64-64-64 forward")
    return None


def time_knn_group_api(batch, dev, reps):
    """The drop-in grouping stage as a caller of pointnet_util runs it: sample_and_group(npoint, 0, 32, xyz, points,
    knn=True) on the encoder's two levels of both clouds (the model's fused path does not materialise the grouped
    tensor, csrc/sapoint.hip; the API does).  Two measurements:
      * `api`: every C-ABI launch inside one pass of the four drop-in calls, each bracketed by its own event pair
        (FPS, centroid gather, search + group);
      * `stage`: the launch the north star names — search + group (pzn_knn_group_f32; ops.knn_group is literally what
        sample_and_group calls after FPS) — replayed `reps` times back to back per level and cloud between ONE event
        pair, so that the average is the kernel's own duration (what rocprofv3 reports), not duration + the idle gap
        an event pair around a single short launch adds.
    Returns (api records, [(ms per launch, launches)] per (cloud, level), cold ms of the four launches)."""
    import puzzlenet_amd.pointnet_util as pu
    from puzzlenet_amd import ops
    g = torch.Generator(device="cpu").manual_seed(7)
    work = []
    for xyz in (batch[0], batch[1]):
        xyz = xyz.contiguous()
        Bc, Nc, _ = xyz.shape
        lvl1 = torch.randn(Bc, Nc, 64, generator=g).to(dev)
        x2 = xyz[:, :512].contiguous()
        lvl2 = torch.randn(Bc, 512, 128, generator=g).to(dev)
        work.append((512, xyz, lvl1))
        work.append((256, x2, lvl2))
    with torch.no_grad():
        for (s, x, f) in work:                       # warm-up of the whole drop-in call
            pu.sample_and_group(s, 0, 32, x, f, False, True)
        torch.cuda.synchronize()
        ops.KernelTimer.start()
        for (s, x, f) in work:
            pu.sample_and_group(s, 0, 32, x, f, False, True)
        api = ops.KernelTimer.stop()
        stage, calls = [], []
        for (s, x, f) in work:
            new_xyz = ops.index_points(x, pu.farthest_point_sample(x, s)).contiguous()
            fused = ops.knn_group_supported(x, f, 32)
            call = ((lambda x=x, f=f, c=new_xyz: ops.knn_group(x, f, c)) if fused else
                    (lambda x=x, f=f, c=new_xyz: ops.group(x, f, c, ops.knn(x, c, 32))))
            calls.append(call)
            call()
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(reps):
                call()
            b.record()
            torch.cuda.synchronize()
            stage.append((a.elapsed_time(b) / reps, 1 if fused else 2))
        # the same launches from COLD caches: a back-to-back replay re-reads its 50 MB of inputs from the 256 MB
        # Infinity Cache; here 512 MB are written elsewhere first, then the FOUR launches of a step-equivalent (distinct
        # inputs: two clouds x two levels) run back to back inside ONE event pair (median of 5)
        flush = torch.empty(128 << 20, dtype=torch.float32, device=dev)
        ts = []
        for _ in range(5):
            flush.fill_(1.0)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for call in calls:
                call()
            b.record()
            torch.cuda.synchronize()
            ts.append(a.elapsed_time(b))
        cold = sorted(ts)[2]
        del flush
    return api, stage, cold


def cpu_baseline(N, pairs, iters):
    """The reference's algorithm on the host cores: oracle/model_ref.py (torch CPU ops in the reference's own sequence) + the
    C EMD restatement.  Bounded sample of the same workload; per-stage milliseconds per step from the oracle's own stage
    clock (forward by stage, EMD forward + backward, everything else of the backward as one figure)."""
    from oracle import model_ref as mr
    # the GPU box gives one job a 16-core share: more threads than that only spin (BASELINE.md section 4 says os.cpu_count())
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
    clock = {"backward": 0.0, "adam": 0.0}

    def step():
        opt.zero_grad()
        loss = model.training_step(batch)
        t0 = time.perf_counter()
        loss.backward()
        t1 = time.perf_counter()
        opt.step()
        clock["backward"] += t1 - t0
        clock["adam"] += time.perf_counter() - t1

    t0 = time.perf_counter()
    step()                                   # warm-up
    print(f"[bench] cpu_baseline warm-up step {time.perf_counter() - t0:.1f}s", file=sys.stderr, flush=True)
    clock.update(backward=0.0, adam=0.0)
    mr.STAGE_CLOCK = {}
    t0 = time.perf_counter()
    for _ in range(iters):
        step()
    dt = time.perf_counter() - t0
    st, mr.STAGE_CLOCK = mr.STAGE_CLOCK, None
    ms = lambda k: 1e3 * st.get(k, 0.0) / iters
    out = {
        "value": pairs * iters / dt, "unit": "pairs/s", "cores": torch.get_num_threads(), "kind": "port",
        "sample": f"{pairs} pairs N={N}, 1 warm-up + {iters} timed fwd+bwd+Adam steps of oracle/model_ref.py, EMD = 1-thread C",
        "ms_per_step": 1e3 * dt / iters,
        # forward, by stage (ms per step of `pairs` pairs): FPS loop; square_distance + argsort; gather + concat; per-point stem,
        # shared MLPs, out projection; the four attention blocks; pose + boundary heads; chamfer x4; loss_tail = the rest of the
        # forward (se3, cross entropies, top-k, sums)
        "fps_ms": ms("fps"), "knn_ms": ms("knn"), "group_ms": ms("group"), "mlp_ms": ms("mlp"), "attention_ms": ms("attention"),
        "heads_ms": ms("heads"), "chamfer_ms": ms("chamfer"),
        "emd_ms": ms("emd") + ms("emd_backward"),       # forward (approxmatch + matchcost) + backward (two gradient kernels)
        "loss_tail_ms": ms("forward_total") - sum(ms(k) for k in ("fps", "knn", "group", "mlp", "attention", "heads", "chamfer", "emd")),
        "backward_ms": 1e3 * clock["backward"] / iters - ms("emd_backward"),     # torch autograd of everything but EMD
        "adam_ms": 1e3 * clock["adam"] / iters,
        "threads_note": "min(cores of this job, 16): a GPU box gives one job a 16-core share",
    }
    return out


def other_workload(dev, B, N, attn, warm=5, steps=10):
    """BASELINE configs[3] / [4] at their per-GPU share, in this process after the headline's timed region: `warm` + `steps`
    optimiser steps of the same runner, then two instrumented steps (EMD time, attention time) and the north-star stage on this
    workload's clouds.  -> dict of scalars"""
    from puzzlenet_amd import _lib, engine, model5_b, ops, synthetic
    lib = _lib.load()
    if attn == "bf16":
        _lib.check(lib.pzn_attn_set_precision(1), "pzn_attn_set_precision")
    try:
        cfg = Cfg()
        cfg.num_points = N
        torch.manual_seed(0)
        model = model5_b.TouchedRegraster(cfg).to(dev)
        batch = synthetic.make_batch(B, N, dev, seed=1234)
        torch.manual_seed(1000)
        runner = engine.TrainStep(model, batch, cfg.lr, world=1)
        for _ in range(warm):
            runner.step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            runner.step()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        runner.close()
        model.two_streams = False
        eager = engine.TrainStep(model, batch, cfg.lr, world=1)
        eager.step()
        torch.cuda.synchronize()
        ops.KernelTimer.start()
        for _ in range(2):
            eager.step()
        kern = ops.KernelTimer.stop()
        flops = dict(ops.KernelTimer.flops)
        eager.close()
        _, stage, cold = time_knn_group_api(batch, dev, 10)
        per_pair = 2 * (knn_group_bytes(N, 512, 32, 64) + knn_group_bytes(512, 256, 32, 128))
        ms_kg = sum(ms for ms, _ in stage)
        at = [k for k in kern if k.startswith("pzn_attn_fused_")]
        at_ms = sum(kern[k][1] for k in at) / 2
        at_fl = sum(flops.get(k, 0) for k in at) / 2
        at_peak = MFMA_BF16_PEAK_TFLOPS if attn == "bf16" else MFMA_X3_PEAK_TFLOPS
        return {
            "pairs_per_s": B * steps / dt, "ms_per_step": dt / steps * 1e3,
            "knn_group_frac": per_pair * B / (ms_kg * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "knn_group_cold_frac": per_pair * B / (cold * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "emd_ms": kern.get("pzn_emd_fused_f32", (0, 0.0))[1] / 2,
            "attention_ms": at_ms,
            "attention_frac": (at_fl / (at_ms * 1e-3) / 1e12 / at_peak) if at_ms > 0 else None,
        }
    finally:
        if attn == "bf16":
            _lib.check(lib.pzn_attn_set_precision(0), "pzn_attn_set_precision")


def from_raw(dev, B, N, raw_points=10000, warm=5, steps=10, blocks=2):
    """SURVEY 8 f2 end to end: a FRESH batch every step, built on the GPU from raw clouds (plane cut, FPS to N, boundary
    labels, random motion: datapipe.PairFeeder, the reference's 64 loader processes train.py:101-104 + dataset.py:1165-1190)
    on a background stream while the previous step trains.  The step's sparse backward kernels move with the training state
    (DESIGN.md section 0), so the resident-batch figure the ratio is taken against is measured HERE, on the same model, in
    blocks of `steps` steps that alternate with the fed ones.  -> (pairs/s fed, ms per fed step, ms per resident step)"""
    import numpy as np
    from puzzlenet_amd import datapipe, engine, model5_b
    rng = np.random.RandomState(0)
    u = rng.randn(B, raw_points, 3).astype(np.float32)
    u /= np.linalg.norm(u, axis=2, keepdims=True)
    raw_h = (u * (0.25 + 0.2 * rng.rand(B, 1, 3).astype(np.float32))).astype(np.float32)   # ellipsoid shells, one per sample
    cfg = Cfg()
    cfg.num_points = N
    torch.manual_seed(0)
    model = model5_b.TouchedRegraster(cfg).to(dev)
    feeder = datapipe.PairFeeder(raw_h, dev, n=N, seed=0)
    fed = res = 0.0
    try:
        runner = engine.TrainStep(model, feeder.next_batch(), cfg.lr, world=1)
        nxt = feeder.next_batch()
        for _ in range(warm):
            runner.step(next_batch=nxt)
            nxt = feeder.next_batch()
        for _ in range(blocks):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                runner.step(next_batch=nxt)
                nxt = feeder.next_batch()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(steps):          # the batch adopted last stays resident
                runner.step()
            torch.cuda.synchronize()
            fed, res = fed + (t1 - t0), res + (time.perf_counter() - t1)
        runner.close()
    finally:
        feeder.close()
    n = blocks * steps
    return B * n / fed, fed / n * 1e3, res / n * 1e3


def build_id():
    """Hash of the sources libpzn.so is built from (csrc/*, include/pzn.h, build.py): PMC byte counts collected on another build
    are not attached to this run's numbers."""
    import hashlib
    h = hashlib.sha256()
    base = os.path.join(ROOT, "puzzlenet_amd", "csrc")
    for f in sorted(os.listdir(base)):
        if f.endswith((".hip", ".h")):
            h.update(open(os.path.join(base, f), "rb").read())
    h.update(open(os.path.join(ROOT, "include", "pzn.h"), "rb").read())
    h.update(open(os.path.join(ROOT, "puzzlenet_amd", "build.py"), "rb").read())      # (compiler flags)
    return h.hexdigest()[:16]


def pmc_traffic(key, B, N):
    """HBM bytes per step of a kernel / stage from the committed PMC passes (tools/pmc_summary.py, collected with
    `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` in separate runs of this same command): only when they were taken
    on THIS build and on this workload; otherwise null.  -> (bytes or None, provenance string)."""
    name = f"{ROUND}_pmc_traffic.json"
    path = os.path.join(ROOT, "profiles", name)
    if not os.path.exists(path):
        return None, "no PMC summary committed for this round"
    doc = json.load(open(path))
    if doc.get("build_id") != build_id():
        return None, f"profiles/{name}: build {doc.get('build_id')}, this is {build_id()}"
    if (doc.get("batch"), doc.get("points")) != (B, N):
        return None, f"profiles/{name}: another workload"
    return doc.get(key), f"profiles/{name} (separate --pmc passes)"


def profile_rows(which, B, N, attn):
    """Rows of a committed `rocprofv3 --kernel-trace --stats` summary — which = "kernel_stats" (the default command: two streams)
    or "kernel_stats_one_stream" (`bench.py --one-stream`) — used only when its sidecar (.meta.json: build id, batch, points,
    attention mode) matches this run.  -> (list of {name, calls, avg_us, pct} or None, why not)"""
    import csv
    base = os.path.join(ROOT, "profiles", f"{ROUND}_{which}")
    if not os.path.exists(base + ".csv") or not os.path.exists(base + ".meta.json"):
        return None, f"no profiles/{ROUND}_{which}.csv with its .meta.json"
    doc = json.load(open(base + ".meta.json"))
    want = {"build_id": build_id(), "batch": B, "points": N, "attn": attn}
    diff = [f"{k} {doc.get(k)} != {v}" for k, v in want.items() if doc.get(k) != v]
    if diff:
        return None, f"profiles/{ROUND}_{which}.csv is of another build / workload: " + ", ".join(diff)
    with open(base + ".csv", newline="") as f:
        rows = [{"name": r["Name"].replace("(anonymous namespace)::", "").replace("void ", ""), "calls": int(r["Calls"]),
                 "avg_us": float(r["AverageNs"]) / 1e3, "pct": float(r["Percentage"])} for r in csv.DictReader(f)]
    return rows, ""


def profile_avg_us(rows, kernel):
    """Call-weighted average duration of the rows whose kernel name starts with `kernel`."""
    hit = [r for r in (rows or []) if r["name"].startswith(kernel)]
    calls = sum(r["calls"] for r in hit)
    return (sum(r["avg_us"] * r["calls"] for r in hit) / calls) if calls else None


def kernel_pass(model, batch, lr, two_streams, steps):
    """`steps` training steps with the library's per-kernel timer on -> {kernel: (launches per step, ms per step)}."""
    from puzzlenet_amd import engine, ops
    model.two_streams = two_streams
    r = engine.TrainStep(model, batch, lr, world=1)
    r.step()
    torch.cuda.synchronize()
    ops.ktimer_start()
    for _ in range(steps):
        r.step()
    rows = ops.ktimer_stop()
    r.close()
    return {k: (n / steps, ms / steps) for k, (n, ms) in rows.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=64, help="pairs per GPU")
    ap.add_argument("--points", type=int, default=2048)
    ap.add_argument("--attn", choices=("f32", "bf16"), default="f32",
                    help="attention products: f32 = the default split-precision path (fp32 results), bf16 = single bf16 "
                         "MFMAs with fp32 softmax / accumulation (BASELINE configs[4])")
    ap.add_argument("--one-stream", action="store_true",
                    help="the two encoders one after the other on ONE stream in the timed loop too: what the committed "
                         "one-stream kernel summary is collected with (a kernel's duration is then its own)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-workloads", action="store_true",
                    help="skip BASELINE configs[3] / [4] at their per-GPU share (5 + 10 steps each) and the from-raw steps")
    ap.add_argument("--cpu-pairs", type=int, default=8)
    ap.add_argument("--cpu-iters", type=int, default=3)
    args = ap.parse_args()

    from puzzlenet_amd import distributed as pdist
    rank, world, local = pdist.init_from_env()        # selects the rank's device before the process group exists
    if world != args.gpus and rank == 0:
        print(f"[bench] warning: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
    dev = torch.device("cuda", torch.cuda.current_device())

    from puzzlenet_amd import _lib, engine, model5_b, ops, synthetic
    _lib.check(_lib.load().pzn_device_check(), "pzn_device_check")      # fail loudly off-gfx950
    if args.attn == "bf16":
        _lib.check(_lib.load().pzn_attn_set_precision(1), "pzn_attn_set_precision")

    cfg = Cfg()
    cfg.num_points = args.points
    torch.manual_seed(0)
    model = model5_b.TouchedRegraster(cfg).to(dev)
    if args.one_stream:
        model.two_streams = False
    pdist.broadcast_parameters(model)
    B, N = args.batch, args.points
    batch = synthetic.make_batch(B, N, dev, seed=1234 + rank)           # inputs resident in HBM before timing
    torch.manual_seed(1000 + rank)                                        # FPS start indices (pointnet_util.py:65)

    runner = engine.TrainStep(model, batch, cfg.lr, world=world)
    if rank == 0:
        print("[bench] runner ready", file=sys.stderr, flush=True)

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
    t_enq = time.perf_counter() - t0      # the host has enqueued the last step (nothing in a step synchronises)
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
    if rank != 0:
        dist.barrier()
        dist.destroy_process_group()
        return

    # ------------------------------------------------------------------ instrumented passes (rank 0, after the timed region)
    # (a) entry points, one stream: a HIP event pair around every C-ABI call -> `stages`, flops per entry point, EMD counters
    prof_steps = 3
    model.two_streams = False
    eager = engine.TrainStep(model, batch, cfg.lr, world=1)
    eager.step()
    torch.cuda.synchronize()
    ops.KernelTimer.start()
    ops.EMD_WALK_STATS = []
    for _ in range(prof_steps):
        eager.step()
    kern = ops.KernelTimer.stop()
    emd_walk, ops.EMD_WALK_STATS = ops.EMD_WALK_STATS, None
    kern_flops = dict(ops.KernelTimer.flops)
    eager.close()
    # (b) kernels, one stream (exclusive times) and (c) two streams (as in the timed loop): the library's own per-kernel timer
    kt1 = kernel_pass(model, batch, cfg.lr, False, prof_steps)
    kt2 = kernel_pass(model, batch, cfg.lr, True, prof_steps)
    model.two_streams = not args.one_stream
    kern_api, stage_kg, cold_kg = time_knn_group_api(batch, dev, 20)

    def per_step(name):
        n, ms = kern.get(name, (0, 0.0))
        return n / prof_steps, ms / prof_steps

    # ------------------------------------------------------------------ roofline: the dominant kernel by exclusive time
    def work_of(name, launches=None):
        return kernel_work(name, B, N)

    cands = {}
    for name, (n, ms) in kt1.items():
        wk = work_of(name, n)
        if wk is not None and n > 0:
            cands[name] = (n, ms, wk)
    rows1, why1 = profile_rows("kernel_stats_one_stream", B, N, args.attn)
    rows2, why2 = profile_rows("kernel_stats", B, N, args.attn)
    # Which one is dominant?  Two kernels are within a percent of each other here (the level-2 max-pool kernel, 2 x 282 us, and
    # the key side of the attention backward, 8 x 71 us), so a live choice flips between runs.  With a committed one-stream
    # summary of THIS build the choice is read from it (largest total time among the kernels with a work model): every run of
    # this build then prices the same kernel, and the live average is checked against the file's.  Without one: live.
    how = "largest exclusive device time per step among kernels with a work model (this run's kernel timer)"
    if not cands:
        raise SystemExit("bench: the kernel timer saw no kernel with a work model (kernel_work): nothing to price")
    dom = max(cands, key=lambda k: cands[k][1])
    if rows1:
        tot = {}
        for r_ in rows1:
            k_ = next((c for c in cands if r_["name"].startswith(c)), None)
            if k_ is not None:
                tot[k_] = tot.get(k_, 0.0) + r_["avg_us"] * r_["calls"]
        if tot:
            dom = max(tot, key=tot.get)
            how = f"largest total time in profiles/{ROUND}_kernel_stats_one_stream.csv among kernels with a work model"
    dn, dms, (dbound, dwork, dunit, dpeak, dpeak_unit, dwhat) = cands[dom]
    davg_us = 1e3 * dms / dn
    scale = 1e12 if dbound == "mfma" else 1e9
    dach = dwork / (davg_us * 1e-6) / scale
    pavg1 = profile_avg_us(rows1, dom)
    pavg2 = profile_avg_us(rows2, dom)
    n2, ms2 = kt2.get(dom, (0, 0.0))
    two_avg_us = 1e3 * ms2 / n2 if n2 else None
    pk, pk_src = pmc_traffic("per_kernel", B, N)
    dom_traffic = None
    if pk:
        hit = [v for k_, v in pk.items() if k_.startswith(dom)]
        if hit:
            dom_traffic = sum(v["hbm_bytes_per_launch"] * v["launches"] for v in hit) / sum(v["launches"] for v in hit)
    ranked = sorted(kt1.items(), key=lambda kv: -kv[1][1])
    roofline = {
        "bound": dbound, "kernel": dom, "achieved": dach, "peak": dpeak, "unit": dpeak_unit, "frac": dach / dpeak,
        "traffic": dom_traffic,
        "what_is_counted": dwhat[:118],
        "how_chosen": how[:118],
        "timed": "library event pair around every launch of the kernel (pzn_ktimer), encoders one after the other",
        "work_per_launch": dwork, "work_unit": dunit, "avg_launch_us": davg_us, "launches_per_step": dn, "ms_per_step": dms,
        "two_stream_avg_launch_us": two_avg_us,
        "two_stream_frac": (dwork / (two_avg_us * 1e-6) / scale / dpeak) if two_avg_us else None,
        "profile_avg_launch_us": pavg1,
        "frac_from_profile_avg": (dwork / (pavg1 * 1e-6) / scale / dpeak) if pavg1 else None,
        "profile_file": f"profiles/{ROUND}_kernel_stats_one_stream.csv" if pavg1 else why1[:118],
        "two_stream_profile_avg_launch_us": pavg2,
        "two_stream_frac_from_profile_avg": (dwork / (pavg2 * 1e-6) / scale / dpeak) if pavg2 else None,
        "kernel_ms_per_step_one_stream": sum(ms for _, ms in kt1.values()),
        "kernel_ms_per_step_two_streams": sum(ms for _, ms in kt2.values()),
    }
    for i, (name, (n, ms)) in enumerate(ranked[:8], 1):      # the ranking the choice was made from, as scalars
        wk = work_of(name, n)
        fr = ""
        if wk is not None:
            fr = f" frac {wk[1] / (1e3 * ms / n * 1e-6) / (1e12 if wk[0] == 'mfma' else 1e9) / wk[3]:.3f} ({wk[0]})"
        roofline[f"top{i}"] = f"{name[:60]}: {ms:.3f} ms/step, {n:.0f} x {1e3 * ms / n:.1f} us{fr}"

    # ------------------------------------------------------------------ the north-star stage: kNN + group (HBM), SURVEY 8(d) bytes
    per_pair = 2 * (knn_group_bytes(N, 512, 32, 64) + knn_group_bytes(512, 256, 32, 128))
    ms_kg = sum(ms for ms, _ in stage_kg)                 # one step-equivalent: 2 clouds x 2 levels
    n_kg = sum(k for _, k in stage_kg)
    kg_traffic, kg_src = pmc_traffic("knn_group_stage_bytes_per_step", B, N)
    kg_ach = per_pair * B / (ms_kg * 1e-3) / 1e9 if ms_kg > 0 else 0.0
    kg_cold = per_pair * B / (cold_kg * 1e-3) / 1e9 if cold_kg > 0 else 0.0
    roofline.update({
        "north_star_stage": "kNN + group = pzn_knn_group_f32, what pointnet_util.sample_and_group(.., knn=True) launches after FPS",
        "north_star_frac": kg_ach / HBM_PEAK_GBS, "north_star_cold_frac": kg_cold / HBM_PEAK_GBS,
        "north_star_ms": ms_kg, "north_star_cold_ms": cold_kg, "north_star_bytes": per_pair * B,
        "north_star_traffic": kg_traffic, "north_star_launches": n_kg,
    })
    n_mk, ms_mk = per_step("pzn_knn_f32")
    roofline_knn_group = {
        "bound": "hbm",
        "kernel": "knn_select_kernel<R, 8, true, D, kp, nt> behind pzn_knn_group_f32 (search + reference-layout [B,S,32,3+D] group "
                  "write in one launch), 4 launches per step-equivalent (2 clouds x 2 levels), each replayed 20x back to back",
        "achieved": kg_ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": kg_ach / HBM_PEAK_GBS,
        "traffic": kg_traffic, "traffic_source": kg_src,
        "algorithmic_bytes_per_step": per_pair * B, "ms_per_step": ms_kg, "launches_per_step": n_kg,
        "launch_ms": {"cloud0_level1": stage_kg[0][0], "cloud0_level2": stage_kg[1][0], "cloud1_level1": stage_kg[2][0],
                      "cloud1_level2": stage_kg[3][0]},
        "cold": {"what": "the same four launches back to back inside ONE event pair right after 512 MB were written elsewhere, "
                         "median of 5", "ms_per_step": cold_kg, "achieved": kg_cold, "frac": kg_cold / HBM_PEAK_GBS},
        "model_path_knn": {"entry": "pzn_knn_f32 (indices only; the encoder gathers per-point rows instead of materialising "
                                    "groups)", "launches_per_step": n_mk, "ms_per_step": ms_mk},
    }

    # ------------------------------------------------------------------ the other stages (fractions as scalars in `roofline`)
    # set-abstraction forward: the generated-row max-pool level (4 launches per step)
    sa_ms = sum(ms for k, (n, ms) in kt1.items() if k.startswith("sa_level_stream_kernel"))
    sa_fl = sum(n * work_of(k, n)[1] for k, (n, ms) in kt1.items() if k.startswith("sa_level_stream_kernel"))
    sa_traffic, _ = pmc_traffic("ws_gemm_maxpool_bytes_per_step", B, N)
    roofline.update({"sa_level_ms": sa_ms, "sa_level_frac": (sa_fl / (sa_ms * 1e-3) / 1e12 / MFMA_X3_PEAK_TFLOPS) if sa_ms else None,
                     "sa_level_traffic": sa_traffic})
    # attention: every pzn_attn_fused_* entry point, algorithmic flops
    at_names = [k for k in kern if k.startswith("pzn_attn_fused_")]
    ms_at = sum(per_step(k)[1] for k in at_names)
    fl_at = sum(kern_flops.get(k, 0) for k in at_names) / prof_steps
    attn_peak = MFMA_BF16_PEAK_TFLOPS if args.attn == "bf16" else MFMA_X3_PEAK_TFLOPS
    roofline.update({"attention_ms": ms_at, "attention_frac": (fl_at / (ms_at * 1e-3) / 1e12 / attn_peak) if ms_at else None,
                     "attention_launches": sum(per_step(k)[0] for k in at_names)})
    # all dense matrix-core entry points together
    dense = [k for k in kern if kern_flops.get(k, 0) > 0]
    d_ms = sum(per_step(k)[1] for k in dense)
    d_fl = sum(kern_flops[k] for k in dense) / prof_steps
    roofline.update({"mfma_family_ms": d_ms, "mfma_family_frac": (d_fl / (d_ms * 1e-3) / 1e12 / MFMA_X3_PEAK_TFLOPS) if d_ms else None})
    # EMD: vector-ALU / transcendental bound.  The reference's schedule is 10 levels x 3 passes x n*m pair evaluations
    # (emd_kernel.cu:46-154); the fused path walks only the cloud-2 points that still hold mass inside the level's x window and
    # counts what it evaluates (uint64 device counters in units of 64 evaluations).  Issue slots per counted evaluation
    # (csrc/emd.hip, measured issue costs tools/valu_rate.hip: v_pk_* 4.4 cycles, v_exp_f32 8): pass B 12 packed + 2 exp per TWO
    # evaluations, pass C + next A 16 + 2 -> (12 + 4 + 16 + 4) / 2 / 2 = 9 (`emd_frac`); round 4 priced 7.25 (`emd_frac_r4_pricing`).
    n_e, ms_e = per_step("pzn_emd_fused_f32")
    ev_exec = ev_ref = 0.0
    for ctr, eb, en, em in emd_walk:
        ev_ref += 30.0 * eb * en * em
        ev_exec += 30.0 * eb * en * em if ctr is None else 64.0 * float(ctr.sum().item())
    ev_exec, ev_ref = ev_exec / prof_steps, ev_ref / prof_steps
    lane_peak = VALU_LANE_SLOTS_PER_S / 1e12
    e_rate = ev_exec / (ms_e * 1e-3) / 1e12 if ms_e > 0 else 0.0
    roofline.update({"emd_ms": ms_e, "emd_frac": 9.0 * e_rate / lane_peak, "emd_frac_r4_pricing": 7.25 * e_rate / lane_peak,
                     "emd_evaluations_per_step": ev_exec, "emd_reference_schedule_over_executed": ev_ref / max(ev_exec, 1.0),
                     "emd_launches": n_e})
    roofline_emd = {
        "bound": "valu", "kernel": "emdf_b_kernel / emdf_k_kernel<0|1|2> (list compaction = device function emdf_compact_wg inside "
                                   "emdf_k_kernel<1>) / emd_sort_x_kernel / emd_small_fused_kernel behind pzn_emd_fused_f32",
        "achieved": 9.0 * e_rate, "peak": lane_peak, "unit": "T lane-slot/s", "frac": 9.0 * e_rate / lane_peak,
        "traffic": pmc_traffic("emd_bytes_per_step", B, N)[0],
        "pair_evaluations_executed_per_step": ev_exec, "pair_evaluations_reference_schedule_per_step": ev_ref,
        "issue_slots_per_evaluation": 9.0, "ms_per_step": ms_e, "launches_per_step": n_e,
        "kernels_ms_per_step": {k: ms for k, (n, ms) in ranked if k.startswith(("emdf_", "emd_"))},
    }
    # pooled layers' backward (csrc/sapool.hip + poolbwd.hip): bytes it must move / lanes it must issue
    n_pb, ms_pb = per_step("pzn_sa_level_bwd_pt_f32")
    lvl = ((B * 512, 128, 128, N), (B * 256, 256, 256, 512))              # (groups R, C1, C2, points per cloud)
    pb_bytes = 2 * sum(2 * 12.0 * R_ * C2_ + 2 * 8.0 * R_ * C2_ + 4.0 * B * n_ * C1_ + 4.0 * C1_ * C2_ for R_, C1_, C2_, n_ in lvl)
    # (round 5 priced more bytes: hit lists read once per 128-column slice, dP written twice - zero fill + rows; kept beside the
    # new figure so that the two rounds compare like for like)
    pb_bytes_r5 = 2 * sum(2 * 12.0 * R_ * C2_ + 8.0 * R_ * C2_ * (1 + C1_ // 128) + 2 * 4.0 * B * n_ * C1_ + 4.0 * C1_ * C2_
                          for R_, C1_, C2_, n_ in lvl)
    pb_fma = 2 * sum(2.0 * R_ * C2_ * C1_ for R_, C1_, C2_, n_ in lvl)         # input-gradient + weight-gradient pass
    pb_traffic = pmc_traffic("pool_bwd_stage_bytes_per_step", B, N)[0]
    roofline.update({"pool_bwd_ms": ms_pb, "pool_bwd_frac": (pb_bytes / (ms_pb * 1e-3) / 1e9 / HBM_PEAK_GBS) if ms_pb else None,
                     "pool_bwd_vector_issue_frac": (pb_fma / (ms_pb * 1e-3) / VALU_LANE_SLOTS_PER_S) if ms_pb else None,
                     "pool_bwd_bytes": pb_bytes, "pool_bwd_bytes_r5_accounting": pb_bytes_r5,
                     "pool_bwd_frac_r5_accounting": (pb_bytes_r5 / (ms_pb * 1e-3) / 1e9 / HBM_PEAK_GBS) if ms_pb else None,
                     "pool_bwd_traffic_over_r5_accounting": (pb_traffic / pb_bytes_r5) if pb_traffic else None,
                     "pool_bwd_traffic": pb_traffic,
                     "pool_bwd_traffic_over_algorithmic": (pb_traffic / pb_bytes) if pb_traffic else None})

    stages = {k: {"launches_per_step": n / prof_steps, "ms_per_step": ms / prof_steps} for k, (n, ms) in sorted(kern.items())}
    stages_api = {k: {"launches": n, "ms": ms} for k, (n, ms) in sorted(kern_api.items())}
    cfg_name = 1 if (B, N) == (64, 2048) else ('4' if N == 8192 else '3' if N == 4096 else '-')
    out = {
        "metric": f"point-cloud pairs/sec (fwd+bwd) at N={N}, B={B}; FPS/kNN idx bit-exact",
        "value": world * B * args.steps / dt, "unit": "pairs/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32" if args.attn == "f32" else "f32 (attention contractions bf16)",
        "data": "synthetic",
        "config": {"workload": f"BASELINE configs[{cfg_name}]: N={N}, {B} pairs/GPU, fp32 train step (predict5 + 4x EMD + bwd + Adam)",
                   "global_batch": world * B, "points": N, "parallelism": f"dp{world}",
                   "encoder_streams": 1 if args.one_stream else 2,
                   "matrix_core_path": "bf16x3 split precision (fp32 results; six bf16 MFMAs per product)",
                   "attention": args.attn, "build_id": build_id(),
                   # host time to ENQUEUE a step against the step itself: close to 1 = this box's host is the bottleneck
                   "host_enqueue_ms_per_step": t_enq / args.steps * 1e3, "host_enqueue_over_step": t_enq / dt,
                   "library_launches_per_step": sum(n for n, _ in kt1.values())},
        "roofline": roofline,
        "roofline_knn_group": roofline_knn_group,
        "roofline_emd": roofline_emd,
        "kernels_one_stream": {k: {"launches_per_step": n, "ms_per_step": ms} for k, (n, ms) in ranked},
        "kernels_two_streams": {k: {"launches_per_step": n, "ms_per_step": ms} for k, (n, ms) in sorted(kt2.items(), key=lambda kv: -kv[1][1])},
        "stages": stages,
        "stages_sample_and_group_dropin": stages_api,
        "loss": loss_val,
    }
    if not args.no_other_workloads and world == 1 and (B, N, args.attn) == (64, 2048, "f32") and not args.one_stream:
        # BASELINE configs[3] and [4] (1-GPU shares) and the step fed from raw clouds, measured by whoever runs this line:
        # same process, after the headline
        del model, batch, runner, eager
        torch.cuda.empty_cache()
        for tag, (b_, n_, at_) in (("n4096_b64", (64, 4096, "f32")), ("n8192_b32", (32, 8192, "f32")),
                                  ("n8192_b32_bf16", (32, 8192, "bf16"))):
            r_ = other_workload(dev, b_, n_, at_)
            for k_, v_ in r_.items():
                out["config"][f"{tag}_{k_}"] = v_
            torch.cuda.empty_cache()
        pps, ms_, res_ = from_raw(dev, B, N)
        out["config"].update({"from_raw_pairs_per_s": pps, "from_raw_ms_per_step": ms_, "from_raw_resident_ms_per_step": res_,
                              "from_raw_over_resident": res_ / ms_,
                              "from_raw_over_value": pps / out["value"],
                              "from_raw_what": "fresh 64 pairs per step cut + sampled from 10000-point raw clouds on a background stream; "
                                               "over_resident: against resident-batch steps of the same model in alternating blocks"})
    if not args.no_cpu_baseline and world == 1:
        out["cpu_baseline"] = cpu_baseline(N, args.cpu_pairs, args.cpu_iters)
    print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
