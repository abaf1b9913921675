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

Prints ONE JSON line on rank 0 (contract in the task statement) carrying `roofline` = the DOMINANT kernel of the step
(the first row of the committed rocprofv3 kernel summary, profiles/<round>_kernel_stats.csv, that a single-kernel entry
point launches - a chained attention kernel, the attention weight-gradient kernel, the streamed set-abstraction level or
the out projection; without a committed summary the largest launch-time sum of those in the instrumented pass; flops,
launches and average launch time printed, the committed rocprofv3 average beside the live one), `roofline_sa_level` (the streamed generated-row max-pool kernel, the
`roofline` of rounds 2-3), `roofline_knn_group` (the stage the north star names: the drop-in sample_and_group's search +
group launch, HBM-bound, SURVEY 8(d) bytes), further per-stage rooflines, and `cpu_baseline` (the torch-CPU + C
restatement in oracle/, kind "port", on the host cores).
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
    Returns (api records, [(ms per launch, launches)] per (cloud, level), launches per step-equivalent)."""
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
        stage = []
        for (s, x, f) in work:
            new_xyz = ops.index_points(x, pu.farthest_point_sample(x, s)).contiguous()
            fused = ops.knn_group_supported(x, f, 32)
            call = (lambda: ops.knn_group(x, f, new_xyz)) if fused else (lambda: ops.group(x, f, new_xyz, ops.knn(x, new_xyz, 32)))
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
        # inputs: two clouds x two levels) run back to back inside ONE event pair — no launch re-reads what another left
        # in a cache, and no per-launch start-up gap is counted (median of 5)
        calls = []
        for (s, x, f) in work:
            new_xyz = ops.index_points(x, pu.farthest_point_sample(x, s)).contiguous()
            fused = ops.knn_group_supported(x, f, 32)
            calls.append((lambda x=x, f=f, c=new_xyz: ops.knn_group(x, f, c)) if fused else
                         (lambda x=x, f=f, c=new_xyz: ops.group(x, f, c, ops.knn(x, c, 32))))
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


def other_workload(dev, B, N, attn, warm=5, steps=10):
    """BASELINE configs[3] / [4] at their per-GPU share, in this process after the headline's timed region: `warm` + `steps`
    optimiser steps of the same runner, then two instrumented steps (EMD time) and the north-star stage on this workload's
    clouds.  -> {workload, pairs_per_s, ms_per_step, knn_group_frac, knn_group_cold_frac, emd_ms}"""
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
        eager.close()
        _, stage, cold = time_knn_group_api(batch, dev, 10)
        per_pair = 2 * (knn_group_bytes(N, 512, 32, 64) + knn_group_bytes(512, 256, 32, 128))
        ms_kg = sum(ms for ms, _ in stage)
        return {
            "workload": f"N={N} points, {B} pairs/GPU, fp32 train step" + (", attention contractions bf16" if attn == "bf16" else ""),
            "points": N, "batch": B, "attention": attn, "steps": steps, "warmup": warm,
            "pairs_per_s": B * steps / dt, "ms_per_step": dt / steps * 1e3,
            "knn_group_frac": per_pair * B / (ms_kg * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "knn_group_cold_frac": per_pair * B / (cold * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "emd_ms": kern.get("pzn_emd_fused_f32", (0, 0.0))[1] / 2,
        }
    finally:
        if attn == "bf16":
            _lib.check(lib.pzn_attn_set_precision(0), "pzn_attn_set_precision")


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


ROUND = "r5"


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
        return None, f"profiles/{name} was collected on build {doc.get('build_id')}, this is {build_id()}"
    if (doc.get("batch"), doc.get("points")) != (B, N):
        return None, f"profiles/{name} was collected on another workload"
    return doc.get(key), f"profiles/{name} (build {doc.get('build_id')}, separate --pmc passes of bench.py)"


def profile_matches(B, N, attn):
    """The committed kernel summary is used only when it was collected on THIS build and workload: tools/collect_profiles.sh
    writes profiles/<round>_kernel_stats.meta.json = {build_id, batch, points, attn} beside it.  -> (ok, why not)"""
    meta = os.path.join(ROOT, "profiles", f"{ROUND}_kernel_stats.meta.json")
    if not os.path.exists(os.path.join(ROOT, "profiles", f"{ROUND}_kernel_stats.csv")) or not os.path.exists(meta):
        return False, f"no profiles/{ROUND}_kernel_stats.csv with its .meta.json committed"
    doc = json.load(open(meta))
    want = {"build_id": build_id(), "batch": B, "points": N, "attn": attn}
    diff = {k: (doc.get(k), v) for k, v in want.items() if doc.get(k) != v}
    if diff:
        return False, f"profiles/{ROUND}_kernel_stats.csv was collected on another build / workload: " + ", ".join(
            f"{k} {a} != {b}" for k, (a, b) in diff.items())
    return True, ""


def profile_top_rows(n):
    """The first n rows of the committed kernel summary (name, calls, average us, share), or None."""
    import csv
    path = os.path.join(ROOT, "profiles", f"{ROUND}_kernel_stats.csv")
    if not os.path.exists(path):
        return None
    with open(path, newline="") as f:
        rows = list(csv.DictReader(f))[:n]
    return [{"name": r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:100], "calls": int(r["Calls"]),
             "avg_us": float(r["AverageNs"]) / 1e3, "percent_of_gpu_time": float(r["Percentage"])} for r in rows]


def profile_kernel_rows(kernel):
    """Rows of the committed `rocprofv3 --kernel-trace --stats` summary (profiles/<round>_kernel_stats.csv) whose kernel
    name contains `kernel`, as (name, calls, average ns, share of GPU time), with the file's rank of the first of them
    (1 = top row); None when no summary of this round is committed."""
    import csv
    path = os.path.join(ROOT, "profiles", f"{ROUND}_kernel_stats.csv")
    if not os.path.exists(path):
        return None
    rows = []
    with open(path, newline="") as f:
        for rank_, r in enumerate(csv.DictReader(f), 1):
            if kernel in r["Name"]:
                rows.append({"rank": rank_, "name": r["Name"][:96], "calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3,
                             "percent_of_gpu_time": float(r["Percentage"])})
    return rows or None


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
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-workloads", action="store_true",
                    help="skip config.other_workloads (BASELINE configs[3] / [4] at their per-GPU share, 5 + 10 steps each)")
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
    # Per-kernel pricing (roofline.achieved, stages): the same step, same shapes, same stream, with a HIP
    # event pair around every C-ABI launch.  Creating ~500 event pairs per step costs host time, so this
    # instrumented pass runs right after the timed region instead of inside it (3 steps).
    kern, prof_steps = {}, 3
    if rank == 0:
        model.two_streams = False      # price kernels one at a time (the timed loop overlaps the two encoders)
        eager = engine.TrainStep(model, batch, cfg.lr, world=1)
        eager.step()
        torch.cuda.synchronize()
        ops.KernelTimer.start()
        ops.EMD_WALK_STATS = []
        ops.SA_ROWMASK_STATS = []
        for _ in range(prof_steps):
            eager.step()
        kern = ops.KernelTimer.stop()
        kern_variants = dict(ops.KernelTimer.variants)
        emd_walk, ops.EMD_WALK_STATS = ops.EMD_WALK_STATS, None
        rowmasks, ops.SA_ROWMASK_STATS = ops.SA_ROWMASK_STATS, None
        # rows of dh that exist (mask bit set) per step, in bytes: what the masked list sum has to read
        def _bits(t):
            v = t.to(torch.int64) & 0xffffffff
            c = torch.zeros_like(v)
            for sh in range(32):
                c += (v >> sh) & 1
            return int(c.sum())
        dh_rows_bytes = sum(_bits(m) * c1 * 4 for m, c1 in rowmasks) / prof_steps if rowmasks else None
        kern_flops = dict(ops.KernelTimer.flops)
        eager.close()
        # the same events with the two encoders on two streams, as in the timed loop (and in a rocprofv3 trace of it): a launch's
        # duration then includes what it loses to the other stream's kernels on the same CUs.  `roofline` (the dominant
        # kernel) is priced from THIS pass, so that its live average is the figure a kernel trace of this command averages to.
        model.two_streams = True
        eager2 = engine.TrainStep(model, batch, cfg.lr, world=1)
        eager2.step()
        torch.cuda.synchronize()
        ops.KernelTimer.start()
        for _ in range(prof_steps):
            eager2.step()
        kern2 = ops.KernelTimer.stop()
        kern2_variants = dict(ops.KernelTimer.variants)
        eager2.close()
        kern_api, stage_kg, cold_kg = time_knn_group_api(batch, dev, 20)

    if rank == 0:
        def per_step(name, table=None):
            n, ms = (table or kern).get(name, (0, 0.0))
            return n / prof_steps, ms / prof_steps

        # (1) `roofline_sa_level`: the generated-row max-pool level (pzn_sa_level_fwd_*: second shared-MLP layer + ReLU + max
        #     over the 32 neighbours on the generated rows of the first, model5_b.py:452-454 / :459-461; 4 launches per
        #     step).  It issues v_mfma_f32_32x32x16_bf16 six times per fp32 product, so it is priced against the bf16
        #     pipe / 6, with the fp32-input MFMA rate beside it.  (`roofline` of rounds 2-3; see (6) for this round's.)
        mp_entry = next((n for n in ("pzn_sa_level_fwd_packed_f32", "pzn_sa_level_fwd_ws_f32", "pzn_sa_level_fwd_f32") if n in kern),
                        "pzn_linear_maxpool_fwd_f32")
        packed = mp_entry == "pzn_sa_level_fwd_packed_f32"     # the kernel alone: the split of W2 is its own entry point (timed beside)
        streamed = packed or (mp_entry == "pzn_sa_level_fwd_ws_f32" and os.environ.get("PZN_SA_STREAM", "1") != "0")
        n_mp, ms_mp = per_step(mp_entry)
        fl_mp = kern_flops.get(mp_entry, 0) / prof_steps
        mp_ach = fl_mp / (ms_mp * 1e-3) / 1e12 if ms_mp > 0 else 0.0
        mp_traffic, mp_src = pmc_traffic("ws_gemm_maxpool_bytes_per_step", B, N)
        roofline_sa_level = {
            "bound": "mfma",
            "kernel": ("sa_level_stream_kernel<C1, CT> (csrc/salevel.hip: bf16x3 MFMA kernel, the rows relu(P'[idx] + Q) of the first "
                       "layer generated once per group in registers, W2 streamed through a three-slot LDS ring by LDS-DMA, "
                       "max / arg-max epilogue in registers" + ("; the split of W2 into planes, sa_pack_w_kernel, is the entry point "
                       "pzn_sa_level_prep_weights_f32: `weight_split_ms_per_step`) behind " if packed else
                       "; + sa_pack_w_kernel, the split of W2 into planes) behind ") if streamed else
                       "ws_gemm_kernel<NT, MAXPOOL=true, ..., GATH> (csrc/wsgemm.hip: weight-stationary bf16x3 kernel, max-pool epilogue; "
                       "GATH: the first layer's rows relu(P'[idx] + Q) are generated in its operand loader) behind ") + mp_entry,
            "achieved": mp_ach, "peak": MFMA_X3_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": mp_ach / MFMA_X3_PEAK_TFLOPS,
            "traffic": mp_traffic, "traffic_source": mp_src,
            "peak_note": "dense bf16 MFMA rate (2500 TFLOP/s) / 6 issued bf16 MFMAs per fp32 product; against the fp32-input "
                         f"MFMA rate ({MFMA_F32_PEAK_TFLOPS} TFLOP/s) the same number reads {mp_ach / MFMA_F32_PEAK_TFLOPS:.3f}",
            "algorithmic_flops_per_step": fl_mp, "ms_per_step": ms_mp, "launches_per_step": n_mp,
            "avg_launch_ms": ms_mp / max(1.0, n_mp),
        }
        if packed:
            roofline_sa_level["weight_split_ms_per_step"] = per_step("pzn_sa_level_prep_weights_f32")[1]
        # (1b) every dense matrix-core entry point together (no sparse vector-ALU passes in the sum)
        dense_names = ("pzn_linear_fwd_f32", "pzn_linear_dgrad_f32", "pzn_linear_wgrad_f32", "pzn_linear_maxpool_fwd_f32",
                       "pzn_sharedmlp_max_fwd_f32", "pzn_attn_fwd_f32", "pzn_attn_bwd_f32", "pzn_attn_block_fwd_f32",
                       "pzn_attn_block_bwd_f32", "pzn_sa_level_fwd_f32", "pzn_sa_level_fwd_ws_f32", "pzn_sa_level_fwd_packed_f32", "pzn_sa_level_prep_weights_f32", "pzn_outproj_maxpts_fwd_f32", "pzn_attn_fused_proj", "pzn_attn_fused_fwd",
                       "pzn_attn_fused_bwd_q", "pzn_attn_fused_bwd_k", "pzn_attn_fused_wgrads", "pzn_linear_slice_fwd_f32",
                       "pzn_point_mlp3_fwd_f32", "pzn_point_mlp3_bwd_f32")
        d_ms = sum(kern.get(k, (0, 0.0))[1] for k in dense_names) / prof_steps
        d_fl = sum(kern_flops.get(k, 0) for k in dense_names) / prof_steps
        d_n = sum(kern.get(k, (0, 0.0))[0] for k in dense_names) / prof_steps
        fam_ach = d_fl / (d_ms * 1e-3) / 1e12 if d_ms > 0 else 0.0
        fam_traffic, fam_src = pmc_traffic("mfma_family_bytes_per_step", B, N)
        roofline_mfma_family = {
            "bound": "mfma",
            "kernel": "all dense matrix-core launches of the step (ws_gemm_kernel, df_wgrad_kernel, gemm_kernel, point_mlp3_* behind "
                      "pzn_linear_* / pzn_attn_* / pzn_point_mlp3_*): algorithmic 2*M*N*K over summed launch time; the sparse vector-ALU "
                      "passes of the pooled backward (pool_dgrad / pool_wgrad) are NOT in this sum",
            "achieved": fam_ach, "peak": MFMA_X3_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": fam_ach / MFMA_X3_PEAK_TFLOPS,
            "traffic": fam_traffic, "traffic_source": fam_src,
            "algorithmic_flops_per_step": d_fl, "ms_per_step": d_ms, "launches_per_step": d_n,
        }
        # (2) the stage the north star names: kNN + group (HBM-bound), SURVEY 8(d) bytes, on the launch the drop-in
        #     sample_and_group really makes; the model path's own search (indices only) is listed beside it.
        per_pair = 2 * (knn_group_bytes(N, 512, 32, 64) + knn_group_bytes(512, 256, 32, 128))
        ms_kg = sum(ms for ms, _ in stage_kg)                 # one step-equivalent: 2 clouds x 2 levels
        n_kg = sum(k for _, k in stage_kg)
        if n_kg == len(stage_kg):
            stage_names = ("knn_select_kernel<R, 8, true, D, kp, nt> behind pzn_knn_group_f32: what pointnet_util."
                           "sample_and_group(npoint, 0, 32, xyz, points, knn=True) launches after FPS (search + "
                           "reference-layout [B,S,32,3+D] group write in one launch), 4 launches per step-equivalent "
                           "(2 clouds x 2 levels), each replayed 20x back to back on the step's clouds")
        else:      # shapes the fused launch does not take (N > 4096): the two single launches
            stage_names = "knn kernels (pzn_knn_f32) + group_fwd_vec_kernel (pzn_group_fwd_f32): N > 4096"
        achieved = per_pair * B / (ms_kg * 1e-3) / 1e9 if ms_kg > 0 else 0.0
        kg_traffic, kg_src = pmc_traffic("knn_group_stage_bytes_per_step", B, N)
        n_mk, ms_mk = per_step("pzn_knn_f32")
        roofline_knn_group = {
            "bound": "hbm", "kernel": stage_names,
            "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "traffic": kg_traffic, "traffic_source": kg_src,
            "algorithmic_bytes_per_step": per_pair * B, "ms_per_step": ms_kg, "launches_per_step": n_kg,
            "avg_launch_ms": ms_kg / max(1, n_kg),
            "launch_ms": {"cloud0_level1": stage_kg[0][0], "cloud0_level2": stage_kg[1][0], "cloud1_level1": stage_kg[2][0],
                          "cloud1_level2": stage_kg[3][0]},
            "cold": {"what": "the same four launches back to back inside ONE event pair right after 512 MB were written elsewhere "
                             "(inputs neither in L2 nor in the Infinity Cache; the four have distinct inputs), median of 5",
                     "ms_per_step": cold_kg, "achieved": per_pair * B / (cold_kg * 1e-3) / 1e9 if cold_kg > 0 else 0.0,
                     "frac": (per_pair * B / (cold_kg * 1e-3) / 1e9 / HBM_PEAK_GBS) if cold_kg > 0 else 0.0},
            "model_path_knn": {"entry": "pzn_knn_f32 (indices only; the encoder gathers per-point rows instead of "
                                        "materialising groups)", "launches_per_step": n_mk, "ms_per_step": ms_mk},
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
        n_gf, ms_gf = per_step("pzn_sa_point_l1_fwd_f32")
        n_gb, ms_gb = per_step("pzn_sa_point_l1_bwd_rm_f32" if "pzn_sa_point_l1_bwd_rm_f32" in kern else "pzn_sa_point_l1_bwd_f32")
        g_bytes_dense = 2 * B * ((gf1 + gf2 if n_gf else 0) + (gb1 + gb2 if n_gb else 0))
        g_bytes = g_bytes_dense
        if dh_rows_bytes is not None and n_gb and not n_gf:
            # with the row mask the list sum reads only the rows that exist: the dense dh term is replaced by what the masks
            # of this very pass count (device popcount), everything else (dP, lists, coordinates) stays
            g_bytes = g_bytes_dense - 2 * B * 4 * 32 * (512 * 128 + 256 * 256) + dh_rows_bytes
        g_ms = ms_gf + ms_gb
        g_ach = g_bytes / (g_ms * 1e-3) / 1e9 if g_ms > 0 else 0.0
        g_traffic, g_src = pmc_traffic("sa_gather_stage_bytes_per_step", B, N)
        if "pzn_sa_level_bwd_pt_f32" in kern:
            roofline_sa_gather = {"bound": "hbm", "kernel": "none: round 5 folds the per-point sums into the input-gradient walk (pool_point_kernel, "
                                                            "csrc/sapool.hip; priced in roofline_pool_bwd); dh and its list sum no longer exist",
                                  "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": None}
        else:
          roofline_sa_gather = {
            "bound": "hbm", "kernel": "sa_point_l1_bwd_kernel (pzn_sa_point_l1_bwd[_rm]_f32: the per-point sums of dh over inverse neighbour "
                                      "lists; with the row mask (_rm, default) rows that won no channel - exactly zero, about half of level 1 and "
                                      "a third of level 2 - are not read: `algorithmic_bytes_per_step` counts the rows that exist (popcount of "
                                      "the masks of the instrumented pass), `dense_bytes_per_step` is the figure without the mask)" + (" + sa_point_l1_fwd_kernel (rows written: PZN_SA_FUSED=0)" if n_gf else
                                                  "; the forward writes no rows any more (generated inside the matrix-core kernel)"),
            "achieved": g_ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": g_ach / HBM_PEAK_GBS,
            "traffic": g_traffic, "traffic_source": g_src,
            "algorithmic_bytes_per_step": g_bytes, "dense_bytes_per_step": g_bytes_dense,
            "avg_launch_ms": {"sa_point_l1_fwd_kernel": ms_gf / max(1.0, n_gf), "sa_point_l1_bwd_kernel": ms_gb / max(1.0, n_gb)},
        }
        # (4) EMD: vector-ALU / transcendental bound.  The reference's schedule is 10 levels x 3 passes x n*m pair
        #     evaluations (emd_kernel.cu:46-154); the fused path walks only the cloud-2 points that still hold mass and lie
        #     inside the level's x window, and counts what it evaluates (pzn_emd_walk_counter_offset: uint64 counters in
        #     units of 64 evaluations), so the evaluations EXECUTED are known (30 n m on the single-workgroup path).  The walks
        #     are packed (v_pk_*: two evaluations per instruction).  Per TWO evaluations, round 5 (csrc/emd.hip): pass B 12 packed
        #     instructions + 2 v_exp_f32, pass C + next A (one walk, counted ONCE by the device counters) 16 + 2 (the sharper
        #     exponential is the softer one squared twice), first A 8 + 2; measured issue costs on this chip (tools/valu_rate.hip):
        #     v_pk_* 4.4 cycles, v_exp_f32 8 = two 4-cycle slots.  Executed evaluations split about evenly between B and C + A, so
        #     (12 + 4 + 16 + 4) / 2 / 2 = 9 issue slots per counted evaluation (round 4 priced 7.25 for three exponentials more per
        #     two walks).  Peak = 256 CUs x 4 SIMDs x 16 lanes x 2.4 GHz = 39.3 T lane-slots/s (one wave64 instruction per 4 cycles).
        n_e, ms_e = per_step("pzn_emd_fused_f32")
        ev_exec, ev_ref = 0.0, 0.0
        for ctr, eb, en, em in emd_walk:
            ev_ref += 30.0 * eb * en * em
            ev_exec += 30.0 * eb * en * em if ctr is None else 64.0 * float(ctr.sum().item())
        ev_exec, ev_ref = ev_exec / prof_steps, ev_ref / prof_steps
        LANE_OPS = 9.0
        e_ach = ev_exec * LANE_OPS / (ms_e * 1e-3) / 1e12 if ms_e > 0 else 0.0
        roofline_emd = {
            "bound": "valu", "kernel": "emdf_b_kernel / emdf_k_kernel<0|1|2> / emdf_compact_kernel / emd_sort_x_kernel / emd_small_fused_kernel "
                                       "behind pzn_emd_fused_f32 (4 calls per step: N x N, B x B, 2 x 128 x 128)",
            "achieved": e_ach, "peak": 39.3, "unit": "T lane-slot/s", "frac": e_ach / 39.3,
            "traffic": pmc_traffic("emd_bytes_per_step", B, N)[0],
            "pair_evaluations_executed_per_step": ev_exec, "pair_evaluations_reference_schedule_per_step": ev_ref,
            "issue_slots_per_evaluation": LANE_OPS, "ms_per_step": ms_e, "launches_per_step": n_e,
            "note": "executed evaluations counted on the device (active lists + x windows); the reference's schedule (30 n m per pair) "
                    f"would be {ev_ref / max(ev_exec, 1.0):.2f}x as many",
        }
        # (5) the attention blocks (4 per encoder): projections on the fp32-accurate bf16x3 path, contractions in --attn
        # (4b) sparse backward of "linear + ReLU + max over 32 neighbours" behind the generated rows (csrc/poolbwd.hip):
        #      one non-zero per (group, channel) = one row axpy of length C1 in each pass.  Two ceilings: the bytes it must
        #      move (dh[R*32, C1] written once, gate rows P'[idx] re-read from L2 not counted) and the vector lanes it must
        #      issue (hits x C1 multiply-adds per pass, one lane-slot each, against the chip's vector issue rate).
        by_point = "pzn_sa_level_bwd_pt_f32" in kern       # round 5 (csrc/sapool.hip): dh is never in memory
        n_pb, ms_pb = per_step("pzn_sa_level_bwd_pt_f32" if by_point else
                               ("pzn_sa_level_bwd_rm_f32" if "pzn_sa_level_bwd_rm_f32" in kern else "pzn_sa_level_bwd_f32"))
        lvl = ((B * 512, 128, 128), (B * 256, 256, 256))              # (groups R, C1, C2) of the two levels
        pb_bytes_dense = 2 * sum(4.0 * R_ * 32 * C1_ + 4.0 * R_ * C2_ * 2 + 4.0 * C1_ * C2_ for R_, C1_, C2_ in lvl)
        pb_bytes = pb_bytes_dense
        if by_point:
            # what the by-point form must move: the pooled tensors (arg-max, out, dout) once for the weight-gradient pass and once
            # for the hit lists, the hit lists (8 bytes per live channel: upper bound = every channel) written once and read
            # once per 128-column slice, dP written once (after its zero fill), the per-point sums; the P' / Q rows the gates are
            # regenerated from are L2-resident re-reads and not counted, as before
            pb_bytes = 2 * sum(2 * 12.0 * R_ * C2_ + 8.0 * R_ * C2_ * (1 + C1_ // 128) + 2 * 4.0 * (R_ // (512 if C1_ == 128 else 256)) *
                               (N if C1_ == 128 else 512) * C1_ + 4.0 * C1_ * C2_ for R_, C1_, C2_ in lvl)
        if dh_rows_bytes is not None and "pzn_sa_level_bwd_rm_f32" in kern:
            # with the row mask only the rows that exist are written (rows are stored in pairs: this counts the rows with a bit)
            pb_bytes = pb_bytes_dense - 2 * sum(4.0 * R_ * 32 * C1_ for R_, C1_, C2_ in lvl) + dh_rows_bytes
        pb_fma = 2 * sum(2.0 * R_ * C2_ * C1_ for R_, C1_, C2_ in lvl)         # input-gradient + weight-gradient pass
        t_hbm, t_valu = pb_bytes / (HBM_PEAK_GBS * 1e9), pb_fma / VALU_LANE_SLOTS_PER_S
        roofline_pool_bwd = {
            "bound": "hbm", "kernel": ("pool_wgrad_kernel (csrc/poolbwd.hip) + pool_hits_kernel + pool_point_kernel (csrc/sapool.hip) behind "
                                       "pzn_sa_level_bwd_pt_f32, 4 calls per step (2 levels x 2 clouds): the rows' gradient dh is computed by "
                                       "point and never written; `dense_bytes_per_step` is what rounds 2-4 priced (dh written and read)") if by_point else
                                      ("pool_dgrad_kernel + pool_wgrad_kernel (csrc/poolbwd.hip) behind pzn_sa_level_bwd[_rm]_f32, "
                                      "4 launches per step (2 levels x 2 clouds); with the row mask (_rm) the dh term of the bytes counts "
                                      "the rows that exist (device popcount of the masks), `dense_bytes_per_step` is the figure without it"),
            "achieved": pb_bytes / (ms_pb * 1e-3) / 1e9 if ms_pb > 0 else 0.0, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": (pb_bytes / (ms_pb * 1e-3) / 1e9 / HBM_PEAK_GBS) if ms_pb > 0 else 0.0,
            "traffic": pmc_traffic("pool_bwd_stage_bytes_per_step", B, N)[0],
            "algorithmic_bytes_per_step": pb_bytes, "dense_bytes_per_step": pb_bytes_dense, "ms_per_step": ms_pb,
            "launches_per_step": n_pb,
            "vector_issue": {"lane_multiply_adds_per_step": pb_fma, "peak_lane_slots_per_s": VALU_LANE_SLOTS_PER_S,
                             "frac": (pb_fma / (ms_pb * 1e-3) / VALU_LANE_SLOTS_PER_S) if ms_pb > 0 else 0.0},
            "floor_ms_per_step": {"hbm": t_hbm * 1e3, "vector_issue": t_valu * 1e3},
        }
        fused_names = ("pzn_attn_fused_prep_weights", "pzn_attn_fused_prep_weights_n", "pzn_attn_fused_proj", "pzn_attn_fused_fwd",
                       "pzn_attn_fused_bwd_q", "pzn_attn_fused_bwd_k", "pzn_attn_fused_wgrads")
        attn_fused = "pzn_attn_fused_fwd" in kern
        attn_names = fused_names if attn_fused else ("pzn_attn_block_fwd_f32", "pzn_attn_block_bwd_f32")
        ms_at = sum(per_step(k)[1] for k in attn_names)
        n_at = sum(per_step(k)[0] for k in attn_names)
        fl_at = sum(kern_flops.get(k, 0) for k in attn_names) / prof_steps
        at_ach = fl_at / (ms_at * 1e-3) / 1e12 if ms_at > 0 else 0.0
        # what the attention kernels' products issue per fp32 product: six bf16 MFMAs (default), ONE in --attn bf16
        # (the chained kernels' single-plane instantiation: every product of the block, priced against the bf16 pipe itself;
        # the weight gradients stay bf16x3 and are a small share)
        attn_peak = MFMA_BF16_PEAK_TFLOPS if (args.attn == "bf16" and attn_fused) else MFMA_X3_PEAK_TFLOPS
        roofline_attention = {
            "bound": "mfma",
            "kernel": ("pzn_attn_fused_{prep_weights_n,proj,fwd,bwd_q,bwd_k,wgrads} (csrc/attnfused.hip): layerAttention (model5_b.py:83-101) as "
                       "chained matrix-core kernels, ONE encoder per launch (128 workgroups at B = 64; the two encoders run on two "
                       "streams in the timed loop and one after the other in this instrumented pass); ALGORITHMIC flops (the "
                       "backward's recomputed scores are not counted)") if attn_fused else
                      ("pzn_attn_block_{fwd,bwd}_f32: layerAttention (model5_b.py:83-101), 8 + 8 launches per step; "
                       f"contractions q k^T / attn v and their backward in {'single bf16 MFMAs, fp32 softmax' if args.attn == 'bf16' else 'bf16x3 split precision (fp32 results)'}"),
            "achieved": at_ach, "peak": attn_peak, "unit": "TFLOP/s", "frac": at_ach / attn_peak,
            "traffic": None, "algorithmic_flops_per_step": fl_at, "ms_per_step": ms_at, "launches_per_step": n_at,
            "note": "256 tokens per cloud whatever N: 64 x (256 x 256 x 64..256) products per encoder and block",
        }
        # (6) `roofline`: the DOMINANT kernel.  Candidates = the entry points that launch exactly ONE kernel each (so that the
        #     event pair around the call times that kernel and nothing else) and whose algorithmic flops the wrapper records;
        #     the winner is the one with the largest launch-time sum per step in this instrumented pass.  (Entry points that
        #     are sequences of kernels - EMD: ~20 passes per N x N call, the sparse pooled backward: two passes - are priced as
        #     stages of their own, roofline_emd / roofline_pool_bwd; their largest single kernels are listed in
        #     `largest_multi_kernel_entries` with the committed profile's figures so that the ranking can be checked.)
        single = {
            "pzn_attn_fused_fwd": ("attn_fwd_kernel", "csrc/attnfused.hip: scores, softmax, PV, x - a, out projection, x + relu(.) for 16 points per wavefront (model5_b.py:67-75, 83-101)"),
            "pzn_attn_fused_bwd_q": ("attn_bwd_q_kernel", "csrc/attnfused.hip: query side of the block's backward (dz, dt, da image, dP, recomputed P, delta, dS, dq)"),
            "pzn_attn_fused_bwd_k": ("attn_bwd_k_kernel", "csrc/attnfused.hip: key side of the block's backward (S, P, dP, dS, dk, dv, dx = u + dq Wq + dk Wk + dv Wv)"),
            "pzn_attn_fused_proj": ("attn_proj_kernel", "csrc/attnfused.hip: q, k, v projection into bf16-plane images"),
            mp_entry: ("sa_level_stream_kernel", "csrc/salevel.hip: generated-row max-pool level (two instantiations: <128,4>, <256,8>)"),
            "pzn_outproj_maxpts_fwd_f32": ("outproj_maxpts_kernel", "csrc/outproj.hip: out projection of the five slices + max over the points"),
            # (pzn_attn_fused_wgrads is a SEQUENCE - a linear weight gradient, zero fills, two df_wgrad_kernel launches of
            #  different shapes - so an event pair around it does not time one kernel: it is priced inside roofline_attention)
        }
        launches_per_call = {}
        cand = {e: per_step(e, kern2) for e in single if e in kern2 and kern_flops.get(e, 0) > 0}
        cand_one_stream = {e: per_step(e) for e in cand}
        cand_fl = {e: kern_flops.get(e, 0) / prof_steps for e in cand}
        # an entry point with several kernel instantiations (the level kernel: <128, 4> and <256, 8>) competes per instantiation,
        # as in the rows of a rocprofv3 kernel summary
        split_entries = {e for (e, _v) in kern2_variants if e in cand}
        for (e, var), (n_, ms_, fl_) in kern2_variants.items():
            if e in split_entries:
                key = e + var
                single[key] = (single[e][0] + var, single[e][1])
                cand[key] = (n_ / prof_steps, ms_ / prof_steps)
                cand_fl[key] = fl_ / prof_steps
                n1, ms1, _ = kern_variants.get((e, var), (0, 0.0, 0))
                cand_one_stream[key] = (n1 / prof_steps, ms1 / prof_steps)
        for e in split_entries:
            cand.pop(e)
        for e, k_ in launches_per_call.items():
            if e in cand:
                cand[e] = (cand[e][0] * k_, cand[e][1])
        # Which one is "dominant"?  The first row of the committed rocprofv3 kernel summary of THIS command (timed loop on two
        # streams + this pass) that one of the candidates launches - rows of multi-kernel entry points (the EMD passes, priced
        # as roofline_emd) are stepped over, so that two rows a tenth of a percent apart swapping places between collections
        # do not change the basis of the choice; without a committed summary: the candidate with the largest launch-time sum
        # per step in this pass.
        how = "largest launch-time sum per step among the single-kernel entry points of the instrumented two-stream pass"
        dom = max(cand, key=lambda e: cand[e][1])
        prof_ok, prof_why = profile_matches(B, N, args.attn)
        top_rows = profile_top_rows(8) if prof_ok else None
        if not prof_ok:
            how += f" ({prof_why})"
        if top_rows:
            by_kernel = {single[e][0]: e for e in cand}
            skipped = []
            for row in top_rows:
                hit = next((by_kernel[k_] for k_ in by_kernel if k_ in row["name"]), None)
                if hit is not None:
                    dom = hit
                    how = (f"first row of profiles/{ROUND}_kernel_stats.csv (rocprofv3 --kernel-trace --stats of this command) that a "
                           f"single-kernel entry point launches: {row['percent_of_gpu_time']} % of GPU time")
                    if skipped:
                        how += "; stepped over (kernels of multi-kernel entry points, priced as stages): " + ", ".join(skipped)
                    break
                skipped.append(f"{row['name'][:40]} ({row['percent_of_gpu_time']} %)")
            else:
                how += f"; none of the first {len(top_rows)} rows of profiles/{ROUND}_kernel_stats.csv belongs to a single-kernel entry point"
        dn, dms = cand[dom]
        dfl = cand_fl[dom]
        dom_peak = attn_peak if dom.startswith("pzn_attn_fused") else MFMA_X3_PEAK_TFLOPS
        dach = dfl / (dms * 1e-3) / 1e12 if dms > 0 else 0.0
        prof_rows = profile_kernel_rows(single[dom][0]) if prof_ok else None
        # HBM bytes per launch of that kernel from this build's committed PMC passes (per-kernel table of tools/pmc_summary.py)
        pk, pk_src = pmc_traffic("per_kernel", B, N)
        dom_traffic = None
        if pk:
            hit = [v for k_, v in pk.items() if k_.startswith(single[dom][0])]
            if hit:
                dom_traffic = sum(v["hbm_bytes_per_launch"] * v["launches"] for v in hit) / sum(v["launches"] for v in hit)
        roofline = {
            "bound": "mfma",
            "kernel": f"{single[dom][0]} behind {dom.split('<')[0]} ({single[dom][1]})",
            "achieved": dach, "peak": dom_peak, "unit": "TFLOP/s", "frac": dach / dom_peak,
            "traffic": dom_traffic, "traffic_source": pk_src + " (bytes per launch, like algorithmic_flops_per_launch)",
            "peak_note": ("dense bf16 MFMA rate (2500 TFLOP/s) / 6 issued bf16 MFMAs per fp32 product" if dom_peak == MFMA_X3_PEAK_TFLOPS
                          else "dense bf16 MFMA rate (one bf16 MFMA per product in --attn bf16)"),
            "algorithmic_flops_per_step": dfl, "algorithmic_flops_per_launch": dfl / max(1.0, dn),
            "ms_per_step": dms, "launches_per_step": dn, "avg_launch_us": 1e3 * dms / max(1.0, dn),
            "timed": "HIP events around every launch of the kernel on its own stream, the two encoders on two streams as in the timed "
                     "loop (3 steps right after it); `one_stream` = the same launches with the encoders one after the other",
            "one_stream": {"avg_launch_us": 1e3 * cand_one_stream[dom][1] / max(1.0, cand_one_stream[dom][0]),
                           "frac": (dfl / (cand_one_stream[dom][1] * 1e-3) / 1e12 / dom_peak) if cand_one_stream[dom][1] > 0 else None},
            "how_chosen": how,
            "candidates_ms_per_step": {single[e][0]: cand[e][1] for e in sorted(cand, key=lambda e: -cand[e][1])},
        }
        # the stage BASELINE.json's north star prices (>= 60 % of the HBM roofline on kNN + group), inside `roofline` so that
        # it travels with the headline object: replayed and cold fractions of 8 TB/s, time per step-equivalent, PMC traffic
        roofline["north_star_stage"] = {
            "stage": "kNN + group: pzn_knn_group_f32, what pointnet_util.sample_and_group(npoint, 0, 32, xyz, points, knn=True) "
                     "launches after FPS (details: roofline_knn_group)",
            "bound": "hbm", "frac": roofline_knn_group["frac"], "cold_frac": roofline_knn_group["cold"]["frac"],
            "ms_per_step": roofline_knn_group["ms_per_step"], "cold_ms_per_step": roofline_knn_group["cold"]["ms_per_step"],
            "algorithmic_bytes_per_step": roofline_knn_group["algorithmic_bytes_per_step"], "traffic": roofline_knn_group["traffic"],
            "peak": HBM_PEAK_GBS, "unit": "GB/s",
        }
        if prof_rows:
            pavg = sum(r["avg_us"] * r["calls"] for r in prof_rows) / sum(r["calls"] for r in prof_rows)
            roofline["profile"] = {
                "file": f"profiles/{ROUND}_kernel_stats.csv", "rows": prof_rows, "profile_avg_launch_us": pavg,
                "frac_from_profile_avg": dfl / max(1.0, dn) / (pavg * 1e-6) / 1e12 / dom_peak,
                "note": "rocprofv3 --kernel-trace --stats of this command: averages over the timed loop, where the two encoders' "
                        "kernels share the chip (two streams), and the one-stream instrumented pass"}
        if top_rows:
            roofline["profile_top_rows"] = top_rows[:6]
        stages = {k: {"launches_per_step": n / prof_steps, "ms_per_step": ms / prof_steps} for k, (n, ms) in sorted(kern.items())}
        stages_api = {k: {"launches": n, "ms": ms} for k, (n, ms) in sorted(kern_api.items())}
        out = {
            "metric": f"point-cloud pairs/sec (fwd+bwd) at N={N}, B={B}; FPS/kNN idx bit-exact",
            "value": world * B * args.steps / dt, "unit": "pairs/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32" if args.attn == "f32" else "f32 (attention contractions bf16)",
            "data": "synthetic",
            "config": {"workload": f"BASELINE configs[{1 if (B, N) == (64, 2048) else ('4' if N == 8192 else '3' if N == 4096 else '-')}] shape: N={N} points, {B} pairs/GPU, fp32 train step "
                                   f"(predict5 + loss_mode 1 losses incl. 4x EMD + backward + Adam)",
                       "global_batch": world * B, "points": N, "parallelism": f"dp{world}", "encoder_streams": 2,
                       "matrix_core_path": "bf16x3 split precision (fp32 results; six bf16 MFMAs per product)",
                       "attention": args.attn, "build_id": build_id(),
                       # host time to ENQUEUE a step against the step itself: close to 1 = this box's host is the bottleneck
                       "host_enqueue_ms_per_step": t_enq / args.steps * 1e3,
                       "host_enqueue_over_step": t_enq / dt},
            "roofline": roofline,
            "roofline_sa_level": roofline_sa_level,
            "roofline_mfma_family": roofline_mfma_family,
            "roofline_knn_group": roofline_knn_group,
            "roofline_sa_gather": roofline_sa_gather,
            "roofline_emd": roofline_emd,
            "roofline_attention": roofline_attention,
            "roofline_pool_bwd": roofline_pool_bwd,
            "stages": stages,
            "stages_sample_and_group_dropin": stages_api,
            "loss": loss_val,
        }
        if not args.no_other_workloads and world == 1 and (B, N, args.attn) == (64, 2048, "f32"):
            # BASELINE configs[3] and [4] (1-GPU shares), measured by whoever runs this line: same process, after the headline
            del model, batch, runner, eager, eager2
            torch.cuda.empty_cache()
            out["config"]["other_workloads"] = [other_workload(dev, 64, 4096, "f32"), other_workload(dev, 32, 8192, "f32"),
                                                other_workload(dev, 32, 8192, "bf16")]
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(N, args.cpu_pairs, args.cpu_iters)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
