#!/usr/bin/env python3
"""Condense rocprofv3 --pmc counter_collection.csv dumps into a per-kernel summary.

    python tools/pmc_summary.py <fetch_dir> <write_dir> <out.json>

<fetch_dir>/<write_dir> are the -d directories of two separate passes:
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d <fetch_dir> -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d <write_dir> -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline
Counters are in KB.  gfx950: FETCH_SIZE tallies 64 B per 128-B request for 16-B/lane streams, so
bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (MI355X_MICROARCH.md, HBM section).
"""
import collections
import csv
import glob
import json
import re
import sys


def load(d):
    rows = list(csv.DictReader(open(glob.glob(d + "/*/*counter_collection.csv")[0])))
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in rows:
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
        name = re.sub(r"[<(].*", "", name)
        agg[name][0] += 1
        agg[name][1] += float(r["Counter_Value"])
    return agg


def main():
    fetch, write, out = sys.argv[1:4]
    f, w = load(fetch), load(write)
    per = {}
    for k in sorted(f):
        if k not in w:
            continue
        fk, wk = f[k][1] / f[k][0], w[k][1] / w[k][0]
        per[k] = {"launches": f[k][0], "FETCH_SIZE_KB_per_launch": round(fk, 1), "WRITE_SIZE_KB_per_launch": round(wk, 1),
                  "hbm_bytes_per_launch": round((2 * fk + wk) * 1024)}
    # kNN + group stage of one step: the fused launch runs 4x per step (2 clouds x 2 levels) with
    # different sizes, so take total bytes / steps; older traces had kNN (2 batched launches) and the
    # padded group (4 launches) as separate kernels.
    stage = None
    if "knn_group_pad_kernel" in per:
        k = per["knn_group_pad_kernel"]
        stage = round(k["hbm_bytes_per_launch"] * 4)
    elif "knn32_reg_kernel" in per and "group_pad_direct_kernel" in per:
        stage = 2 * per["knn32_reg_kernel"]["hbm_bytes_per_launch"] + 4 * per["group_pad_direct_kernel"]["hbm_bytes_per_launch"]
    # first set-abstraction layer of the model path (csrc/sapoint.hip): gather forward + list-sum backward, 4 launches
    # each per step (2 clouds x 2 levels)
    sa = None
    if "sa_point_l1_fwd_kernel" in per and "sa_point_l1_bwd_kernel" in per:
        sa = round(4 * (per["sa_point_l1_fwd_kernel"]["hbm_bytes_per_launch"] + per["sa_point_l1_bwd_kernel"]["hbm_bytes_per_launch"]))
    # the matrix-core family behind bench.py's `roofline` (+ the sparse pooled passes its time includes): bytes per
    # training step = sum over launches / passes in the trace (1 warm-up + 2 timed + 1 + 3 instrumented = 7)
    passes = 7
    fam = ("ws_gemm_kernel", "df_wgrad_kernel", "gemm_kernel", "pool_dgrad_kernel", "pool_wgrad_kernel")
    mfma = round(sum(per[k]["hbm_bytes_per_launch"] * per[k]["launches"] for k in fam if k in per) / passes) or None
    doc = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes, python bench.py --steps 2 --warmup 1 "
                     "--no-cpu-baseline (B=64, N=2048)",
           "correction": "bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950 FETCH_SIZE half-count, MI355X_MICROARCH.md)",
           "knn_group_stage_bytes_per_step": stage,
           "knn_group_stage_algorithmic_bytes_per_step": 1242431488,
           "mfma_family_bytes_per_step": mfma,
           "sa_gather_stage_bytes_per_step": sa,
           "sa_gather_stage_algorithmic_bytes_per_step": 4746904576,
           "per_kernel": {k: v for k, v in per.items() if not k.startswith("at::") and "rocclr" not in k}}
    json.dump(doc, open(out, "w"), indent=1)
    print(json.dumps({k: doc[k] for k in ("knn_group_stage_bytes_per_step", "knn_group_stage_algorithmic_bytes_per_step",
                                          "sa_gather_stage_bytes_per_step", "sa_gather_stage_algorithmic_bytes_per_step")}))


if __name__ == "__main__":
    main()
