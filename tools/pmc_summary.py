#!/usr/bin/env python3
"""Condense rocprofv3 --pmc counter_collection.csv dumps into the per-kernel HBM traffic summary bench.py attaches to its
roofline objects (profiles/<round>_pmc_traffic.json).

    python tools/pmc_summary.py <fetch_dir> <write_dir> <out.json> [--batch 64 --points 2048]

<fetch_dir> / <write_dir> are the -d directories of two SEPARATE passes of the same command:
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d <fetch_dir> -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d <write_dir> -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline
Counters are in KB.  gfx950: FETCH_SIZE tallies 64 B per 128-B request of a wide streaming read, so
bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (MI355X_MICROARCH.md, HBM section).  The summary carries the hash of the
kernel sources it was taken on (bench.build_id): bench.py reports these bytes only for that build.
"""
import argparse
import collections
import csv
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def load(d):
    """-> {kernel name incl. template arguments: [launches, counter sum]}"""
    rows = list(csv.DictReader(open(glob.glob(d + "/*/*counter_collection.csv")[0])))
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in rows:
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
        name = re.sub(r"\(.*", "", name).strip()
        agg[name][0] += 1
        agg[name][1] += float(r["Counter_Value"])
    return agg


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("fetch")
    ap.add_argument("write")
    ap.add_argument("out")
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--points", type=int, default=2048)
    ap.add_argument("--steps-in-trace", type=int, default=15,
                    help="1 warm-up + 2 timed + (1 + 3) entry-point pass + (1 + 3) one-stream kernel pass + (1 + 3) two-stream kernel pass")
    a = ap.parse_args()
    import bench
    f, w = load(a.fetch), load(a.write)
    per = {}
    for k in sorted(f):
        if k not in w or k.startswith("at::") or "rocclr" in k or k.startswith("Cijk"):
            continue
        fk, wk = f[k][1] / f[k][0], w[k][1] / w[k][0]
        per[k] = {"launches": f[k][0], "FETCH_SIZE_KB_per_launch": round(fk, 1), "WRITE_SIZE_KB_per_launch": round(wk, 1),
                  "hbm_bytes_per_launch": round((2 * fk + wk) * 1024)}

    def sel(pattern):
        return {k: v for k, v in per.items() if re.search(pattern, k)}

    # the drop-in stage (search + group, one launch per level and cloud): one step-equivalent = 2 clouds x (level 1 + level 2)
    kg = sel(r"^knn_select_kernel<\d+, \d+, true")
    stage = round(2 * sum(v["hbm_bytes_per_launch"] for v in kg.values())) if len(kg) == 2 else None
    # the named matrix-core kernel (max-pool variant with the generated operand): 2 clouds x (level 1 + level 2)
    # (the streamed-weights kernel of salevel.hip; the weight-stationary GATH variant for shapes it does not take)
    mp = sel(r"^sa_level_stream_kernel<") or sel(r"^ws_gemm_kernel<\d+, true, false, \d+, true, (true|false), true>")
    maxpool = round(2 * sum(v["hbm_bytes_per_launch"] for v in mp.values())) if len(mp) == 2 else None
    fam = sel(r"^(ws_gemm_kernel|df_wgrad_kernel|attn_wgrad_kernel|gemm_kernel|sa_level_stream_kernel|outproj_maxpts_kernel|point_mlp3_(fwd|bwd)_kernel|attn_(proj|fwd|bwd_q|bwd_k)_kernel)")
    mfma = round(sum(v["hbm_bytes_per_launch"] * v["launches"] for v in fam.values()) / a.steps_in_trace) or None
    sa_b = None      # (rounds 2-4 priced the list sum of dh here; dh has not existed since round 5)
    # the set-abstraction backward by point (round 5): weight-gradient pass, hit lists, the walk by point
    pb = sel(r"^(pool_wgrad_kernel|pool_hits_kernel|pool_point_kernel)")
    pb_b = round(sum(v["hbm_bytes_per_launch"] * v["launches"] for v in pb.values()) / a.steps_in_trace) or None
    emd = sel(r"^(emdf_|emd_)")
    emd_b = round(sum(v["hbm_bytes_per_launch"] * v["launches"] for v in emd.values()) / a.steps_in_trace) or None
    prep = sel(r"^sa_prep_kernel")
    prep_b = round(sum(v["hbm_bytes_per_launch"] * v["launches"] for v in prep.values()) / a.steps_in_trace) or None
    doc = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes, python3 bench.py --steps 2 --warmup 1 "
                     "--no-cpu-baseline --no-other-workloads",
           "correction": "bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950 FETCH_SIZE half-count, MI355X_MICROARCH.md)",
           "build_id": bench.build_id(), "batch": a.batch, "points": a.points,
           "knn_group_stage_bytes_per_step": stage,
           "knn_group_stage_algorithmic_bytes_per_step": a.batch * 2 * (bench.knn_group_bytes(a.points, 512, 32, 64) +
                                                                          bench.knn_group_bytes(512, 256, 32, 128)),
           "knn_group_stage_kernels": kg,
           "ws_gemm_maxpool_bytes_per_step": maxpool, "ws_gemm_maxpool_kernels": mp,
           "mfma_family_bytes_per_step": mfma,
           "sa_gather_stage_bytes_per_step": sa_b,
           "sa_prep_bytes_per_step": prep_b,
           "pool_bwd_stage_bytes_per_step": pb_b,
           "emd_bytes_per_step": emd_b,
           "per_kernel": per}
    json.dump(doc, open(a.out, "w"), indent=1)
    print(json.dumps({k: doc[k] for k in ("build_id", "knn_group_stage_bytes_per_step", "knn_group_stage_algorithmic_bytes_per_step",
                                          "ws_gemm_maxpool_bytes_per_step", "mfma_family_bytes_per_step",
                                          "sa_gather_stage_bytes_per_step")}))


if __name__ == "__main__":
    main()
