"""bench.py's bookkeeping that runs without a GPU: the work model of the kernels `roofline` can name, and the committed profile
summaries (used only when their sidecar names this build and workload)."""
import csv
import json
import os

import bench

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_kernel_work_model():
    # level 2 of the set abstraction = 2 x (B 256 groups x 32 rows) x 256 x 256 flop; bf16x3 ceiling = 2500 / 6
    b, w, unit, peak, punit, _ = bench.kernel_work("sa_level_stream_kernel<256, 8>", 64, 2048)
    assert (b, unit, punit) == ("mfma", "flop", "TFLOP/s") and w == 2.0 * 64 * 256 * 32 * 256 * 256 == 68719476736.0
    assert abs(peak - 2500.0 / 6) < 1e-9
    assert bench.kernel_work("attn_bwd_k_kernel<1>", 64, 2048)[3] == 2500.0      # single-plane instantiation: the bf16 pipe itself
    assert bench.kernel_work("attn_bwd_k_kernel<3>", 64, 2048)[3] == peak
    assert bench.kernel_work("outproj_maxpts_kernel", 64, 2048)[1] == 2.0 * 64 * 256 * 1280 * 1024
    b, w, unit, peak, punit, _ = bench.kernel_work("pool_point_kernel<16, 8>", 64, 2048)
    assert (b, unit, peak, punit) == ("hbm", "B", 8000.0, "GB/s") and w > 0
    assert bench.kernel_work("attn_wgrad_kernel", 64, 2048)[1] == 2.0 * 64 * 256 * 256 * (2 * 64 + 2 * 256)
    b, w, *_ = bench.kernel_work("attn_wgrad_reduce_kernel", 64, 2048)               # 24 partial tiles + the gradients in and out
    assert b == "hbm" and w == 4.0 * (640 * 256 + 640) * 26
    assert bench.kernel_work("emdf_k_kernel<1, true>", 64, 2048) is None          # priced as a stage (vector issue), not here
    # the north-star stage's bytes per pair
    assert 2 * (bench.knn_group_bytes(2048, 512, 32, 64) + bench.knn_group_bytes(512, 256, 32, 128)) == 19412992


def test_committed_profiles_carry_their_sidecars():
    """Every committed kernel summary of this round has a sidecar naming build, batch, points and attention mode; bench.py uses
    the file only when they match the run (here: when the sidecar's build is the current one), and the kernel the summary ranks
    first among those with a work model is one bench.py can price."""
    for which in ("kernel_stats", "kernel_stats_one_stream"):
        base = os.path.join(ROOT, "profiles", f"{bench.ROUND}_{which}")
        if not os.path.exists(base + ".csv"):
            continue
        meta = json.load(open(base + ".meta.json"))
        assert set(meta) >= {"build_id", "batch", "points", "attn", "command"}
        assert ("--one-stream" in meta["command"]) == which.endswith("one_stream")
        rows, why = bench.profile_rows(which, meta["batch"], meta["points"], meta["attn"])
        if meta["build_id"] != bench.build_id():
            assert rows is None and "another build" in why
            continue
        assert rows and why == ""
        names = [r["name"] for r in rows if bench.kernel_work(r["name"], meta["batch"], meta["points"]) is not None]
        assert names, "no row of the summary has a work model"
        avg = bench.profile_avg_us(rows, names[0].split("(")[0])
        assert avg and avg > 0
        with open(base + ".csv", newline="") as f:
            assert {"Name", "Calls", "AverageNs", "Percentage"} <= set(next(csv.DictReader(f)))
