"""Per-launch table of one fused EMD call (pzn_emd_fused_f32): executed pair evaluations from the device's per-launch
counters and, from a rocprofv3 kernel trace of the same process, duration and the idle gap in front of every launch.

    cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $O/emdlv -- python3 $R/tools/emd_levels.py run 64 2048 rigid
    python3 tools/emd_levels.py parse $O/emdlv gpurun_out/emd_levels_counts.json

Regimes: `indep` two independent uniform clouds; `rigid` a uniform cloud against its copy moved by a twist of norm 0.8 (what
the loss term sees with an untrained pose head: `earth_mover_distance(de_mrpc, rpc)`, model5_b.py:1002); `near` the copy
moved by a twist of norm 0.08 (a trained pose head)."""
import csv, glob, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def launch_names():
    names = ["A(7)"]
    for j in range(7, -3, -1):
        names += ["B(%d)" % j, ("CA(%d)" % j) if j > -2 else "C(-2)"]
    return names


def run(B, n, regime, out):
    import torch
    from puzzlenet_amd import ops, se3
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    a = torch.rand(B, n, 3, generator=g).to(dev)
    if regime == "indep":
        b = torch.rand(B, n, 3, generator=g).to(dev)
    else:
        x = torch.randn(B, 6, generator=g)
        x = (0.8 if regime == "rigid" else 0.08) * x / x.norm(dim=1, keepdim=True)
        T = se3.exp(x.to(dev))
        b = se3.transform(T, a.permute(0, 2, 1)).permute(0, 2, 1).contiguous()
    ops.EMD_WALK_STATS = []
    for _ in range(3):
        cost = ops.emd_fused(b, a)
    torch.cuda.synchronize()
    ctr = ops.EMD_WALK_STATS[-1][0].cpu().view(32, -1).sum(1).tolist()
    full = float(B) * n * n
    res = {"B": B, "n": n, "regime": regime, "cost_mean": float(cost.mean()),
           "evals": {nm: 64.0 * c for nm, c in zip(launch_names(), ctr)}, "full_pass": full}
    json.dump(res, open(out, "w"))
    print("total executed evaluations %.3f G = %.2f full passes" % (64.0 * sum(ctr) / 1e9, 64.0 * sum(ctr) / full))


def parse(d, counts):
    res = json.load(open(counts))
    f = glob.glob(d + "/*/*kernel_trace.csv")[0]
    rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
    rows.sort()
    rows = [r for r in rows if "emd" in r[2]]
    # the last call: from the last emd_sort_x_kernel on
    i0 = max(i for i, r in enumerate(rows) if "emd_sort_x" in r[2])
    call = rows[i0:]
    t0, t1 = call[0][0], call[-1][1]
    print("%s  B=%d n=%d  call %.1f us wall, %.1f us of kernels, %d launches" % (
        res["regime"], res["B"], res["n"], (t1 - t0) / 1e3, sum(e - s for s, e, _ in call) / 1e3, len(call)))
    names = iter(launch_names())
    prev = None
    full = res["full_pass"]
    print("%-10s %-26s %9s %8s %10s %12s" % ("launch", "kernel", "us", "gap us", "evals/full", "Gevals/s"))
    for s, e, k in call:
        k = k.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:26]
        nm = next(names) if ("emdf_k" in k or "emdf_b" in k) else "-"
        ev = res["evals"].get(nm, 0.0)
        print("%-10s %-26s %9.1f %8.1f %10.3f %12.1f" % (nm, k, (e - s) / 1e3, 0.0 if prev is None else (s - prev) / 1e3,
                                                       ev / full, ev / max(1, e - s)))
        prev = e


def pmc(d):
    """per-dispatch counter table of the last fused call in a `rocprofv3 --pmc ... --kernel-trace` dump of `run`"""
    import collections
    f = glob.glob(d + "/*/*counter_collection.csv")[0]
    disp = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        if "emd" not in r["Kernel_Name"]:
            continue
        e = disp.setdefault(int(r["Dispatch_Id"]), {"name": r["Kernel_Name"], "t": int(r["End_Timestamp"]) - int(r["Start_Timestamp"])})
        e[r["Counter_Name"]] = e.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    ids = sorted(disp)
    i0 = max(i for i in ids if "emd_sort_x" in disp[i]["name"])
    call = [disp[i] for i in ids if i >= i0]
    ctrs = [c for c in call[0] if c not in ("name", "t")]
    names = iter(launch_names())
    print("%-8s %-24s %8s | " % ("launch", "kernel", "us") + " ".join("%14s" % c[-14:] for c in ctrs))
    for e in call:
        k = e["name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:24]
        nm = next(names) if ("emdf_k" in k or "emdf_b" in k) else "-"
        print("%-8s %-24s %8.1f | " % (nm, k, e["t"] / 1e3) + " ".join("%14.5g" % e.get(c, 0.0) for c in ctrs))


if __name__ == "__main__":
    if sys.argv[1] == "run":
        B, n, regime = int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
        out = sys.argv[5] if len(sys.argv) > 5 else os.path.join(ROOT, "gpurun_out", "emd_levels_counts.json")
        os.makedirs(os.path.dirname(out), exist_ok=True)
        run(B, n, regime, out)
    elif sys.argv[1] == "pmc":
        pmc(sys.argv[2])
    else:
        parse(sys.argv[2], sys.argv[3])
