"""The streamed-weights set-abstraction forward (salevel.hip) against the weight-stationary kernel on the same inputs:
bit-level comparison of the max and arg-max, and the launch time of each on the production shapes.

    python tools/check_sa_stream.py [B]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from puzzlenet_amd import _lib, ops  # noqa: E402


def run(B, N, S, C1, C2, dev):
    g = torch.Generator().manual_seed(C1 + C2 + N)
    P = torch.randn(B * N, C1, generator=g).to(dev)
    Q = (0.3 * torch.randn(B * S, C1, generator=g)).to(dev)
    idx = torch.randint(0, N, (B, S, 32), generator=g).to(dev)
    w2 = (torch.randn(C2, C1, generator=g) / C1 ** 0.5).to(dev)
    b2 = (0.1 * torch.randn(C2, generator=g)).to(dev)
    lib = _lib.load()
    st = torch.cuda.current_stream().cuda_stream
    R = B * S
    outs = []
    ws = torch.empty(lib.pzn_sa_level_fwd_workspace_bytes(C1, C2), dtype=torch.uint8, device=dev)
    for use_ws in (False, True):
        out = torch.empty(R, C2, device=dev)
        arg = torch.empty(R, C2, dtype=torch.int32, device=dev)

        def launch():
            if use_ws:
                _lib.call("pzn_sa_level_fwd_ws_f32", P.data_ptr(), Q.data_ptr(), idx.data_ptr(), w2.data_ptr(), b2.data_ptr(),
                          B, N, S, C1, C2, out.data_ptr(), arg.data_ptr(), ws.data_ptr(), st)
            else:
                _lib.call("pzn_sa_level_fwd_f32", P.data_ptr(), Q.data_ptr(), idx.data_ptr(), w2.data_ptr(), b2.data_ptr(),
                          B, N, S, C1, C2, out.data_ptr(), arg.data_ptr(), st)
        launch()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            launch()
        e1.record()
        torch.cuda.synchronize()
        outs.append((out, arg, e0.elapsed_time(e1) / 20))
    (o0, a0, t0), (o1, a1, t1) = outs
    # float64 restatement of 64 sampled groups
    gs = torch.randint(0, R, (64,), generator=g)
    worst = 0.0
    for gi in gs.tolist():
        b = gi // S
        rows = torch.relu(P[b * N + idx.view(R, 32)[gi]].double() + Q[gi].double())
        ref = torch.relu(rows @ w2.double().T + b2.double()).max(dim=0).values
        worst = max(worst, float((o1[gi].double() - ref).abs().max() / (ref.abs().max() + 1e-30)))
    fl = 2.0 * R * 32 * C1 * C2
    print(f"B={B} N={N} S={S} C1={C1} C2={C2}: stationary {t0 * 1e3:.0f} us ({fl / t0 / 1e9:.0f} TF/s)  streamed {t1 * 1e3:.0f} us "
          f"({fl / t1 / 1e9:.0f} TF/s)  max|out diff| {float((o0 - o1).abs().max()):.2e}  arg-max differs at "
          f"{int((a0 != a1).sum())} of {a0.numel()}  vs fp64 {worst:.2e}", flush=True)


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    dev = torch.device("cuda:0")
    run(2, 64, 20, 128, 128, dev)          # ragged: 40 groups, the last round half empty
    run(3, 64, 21, 128, 128, dev)          # 63 groups: an odd count (the last pair's second group is its first)
    run(B, 2048, 512, 128, 128, dev)
    run(B, 512, 256, 256, 256, dev)
    run(B, 4096, 512, 128, 128, dev)


if __name__ == "__main__":
    main()
