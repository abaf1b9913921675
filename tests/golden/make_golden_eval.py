#!/usr/bin/env python3
"""Eval-path and checkpoint-interop golden fixtures: runs the REFERENCE on CPU.

    python tests/golden/make_golden_eval.py      ->  tests/golden/eval.npz

* the reference's `metrics.py` functions and `TouchedRegraster.compute_metrics` on seeded poses;
* `TouchedRegraster.test_step` (model5_b.py:1279-1366) on the seeded B=4, N=1024 batch of model.npz with the
  closed-form parameter fill (10 scores);
* the manifest of `TouchedRegraster(config).state_dict()` (names, shapes, dtypes) = what a Lightning `.ckpt`
  of the reference holds under "state_dict".
Same harness rules as make_golden_model.py (sys.modules placeholders only, nothing of the reference is copied).
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_golden_model as gm  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def main():
    mb = gm.import_reference_model()
    import metrics as ref_metrics
    import se_math.se3 as se3
    torch.set_num_threads(8)
    rng = np.random.default_rng(777)
    G = {}

    # ---- metrics.py on seeded poses (B = 6): predicted (R, t) and ground-truth igt
    x = torch.from_numpy((0.7 * rng.standard_normal((6, 6))).astype(np.float32))
    y = torch.from_numpy((0.7 * rng.standard_normal((6, 6))).astype(np.float32))
    pred, igt = se3.exp(x), se3.exp(y)
    R, t = pred[:, :3, :3].contiguous(), pred[:, :3, 3].contiguous()
    G["m_R"], G["m_t"], G["m_igt"] = R.numpy(), t.numpy(), igt.numpy()
    inv_R, inv_t = ref_metrics.inv_R_t(igt[:, :3, :3], igt[:, :3, 3])
    G["m_inv_R"], G["m_inv_t"] = inv_R.numpy(), inv_t.numpy()
    r_mse, r_mae = ref_metrics.anisotropic_R_error(R, inv_R)
    t_mse, t_mae = ref_metrics.anisotropic_t_error(t, inv_t)
    G["m_r_mse"], G["m_r_mae"], G["m_t_mse"], G["m_t_mae"] = r_mse, r_mae, t_mse, t_mae
    G["m_r_iso"] = ref_metrics.isotropic_R_error(R, inv_R).numpy()
    G["m_t_iso"] = ref_metrics.isotropic_t_error(t, inv_t, inv_R).numpy()

    # ---- compute_metrics + test_step on the model.npz batch (B=4, N=1024), eval mode, closed-form fill
    mb.pl.LightningModule.device = property(lambda self: torch.device("cpu"))
    model = mb.TouchedRegraster(gm.Cfg())
    gm.fill_params(model)
    M = np.load(os.path.join(OUT, "model.npz"))
    batch = [torch.from_numpy(M[f"p5_batch{i}"]) for i in range(8)]
    cm = model.compute_metrics(R[:4], t[:4], igt[:4])
    for name, v in zip(("r_mse", "r_mae", "t_mse", "t_mae", "r_iso", "t_iso"), cm):
        G["cm_" + name] = np.asarray(v.detach().numpy() if isinstance(v, torch.Tensor) else v)
    for e in (model.Encoder, model.Encoder2):
        e.bn1.reset_running_stats()
        e.bn2.reset_running_stats()
    model.eval()
    torch.manual_seed(2024)
    with torch.no_grad():
        scores = model.test_step(batch, 0)
    G["ts_scores"] = scores.numpy()
    G["ts_seed"] = np.array([2024])

    # ---- state_dict manifest (checkpoint interop)
    sd = model.state_dict()
    G["sd_names"] = np.array(list(sd.keys()))
    G["sd_shapes"] = np.array([",".join(str(int(s)) for s in v.shape) for v in sd.values()])
    G["sd_dtypes"] = np.array([str(v.dtype) for v in sd.values()])

    np.savez_compressed(os.path.join(OUT, "eval.npz"), **G)
    print("eval.npz:", len(G), "arrays,", os.path.getsize(os.path.join(OUT, "eval.npz")) >> 10, "KiB")
    print("test_step scores:", G["ts_scores"])


if __name__ == "__main__":
    main()
