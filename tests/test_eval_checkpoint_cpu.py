"""CPU: eval-path host glue (puzzlenet_amd/metrics.py) and checkpoint interop (puzzlenet_amd/checkpoint.py) against
fixtures produced by the reference itself (tests/golden/make_golden_eval.py -> eval.npz): its metrics.py on seeded
poses, TouchedRegraster.compute_metrics, and the manifest of TouchedRegraster.state_dict().  SURVEY §8 rows f3, f4."""
import argparse
import os

import numpy as np
import torch

from oracle import model_ref as mr
from puzzlenet_amd import checkpoint, metrics
from puzzlenet_amd import model5_b as mb


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def test_metrics_match_reference(golden_eval):
    G = golden_eval
    R, t, igt = _t(G["m_R"]), _t(G["m_t"]), _t(G["m_igt"])
    inv_R, inv_t = metrics.inv_R_t(igt[:, :3, :3], igt[:, :3, 3])
    np.testing.assert_allclose(inv_R.numpy(), G["m_inv_R"], rtol=0, atol=1e-7)
    np.testing.assert_allclose(inv_t.numpy(), G["m_inv_t"], rtol=0, atol=1e-7)
    r_mse, r_mae = metrics.anisotropic_R_error(R, inv_R)
    t_mse, t_mae = metrics.anisotropic_t_error(t, inv_t)
    np.testing.assert_allclose(r_mse, G["m_r_mse"], rtol=1e-6)
    np.testing.assert_allclose(r_mae, G["m_r_mae"], rtol=1e-6)
    np.testing.assert_allclose(t_mse, G["m_t_mse"], rtol=1e-6)
    np.testing.assert_allclose(t_mae, G["m_t_mae"], rtol=1e-6)
    np.testing.assert_allclose(metrics.isotropic_R_error(R, inv_R).numpy(), G["m_r_iso"], rtol=1e-5, atol=1e-4)
    np.testing.assert_allclose(metrics.isotropic_t_error(t, inv_t, inv_R).numpy(), G["m_t_iso"], rtol=1e-5)


def test_compute_metrics_matches_reference(golden_eval):
    G = golden_eval
    m = mb.TouchedRegraster(mr.Cfg())
    got = m.compute_metrics(_t(G["m_R"])[:4], _t(G["m_t"])[:4], _t(G["m_igt"])[:4])
    for name, v in zip(("r_mse", "r_mae", "t_mse", "t_mae", "r_iso", "t_iso"), got):
        v = v.numpy() if isinstance(v, torch.Tensor) else v
        np.testing.assert_allclose(v, G["cm_" + name], rtol=1e-5, atol=1e-4, err_msg=name)


def test_state_dict_manifest_equals_reference(golden_eval):
    """What a reference Lightning checkpoint holds under "state_dict": same names, order, shapes, dtypes."""
    G = golden_eval
    sd = mb.TouchedRegraster(mr.Cfg()).state_dict()
    assert list(sd.keys()) == [str(n) for n in G["sd_names"]]
    assert [",".join(str(int(s)) for s in v.shape) for v in sd.values()] == [str(s) for s in G["sd_shapes"]]
    assert [str(v.dtype) for v in sd.values()] == [str(d) for d in G["sd_dtypes"]]


def test_lightning_checkpoint_round_trip(tmp_path):
    """A file in the reference trainer's format (state_dict + hyper_parameters.config as an argparse.Namespace)
    loads through checkpoint.load_reference_checkpoint / build_from_reference_checkpoint, buffers included."""
    src = mb.TouchedRegraster(mr.Cfg())
    mr.fill_params(src)
    with torch.no_grad():
        src.Encoder.bn1.running_mean.add_(0.25)
        src.Encoder2.bn2.num_batches_tracked.add_(7)
    ns = argparse.Namespace(dataset="cad", loss_mode=1, loss_sum=False, use_emd2=False, use_cd2=False, use_emd3=False,
                            pretrain_epochs=0, lr=0.9e-3, m="ckpt", output_path=str(tmp_path))
    path = os.path.join(tmp_path, "epoch=3.ckpt")
    torch.save({"epoch": 3, "global_step": 100, "state_dict": src.state_dict(), "hyper_parameters": {"config": ns}}, path)

    dst = mb.TouchedRegraster(mr.Cfg())
    before = dst.Encoder.mlp1.weight.data_ptr()
    cfg = checkpoint.load_reference_checkpoint(dst, path)
    assert cfg.m == "ckpt" and dst.Encoder.mlp1.weight.data_ptr() == before      # in place
    for (k, a), (_, b) in zip(src.state_dict().items(), dst.state_dict().items()):
        assert torch.equal(a, b), k
    built = checkpoint.build_from_reference_checkpoint(path)
    assert built.num_points == 1024 and built.C.output_path == str(tmp_path)
    assert torch.equal(built.Encoder2.bn2.num_batches_tracked, src.Encoder2.bn2.num_batches_tracked)
    # a bare state dict works too, and a wrong one is refused
    assert checkpoint.load_reference_checkpoint(dst, src.state_dict()) is None
    bad = dict(src.state_dict())
    bad.pop("dt")
    try:
        checkpoint.load_reference_checkpoint(dst, bad)
        raise AssertionError("missing key accepted")
    except RuntimeError:
        pass


def test_save_reference_checkpoint_round_trip(tmp_path):
    """checkpoint.save_reference_checkpoint writes what the reference's load_from_checkpoint reads (state_dict with the
    reference's keys + hyper_parameters.config as an argparse.Namespace); the file loads back under weights_only=True."""
    src = mb.TouchedRegraster(mr.Cfg(num_points=1024, m="saved"))
    mr.fill_params(src)
    path = checkpoint.save_reference_checkpoint(src, src.C, os.path.join(tmp_path, "out.ckpt"), epoch=2, global_step=40)
    raw = torch.load(path, map_location="cpu", weights_only=False)
    assert set(raw) >= {"state_dict", "hyper_parameters", "epoch", "global_step"}
    assert isinstance(raw["hyper_parameters"]["config"], argparse.Namespace)
    assert not hasattr(raw["hyper_parameters"]["config"], "num_points") and raw["hyper_parameters"]["config"].m == "saved"
    assert raw["hyper_parameters"]["config"].loss_mode == 1 and raw["hyper_parameters"]["config"].lr == 0.9e-3
    assert list(raw["state_dict"].keys()) == list(src.state_dict().keys())
    built = checkpoint.build_from_reference_checkpoint(path)                 # the safe (weights_only) reader
    for (k, a), (_, b) in zip(src.state_dict().items(), built.state_dict().items()):
        assert torch.equal(a, b), k


class Payload:
    pass


def test_untrusted_pickles_are_refused(tmp_path):
    """A checkpoint that smuggles an arbitrary class is not unpickled unless the caller says the file is trusted."""
    path = os.path.join(tmp_path, "evil.ckpt")
    torch.save({"state_dict": {}, "hyper_parameters": {"config": Payload()}}, path)
    try:
        checkpoint.read_reference_checkpoint(path)
        raise AssertionError("arbitrary class unpickled")
    except Exception as e:      # noqa: BLE001
        assert "Payload" in str(e) or "Unsupported" in str(e) or "weights_only" in str(e), e
