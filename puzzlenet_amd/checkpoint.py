"""Checkpoint interop: load what the reference's Lightning trainer saves.

A reference checkpoint (`train.py:119-124`, loaded by `test.py:110-112` with
`TouchedRegraster.load_from_checkpoint`) is a `torch.save`d dict with
  "state_dict"        -> the module's parameters and BatchNorm buffers, keys as in the reference model
                         (`Encoder.mlp1.weight`, ..., `tfMLP.0.weight`, `dt`; manifest pinned by
                         tests/golden/eval.npz), and
  "hyper_parameters"  -> {"config": <argparse.Namespace>} (`save_hyperparameters()`, model5_b.py:522).
`TouchedRegraster` here keeps the reference's parameter names and shapes, so the state dict loads as is; the only
field the reference config lacks is `num_points` (every hard-wired 1024 of the reference follows it).
`save_reference_checkpoint` writes the same layout, so weights trained here go back to the reference's eval driver.

Files are read with `torch.load(weights_only=True)` and `argparse.Namespace` allow-listed (the only non-tensor class
a reference checkpoint needs): unpickling arbitrary objects from a checkpoint path executes code, so that takes an
explicit `trusted=True`.
"""
import argparse

import torch


def read_reference_checkpoint(path_or_dict, map_location="cpu", trusted=False):
    """-> (state_dict, config or None).  Accepts a path or an already loaded dict; also accepts a bare state dict.
    trusted=True: full unpickling (only for files you wrote yourself: a Lightning checkpoint with callback state or
    custom classes in it needs that)."""
    ck = path_or_dict
    if not isinstance(ck, dict):
        if trusted:
            ck = torch.load(path_or_dict, map_location=map_location, weights_only=False)
        else:
            with torch.serialization.safe_globals([argparse.Namespace]):
                ck = torch.load(path_or_dict, map_location=map_location, weights_only=True)
    if "state_dict" not in ck:
        return ck, None
    hp = ck.get("hyper_parameters") or {}
    cfg = hp.get("config") if isinstance(hp, dict) else getattr(hp, "config", None)
    return ck["state_dict"], cfg


def load_reference_checkpoint(model, path_or_dict, strict=True, trusted=False):
    """Copy a reference checkpoint into `model` (a puzzlenet_amd.model5_b.TouchedRegraster); returns the config stored
    in the checkpoint (or None).  Parameter storage is updated in place, so flat-buffer views (FlatAdam) stay valid."""
    sd, cfg = read_reference_checkpoint(path_or_dict, trusted=trusted)
    missing, unexpected = model.load_state_dict(sd, strict=False)
    if strict and (missing or unexpected):
        raise RuntimeError(f"checkpoint does not match the model: missing {list(missing)}, unexpected {list(unexpected)}")
    return cfg


def build_from_reference_checkpoint(path_or_dict, num_points=None, device=None, trusted=False):
    """TouchedRegraster built from the checkpoint's own config (+ num_points: BatchNorm width = points per cloud,
    taken from the checkpoint's bn1 buffers when not given)."""
    from . import model5_b
    sd, cfg = read_reference_checkpoint(path_or_dict, trusted=trusted)
    if cfg is None:
        raise ValueError("checkpoint carries no hyper_parameters.config; build the model yourself and use load_reference_checkpoint")
    if num_points is None:
        num_points = int(sd["Encoder.bn1.weight"].shape[0])
    if not hasattr(cfg, "num_points"):
        cfg.num_points = num_points
    model = model5_b.TouchedRegraster(cfg)
    load_reference_checkpoint(model, {"state_dict": sd})
    return model.to(device) if device is not None else model


def save_reference_checkpoint(model, config, path, epoch=0, global_step=0):
    """Write `model` in the layout the reference's `TouchedRegraster.load_from_checkpoint` (test.py:110-112) reads:
    "state_dict" with the reference's keys (CPU tensors) and "hyper_parameters" = {"config": argparse.Namespace} without
    the one field the reference does not know (`num_points`), plus the bookkeeping keys Lightning expects to find."""
    if isinstance(config, dict):
        fields = dict(config)
    else:      # Namespace, or a class used as one (bench.Cfg, model_ref.Cfg): class attributes, then instance attributes
        fields = {k: getattr(config, k) for k in dir(config) if not k.startswith("_") and not callable(getattr(config, k))}
    fields.pop("num_points", None)
    ck = {"epoch": int(epoch), "global_step": int(global_step), "pytorch-lightning_version": "1.5.10",
          "state_dict": {k: v.detach().cpu().clone() for k, v in model.state_dict().items()},
          "hyper_parameters": {"config": argparse.Namespace(**fields)}}
    torch.save(ck, path)
    return path
