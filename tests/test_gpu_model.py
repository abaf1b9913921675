"""GPU parity of the model path (puzzlenet_amd.model5_b drop-in) against the reference's
own outputs (tests/golden/model.npz, loss.npz).  Tolerance: fp32 pose / logits / loss within
1e-4 relative (BASELINE.json north_star); FPS-selected points bit-exact."""
import os

import numpy as np
import pytest
import torch

from oracle import model_ref as mr

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _close(got, want, rtol=1e-4, atol=1e-5, msg=""):
    np.testing.assert_allclose(got.detach().cpu().numpy() if isinstance(got, torch.Tensor) else got, want,
                               rtol=rtol, atol=atol, err_msg=msg)


def test_state_dict_names_match_reference(golden_loss, golden_model):
    from puzzlenet_amd import model5_b as mb
    m = mb.TouchedRegraster(mr.Cfg())
    assert sorted(n for n, _ in m.named_parameters()) == list(golden_loss["ts0_grad_names"])
    assert sum(p.numel() for p in m.parameters()) == int(golden_model["n_params"][0])
    # a state_dict of the (reference-named) oracle model loads strictly
    m.load_state_dict(mr.RefModel(mr.Cfg()).state_dict(), strict=True)


def test_layer_attention(golden_model, dev):
    from puzzlenet_amd import model5_b as mb
    G = golden_model
    att = mb.layerAttention(mr.Cfg(), 256)
    mr.fill_params(att)
    att.to(dev)
    x = _t(G["att_x"], dev).requires_grad_(True)
    r, a = att(x)
    _close(r, G["att_r"])
    _close(a, G["att_a"], atol=1e-6)
    ((r * _t(G["att_wr"], dev)).sum() + (a * _t(G["att_wa"], dev)).sum()).backward()
    _close(x.grad, G["att_gx"], rtol=1e-4, atol=2e-5)
    for n, p in att.named_parameters():
        _close(p.grad, G["att_g_" + n], rtol=1e-4, atol=2e-4, msg=n)


@pytest.mark.parametrize("N", [1024, 2048])
@pytest.mark.parametrize("mode", ["train", "eval"])
def test_encoder(golden_model, dev, N, mode):
    from puzzlenet_amd import model5_b as mb
    G = golden_model
    enc = mb.PCTransformer_nonsort(mr.Cfg(), num_points=N)
    mr.fill_params(enc)
    enc.to(dev).train(mode == "train")
    torch.manual_seed(1000 + N)
    with torch.no_grad():
        f_global, x2, attention, out, xf = enc(_t(G[f"enc{N}_xyz"], dev))
    tag = f"enc{N}_{mode}_"
    assert np.array_equal(x2.cpu().numpy(), G[tag + "x2"])          # same 256 points as the reference, bit for bit
    _close(xf[:, ::8], G[tag + "x_feature_s8"])
    _close(attention[:, ::8], G[tag + "attention_s8"], atol=1e-6)
    _close(f_global, G[tag + "f_global"], atol=1e-4)
    _close(out[:, ::16, ::8], G[tag + "out_sample"], atol=1e-4)
    if mode == "train":
        _close(enc.bn1.running_mean, G[tag + "bn1_running_mean"])
        _close(enc.bn1.running_var, G[tag + "bn1_running_var"])


@pytest.mark.parametrize("mode", ["eval", "train"])
def test_predict5(golden_model, dev, mode):
    from puzzlenet_amd import model5_b as mb
    G = golden_model
    m = mb.TouchedRegraster(mr.Cfg())
    mr.fill_params(m)
    m.to(dev)
    batch = [_t(G[f"p5_batch{i}"], dev) for i in range(8)]
    torch.manual_seed(2024)
    with torch.no_grad():
        out = m.predict5(batch, 4, need=True, training=(mode == "train"))
    _close(out[0], G[f"p5_{mode}_out"])                                   # pose twist [4,6]
    g_ref = mr.se3_exp(torch.from_numpy(G[f"p5_{mode}_out"]))
    from puzzlenet_amd import se3
    _close(se3.exp(out[0]), g_ref.numpy())                                 # SE(3) [4,4,4]
    assert np.array_equal(out[2].cpu().numpy(), G[f"p5_{mode}_x2"])
    assert np.array_equal(out[4].cpu().numpy(), G[f"p5_{mode}_mrpc_x2"])
    _close(out[3][:, ::8], G[f"p5_{mode}_attention_s8"], atol=1e-6)
    _close(out[6], G[f"p5_{mode}_de_fpcb"], atol=1e-4)                     # boundary logits [4,2,1024]
    _close(out[7], G[f"p5_{mode}_de_mrpcb"], atol=1e-4)
    o4 = m.predict5(batch, 4, need=False, training=False)
    assert len(o4) == 4 and o4[0].shape == (4, 6) and o4[2].shape == (4, 2, 1024)


def test_se3_chamfer_comp(golden_loss, dev):
    from puzzlenet_amd import model5_b as mb, se3
    G = golden_loss
    tw = _t(G["se3_twist"], dev).requires_grad_(True)
    g = se3.exp(tw)
    _close(g, G["se3_exp"], rtol=1e-5, atol=1e-6)
    _close(se3.transform(g.detach(), _t(G["se3_pts"], dev)), G["se3_transform"], rtol=1e-5, atol=1e-6)
    (g * _t(G["se3_w"], dev)).sum().backward()
    _close(tw.grad, G["se3_exp_grad"], rtol=1e-4, atol=1e-5)
    m = mb.TouchedRegraster(mr.Cfg())
    a, b = _t(G["cd_a"], dev).requires_grad_(True), _t(G["cd_b"], dev).requires_grad_(True)
    d1, d2 = m.chamfer_loss(a, b)
    _close(d1, G["cd_d1"], rtol=1e-4, atol=2e-6)
    _close(d2, G["cd_d2"], rtol=1e-4, atol=2e-6)
    (d1.mean() + 2 * d2.mean()).backward()
    _close(a.grad, G["cd_ga"], rtol=1e-3, atol=1e-6)
    _close(b.grad, G["cd_gb"], rtol=1e-3, atol=1e-6)
    _close(m.comp(_t(G["comp_g"], dev), _t(G["comp_igt"], dev)), G["comp_out"], rtol=1e-5)


@pytest.mark.parametrize("loss_mode", [0, 1])
def test_training_step_vs_reference(golden_loss, dev, loss_mode):
    """Loss value and per-parameter gradients of one whole training_step against the reference's
    own training_step run on CPU.  loss_mode 0 has no EMD term in the loss (tight); loss_mode 1
    adds all four EMD terms, whose gradients carry the auction's conditioning (see test_gpu_emd)."""
    from puzzlenet_amd import model5_b as mb
    G = golden_loss
    flags = {} if loss_mode == 0 else dict(use_emd2=True, use_cd2=True, use_emd3=True)
    m = mb.TouchedRegraster(mr.Cfg(loss_mode=loss_mode, **flags))
    mr.fill_params(m)
    m.to(dev)
    batch = [_t(G[f"ts_batch{i}"], dev) for i in range(8)]
    torch.manual_seed(99)
    loss = m.training_step(batch, 0)["loss"]
    tag = f"ts{loss_mode}_"
    np.testing.assert_allclose(loss.item(), G[tag + "loss"][0], rtol=1e-4)
    loss.backward()
    params = dict(m.named_parameters())
    # Gradients pass through arg-min (chamfer) and top-k selections: a near-tie that resolves
    # differently under another GEMM summation order moves a few rows of gradient, so per-tensor
    # norms are held to 1 % (3 % with the four EMD terms), with a noise floor for tensors whose
    # true gradient is zero (e.g. a bias in front of BatchNorm).
    rt_norm, rt_s = (1e-2, 3e-2) if loss_mode == 0 else (3e-2, 6e-2)
    total = float(np.sqrt((G[tag + "grad_norms"] ** 2).sum()))
    for name, norm, samp in zip(G[tag + "grad_names"], G[tag + "grad_norms"], G[tag + "grad_samples"]):
        p = params[str(name)]
        g = p.grad if p.grad is not None else torch.zeros_like(p)
        np.testing.assert_allclose(g.norm().item(), norm, rtol=rt_norm, atol=1e-5 * total, err_msg=str(name))
        flat = g.flatten()
        idx = torch.linspace(0, flat.numel() - 1, 8).long().to(dev)
        rms = float(norm) / max(1.0, flat.numel() ** 0.5)
        np.testing.assert_allclose(flat[idx].cpu().numpy(), samp, rtol=rt_s, atol=1e-6 * total + 0.05 * rms,
                                   err_msg=str(name))


def _device_winners(m, capture):
    """ops.WINNER_CAPTURE of one forward of the product model -> the keys oracle.model_ref.RefModel.pin_winners takes."""
    from puzzlenet_amd import _lib
    if _lib.load().pzn_gemm_get_precision() == 0:
        # (PZN_GEMM_PRECISION=f32: the encoders' global max then goes through the plain max over points, whose captures carry no
        # owner - three of them could not be told apart below; the fingerprint test covers that mode)
        pytest.skip("the winner pinning needs the split-precision paths' captures (default mode)")
    owner = {}
    for tag, enc in (("Encoder.", m.Encoder), ("Encoder2.", m.Encoder2)):
        owner[enc.mlp4.weight.data_ptr()] = tag + "sa1"
        owner[enc.mlp6.weight.data_ptr()] = tag + "sa2"
        owner[enc.out.weight.data_ptr()] = tag + "gmax"
    win = {}
    for kind, ptr, arg in capture:
        key = "heads.gmax" if kind == "maxpts" else owner[ptr]
        assert key not in win, key
        win[key] = arg.detach().cpu().to(torch.long)
    assert len(win) == 7, sorted(win)
    return win


def _check_flips(pins, max_flips=16, max_gap=1e-5):
    """The oracle's own arg-max differs from the pinned (device) winner only on near-ties: a handful of entries, each with
    the two candidates closer than fp32 rounding of the sums in front of them."""
    total = 0
    for key, (n, gap, of) in sorted(pins["flips"].items()):
        total += n
        assert gap <= max_gap, (key, n, gap)
    print("max-pool winners that differ from the oracle's own:", {k: v[0] for k, v in pins["flips"].items() if v[0]}, "of",
          sum(v[2] for v in pins["flips"].values()))
    assert total <= max_flips, pins["flips"]
    return total


@pytest.mark.parametrize("loss_mode", [0, 1])
def test_training_step_full_gradients_vs_oracle(golden_loss, dev, loss_mode):
    """The whole gradient of one training_step, every entry of every parameter, against the torch-CPU restatement run
    on the same batch (oracle/model_ref.py, itself pinned to the reference's training_step by tests/test_oracle_model.py):
    the fingerprint test above holds norms and 8 samples per tensor to percents; this one holds the full 8,059,220-entry
    gradient to 2e-4 in L2 and every tensor to 1e-2 of its own norm, above a floor of 1e-6 of the total (the key biases of
    the attention blocks have a mathematically zero gradient: noise on both sides).
    The function is not continuous: the global max over points sends a channel's whole gradient through ONE point, and a
    max-pool winner that is ahead by less than fp32 rounding may fall the other way under a different summation order - one
    such flip on a heavy path moves ~1 % of one encoder's gradients (tests/_grad_rows.py prints the rows).  The comparison is
    therefore made flip-aware instead of loose: the product's arg-max tensors (both set-abstraction levels and the global max
    of both encoders, the heads' max: ops.WINNER_CAPTURE) are handed to the oracle, which evaluates the function OF THOSE
    WINNERS (model_ref._pool) and reports where its own arg-max differs: at most a handful of entries (printed), each a
    near-tie (gap <= 1e-5 of the tensor's largest maximum) - and then the old bounds hold for every tensor."""
    from puzzlenet_amd import model5_b as mb, ops
    G = golden_loss
    flags = {} if loss_mode == 0 else dict(use_emd2=True, use_cd2=True, use_emd3=True)
    cfg = mr.Cfg(loss_mode=loss_mode, **flags)
    ops.clear_grad_sinks()
    m = mb.TouchedRegraster(cfg)
    mr.fill_params(m)
    ref = mr.RefModel(cfg)
    ref.load_state_dict(m.state_dict(), strict=True)
    m.to(dev)
    batch = [_t(G[f"ts_batch{i}"], dev) for i in range(8)]
    torch.manual_seed(99)
    ops.WINNER_CAPTURE = []
    try:
        loss = m.training_step(batch, 0)["loss"]
        capture, ops.WINNER_CAPTURE = ops.WINNER_CAPTURE, None
    finally:
        ops.WINNER_CAPTURE = None
    pins = ref.pin_winners(_device_winners(m, capture))
    torch.manual_seed(99)
    ref_loss = ref.training_step([t.cpu() for t in batch])
    ref_loss = ref_loss[0] if isinstance(ref_loss, tuple) else ref_loss
    _check_flips(pins)
    assert abs(loss.item() - ref_loss.item()) <= 1e-4 * abs(ref_loss.item())
    loss.backward()
    ref_loss.backward()
    rp = dict(ref.named_parameters())
    err2 = ref2 = 0.0
    rows = []
    for name, p in m.named_parameters():
        g = (p.grad if p.grad is not None else torch.zeros_like(p)).cpu().double()
        gr = rp[name].grad
        gr = (gr if gr is not None else torch.zeros_like(rp[name])).double()
        e, r = float((g - gr).norm()), float(gr.norm())
        err2, ref2 = err2 + e * e, ref2 + r * r
        rows.append((name, e, r))
    total = ref2 ** 0.5
    assert err2 ** 0.5 <= 2e-4 * total, (err2 ** 0.5 / total)
    for name, e, r in rows:
        assert e <= 1e-2 * r + 1e-6 * total, (name, e, r)


def test_gradient_sinks_match_autograd_accumulation(golden_loss, dev):
    """With the flat gradient bucket registered as sinks (what bench.py / engine.TrainStep use), the
    weight-gradient kernels add straight into the bucket; the result must equal ordinary autograd
    accumulation, also over two backward passes without zeroing in between."""
    from puzzlenet_amd import distributed as pdist
    from puzzlenet_amd import model5_b as mb
    from puzzlenet_amd import ops
    G = golden_loss
    batch = [_t(G[f"ts_batch{i}"], dev) for i in range(8)]

    def run(use_sinks):
        m = mb.TouchedRegraster(mr.Cfg(loss_mode=1))
        mr.fill_params(m)
        m.to(dev)
        ops.clear_grad_sinks()
        bucket = pdist.FlatGradAllReduce(m.parameters()) if use_sinks else None
        if not use_sinks:
            ops.clear_grad_sinks()
        for _ in range(2):
            torch.manual_seed(99)
            m.training_step(batch, 0)["loss"].backward()
        grads = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).flatten() for p in m.parameters()])
        if use_sinks:
            assert len(ops._GRAD_SINKS) == sum(1 for _ in m.parameters())
            packed = torch.cat([bucket.flat[o:o + p.numel()] for p, o in zip(bucket.params, bucket.offsets)])
            assert torch.equal(grads, packed)                    # p.grad ARE the (16-byte aligned) bucket slots
        ops.clear_grad_sinks()
        return grads

    a, b = run(True), run(False)
    assert float((a - b).norm() / b.norm()) < 2e-3        # (EMD terms in the loss: see test_gpu_emd)
    assert float(a.abs().max()) > 0


def test_emd_side_stream_and_deferred_root(golden_loss, dev):
    """training_step runs the N x N EMD on the side stream; with `defer_emd_loss` (engine.TrainStep, eager mode) the EMD
    term is a separate backward root and the caller sums the parts.  Loss and gradients must be those of the
    sequential step (one stream, one root)."""
    from puzzlenet_amd import model5_b as mb
    from puzzlenet_amd import ops
    G = golden_loss
    batch = [_t(G[f"ts_batch{i}"], dev) for i in range(8)]

    def run(two_streams, defer):
        m = mb.TouchedRegraster(mr.Cfg(loss_mode=1, use_emd2=True, use_cd2=True, use_emd3=True))
        mr.fill_params(m)
        m.to(dev)
        ops.clear_grad_sinks()
        m.two_streams = two_streams
        m.defer_emd_loss = defer
        torch.manual_seed(5)
        out = m.training_step(batch, 0)
        if "loss" in out:
            assert not defer
            out["loss"].backward()
            loss = out["loss"].detach()
        else:
            assert defer and two_streams
            terms = list(out["loss_terms"])
            torch.autograd.backward(terms)
            torch.cuda.current_stream().wait_stream(out["join_stream"])
            loss = (terms[0] + terms[1]).detach()
        torch.cuda.synchronize()
        grads = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).flatten() for p in m.parameters()])
        return float(loss), grads

    l0, g0 = run(False, False)
    l1, g1 = run(True, False)
    l2, g2 = run(True, True)
    # (the pose head's few-row layers sum their K splits with atomics: the forward itself repeats to ~1e-6 only)
    assert abs(l1 - l0) <= 2e-5 * abs(l0) and abs(l2 - l0) <= 2e-5 * abs(l0)
    # (gradients carry the run-to-run rounding of the atomic weight-gradient sums; EMD terms: see test_gpu_emd)
    assert float((g1 - g0).norm() / g0.norm()) < 2e-3
    assert float((g2 - g0).norm() / g0.norm()) < 2e-3


def test_flat_adam_matches_torch_adam(dev):
    """distributed.FlatAdam (one pzn_adam_step_f32 launch over flat buffers, StepLR folded in) against
    torch.optim.Adam + StepLR on CPU, same gradients, 120 steps (crosses two scheduler boundaries)."""
    from puzzlenet_amd import distributed as pdist
    torch.manual_seed(3)
    ref = [torch.nn.Parameter(torch.randn(37, 5)), torch.nn.Parameter(torch.randn(11)), torch.nn.Parameter(torch.randn(4, 4, 3))]
    mine = [torch.nn.Parameter(p.detach().clone().to(dev)) for p in ref]
    grads = pdist.FlatGradAllReduce(mine)
    opt = pdist.FlatAdam(grads, 1e-3, sched_step=50, sched_gamma=0.999)
    ropt = torch.optim.Adam(ref, lr=1e-3)
    rsch = torch.optim.lr_scheduler.StepLR(ropt, 50, 0.999)
    g = torch.Generator().manual_seed(5)
    for _ in range(120):
        for p, q in zip(ref, mine):
            gr = torch.randn(p.shape, generator=g) * 0.1
            p.grad = gr.clone()
            q.grad.copy_(gr.to(dev))
        ropt.step(); rsch.step()
        opt.step()
    for p, q in zip(ref, mine):
        assert (q.detach().cpu() - p.detach()).abs().max() < 2e-6
        assert q.data_ptr() >= opt.flat.data_ptr()          # parameters live inside the flat buffer
    from puzzlenet_amd import ops
    ops.clear_grad_sinks()


def test_test_step_matches_reference(golden_model, golden_eval, dev, tmp_path):
    """Eval path: TouchedRegraster.test_step on the seeded B=4, N=1024 batch against the ten scores
    the reference's own test_step produced on CPU (tests/golden/make_golden_eval.py), and test_epoch_end's file."""
    from puzzlenet_amd import model5_b as mb
    G, E = golden_model, golden_eval
    cfg = mr.Cfg()
    cfg.output_path = str(tmp_path)
    m = mb.TouchedRegraster(cfg)
    mr.fill_params(m)
    m.to(dev).eval()
    batch = [_t(G[f"p5_batch{i}"], dev) for i in range(8)]
    torch.manual_seed(int(E["ts_seed"][0]))
    with torch.no_grad():
        scores = m.test_step(batch, 0)
    assert scores.shape == (1, 10)
    got, want = scores.cpu().numpy()[0], E["ts_scores"][0]
    np.testing.assert_allclose(got[:6], want[:6], rtol=2e-4, atol=1e-5)      # pose errors (Euler angles amplify)
    np.testing.assert_allclose(got[6:8], want[6:8], rtol=0, atol=1e-6)        # IoU of the top-128 boundary picks
    np.testing.assert_allclose(got[8:], want[8:], rtol=1e-4)                  # boundary chamfer distances
    # un-batched sample path + the metrics file
    single = [b[0] for b in batch]
    with torch.no_grad():
        s1 = m.test_step(single, 0)
    assert s1.shape == (1, 10) and torch.isfinite(s1).all()
    mean = m.test_epoch_end([scores, scores])
    np.testing.assert_allclose(mean.cpu().numpy(), got, rtol=1e-6)
    files = [f for f in os.listdir(tmp_path) if f.endswith("metrics.txt")]
    assert len(files) == 1
    lines = open(os.path.join(tmp_path, files[0])).read().splitlines()
    assert lines[0].startswith("r_mse,") and len(lines[1].split()) == 10


def test_se3_exp_kernel_vs_tensor_ops(dev):
    """csrc/se3.hip (forward and hand-derived backward) against the tensor-op restatement of se3.exp evaluated in
    float64 on the CPU with autograd: generic twists, |w| on both sides of the 0.01 series switch, w = 0."""
    from puzzlenet_amd import se3
    g = torch.Generator().manual_seed(11)
    x = torch.randn(40, 6, generator=g)
    x[8:16, :3] *= 1e-3                      # |w| ~ 1e-3: series branch
    x[16:20, :3] *= 0.004                    # around the switch
    x[20, :3] = 0                            # exact zero rotation
    x[21:24, :3] *= 3.0                      # large angles
    wgt = torch.randn(40, 4, 4, generator=g)
    xr = x.double().requires_grad_(True)
    gr = se3.exp(xr)
    (gr * wgt.double()).sum().backward()
    xd = x.to(dev).requires_grad_(True)
    gd = se3.exp(xd)
    assert gd.shape == (40, 4, 4)
    np.testing.assert_allclose(gd.detach().cpu().numpy(), gr.detach().numpy(), rtol=2e-6, atol=2e-7)
    (gd * wgt.to(dev)).sum().backward()
    np.testing.assert_allclose(xd.grad.cpu().numpy(), xr.grad.numpy(), rtol=1e-5, atol=2e-6)


def test_train_step_with_prefetched_sampling_plans(golden_loss, dev):
    """engine.TrainStep enqueues the coordinate-only part of the NEXT step (FPS, centroid gathers, neighbour searches:
    TouchedRegraster.prefetch_plans) on a background stream at the start of a step.  Same start-index draws in the same
    order, same kernels, another place in the queue: the losses of three optimiser steps must be those of the runner
    without it (to the run-to-run rounding of the atomically summed layers), and a runner fed a different batch for the
    following step must pick that batch's plan up."""
    from puzzlenet_amd import engine
    from puzzlenet_amd import model5_b as mb
    from puzzlenet_amd import ops
    G = golden_loss
    batch = [_t(G[f"ts_batch{i}"], dev) for i in range(8)]
    batch2 = [t.clone() for t in batch]
    batch2[0], batch2[1] = batch[1].clone(), batch[0].clone()       # another pair of clouds for the second step

    def run(prefetch):
        m = mb.TouchedRegraster(mr.Cfg(loss_mode=1, use_emd2=True, use_cd2=True, use_emd3=True))
        mr.fill_params(m)
        m.to(dev)
        ops.clear_grad_sinks()
        torch.manual_seed(11)
        losses, took = [], []
        orig = m._take_plans

        def spy(fpc, mrpc, streams):
            r = orig(fpc, mrpc, streams)
            took.append(r is not None)
            return r
        m._take_plans = spy
        with engine.TrainStep(m, batch, 1e-3, world=1, prefetch=prefetch) as r:
            losses.append(float(r.step(next_batch=batch2)))
            losses.append(float(r.step(next_batch=batch)))
            losses.append(float(r.step()))
        torch.cuda.synchronize()
        assert m._plan_cache is None
        return losses, took

    l0, t0 = run(False)
    l1, t1 = run(True)
    assert t0 == [False] * 3 and t1 == [True] * 3
    for a, b in zip(l0, l1):
        assert abs(a - b) <= 1e-4 * abs(a), (l0, l1)


def test_train_step_private_fps_generator_leaves_the_global_one_alone(golden_loss, dev):
    """ADVICE (round 3): with the plan prefetch the FPS start indices of step k + 1 are drawn before step k runs.  A runner
    given its own generator draws them from it: torch's global CPU generator is not advanced by the steps, two runs with
    the same private seed give the same losses, `step(last=True)` prefetches nothing (no draws for a step that never
    runs), and close() gives the model its attribute back."""
    from puzzlenet_amd import engine
    from puzzlenet_amd import model5_b as mb
    from puzzlenet_amd import ops
    G = golden_loss
    batch = [_t(G[f"ts_batch{i}"], dev) for i in range(8)]

    def run(seed):
        m = mb.TouchedRegraster(mr.Cfg(loss_mode=1, use_emd2=True, use_cd2=True, use_emd3=True))
        mr.fill_params(m)
        m.to(dev)
        ops.clear_grad_sinks()
        gen = torch.Generator().manual_seed(seed)
        torch.manual_seed(5)
        before = torch.get_rng_state()
        ops.WINNER_CAPTURE = []
        try:
            with engine.TrainStep(m, batch, 1e-3, world=1, prefetch=True, fps_generator=gen) as r:
                losses = [float(r.step()), float(r.step(last=True))]
                assert r._plans_ahead is None                   # nothing prefetched behind the last step
                state_after_last = gen.get_state()
            torch.cuda.synchronize()
            winners = [_device_winners(m, ops.WINNER_CAPTURE[7 * i: 7 * i + 7]) for i in range(2)]
        finally:
            ops.WINNER_CAPTURE = None
        assert torch.equal(torch.get_rng_state(), before)    # the global generator was never drawn from
        assert m.fps_generator is None                       # handed back
        return losses, state_after_last, winners

    (la, sa, wa), (lb, sb, wb) = run(21), run(21)
    assert torch.equal(sa, sb)
    # the same draws -> the same first loss to the rounding of the kernels' atomic sums, and the same max-pool winners
    assert abs(la[0] - lb[0]) <= 1e-5 * abs(la[0]), (la, lb)
    assert all(torch.equal(wa[0][k], wb[0][k]) for k in wa[0]), "first-step max-pool winners differ between two seeded runs"
    # The second step sits behind an Adam update, which turns the rounding noise of atomic sums in near-zero gradients into
    # +-lr: a max-pool winner of this batch that is ahead by less than that may fall either way, and one flip on a heavy path
    # moves the loss by ~1e-4 (17236.2 / 17238.1 were seen, with 121 of 1 057 024 winners different).  So: count the winners
    # that differ between the two runs' second steps - at most one in a thousand - and hold the loss to 1e-4 when there is
    # none, to 1e-3 when there are some (wrong draws pick other centroids: every winner of the cloud changes and the loss
    # moves by percents).
    flips = sum(int((wa[1][k] != wb[1][k]).sum()) for k in wa[1])
    total = sum(wa[1][k].numel() for k in wa[1])
    print("second-step max-pool winners that differ between the two runs:", flips, "of", total, "losses", la[1], lb[1])
    assert flips <= total // 1000, (flips, total)
    assert abs(la[1] - lb[1]) <= (1e-4 if flips == 0 else 1e-3) * abs(la[1]), (la, lb, flips)
    # two steps = two sets of four draws, not three: the generator is where 2 x 4 draws of 64 leave it
    ref = torch.Generator().manual_seed(21)
    for _ in range(2):
        for n_ in (batch[0].shape[1], 512, batch[0].shape[1], 512):
            torch.randint(0, n_, (batch[0].shape[0],), dtype=torch.long, generator=ref)
    assert torch.equal(sa, ref.get_state())


def test_memory_is_flat_over_steps(dev):
    """ADVICE (round 5): nothing a step allocates outlives it - neither TrainStep.step (fused stem under two names, saved
    statistics, prefetched plans) nor predict5 under no_grad.  The stem's second name used to hang in the first one's
    __dict__ (a cycle through the view's base that Python's collector cannot see): 2 x 4 MB per step at this shape."""
    import gc
    from puzzlenet_amd import engine, model5_b as mb, ops, synthetic
    B, N = 8, 1024
    ops.clear_grad_sinks()
    m = mb.TouchedRegraster(mr.Cfg(num_points=N, loss_mode=1)).to(dev)
    batch = synthetic.make_batch(B, N, dev, seed=3)
    assert mb._STEM_FUSED and ops.stem_supported(batch[0], m.Encoder.mlp1, m.Encoder.bn1, m.Encoder.mlp2, m.Encoder.bn2)

    def settled():
        torch.cuda.synchronize()
        gc.collect()
        return torch.cuda.memory_allocated()

    with engine.TrainStep(m, batch, 1e-3, world=1) as r:
        for _ in range(4):
            r.step()
        base = settled()
        for _ in range(20):
            r.step()
        after = settled()
    assert after - base <= (1 << 20), (base, after)      # (2 x B x N x 64 x 4 = 4 MB per step would be 84 MB here)
    ops.clear_grad_sinks()
    with torch.no_grad():
        for _ in range(3):
            m.predict5(batch, B, training=False)
        base = settled()
        for _ in range(20):
            m.predict5(batch, B, training=False)
        after = settled()
    assert after - base <= (1 << 20), (base, after)
