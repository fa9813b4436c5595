"""GPU parity, round 5.

* autograd.FilmPosFramesFn -- the frame-encoding gradient accumulators are keyed to the backward pass, not to the tensors: a loss on some
  of the rollout's calls, and a second pass over a retained graph (trainer/trainer.py:144-159's sliding window, tante.py:136-141).
* csrc/block_bwd_fs.hip -- the WHOLE backward of a TransformerBlock in one launch (attn_backbone.py:59-83 backwards) against the three
  launches it replaces, against a float64 autograd of the same block, and inside the production-shape train step (fixture g14).

Bars: fp32 compute 1e-5 / gradients 2e-4, bf16 compute 1e-2 / gradients 4e-2, relative to the reference's fp32 CPU result.
"""
import ctypes as Ct

import numpy as np
import pytest
import torch

from conftest import rel_err, max_rel, record_parity

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


# ---------------------------------------------------------------------------------------------------
# FilmPosFramesFn: whichever windows take part in a backward pass, the frame's encoder receives their sum
# ---------------------------------------------------------------------------------------------------
def _g14_grads(dev, monkeypatch, frame_film, n_loss_calls, passes=1):
    """Gradients of the g14 model (C = 256, fp32 compute, per-operator nodes) for a loss on the first `n_loss_calls` predicted frames."""
    import tante_amd
    from tante_amd import autograd as A
    from tante_amd import train_forward as TF
    from conftest import g14_setup, G14_FIELDS, G14_RES
    monkeypatch.setattr(TF, "FRAME_FILM", frame_film)
    m, batch, _, names = g14_setup()
    m = m.to(dev).train().set_compute("fp32")
    opt = tante_amd.FlatAdamW(m.parameters(), lr=1e-4)
    opt.zero_grad()
    md = tante_amd.TanteMetadata(n_fields=G14_FIELDS, spatial_resolution=G14_RES)
    fmt = tante_amd.DefaultChannelsFirstFormatter(md)
    b = {k: v.to(dev) for k, v in batch.items()}
    y_pred, y_ref = tante_amd.rollout_model(m, b, fmt, 4)
    loss = A.MseMeanFn.apply(y_pred[:, :n_loss_calls].contiguous(), y_ref[:, :n_loss_calls].contiguous())
    A.reset_backward_state()
    try:
        for i in range(passes):
            loss.backward(retain_graph=i + 1 < passes)
    finally:
        A.reset_backward_state(after=True)
    return {k: p.grad.detach().clone() for k, p in m.named_parameters()}, names


@pytest.mark.parametrize("n_loss_calls", [2, 1])
def test_frame_gradients_with_a_loss_on_some_calls_only(dev, monkeypatch, n_loss_calls):
    """The loss reads the first calls' frames only: the later calls' FilmPosFramesFn nodes never run, and the frames they share with the
    earlier windows must still hand their (partial) sums to the encoder.  Against the plain path (one stacked window per call, autograd
    sums the frame gradients): fp32, 2e-5 of each tensor's largest entry.  (Round 4's per-tensor window counters never completed here
    and the encoder silently received no gradient.)"""
    g_acc, names = _g14_grads(dev, monkeypatch, True, n_loss_calls)
    g_ref, _ = _g14_grads(dev, monkeypatch, False, n_loss_calls)
    worst = 0.0
    for k in names:
        e = max_rel(g_acc[k].cpu(), g_ref[k].cpu())
        worst = max(worst, e)
        assert e < 2e-5, (k, e)
    enc = [k for k in names if k.startswith("encoder.")]
    assert enc and all(float(g_acc[k].abs().max()) > 0.0 for k in enc), "the encoder received no gradient"
    record_parity(worst, worst, 2e-5, "fp32", f"FilmPosFramesFn: loss on {n_loss_calls} of 4 calls vs the stacked-window path")


def test_frame_gradients_second_pass_over_a_retained_graph(dev, monkeypatch):
    """backward(retain_graph=True) followed by a second backward over the same graph: the gradient slots hold exactly twice one pass's
    gradients (round 4's node cleared its records in backward and crashed on re-entry)."""
    g2, names = _g14_grads(dev, monkeypatch, True, 4, passes=2)
    g1, _ = _g14_grads(dev, monkeypatch, True, 4, passes=1)
    worst = 0.0
    for k in names:
        e = max_rel(g2[k].cpu(), 2.0 * g1[k].cpu())
        worst = max(worst, e)
        assert e < 2e-5, (k, e)
    record_parity(worst, worst, 2e-5, "fp32", "FilmPosFramesFn: two passes over a retained graph = 2 x one pass")
