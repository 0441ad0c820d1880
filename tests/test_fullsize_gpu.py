"""Full-size BASELINE configurations on the GPU, checked through size-independent properties (the oracle is too slow here):
mask structure, finiteness, a directional-derivative check of the hand-written backward, patch round trips, and that one
AdamW step lowers the loss.  Also runs every BASELINE config (C2, C3, C3', C4, C5 shapes) for one step."""

import pytest
import torch

pytestmark = pytest.mark.gpu


def _engine(config, B, seed=0):
    import bench
    from maestro_amd.train.trainer import synthetic_batch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    dev = torch.device("cuda:0")
    torch.manual_seed(seed)
    ds, model = bench.build_model(config)
    eng = model.engine(B, dev, loss="l2_norm")
    batch = synthetic_batch(ds.dataset, B, dev)
    return ds, model, eng, batch


def test_c3_full_size_properties():
    from maestro_amd import hip
    from maestro_amd.train.optim import FusedAdamW
    B = 4
    ds, model, eng, batch = _engine("c3", B)
    torch.manual_seed(1)
    noise, struct = eng.draw_masks()
    loss = eng.forward(batch, noise=noise, struct=struct).item()
    eng.zero_grad()
    eng.backward()
    torch.cuda.synchronize()
    assert loss == loss and 0.1 < loss < 10.0
    # ---- mask structure: exactly k masked per row, ascending disjoint index lists, consistent inverse map
    for g in eng.groups:
        gb = eng.gb[g.name]
        mask, vis, msk, inv = gb["mask"].cpu().bool(), gb["vis"].cpu().long(), gb["msk"].cpu().long(), gb["inv"].cpu().long()
        assert (mask.sum(1) == g.k).all() and vis.shape == (B, g.N) and msk.shape == (B, g.k)
        assert (vis[:, 1:] > vis[:, :-1]).all() and (msk[:, 1:] > msk[:, :-1]).all()
        assert mask.gather(1, msk).all() and not mask.gather(1, vis).any()
        assert torch.equal(inv.gather(1, vis), torch.arange(g.N).expand(B, -1)) and (inv.gather(1, msk) == -1).all()
        st = struct[g.name]
        forced = st & ~mask   # structurally masked tokens may only stay visible if more than k were structurally masked
        assert not forced.any() or (st.sum(1) > g.k).any()
    # ---- gradients: finite, none identically zero on the pretrain path
    grad = eng.store.grad
    assert torch.isfinite(grad).all()
    for name, p in model.named_parameters():
        assert eng.store.g(p).abs().max() > 0, f"{name} received no gradient"
    # ---- directional derivative of the whole hand-written backward (same masks): L(w + e d) - L(w - e d) ~ 2 e <g, d>
    gnorm = grad.norm().item()
    d = grad / gnorm
    eps = 0.02 / gnorm   # predicted total change 0.04
    w0 = eng.store.flat.clone()
    vals = []
    for sgn in (+1.0, -1.0):
        eng.store.flat.copy_(w0 + sgn * eps * d)
        vals.append(eng.forward(batch, noise=noise, struct=struct).item())
    eng.store.flat.copy_(w0)
    measured, predicted = vals[0] - vals[1], 2 * eps * gnorm
    assert abs(measured - predicted) < 0.15 * predicted, (measured, predicted)
    # ---- patch round trip at full size: patchify(raw target) -> depatchify == input raster
    s = eng.mods["aerial"]
    cols = torch.empty(B * s.n_tok, s.Kpad, device=batch["aerial"].device, dtype=torch.bfloat16)
    tgt = torch.empty(B * s.n_tok, s.K, device=cols.device)
    nb = eng.mb["aerial"]["norm_bands"]
    hip.patchify(batch["aerial"], cols, tgt, B, s.C, s.S, s.P, s.Kpad, nb, len(s.norm_bands), False, False)
    img = torch.empty_like(batch["aerial"])
    hip.depatchify(tgt, img, B, s.C, s.S, s.P)
    assert torch.equal(img, batch["aerial"])
    # ---- one optimizer step with the same masks lowers the loss
    eng.forward(batch, noise=noise, struct=struct)
    eng.zero_grad()
    eng.backward()
    FusedAdamW(eng, 1e-4).step()
    assert eng.forward(batch, noise=noise, struct=struct).item() < loss


@pytest.mark.parametrize("config,B", [("c2", 4), ("c3p", 2), ("c4", 2), ("c5", 2)])
def test_every_baseline_config_runs(config, B):
    ds, model, eng, batch = _engine(config, B)
    loss = eng.forward(batch).item()
    eng.zero_grad()
    eng.backward()
    torch.cuda.synchronize()
    assert loss == loss and loss < 20.0
    assert torch.isfinite(eng.store.grad).all()
    pixels, masks = eng.reconstructions()
    for m, c in ds.dataset.inputs.items():
        assert pixels[m].shape == batch[m].shape and masks[m].shape == batch[m].shape and masks[m].dtype == torch.bool


@pytest.mark.parametrize("config,phase", [("c3", "finetune"), ("c3", "probe"), ("c4", "finetune")])
def test_supervised_full_size_properties(config, phase):
    """The probe / finetune branch at the full BASELINE shapes (C3: FLAIR segmentation, 15 classes at 512 x 512 on top of
    1424 tokens per tile; C4: ViT-L multilabel): finite loss and gradients, probe leaves the encoder's gradient slice
    untouched, the loss starts near ln(classes) / ln 2 for random heads, and a few AdamW steps on a fixed batch lower it."""
    import math

    import bench
    from maestro_amd.train.trainer import SupervisedLoop, synthetic_batch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    dev = torch.device("cuda:0")
    B = 2
    torch.manual_seed(0)
    ds, model = bench.build_model(config, phase)
    loop = SupervisedLoop(model, B, dev, phase=phase, base_lr=1e-3, total_steps=20)
    batch = synthetic_batch(ds.dataset, B, dev)
    batch.update(bench.synthetic_targets(ds.dataset, B, dev))
    eng = loop.engine
    first = eng.forward(batch).item()
    n_cls = {t: c.num_classes for t, c in ds.dataset.targets.items()}
    kinds = {c.type_target for c in ds.dataset.targets.values()}
    expect = sum(math.log(n) if ds.dataset.targets[t].type_target != "multilabel_classif" else math.log(2.0)
                 for t, n in n_cls.items())
    assert math.isfinite(first) and 0.3 * expect < first < 3.0 * expect, (first, expect, kinds)
    eng.zero_grad()
    eng.backward()
    torch.cuda.synchronize()
    g = eng.store.grad
    assert torch.isfinite(g).all()
    lo, hi = eng.trainable_span
    assert float(g[lo:hi].abs().sum()) > 0
    if phase == "probe":
        assert lo > 0 and float(g[:lo].abs().sum()) == 0.0
    else:
        assert lo == 0 and float(g[: g.numel() // 2].abs().sum()) > 0
    losses = [float(loop.step(batch)) for _ in range(8)]
    assert all(math.isfinite(x) for x in losses) and losses[-1] < losses[0], losses
    logits = eng.logits()
    for t, c in ds.dataset.targets.items():
        if c.type_target == "segment":
            assert logits[t].shape[0] == B and logits[t].shape[-3] == c.num_classes
        else:
            assert tuple(logits[t].shape) == (B, c.num_classes)
