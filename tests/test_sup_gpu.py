"""Probe / finetune branch on the GPU: SupervisedEngine vs the oracle AND the reference's golden
vectors (tests/golden/sup_*.npz), same weights, inputs and targets.

Tolerances (bf16 GEMMs / attention with fp32 accumulation vs the fp32 CPU oracle):
  loss_pred ............... |d| <= 1.2e-3 * |loss|
  logits .................. relative L2 error <= 1.6e-2 per target
  parameter gradients ..... relative L2 error <= 4.2e-2 per parameter (plus an absolute floor for ~zero grads);
                            probe: every non-head gradient is exactly zero (features detached, head.py:17-25)
"""

import numpy as np
import pytest
import torch

import maestro_amd.conf as conf
from maestro_amd.ssl import mae as pmae
from oracle import heads as oh
from tests.test_oracle_sup import CASES, build_sup_case

pytestmark = pytest.mark.gpu
LOSS_TOL, LOGIT_TOL, GRAD_TOL = 8e-4, 1.1e-2, 2.9e-2    # <= 2x observed on MI355X (round 3): 3.8e-4, 5.3e-3, 1.42e-2


def _rel(a, b):
    return ((a - b).double().norm() / b.double().norm().clamp(min=1e-12)).item()


def _setup(name):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    dev = torch.device("cuda:0")
    case, ds, oracle, _, batch = build_sup_case(name)
    model = getattr(pmae, f"mae_{case['size']}")(
        datasets=ds, mask=conf.MaskConfig(), interpolate="nearest", fusion_mode=case["fusion"], inter_depth=case["inter_depth"],
        model="mae", num_levels=1, type_head=case["type_head"], fac_abs_enc=1.0, fac_date_enc=1.0, **case["model_kw"])
    missing, unexpected = model.load_state_dict(oracle.state_dict(), strict=True)   # heads included, key for key
    assert not missing and not unexpected
    return dev, case, ds, oracle, model, batch


@pytest.mark.parametrize("phase", ["probe", "finetune"])
@pytest.mark.parametrize("name", list(CASES))
def test_supervised_engine_matches_oracle_and_reference(golden_dir, name, phase, observed):
    dev, case, ds, oracle, model, batch = _setup(name)
    gold = np.load(golden_dir / f"{name}.npz", allow_pickle=False)
    eng = model.sup_engine(case["B"], dev, phase)
    dbatch = {k: v.to(dev) for k, v in batch.items()}
    for it in range(3):          # eager, hipGraph capture, replay: identical results
        loss = eng.forward(dbatch)
        eng.zero_grad()
        eng.backward()
        torch.cuda.synchronize()
        if it == 0:
            first, grad0 = loss.item(), eng.store.grad.clone()
        else:
            assert abs(loss.item() - first) <= 1e-5 * abs(first)
            rel = ((eng.store.grad - grad0).norm() / grad0.norm()).item()
            if not rel < 1e-4:      # which parameters differ between the eager and the replayed backward?
                bad = []
                for k, p in model.named_parameters():
                    if id(p) not in eng.store.offset:
                        continue
                    o = eng.store.offset[id(p)]
                    a, b0 = eng.store.grad[o: o + p.numel()], grad0[o: o + p.numel()]
                    d = ((a - b0).norm() / b0.norm().clamp(min=1e-20)).item()
                    if not d < 1e-4:
                        bad.append((k, d, a.flatten()[:3].tolist(), b0.flatten()[:3].tolist()))
                info = {n: (float(b["dw_conv"].abs().max()), float(b["dyc"].float().abs().max()), float(b["cols"].float().abs().max()))
                        for n, b in eng.mb.items()}
                raise AssertionError(f"iteration {it}: replayed gradients differ from the eager ones (rel {rel:.3e}); graphs "
                                     f"{list(eng._graphs)}; parameters: {bad[:8]}; |dw_conv|, |dyc|, |cols| max per modality: {info}")
    logits = eng.logits()

    ob, _, _, ologits = oracle({k: v.clone() for k, v in batch.items()}, phase)
    oloss = oh.compute_loss_pred(oracle.dataset, ob, ologits)
    oracle.zero_grad()
    oloss.backward()
    observed(f"sup/{name}/{phase}", "loss", abs(loss.item() - oloss.item()) / abs(oloss.item()))
    assert abs(loss.item() - oloss.item()) < LOSS_TOL * abs(oloss.item()), (loss.item(), oloss.item())
    assert abs(loss.item() - float(gold[f"{phase}/loss"])) < LOSS_TOL * abs(float(gold[f"{phase}/loss"]))
    for t in ologits:
        assert logits[t].shape == ologits[t].shape
        observed(f"sup/{name}/{phase}", f"logits/{t}", _rel(logits[t].cpu(), ologits[t].detach()))
        assert _rel(logits[t].cpu(), ologits[t].detach()) < LOGIT_TOL, (t, _rel(logits[t].cpu(), ologits[t].detach()))
        flat = logits[t].cpu().reshape(logits[t].shape[0], -1)
        stride = max(1, flat.shape[1] // 4096)
        assert _rel(flat[:, ::stride], torch.from_numpy(gold[f"{phase}/logits/{t}"])) < LOGIT_TOL   # the reference's own logits
    ograds = {k: p.grad for k, p in oracle.named_parameters() if p.grad is not None}
    gmax = max(g.abs().max().item() for g in ograds.values())
    worst = (0.0, None)
    for k, p in model.named_parameters():
        if id(p) not in eng.store.offset:
            continue
        got = eng.store.g(p).cpu()
        if k not in ograds:
            assert float(got.abs().max()) == 0.0, f"{k}: gradient on a detached / unused parameter"
            continue
        want = ograds[k]
        err, ref = (got - want).double().norm().item(), want.double().norm().item()
        assert err <= GRAD_TOL * ref + 1e-5 * gmax * want.numel() ** 0.5, f"{k}: grad rel err {err / max(ref, 1e-12):.3e}"
        worst = max(worst, (err / max(ref, 1e-12), k))
        assert abs(got.double().norm().item() - float(gold[f"{phase}/gradnorm/{k}"])) <= GRAD_TOL * ref + 1e-5 * gmax * want.numel() ** 0.5
    observed(f"sup/{name}/{phase}", f"grad_worst/{worst[1]}", worst[0])
    print(f"[{name}/{phase}] loss hip={loss.item():.6f} oracle={oloss.item():.6f} worst grad rel err {worst}")


def test_model_forward_contract_probe(golden_dir):
    """``MAE.forward(batch, ssl_phase)`` returns ``(batch, None, None, logits)`` in probe / finetune (mim.py:503-505)."""
    dev, case, ds, oracle, model, batch = _setup("sup_pastis_two")
    dbatch = {k: v.to(dev) for k, v in batch.items()}
    out_batch, rec, msk, logits = model(dbatch, ssl_phase="probe")
    assert rec is None and msk is None and set(logits) == set(ds.dataset.targets)
    _, _, _, ologits = oracle({k: v.clone() for k, v in batch.items()}, "probe")
    for t in logits:
        assert logits[t].shape == ologits[t].shape and _rel(logits[t].cpu(), ologits[t].detach()) < LOGIT_TOL


@pytest.mark.parametrize("phase", ["probe", "finetune"])
def test_ssl_module_supervised_step_and_loop(phase):
    """Lightning-style step (``loss.backward()`` fills ``param.grad``; probe leaves the encoder without gradients) and the
    built-in loop: the loss of a fixed batch goes down, and probe does not move a single encoder weight."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from types import SimpleNamespace

    from maestro_amd.train.model import SSLModule
    from maestro_amd.train.trainer import SupervisedLoop
    from oracle.gen_golden import build_datasets, make_batch, make_targets
    dev = torch.device("cuda:0")
    case = CASES["sup_treesat_mlc"]
    ds = build_datasets(case, conf)
    torch.manual_seed(0)
    module = SSLModule(datasets=ds, mask=conf.MaskConfig(), interpolate="nearest", fusion_mode="group", inter_depth=1,
                       model="mae", model_size="tiny", type_head="attentive", loss="l2_norm")
    module.trainer = SimpleNamespace(ssl_phase=phase)
    batch = make_batch(ds.dataset, 4, 1)
    batch.update(make_targets(ds.dataset, 4, 1))
    batch = {k: v.to(dev) for k, v in batch.items()}
    out = module.training_step(batch, 0)
    assert set(out) == {"loss", "log_inputs", "log_preds", "log_targets"} and out["loss"].requires_grad
    out["loss"].backward()
    named = dict(module.model.named_parameters())
    assert all(named[k].grad is not None and float(named[k].grad.abs().sum()) > 0 for k in named if k.startswith("heads."))
    enc = [k for k in named if k.startswith("encoder.")]
    if phase == "probe":
        assert all(named[k].grad is None for k in enc)
    else:
        assert all(named[k].grad is not None for k in enc)
    before = {k: named[k].detach().clone() for k in enc[:4]}
    loop = SupervisedLoop(module.model, 4, dev, phase=phase, base_lr=3e-3, total_steps=40)
    losses = [float(loop.step(batch)) for _ in range(25)]
    assert losses[-1] < 0.8 * losses[0], losses
    moved = [float((named[k].detach() - before[k]).abs().max()) for k in before]
    assert (max(moved) == 0.0) if phase == "probe" else (min(moved) > 0.0)


def test_finetune_validation_uses_ema_weights():
    """``base.py:195-202``: in finetune, val / test steps run the EMA model.  The EMA copy stays on the CPU; its values
    are swapped into the engine's flat buffer around the forward and the trained weights are back afterwards."""
    from types import SimpleNamespace

    from maestro_amd.train.model import SSLModule
    from oracle.gen_golden import build_datasets, make_batch, make_targets
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    DEV = torch.device("cuda:0")  # noqa: N806
    case = CASES["sup_treesat_mlc"]
    ds = build_datasets(case, conf)
    batch = make_batch(ds.dataset, 4, 1)
    batch.update(make_targets(ds.dataset, 4, 1))
    torch.manual_seed(3)
    module = SSLModule(datasets=ds, mask=conf.MaskConfig(), interpolate="nearest", fusion_mode="group", inter_depth=1,
                       model="mae", model_size="tiny", type_head="attentive", use_ema=True)
    module.trainer = SimpleNamespace(ssl_phase="finetune", max_epochs=10)
    dbatch = {k: v.to(DEV) for k, v in batch.items()}
    train_loss = module.training_step(dbatch, 0)["loss"].item()
    val_same = module.validation_step(dbatch, 0)["loss"].item()         # EMA == model right after construction
    assert abs(val_same - train_loss) <= 1e-5 * abs(train_loss)
    with torch.no_grad():                                                # make the EMA copy differ, the way update_ema would
        for p in module.ema_model.parameters():
            p.mul_(0.5)
    eng = module.model._sup_engine
    before = eng.store.flat.clone()
    val_ema = module.validation_step(dbatch, 0)["loss"].item()
    assert abs(val_ema - train_loss) > 1e-4 * abs(train_loss), "validation did not see the EMA weights"
    assert torch.equal(eng.store.flat, before), "the trained weights must be back after the EMA evaluation"
    assert abs(module.training_step(dbatch, 1)["loss"].item() - train_loss) <= 1e-5 * abs(train_loss)
    # oracle check of the EMA forward: load the halved weights into a plain module and compare the loss
    module2 = SSLModule(datasets=ds, mask=conf.MaskConfig(), interpolate="nearest", fusion_mode="group", inter_depth=1,
                        model="mae", model_size="tiny", type_head="attentive", use_ema=False)
    module2.model.load_state_dict(module.ema_model.state_dict())
    module2.trainer = SimpleNamespace(ssl_phase="finetune", max_epochs=10)
    want = module2.validation_step(dbatch, 0)["loss"].item()
    assert abs(val_ema - want) <= 1e-5 * abs(want), (val_ema, want)
    module.update_ema()                                                  # invalidates the cached flat copy
    assert module._ema_cache is None
    del case
