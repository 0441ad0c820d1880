"""Probe / finetune branch: oracle vs the REFERENCE's own logits, loss_pred and gradient norms
(golden vectors sup_*.npz from oracle/gen_golden.py: maestro/ssl/mim.py:343-394, maestro/layers/head.py,
maestro/train/base.py:98-151 run on seeded weights and inputs).  CPU only."""

import numpy as np
import pytest
import torch

import maestro_amd.conf as conf
from oracle import heads as oh
from oracle import mae as om
from oracle.gen_golden import build_datasets, init_weights, make_batch, make_targets, sup_case_table

CASES = sup_case_table()


def build_sup_case(name):
    case = CASES[name]
    ds = build_datasets(case, conf)
    oracle = om.build_oracle(ds, conf.MaskConfig(), model_size=case["size"], fusion_mode=case["fusion"],
                             inter_depth=case["inter_depth"], interpolate="nearest", model="mae", num_levels=1,
                             type_head=case["type_head"], fac_abs_enc=1.0, fac_date_enc=1.0, **case["model_kw"])
    chk = init_weights(oracle, case["seed"])
    batch = make_batch(ds.dataset, case["B"], case["seed"], stress=case.get("stress", False))
    batch.update(make_targets(ds.dataset, case["B"], case["seed"]))
    return case, ds, oracle, chk, batch


@pytest.mark.parametrize("phase", ["probe", "finetune"])
@pytest.mark.parametrize("name", list(CASES))
def test_oracle_supervised_branch_matches_reference(golden_dir, name, phase):
    gold = np.load(golden_dir / f"{name}.npz", allow_pickle=False)
    case, ds, oracle, chk, batch = build_sup_case(name)
    assert abs(chk - float(gold["weights_checksum"])) < 1e-6 * chk
    ob, _, _, logits = oracle({k: v.clone() for k, v in batch.items()}, phase)
    loss = oh.compute_loss_pred(oracle.dataset, ob, logits)
    assert abs(loss.item() - float(gold[f"{phase}/loss"])) < 1e-5 * abs(float(gold[f"{phase}/loss"]))
    for t, lg in logits.items():
        flat = lg.detach().reshape(lg.shape[0], -1)
        stride = max(1, flat.shape[1] // 4096)
        np.testing.assert_allclose(flat[:, ::stride].numpy(), gold[f"{phase}/logits/{t}"], atol=2e-5)
        assert abs(lg.detach().double().sum().item() - float(gold[f"{phase}/logits_sum/{t}"])) < 1e-3 * lg.numel() ** 0.5
    oracle.zero_grad()
    loss.backward()
    got = {k: p.grad.double().norm().item() for k, p in oracle.named_parameters() if p.grad is not None}
    want = {k.split("/", 2)[2]: float(gold[k]) for k in gold.files if k.startswith(f"{phase}/gradnorm/")}
    assert set(got) == set(want), set(got) ^ set(want)       # probe: head parameters only; finetune: encoder + heads
    for k, v in want.items():
        assert abs(got[k] - v) <= 2e-4 * v + 1e-7, (k, got[k], v)
    if phase == "probe":
        assert all(k.startswith("heads.") for k in got)


def test_loss_pred_skips_missing_and_handles_empty_selection():
    case, ds, oracle, _, batch = build_sup_case("sup_flair_seg")
    _, _, _, logits = oracle({k: v.clone() for k, v in batch.items()}, "probe")
    all_missing = dict(batch, cosia=torch.full_like(batch["cosia"], -1))
    loss = oh.compute_loss_pred(oracle.dataset, all_missing, logits)      # base.py:147-148: 0 * mean(logits)
    assert loss.item() == 0.0 and loss.requires_grad
