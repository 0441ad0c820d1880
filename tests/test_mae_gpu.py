"""End-to-end GPU parity of the HIP engine: forward, masks, loss and every parameter gradient vs the oracle and vs
the REFERENCE's golden vectors (tests/golden/*.npz), same weights, same inputs, same injected RNG draws.

Tolerances (bf16 MFMA GEMMs/attention with fp32 accumulation and fp32 residual stream vs an fp32 CPU oracle), set to <= 2x
the worst error observed over all golden cases on MI355X (round 3, profiles/r03_observed_errors.jsonl: loss 1.04e-3,
pixels_rec 4.9e-3, worst parameter gradient 1.51e-2):
  mask indices ............ bit-exact
  loss .................... |d| <= 2.1e-3 * |loss|
  pixels_rec .............. relative L2 error <= 1e-2 per modality
  parameter gradients ..... relative L2 error <= 3e-2 per parameter (plus an absolute floor for ~zero grads)
"""

import numpy as np
import pytest
import torch

import maestro_amd.conf as conf
from maestro_amd.ssl import mae as pmae
from oracle import mae as om
from oracle.gen_golden import build_datasets, case_table, init_weights, make_batch, resize_case_table, tie_case_table, token_masks

pytestmark = pytest.mark.gpu
CASES = case_table()
TIE_CASES = tie_case_table()
RESIZE_CASES = resize_case_table()
LOSS_TOL, PIX_TOL, GRAD_TOL = 2.1e-3, 1e-2, 3e-2
COMMON = dict(interpolate="nearest", model="mae", num_levels=1, type_head="attentive", fac_abs_enc=1.0, fac_date_enc=1.0)


def _rel(a, b):
    return ((a - b).double().norm() / b.double().norm().clamp(min=1e-12)).item()


def _setup(name, golden_dir, table=None):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    dev = torch.device("cuda:0")
    case = (table or {**CASES, **RESIZE_CASES})[name]
    gold = np.load(golden_dir / f"{name}.npz", allow_pickle=False)
    ds = build_datasets(case, conf)
    kw = dict(fusion_mode=case["fusion"], inter_depth=case["inter_depth"], **{**COMMON, "interpolate": case.get("interpolate", "nearest")},
              **case["model_kw"])
    mask_cfg = conf.MaskConfig(**case.get("mask_kw", {}))
    oracle = om.build_oracle(ds, mask_cfg, model_size=case["size"], **kw)
    init_weights(oracle, case["seed"])
    model = getattr(pmae, f"mae_{case['size']}")(datasets=ds, mask=mask_cfg, **kw)
    missing, unexpected = model.load_state_dict(oracle.state_dict(), strict=True)
    batch = make_batch(ds.dataset, case["B"], case["seed"], stress=case.get("stress", False), sizes=case.get("raster_size"))
    noise, struct = {}, {}
    for key in gold.files:
        if key.startswith("noise/"):
            g = key.split("/", 1)[1]
            noise[g] = torch.from_numpy(gold[key])
            struct[g] = torch.from_numpy(np.unpackbits(gold[f"struct/{g}"], axis=1)[:, : noise[g].shape[1]].astype(bool))
    return dev, case, gold, ds, oracle, model, batch, noise, struct


@pytest.mark.parametrize("wgrad", ["fused", "deferred"])
@pytest.mark.parametrize("name", list(CASES))
def test_engine_matches_oracle_and_reference(golden_dir, name, wgrad, observed):
    """``wgrad``: weight gradients in line with the dgrad chain (split-K atomics) / deferred into one grouped launch."""
    _check_case(golden_dir, name, wgrad, observed, f"tiny/{name}/{wgrad}")


@pytest.mark.parametrize("name", list(CASES))
def test_goldens_through_the_lds_dma_tiles(golden_dir, name, observed, monkeypatch):
    """The same 11 reference goldens with every eligible NT / NN GEMM forced onto the 256 x 256 LDS-DMA tile
    (``MH_GEMM_DMA=1``, resolved in maestro_amd/hip.py): the kernel the M = 32768 decoder problems of the bench
    configuration run on, which the tiny shapes would never pick by themselves."""
    monkeypatch.setenv("MH_GEMM_DMA", "1")
    _check_case(golden_dir, name, "deferred", observed, f"tiny_dma/{name}")


def _check_case(golden_dir, name, wgrad, observed, tag):
    dev, case, gold, ds, oracle, model, batch, noise, struct = _setup(name, golden_dir)
    eng = model.engine(case["B"], dev, loss="l2_norm")
    eng.wgrad_mode = wgrad
    dbatch = {k: v.to(dev) for k, v in batch.items()}
    loss = eng.forward(dbatch, noise=noise, struct=struct)
    eng.zero_grad()
    eng.backward()
    torch.cuda.synchronize()
    pixels, masks = eng.reconstructions()

    ob = {k: v.clone() for k, v in batch.items()}
    ob, orec, omsk, _ = oracle(ob, "pretrain", noise=noise, struct_masks={g: s[:, :, None] for g, s in struct.items()})
    oloss = om.compute_loss_rec(ob, orec, omsk, oracle.out_grid_size, om.norm_bands_of(ds.dataset), "l2_norm")
    oracle.zero_grad()
    oloss.backward()

    group_of = dict(ds.dataset.groups) if case["fusion"] == "group" else {m: m for m in ds.dataset.inputs}
    multi = {g for g in set(group_of.values()) if sum(1 for v in group_of.values() if v == g) > 1}
    # (several band-groups = several different mask tokens inside one modality: tie-dependent in the reference as well, Q5)
    multi |= {group_of[m] for m, c in ds.dataset.inputs.items() if not isinstance(c.bands, int) and len(c.bands) > 1}
    for m in orec:
        assert torch.equal(masks[m].cpu(), omsk[m]), f"{m}: mask differs from oracle"
        tok = token_masks(masks[m].cpu(), ds.dataset.inputs[m]).numpy()
        ref_tok = np.unpackbits(gold[f"mask_tok/{m}"], axis=2)[:, :, : tok.shape[2]].astype(bool)
        assert np.array_equal(tok, ref_tok), f"{m}: mask indices differ from the reference"
        observed(tag, f"pixels/{m}", _rel(pixels[m].cpu(), orec[m].detach()))
        assert _rel(pixels[m].cpu(), orec[m].detach()) < PIX_TOL, (m, _rel(pixels[m].cpu(), orec[m].detach()))
        if group_of[m] not in multi:  # reference value independent of its implementation-defined tie order
            assert _rel(pixels[m].cpu(), torch.from_numpy(gold[f"pixels_rec/{m}"])) < PIX_TOL
    observed(tag, "loss", abs(loss.item() - oloss.item()) / abs(oloss.item()))
    assert abs(loss.item() - oloss.item()) < LOSS_TOL * abs(oloss.item()), (loss.item(), oloss.item())
    if not multi:
        assert abs(loss.item() - float(gold["loss_l2_norm"])) < LOSS_TOL * abs(float(gold["loss_l2_norm"]))

    ograds = {k: p.grad for k, p in oracle.named_parameters() if p.grad is not None}
    worst = (0.0, None)
    gmax = max(g.abs().max().item() for g in ograds.values())
    for k, p in model.named_parameters():
        if k not in ograds:
            continue
        got, want = eng.store.g(p).cpu(), ograds[k]
        err = (got - want).double().norm().item()
        ref = want.double().norm().item()
        ok = err <= GRAD_TOL * ref + 1e-3 * gmax * want.numel() ** 0.5 * 1e-2
        if err / max(ref, 1e-12) > worst[0]:
            worst = (err / max(ref, 1e-12), k)
        assert ok, f"{k}: grad rel err {err / max(ref, 1e-12):.3e} (|ref|={ref:.3e})"
    observed(tag, f"grad_worst/{worst[1]}", worst[0])
    print(f"[{name}] loss hip={loss.item():.6f} oracle={oloss.item():.6f} worst grad rel err {worst}")


@pytest.mark.parametrize("name", list(RESIZE_CASES))
def test_resized_inputs_match_the_reference(golden_dir, name, observed):
    """Round 6: rasters that do NOT arrive at ``image_size`` go through ``mh_resize`` (bilinear / bicubic,
    PyTorch's align_corners=False maps) and, for ``dem``, the elevation rescale AFTER the resize -- against goldens generated by the
    reference's own ``resize_and_rescale`` (mim.py:425-437): masks bit-exact, reconstructions, loss, every gradient (``_check_case``),
    and the returned batch (the loss target) against the reference's resized rasters."""
    _check_case(golden_dir, name, "deferred", observed, f"tiny_resize/{name}")
    dev, case, gold, ds, oracle, model, batch, noise, struct = _setup(name, golden_dir)
    eng = model.engine(case["B"], dev, loss="l2_norm")
    dbatch = {k: v.to(dev) for k, v in batch.items()}
    eng.forward(dbatch, noise=noise, struct=struct)
    returned = eng.returned_batch(dbatch)
    for m in case["raster_size"]:
        want = torch.from_numpy(gold[f"target/{m}"])
        assert returned[m].shape == want.shape and want.shape[-1] == ds.dataset.inputs[m].image_size
        err = (returned[m].cpu() - want).abs().max().item()
        observed(f"tiny_resize/{name}", f"target/{m}", err)
        assert err < 2e-5 * max(1.0, want.abs().max().item()), (m, err)


@pytest.mark.parametrize("name", list(TIE_CASES))
def test_reference_tie_order(golden_dir, name, observed):
    """``MAEEngine.tie_order = "torch"`` on the tie-DEPENDENT reference goldens (more than k structurally masked tokens in a
    (sample, group), and the two-modality s1 group: SURVEY Q5, maestro/ssl/mae.py:241, 274-286): the engine reissues the
    reference's two unstable argsort calls on the host, so the masked set is the reference's bit for bit, every modality's
    reconstruction matches the reference's stored one (including the s1 modalities, where mask tokens land on the other
    modality's positions), and every parameter gradient matches the oracle run in the same mode."""
    dev, case, gold, ds, oracle, model, batch, noise, struct = _setup(name, golden_dir, table=TIE_CASES)
    assert not all(bool(gold[k]) for k in gold.files if k.startswith("tie_free/"))
    eng = model.engine(case["B"], dev, loss="l2_norm")
    eng.tie_order = "torch"
    loss = eng.forward({k: v.to(dev) for k, v in batch.items()}, noise=noise, struct=struct)
    eng.zero_grad()
    eng.backward()
    torch.cuda.synchronize()
    pixels, masks = eng.reconstructions()
    for m in pixels:
        tok = token_masks(masks[m].cpu(), ds.dataset.inputs[m]).numpy()
        ref_tok = np.unpackbits(gold[f"mask_tok/{m}"], axis=2)[:, :, : tok.shape[2]].astype(bool)
        assert np.array_equal(tok, ref_tok), f"{m}: masked set differs from the reference's"
        e = _rel(pixels[m].cpu(), torch.from_numpy(gold[f"pixels_rec/{m}"]))
        observed(f"ties/{name}", f"pixels/{m}", e)
        assert e < PIX_TOL, (m, e)
    want = float(gold["loss_l2_norm"])
    observed(f"ties/{name}", "loss", abs(loss.item() - want) / abs(want))
    assert abs(loss.item() - want) < LOSS_TOL * abs(want), (loss.item(), want)
    oracle.reference_tie_order = True
    ob, orec, omsk, _ = oracle({k: v.clone() for k, v in batch.items()}, "pretrain", noise=noise,
                               struct_masks={g: s[:, :, None] for g, s in struct.items()})
    oracle.zero_grad()
    om.compute_loss_rec(ob, orec, omsk, oracle.out_grid_size, om.norm_bands_of(ds.dataset), "l2_norm").backward()
    ograds = {k: p.grad for k, p in oracle.named_parameters() if p.grad is not None}
    gmax = max(g.abs().max().item() for g in ograds.values())
    worst = 0.0
    for k, p in model.named_parameters():
        if k in ograds:
            got, ref = eng.store.g(p).cpu(), ograds[k]
            err, nrm = (got - ref).double().norm().item(), ref.double().norm().item()
            if nrm > 1e-3 * gmax * ref.numel() ** 0.5:
                worst = max(worst, err / nrm)
            assert err <= GRAD_TOL * nrm + 1e-5 * gmax * ref.numel() ** 0.5, (k, err / max(nrm, 1e-12))
    observed(f"ties/{name}", "grad_worst", worst)
    # ... and the default (stable) mode does NOT reproduce the reference here: the divergence is real and opt-in to close
    eng2 = getattr(pmae, f"mae_{case['size']}")(datasets=ds, mask=conf.MaskConfig(**case.get("mask_kw", {})), fusion_mode=case["fusion"],
                                              inter_depth=case["inter_depth"], **COMMON, **case["model_kw"]).engine(case["B"], dev)
    eng2.forward({k: v.to(dev) for k, v in batch.items()}, noise=noise, struct=struct)
    _, masks2 = eng2.reconstructions()
    assert any(not torch.equal(masks2[m], masks[m]) for m in masks), "the stable order should differ on a tie case"


@pytest.mark.parametrize("loss", ["l1", "l2", "l1_norm"])
def test_loss_variants(golden_dir, loss, observed):
    """Forward value vs the reference's golden loss, and every parameter gradient vs the oracle, for the other losses."""
    dev, case, gold, ds, oracle, model, batch, noise, struct = _setup("c3_aerial_s2", golden_dir)
    eng = model.engine(case["B"], dev, loss=loss)
    out = eng.forward({k: v.to(dev) for k, v in batch.items()}, noise=noise, struct=struct)
    want = float(gold[f"loss_{loss}"])
    assert abs(out.item() - want) < LOSS_TOL * abs(want), (out.item(), want)
    eng.zero_grad()
    eng.backward()
    ob = {k: v.clone() for k, v in batch.items()}
    ob, orec, omsk, _ = oracle(ob, "pretrain", noise=noise, struct_masks={g: s[:, :, None] for g, s in struct.items()})
    oracle.zero_grad()
    om.compute_loss_rec(ob, orec, omsk, oracle.out_grid_size, om.norm_bands_of(ds.dataset), loss).backward()
    ograds = {k: p.grad for k, p in oracle.named_parameters() if p.grad is not None}
    # (l1: d|x| = sign(x) flips for the few elements whose bf16 reconstruction lands on the other side of the target; the
    # observed relative L2 errors are nevertheless the same as for l2: 1.0e-2 .. 1.1e-2)
    tol = GRAD_TOL
    gmax = max(g.abs().max().item() for g in ograds.values())
    worst = 0.0
    for k, p in model.named_parameters():
        if k in ograds:
            got, ref = eng.store.g(p).cpu(), ograds[k]
            err, nrm = (got - ref).double().norm().item(), ref.double().norm().item()
            if nrm > 1e-3 * gmax * ref.numel() ** 0.5:
                worst = max(worst, err / nrm)
            assert err <= tol * nrm + 1e-5 * gmax * ref.numel() ** 0.5, (k, err / max(nrm, 1e-12))
    observed(f"tiny/loss_variant/{loss}", "loss", abs(out.item() - want) / abs(want))
    observed(f"tiny/loss_variant/{loss}", "grad_worst", worst)


def test_forward_api_and_seeded_rng(golden_dir):
    """``MAE.forward`` contract + host RNG order: same global seed as the golden run -> identical masks."""
    dev, case, gold, ds, oracle, model, batch, noise, struct = _setup("c3p_dem_s1", golden_dir)
    torch.manual_seed(4242 + case["seed"])
    dbatch = {k: v.to(dev) for k, v in batch.items()}
    out_batch, pixels, masks, logits = model(dbatch, ssl_phase="pretrain")
    assert logits is None and set(pixels) == set(ds.dataset.inputs)
    for m in pixels:
        P = ds.dataset.inputs[m].patch_size.mae
        tok = masks[m][:, :, 0, ::P, ::P].flatten(2).cpu().numpy()
        ref_tok = np.unpackbits(gold[f"mask_tok/{m}"], axis=2)[:, :, : tok.shape[2]].astype(bool)
        assert np.array_equal(tok, ref_tok)
    np.testing.assert_allclose(out_batch["dem"].cpu().numpy(), gold["target/dem"], atol=1e-6)  # rescale_elev
    assert torch.equal(dbatch["dem"].cpu(), batch["dem"])  # caller's tensor is left untouched


@pytest.mark.parametrize("wgrad", ["fused", "deferred"])
def test_graph_replay_and_streams_match_eager(golden_dir, wgrad):
    """Steps 2+ replay captured hipGraphs with group-parallel streams; results must match the eager single-stream run."""
    dev, case, gold, ds, oracle, model, batch, noise, struct = _setup("c4_treesat", golden_dir)
    dbatch = {k: v.to(dev) for k, v in batch.items()}
    eng = model.engine(case["B"], dev, loss="l2_norm")
    eng.wgrad_mode = "fused"
    eng.use_graphs, eng.multi_stream = False, False
    ref_loss = eng.forward(dbatch, noise=noise, struct=struct).item()
    eng.zero_grad()
    eng.backward()
    ref_grad = eng.store.grad.clone()
    eng.use_graphs, eng.multi_stream, eng.wgrad_mode = True, True, wgrad
    for it in range(4):   # eager+streams, capture, replay, replay
        loss = eng.forward(dbatch, noise=noise, struct=struct).item()
        eng.zero_grad()
        eng.backward()
        torch.cuda.synchronize()
        assert abs(loss - ref_loss) < 1e-5 * abs(ref_loss), (it, loss, ref_loss)
        rel = ((eng.store.grad - ref_grad).norm() / ref_grad.norm()).item()
        assert rel < 1e-4, (it, rel)
    plan = "fused" if wgrad == "fused" else "all"
    assert set(eng._graphs) >= {"forward"} | {f"{seg}:{plan}:n" for seg in ("bwd_dec", "bwd_joint", "bwd_enc0")}


def test_overlapped_wgrad_plan_matches_inline(golden_dir, monkeypatch):
    """Opt-in plan ``MAESTRO_WGRAD_OVERLAP=1``: five backward segments, every segment's grouped weight gradients on a side
    stream under the next segment's dgrad chain (eager, captured and replayed) -- same gradients as the in-line plan."""
    monkeypatch.setenv("MAESTRO_WGRAD_OVERLAP", "1")
    dev, case, gold, ds, oracle, model, batch, noise, struct = _setup("c3_aerial_s2", golden_dir)
    dbatch = {k: v.to(dev) for k, v in batch.items()}
    eng = model.engine(case["B"], dev, loss="l2_norm")
    eng.wgrad_mode = "fused"
    eng.use_graphs, eng.multi_stream = False, False
    eng.forward(dbatch, noise=noise, struct=struct)
    eng.zero_grad()
    eng.backward()
    ref_grad = eng.store.grad.clone()
    eng.use_graphs, eng.multi_stream, eng.wgrad_mode = True, True, "deferred"
    for it in range(3):
        eng.forward(dbatch, noise=noise, struct=struct)
        eng.zero_grad()
        eng.backward()
        torch.cuda.synchronize()
        assert eng._plan == "ovl"
        rel = ((eng.store.grad - ref_grad).norm() / ref_grad.norm()).item()
        assert rel < 1e-4, (it, rel)
    assert {"bwd_dec:ovl:n", "bwd_joint:ovl:n", "bwd_enc0:ovl:n"} <= set(eng._graphs)


@pytest.mark.parametrize("interpolate", ["nearest", "bilinear", "bicubic"])
def test_input_resize_staging_matches_oracle(golden_dir, interpolate):
    """Rasters arriving at another resolution are resized on the GPU (mim.py:427-432) exactly like the oracle/reference."""
    dev, case, gold, ds, oracle, model, batch, noise, struct = _setup("c3_aerial_s2", golden_dir)
    model.interpolate = oracle.interpolate = interpolate
    small = dict(batch)
    small["aerial"] = torch.nn.functional.avg_pool2d(batch["aerial"].flatten(0, 1), 2).unflatten(0, batch["aerial"].shape[:2]).contiguous()
    eng = model.engine(case["B"], dev, loss="l2_norm")
    loss = eng.forward({k: v.to(dev) for k, v in small.items()}, noise=noise, struct=struct)
    ob = {k: v.clone() for k, v in small.items()}
    ob, orec, omsk, _ = oracle(ob, "pretrain", noise=noise, struct_masks={g: s[:, :, None] for g, s in struct.items()})
    oloss = om.compute_loss_rec(ob, orec, omsk, oracle.out_grid_size, om.norm_bands_of(ds.dataset), "l2_norm")
    assert abs(loss.item() - oloss.item()) < LOSS_TOL * abs(oloss.item())
    returned = eng.returned_batch({k: v.to(dev) for k, v in small.items()})
    assert (returned["aerial"].cpu() - ob["aerial"]).abs().max() < 2e-6 and returned["aerial"].shape[-1] == 64


@pytest.mark.parametrize("wgrad", ["fused", "deferred"])
def test_segmented_backward_with_grad_hook_matches_unsegmented(wgrad):
    """With a gradient hook (data parallel) the encoder-side backward is cut into layer ranges; gradients must not change,
    and the reported slices must tile the trainable flat buffer exactly once."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from maestro_amd.train.trainer import synthetic_batch
    dev = torch.device("cuda:0")
    ds = conf.DatasetsConfig(name_dataset="treesatai_ts", treesatai_ts=conf.TreeSatAITSConfig(
        filter_targets=[], aerial=conf.InputRasterConfig(image_size=60, patch_size=conf.PatchSizeConfig(mae=20), bands=4,
                                                         norm_bands=[1, 3], norm_fac=255.0)))
    torch.manual_seed(0)
    model = pmae.mae_tiny(datasets=ds, mask=conf.MaskConfig(), depth=7, inter_depth=1, fusion_mode="group", **{
        k: v for k, v in COMMON.items()})
    B = 2
    eng = model.engine(B, dev)
    batch = synthetic_batch(ds.dataset, B, dev)
    torch.manual_seed(5)
    noise, struct = eng.draw_masks()
    eng.use_graphs, eng.wgrad_mode = False, "fused"
    eng.forward(batch, noise=noise, struct=struct)
    eng.zero_grad()
    eng.backward()
    ref = eng.store.grad.clone()
    spans = []
    eng.grad_hook = lambda lo, hi: spans.append((lo, hi))
    eng.use_graphs, eng.wgrad_mode = True, wgrad
    for it in range(3):
        spans.clear()
        eng.forward(batch, noise=noise, struct=struct)
        eng.zero_grad()
        eng.backward()
        torch.cuda.synchronize()
        rel = ((eng.store.grad - ref).norm() / ref.norm()).item()
        assert rel < 1e-4, (it, rel)
        covered = sorted(set(spans))
        assert covered[0][0] == 0 and covered[-1][1] == eng.store.total
        assert all(a[1] == b[0] for a, b in zip(covered, covered[1:])), "reported gradient slices overlap or leave gaps"
    plan = "fused" if wgrad == "fused" else "enc"
    assert {f"bwd_enc{i}:{plan}:h" for i in range(3)} <= set(eng._graphs)


def test_ssl_module_lightning_style_step(golden_dir):
    """The drop-in surface as Lightning drives it: training_step -> loss.backward() -> torch optimizer step."""
    from types import SimpleNamespace

    from maestro_amd.train.model import SSLModule
    from maestro_amd.train.trainer import synthetic_batch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    dev = torch.device("cuda:0")
    ds = conf.DatasetsConfig(name_dataset="treesatai_ts", treesatai_ts=conf.TreeSatAITSConfig(
        filter_targets=[], aerial=conf.InputRasterConfig(image_size=60, patch_size=conf.PatchSizeConfig(mae=20), bands=4,
                                                         norm_bands=[1, 3], norm_fac=255.0)))
    torch.manual_seed(0)
    mod = SSLModule(datasets=ds, mask=conf.MaskConfig(), interpolate="nearest", fusion_mode="group", inter_depth=3,
                    model="mae", model_size="tiny", loss="l1_norm", use_ema=False)
    mod.trainer = SimpleNamespace(ssl_phase="pretrain", train_dataloader=SimpleNamespace(batch_size=2),
                                  accumulate_grad_batches=1, num_nodes=1, num_devices=1, base_lr=3e-3, wd=0.01, b1=0.9, b2=0.99,
                                  final_factor=1e7, estimated_stepping_batches=20, max_epochs=5)
    batch = synthetic_batch(ds.dataset, 2, dev)
    cfg = mod.configure_optimizers()
    opt, sched = cfg["optimizer"], cfg["lr_scheduler"]["scheduler"]
    losses = []
    for step in range(6):
        torch.manual_seed(11)                      # same masks every step -> the loss must go down
        out = mod.training_step(batch, step)
        assert set(out) == {"loss", "log_inputs", "log_preds", "log_targets"} and out["loss"].requires_grad
        opt.zero_grad(set_to_none=True)            # Lightning clears grads to None; backward must re-attach the views
        out["loss"].backward()
        eng = mod.model._engine
        for p in eng.store.params:
            assert p.grad is not None and p.grad.data_ptr() == eng.store.g(p).data_ptr()
        opt.step()
        sched.step()
        losses.append(out["loss"].item())
    assert losses[-1] < losses[0], losses
    assert mod.metrics["loss_rec_train"].count == 6
    name, img = next(iter(out["log_preds"].items()))
    assert name.startswith("pretrain_train/_aerial") and isinstance(img, torch.Tensor) and img.shape == (4, 60, 60)
    # torch.optim.AdamW updates the parameters behind the engine's back -- the bf16 shadows the GEMMs read
    # must follow (they are re-cast at the start of the next forward)
    mod.training_step(batch, 99)
    st = eng.store
    assert torch.equal(st.half.float(), st.flat.bfloat16().float()), "bf16 weight shadow is stale after a torch optimizer step"


@pytest.mark.parametrize("name", ["c3_aerial_s2", "c3p_dem_s1", "c5_s2naip_stress"])
def test_log_tensors_match_reference(golden_dir, name):
    """``training_step`` returns the image logs as TENSORS the reference's ``ImageLogger.to_numpy`` can consume
    (``maestro/train/logger.py:52-59``); values against what the reference's ``compute_logs_rec`` returned for the same
    weights, inputs and seed (``maestro/train/model.py:160-193``): inputs / targets come from the RETURNED batch
    (elevation-rescaled for ``dem``), reconstructions are blended in where the mask is set."""
    from types import SimpleNamespace

    from maestro_amd.train.model import SSLModule
    dev, case, gold, ds, oracle, model, batch, noise, struct = _setup(name, golden_dir)
    mod = SSLModule(datasets=ds, mask=conf.MaskConfig(), interpolate="nearest", fusion_mode=case["fusion"],
                    inter_depth=case["inter_depth"], model="mae", model_size="tiny", loss="l2_norm")
    mod.model = model                       # the case's reduced-depth model with the golden run's weights
    mod.trainer = SimpleNamespace(ssl_phase="pretrain")
    torch.manual_seed(4242 + case["seed"])  # the golden run's seed: same host draws, same masks
    out = mod.training_step({k: v.to(dev) for k, v in batch.items()}, 0)
    torch.cuda.synchronize()
    group_of = dict(ds.dataset.groups)
    multi = {g for g in set(group_of.values()) if sum(1 for v in group_of.values() if v == g) > 1}
    seen = 0
    for part in ("log_inputs", "log_preds", "log_targets"):
        for key, img in out[part].items():
            want = gold[f"logs/{key}"]
            got = img.detach().cpu().numpy().astype(np.float32)      # what ImageLogger.to_numpy does
            assert got.shape == want.shape, key
            mod_name = key.split("/_", 1)[1].rsplit("_", 1)[0]
            if part == "log_preds":
                if group_of[mod_name] in multi:
                    continue             # reconstruction depends on the reference's tie order there
                err = np.linalg.norm(got - want) / max(np.linalg.norm(want), 1e-12)
                assert err < PIX_TOL, (key, err)
            else:
                np.testing.assert_allclose(got, want, rtol=1e-6, atol=1e-6, err_msg=key)
            seen += 1
    assert seen >= 3 and len(out["log_inputs"]) == sum(1 for k in gold.files if k.endswith("_input"))


def test_loader_fed_batches_keep_graph_replay(golden_dir):
    """Batches arriving at a new device address every step (a data loader) are staged into engine-owned buffers, so the
    hipGraphs are captured once and replayed; results equal the resident-batch run."""
    dev, case, gold, ds, oracle, model, batch, noise, struct = _setup("c3_aerial_s2", golden_dir)
    dbatch = {k: v.to(dev) for k, v in batch.items()}
    eng = model.engine(case["B"], dev, loss="l2_norm")
    ref = []
    for _ in range(3):
        ref.append(eng.forward(dbatch, noise=noise, struct=struct).item())
    model._engine = None
    eng2 = model.engine(case["B"], dev, loss="l2_norm")
    keep = []
    for it in range(5):
        fresh = {k: v.clone() for k, v in dbatch.items()}     # new addresses every step
        keep.append(fresh)                                    # (kept alive so the allocator cannot hand the same block back)
        loss = eng2.forward(fresh, noise=noise, struct=struct).item()
        eng2.zero_grad()
        eng2.backward()
        assert abs(loss - ref[0]) <= 1e-5 * abs(ref[0]), (it, loss, ref[0])
    assert all(eng2._inputs[k]["buf"] is not None for k in ("aerial", "s2"))
    assert "forward" in eng2._graphs and any(k.startswith("bwd_dec") for k in eng2._graphs)


def test_gradient_accumulation_sums_micro_batches():
    """accumulate_grad_batches (reference ``conf/trainer.py``, lr rule ``model.py:120-128``): the engine stores its gradients,
    so both surfaces add micro-batches explicitly -- Lightning style (two backward() calls without zero_grad in between,
    loss divided by 2) and ``PretrainLoop.step([mb1, mb2])``.  Both must equal the hand-made mean of the two gradients."""
    from types import SimpleNamespace

    from maestro_amd.train.model import SSLModule
    from maestro_amd.train.trainer import PretrainLoop, synthetic_batch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    dev = torch.device("cuda:0")
    ds = conf.DatasetsConfig(name_dataset="treesatai_ts", treesatai_ts=conf.TreeSatAITSConfig(
        filter_targets=[], aerial=conf.InputRasterConfig(image_size=60, patch_size=conf.PatchSizeConfig(mae=20), bands=4,
                                                         norm_bands=[1, 3], norm_fac=255.0)))
    torch.manual_seed(0)
    mod = SSLModule(datasets=ds, mask=conf.MaskConfig(), interpolate="nearest", fusion_mode="group", inter_depth=3,
                    model="mae", model_size="tiny", loss="l2_norm", use_ema=False)
    mod.trainer = SimpleNamespace(ssl_phase="pretrain")
    mbs = [synthetic_batch(ds.dataset, 2, dev, seed=s) for s in (1, 2)]
    eng = mod.model.engine(2, dev, loss="l2_norm")
    st = eng.store
    single = []
    for i, mb in enumerate(mbs):                  # reference: each micro-batch on its own, fixed mask seeds
        torch.manual_seed(20 + i)
        eng.forward(mb)
        eng.zero_grad()
        eng.backward()
        single.append(st.grad.clone())
    want = 0.5 * (single[0] + single[1])
    assert (single[0] - single[1]).norm() > 1e-3 * want.norm()
    # Lightning style
    for p in st.params:
        p.grad = None
    for i, mb in enumerate(mbs):
        torch.manual_seed(20 + i)
        (mod.training_step(mb, i)["loss"] / 2).backward()
    torch.cuda.synchronize()
    rel = ((st.grad - want).norm() / want.norm()).item()
    assert rel < 1e-5, rel
    assert all(p.grad.data_ptr() == st.g(p).data_ptr() for p in st.params)
    # own loop: the optimizer sees the SUM with grad_scale 1/2
    loop = PretrainLoop(mod.model, 2, dev, loss="l2_norm", accumulate=2)
    assert loop.engine is eng
    seen = {}
    loop.opt.step = lambda lr=None, grad_scale=1.0, **kw: seen.update(g=st.grad.clone(), scale=grad_scale)
    torch.manual_seed(20)                         # one seed for the pair: the manual replay below draws the same masks
    loop.step(mbs)
    assert seen["scale"] == 0.5
    torch.manual_seed(20)
    eng.forward(mbs[0]); eng.zero_grad(); eng.backward(); g0 = st.grad.clone()
    eng.forward(mbs[1]); eng.zero_grad(); eng.backward(); g1 = st.grad.clone()
    rel = ((seen["g"] - (g0 + g1)).norm() / (g0 + g1).norm()).item()
    assert rel < 1e-5, rel


@pytest.mark.parametrize("fusion,inter", [("group", 1), ("shared", 0)])
def test_optimizer_overlapped_with_next_forward_matches_classic_step(fusion, inter):
    """``PretrainLoop(overlap_optimizer=True)`` queues AdamW of step t and runs it, cut into per-layer stages on a side
    stream, inside the (captured) forward of step t+1.  Same arithmetic, same order per element: after ``flush()`` the
    parameters, the optimizer moments and every step's loss equal the classic end-of-step update."""
    from maestro_amd.train.trainer import PretrainLoop, synthetic_batch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    dev = torch.device("cuda:0")
    ds = conf.DatasetsConfig(name_dataset="treesatai_ts", treesatai_ts=conf.TreeSatAITSConfig(
        filter_targets=[], aerial=conf.InputRasterConfig(image_size=60, patch_size=conf.PatchSizeConfig(mae=20), bands=4,
                                                         norm_bands=[1, 3], norm_fac=255.0)))
    batch = synthetic_batch(ds.dataset, 2, dev, seed=3)
    runs = {}
    for overlap in (False, True, "again"):      # "again": the classic step a second time = the run-to-run noise floor
        key, overlap = overlap, overlap is True
        torch.manual_seed(0)
        model = pmae.mae_tiny(datasets=ds, mask=conf.MaskConfig(), depth=3, inter_depth=inter, fusion_mode=fusion,
                              **{k: v for k, v in COMMON.items()})
        loop = PretrainLoop(model, 2, dev, loss="l2_norm", base_lr=3e-3, total_steps=12, overlap_optimizer=overlap)
        torch.manual_seed(7)
        losses = [loop.step(batch).item() for _ in range(4)]   # eager, capture + replay, replay, replay
        if overlap:
            assert loop.engine._opt_pending is not None and "forward:opt" in loop.engine._graphs
        loop.flush()
        torch.cuda.synchronize()
        runs[key] = (losses, loop.engine.store.flat.clone(), loop.opt.m.clone(), loop.opt.v.clone(), loop.opt.t,
                     loop.engine.store.half.float().clone())
    (l0, p0, m0, v0, t0, h0), (l1, p1, m1, v1, t1, h1), (l2, p2, m2, v2, _, _) = runs[False], runs[True], runs["again"]
    assert t0 == t1 == 4
    assert l0[0] != l0[-1]
    for a, b in zip(l0, l1):
        assert abs(a - b) <= 2e-4 * abs(a), (l0, l1)
    # Same arithmetic per element (bias corrections included), so the only difference is the run-to-run order of the atomic
    # gradient sums, which Adam amplifies (a sign flip of a near-zero gradient is a full +-lr step): typically 1e-9..1e-7
    # after a few steps (up to 5e-5 late in a long test session), also between two classic runs.  A wrong stage or scalar shows up at >= 3e-4.
    rel = lambda a, b: ((a - b).norm() / b.norm()).item()  # noqa: E731
    noise = (rel(p2, p0), rel(m2, m0), rel(v2, v0))
    got = (rel(p1, p0), rel(m1, m0), rel(v1, v0))
    # (a float-vs-double bias correction, the one real discrepancy found while writing this, gave 3.7e-4 after four steps)
    for g, n, floor in zip(got, noise, (1e-4, 1e-3, 1e-3)):   # the bar is the larger of a fixed floor and the measured spread
        assert g < max(floor, 3.0 * n), (got, noise)
    assert torch.equal(h1, p1.bfloat16().float()), "bf16 shadow out of date after the overlapped update"


@pytest.mark.parametrize("inter_depth", [1, 0])
def test_identity_enc_to_dec_when_widths_match(golden_dir, inter_depth):
    """``embed_dim == decoder_dim``: the reference builds ``nn.Identity`` instead of a Linear (``maestro/ssl/mae.py:145-154``).
    No shipped size has it; built with ``decoder_dim = 192`` on the tiny preset and checked against the oracle (which restates
    the same rule) with and without the joint encoder in front."""
    dev, case, gold, ds, _, _, batch, noise, struct = _setup("c3_aerial_s2", golden_dir)
    kw = dict(fusion_mode="group", inter_depth=inter_depth, depth=3, decoder_dim=192, decoder_heads=6, decoder_dim_head=32, **COMMON)
    oracle = om.build_oracle(ds, conf.MaskConfig(), model_size="tiny", **kw)
    init_weights(oracle, 19)
    model = pmae.mae_tiny(datasets=ds, mask=conf.MaskConfig(), **kw)
    assert all(isinstance(mod, torch.nn.Identity) for mod in model.enc_to_dec.values())
    model.load_state_dict(oracle.state_dict(), strict=True)
    eng = model.engine(case["B"], dev, loss="l2_norm")
    loss = eng.forward({k: v.to(dev) for k, v in batch.items()}, noise=noise, struct=struct).clone()
    eng.zero_grad()
    eng.backward()
    torch.cuda.synchronize()
    ob, orec, omsk, _ = oracle({k: v.clone() for k, v in batch.items()}, "pretrain", noise=noise,
                               struct_masks={g: s[:, :, None] for g, s in struct.items()})
    oloss = om.compute_loss_rec(ob, orec, omsk, oracle.out_grid_size, om.norm_bands_of(ds.dataset), "l2_norm")
    oracle.zero_grad()
    oloss.backward()
    assert abs(loss.item() - oloss.item()) < LOSS_TOL * abs(oloss.item()), (loss.item(), oloss.item())
    pixels, masks = eng.reconstructions()
    for m in orec:
        assert torch.equal(masks[m].cpu(), omsk[m]) and _rel(pixels[m].cpu(), orec[m].detach()) < PIX_TOL
    ograds = {k: p.grad for k, p in oracle.named_parameters() if p.grad is not None}
    gmax = max(g.abs().max().item() for g in ograds.values())
    for k, p in model.named_parameters():
        if k in ograds:
            got, want = eng.store.g(p).cpu(), ograds[k]
            err, ref = (got - want).double().norm().item(), want.double().norm().item()
            assert err <= GRAD_TOL * ref + 1e-5 * gmax * want.numel() ** 0.5, (k, err / max(ref, 1e-12))


@pytest.mark.parametrize("name", ["bg_aerial_s2", "bg_ts_monotemp"])
def test_band_groups_under_the_exchange_plan(golden_dir, name):
    """Several band-groups per modality inside the data-parallel launch plan (no process group: the plan without the
    collectives): the reported gradient slices tile the buffer exactly once per step although a modality's band-groups share
    its patch-embed / pixelify / mask-token parameters, and the losses equal the plain loop's."""
    from maestro_amd.train.trainer import PretrainLoop
    dev, case, gold, ds, oracle, model, batch, noise, struct = _setup(name, golden_dir)
    dbatch = {k: v.to(dev) for k, v in batch.items()}
    runs = []
    for exchange in (False, True):
        model.load_state_dict(oracle.state_dict(), strict=True)
        model._engine = None
        loop = PretrainLoop(model, case["B"], dev, total_steps=8, exchange=exchange or None, bucket_mb=1)
        torch.manual_seed(21)
        runs.append([float(loop.step(dbatch).item()) for _ in range(4)])
        if exchange:
            covered = sorted(loop.sync.launched)
            assert covered[0][0] == 0 and covered[-1][1] == loop.engine.store.grad_all.numel()
            assert all(a[1] == b[0] for a, b in zip(covered, covered[1:])), covered
    assert all(abs(a - b) < 1e-3 * abs(a) for a, b in zip(*runs)), runs


def test_engine_adamw_equals_torch_adamw_and_interchanges_state(golden_dir):
    """``configure_optimizers`` returns ``EngineAdamW``: a ``torch.optim.AdamW`` whose step is one fused launch over the flat buffer.
    Same trajectory as torch's own implementation on the Lightning-style loop (same masks every step), and its ``state_dict``
    loads into a plain ``torch.optim.AdamW`` (the reference's optimizer, ``maestro/train/model.py:135-140``) and back."""
    from types import SimpleNamespace

    from maestro_amd.train.model import SSLModule
    from maestro_amd.train.optim import EngineAdamW
    from maestro_amd.train.trainer import synthetic_batch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    dev = torch.device("cuda:0")
    ds = conf.DatasetsConfig(name_dataset="treesatai_ts", treesatai_ts=conf.TreeSatAITSConfig(
        filter_targets=[], aerial=conf.InputRasterConfig(image_size=60, patch_size=conf.PatchSizeConfig(mae=20), bands=4,
                                                         norm_bands=[1, 3], norm_fac=255.0)))
    batch = synthetic_batch(ds.dataset, 2, dev)

    def build():
        torch.manual_seed(0)
        mod = SSLModule(datasets=ds, mask=conf.MaskConfig(), interpolate="nearest", fusion_mode="group", inter_depth=3,
                        model="mae", model_size="tiny", loss="l2_norm", use_ema=False)
        mod.trainer = SimpleNamespace(ssl_phase="pretrain", train_dataloader=SimpleNamespace(batch_size=2),
                                      accumulate_grad_batches=1, num_nodes=1, num_devices=1, base_lr=3e-3, wd=0.01, b1=0.9, b2=0.99,
                                      final_factor=1e7, estimated_stepping_batches=20, max_epochs=5)
        return mod

    def run(mod, opt, sched, steps, first=0):
        for step in range(first, first + steps):
            torch.manual_seed(11 + step)
            out = mod.training_step(batch, step)
            opt.zero_grad(set_to_none=True)
            out["loss"].backward()
            opt.step()
            sched.step()
        torch.cuda.synchronize()
        return torch.cat([p.detach().reshape(-1) for _, p in sorted(mod.named_parameters()) if p.numel() > 1]).cpu()

    def sched_of(opt, mod):  # noqa: ARG001
        return torch.optim.lr_scheduler.OneCycleLR(opt, max_lr=max_lr, total_steps=20, pct_start=0.2,
                                                   cycle_momentum=False, div_factor=1000, final_div_factor=1e4)

    # (a) fused step against torch's own AdamW, 4 steps
    mod_a = build()
    cfg = mod_a.configure_optimizers()
    opt_a, sched_a = cfg["optimizer"], cfg["lr_scheduler"]["scheduler"]
    max_lr = opt_a.defaults["lr"]
    assert isinstance(opt_a, EngineAdamW) and isinstance(opt_a, torch.optim.AdamW)
    init = torch.cat([p.detach().reshape(-1) for _, p in sorted(mod_a.named_parameters()) if p.numel() > 1]).cpu().clone()
    # step by step against torch.optim.AdamW fed the SAME gradients and the same weights (deterministic comparison)
    params_a = [p for n, p in mod_a.named_parameters() if n != "_anchor"]
    twins = [torch.nn.Parameter(p.detach().clone().to(dev)) for p in params_a]   # (the engine moves the trained parameters to the GPU)
    opt_t = torch.optim.AdamW(twins, lr=opt_a.defaults["lr"], weight_decay=0.01, betas=(0.9, 0.99))
    sched_t = sched_of(opt_t, None)
    for step in range(4):
        torch.manual_seed(11 + step)
        out = mod_a.training_step(batch, step)
        opt_a.zero_grad(set_to_none=True)
        out["loss"].backward()
        for t, p in zip(twins, params_a):
            t.data.copy_(p.detach())
            t.grad = None if p.grad is None else p.grad.detach().clone().to(dev)
        opt_a.step()
        sched_a.step()
        opt_t.step()
        sched_t.step()
        worst = max(((p.detach().to(dev) - t.detach()).abs().max() / t.detach().abs().max().clamp(min=1e-3)).item() for t, p in zip(twins, params_a))
        assert worst < 2e-6, (step, worst)
    torch.cuda.synchronize()
    got = torch.cat([p.detach().reshape(-1) for _, p in sorted(mod_a.named_parameters()) if p.numel() > 1]).cpu()
    assert opt_a._fused is not None and opt_a._fused.t == 4, "the fused launch was not used"
    mod_b = build()
    params_b = [p for n, p in mod_b.named_parameters() if n != "_anchor"]
    opt_b = torch.optim.AdamW(params_b, lr=opt_a.defaults["lr"], weight_decay=0.01, betas=(0.9, 0.99))
    sched_b = sched_of(opt_b, mod_b)
    want = run(mod_b, opt_b, sched_b, 4)
    upd_got, upd_want = got - init, want - init
    rel = ((upd_got - upd_want).norm() / upd_want.norm()).item()
    # (two separate RUNS: the early AdamW updates are ~ lr * sign(g), so the atomics' summation order shows at the 1e-3 level)
    assert upd_want.abs().max() > 0 and rel < 2e-2, rel
    # (b) state interchange, again step by step on identical gradients:
    #     torch.optim.AdamW's state -> EngineAdamW (adopted into the flat moment buffers at the next step) ...
    def lockstep(opt_fused, sched_fused, opt_torch, sched_torch, first, steps):
        for step in range(first, first + steps):
            torch.manual_seed(11 + step)
            out = mod_a.training_step(batch, step)
            opt_fused.zero_grad(set_to_none=True)
            out["loss"].backward()
            for t, p in zip(twins, params_a):
                t.data.copy_(p.detach())
                t.grad = None if p.grad is None else p.grad.detach().clone().to(dev)
            opt_fused.step()
            sched_fused.step()
            opt_torch.step()
            sched_torch.step()
            worst = max(((p.detach().to(dev) - t.detach()).abs().max() / t.detach().abs().max().clamp(min=1e-3)).item()
                        for t, p in zip(twins, params_a))
            assert worst < 2e-6, (step, worst)

    import io

    def through_a_file(sd):          # (load_state_dict aliases the tensors it is given: interchange goes through a checkpoint file)
        buf = io.BytesIO()
        torch.save(sd, buf)
        buf.seek(0)
        return torch.load(buf, weights_only=False)

    sd_b = opt_b.state_dict()
    assert set(sd_b["state"][0]) == {"step", "exp_avg", "exp_avg_sq"}
    opt_a.load_state_dict(through_a_file(sd_b))
    opt_t.load_state_dict(through_a_file(sd_b))
    assert opt_a._bound is None
    lockstep(opt_a, sched_a, opt_t, sched_t, 4, 2)
    assert opt_a._fused.t == 6 and float(opt_a.state[params_a[0]]["step"]) == 6.0
    #     ... and EngineAdamW's state -> a fresh torch.optim.AdamW (what the reference would load from our checkpoint)
    sd_a = opt_a.state_dict()
    assert set(sd_a["state"][0]) == {"step", "exp_avg", "exp_avg_sq"} and float(sd_a["state"][0]["step"]) == 6.0
    assert len(sd_a["state"]) == len(sd_b["state"])          # the same parameters carry state (those that received gradients)
    opt_f = torch.optim.AdamW(twins, lr=opt_a.defaults["lr"], weight_decay=0.01, betas=(0.9, 0.99))
    opt_f.load_state_dict(through_a_file(sd_a))
    sched_f = sched_of(opt_f, None)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")     # ("scheduler stepped before the optimizer": it is being restored to position 6)
        sched_f.last_epoch = 5
        sched_f.step()
    lockstep(opt_a, sched_a, opt_f, sched_f, 6, 2)


def test_instep_tile_tuner_keeps_the_step_intact(golden_dir):
    """The opt-in first-step tuner (``MAESTRO_INSTEP_TUNE=1`` / ``engine.instep_tune``) recomputes the step once per candidate tile:
    the result of the step is the one a plain engine gives (any tile computes the same GEMM), exactly one set of draws is used,
    and every GEMM signature of the step got a decision."""
    from maestro_amd import hip
    dev, case, gold, ds, oracle, model, batch, noise, struct = _setup("c3_aerial_s2", golden_dir)
    dbatch = {k: v.to(dev) for k, v in batch.items()}
    eng = model.engine(case["B"], dev, loss="l2_norm")
    eng.forward(dbatch, noise=noise, struct=struct)
    eng.zero_grad()
    eng.backward()
    torch.cuda.synchronize()
    want_loss, want_grad = eng.loss_acc.clone(), eng.store.grad.clone()
    before = dict(hip.gemm_tile_choices())
    try:
        model._engine = None                      # a fresh engine on the same parameters
        eng2 = model.engine(case["B"], dev, loss="l2_norm")
        draws = []
        inner = eng2.draw_masks
        eng2.draw_masks = lambda *a, **k: draws.append(1) or inner(*a, **k)
        eng2.instep_tune = True
        eng2.forward(dbatch, noise=noise, struct=struct)
        eng2.zero_grad()
        eng2.backward()
        torch.cuda.synchronize()
        assert eng2.instep_tune is False and eng2.tile_report, "the tuning passes did not run"
        assert not draws, "injected draws must be reused by every pass"
        assert all(pick in hip.InStepTuner.CANDIDATES and hip.TILE_AUTO in ms for pick, ms in eng2.tile_report.values())
        assert abs(eng2.loss_acc.item() - want_loss.item()) < 2e-3 * abs(want_loss.item())
        rel = ((eng2.store.grad - want_grad).norm() / want_grad.norm()).item()
        assert rel < 5e-3, rel                     # other tiles: the GELU differs by one bf16 ulp in a few elements
        # a second step replays the captured graphs with the chosen tiles
        for _ in range(2):
            eng2.forward(dbatch, noise=noise, struct=struct)
            eng2.zero_grad()
            eng2.backward()
        torch.cuda.synchronize()
        assert ((eng2.store.grad - want_grad).norm() / want_grad.norm()).item() < 5e-3 and eng2._graphs
    finally:
        hip._tile_choice.clear()
        hip._tile_choice.update(before)


def _tiny_treesat():
    ds = conf.DatasetsConfig(name_dataset="treesatai_ts", treesatai_ts=conf.TreeSatAITSConfig(
        filter_targets=[], aerial=conf.InputRasterConfig(image_size=60, patch_size=conf.PatchSizeConfig(mae=20), bands=4,
                                                         norm_bands=[1, 3], norm_fac=255.0)))
    torch.manual_seed(0)
    model = pmae.mae_tiny(datasets=ds, mask=conf.MaskConfig(), interpolate="nearest", fusion_mode="group", inter_depth=1,
                          model="mae", num_levels=1, depth=2)
    return ds, model


def test_rebuilt_engines_do_not_accumulate_retired_graphs():
    """an engine that captured hipGraphs and is dropped (``ssl/mae.py`` rebuilds the engine whenever the batch size
    changes: partial last batch, validation) hands its graphs to a retire list; the NEXT engine's constructor destroys them with
    the device idle -- the list, and with it the dead engines' private memory pools, must not grow with every rebuild."""
    import gc

    from maestro_amd import engine as E
    from maestro_amd.train.trainer import PretrainLoop, synthetic_batch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    dev = torch.device("cuda:0")
    ds, model = _tiny_treesat()
    losses = []
    for B in (2, 3, 2):  # noqa: N806
        loop = PretrainLoop(model, B, dev, total_steps=20)
        batch = synthetic_batch(ds.dataset, B, dev)
        torch.manual_seed(5)
        for _ in range(4):
            losses.append(float(loop.step(batch).item()))
        assert len(loop.engine._graphs) >= 2, "the segments should have been captured by now"
        assert not E._RETIRED_GRAPHS, "the previous engine's graphs were not destroyed when this engine was built"
        model._engine = None
        del loop
        gc.collect()
        assert len(E._RETIRED_GRAPHS) <= 1          # retired, not destroyed, by the collector ...
    E.drain_retired_graphs()
    assert not E._RETIRED_GRAPHS and all(x == x for x in losses)


def test_a_bare_forward_never_runs_backward_passes():
    """the start-up passes (forward + zero_grad + backward, repeated) are opt-in for the explicit training loops;
    a forward of a bare engine -- Lightning's sanity-check validation, predict-only runs -- leaves the gradient buffer alone."""
    from maestro_amd.train.trainer import synthetic_batch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    dev = torch.device("cuda:0")
    ds, model = _tiny_treesat()
    eng = model.engine(2, dev, loss="l2_norm")
    assert eng.warm_passes == 0
    eng.store.grad.fill_(7.0)
    eng.forward(synthetic_batch(ds.dataset, 2, dev))
    torch.cuda.synchronize()
    assert bool((eng.store.grad == 7.0).all()), "a plain forward touched the gradient buffer"


def test_engine_adamw_leaves_stale_gradients_outside_the_span_to_torch():
    """a parameter of the optimizer's group that the engine's trainable span does not cover but that carries a
    gradient (a phase change with ``set_to_none=False``, a parameter the engine does not own) is updated by
    ``torch.optim.AdamW.step``; the fused launch would skip it, so ``EngineAdamW`` must take torch's path then."""
    from maestro_amd.train.optim import EngineAdamW
    from maestro_amd.train.trainer import synthetic_batch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    dev = torch.device("cuda:0")
    ds, model = _tiny_treesat()
    eng = model.engine(2, dev, loss="l2_norm")
    extra = torch.nn.Parameter(torch.ones(8, device=dev))
    opt = EngineAdamW(list(eng.store.params) + [extra], lambda: eng, lr=1e-2, weight_decay=0.0)
    eng.forward(synthetic_batch(ds.dataset, 2, dev))
    eng.zero_grad()
    eng.backward()
    for p in eng.store.params:
        p.grad = eng.store.g(p)
    eng.store.fresh = False
    assert opt._eligible(eng), "the fused launch should serve the plain case"
    extra.grad = torch.full_like(extra, 0.5)
    assert not opt._eligible(eng)
    before = extra.detach().clone()
    opt.step()
    torch.cuda.synchronize()
    assert not torch.equal(extra.detach(), before), "the stale-gradient parameter was not updated"
