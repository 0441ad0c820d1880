"""Host-side logic of the product (configs, geometry, RNG order, schedules) on CPU."""

import itertools

import pytest
import torch

import maestro_amd.conf as conf
from maestro_amd.layers.utils import draw_struct_masks, pool_pos_table, posemb_sincos_2d
from maestro_amd.ssl.mae import mae_large, mae_medium, mae_small, mae_tiny
from maestro_amd.train.model import SSLModule
from maestro_amd.train.optim import OneCycle, scaled_lr
from oracle import layers as ol
from oracle import mae as om

ARGS = dict(interpolate="nearest", model="mae", num_levels=1, type_head="attentive", fac_abs_enc=1.0, fac_date_enc=1.0)


def treesat():
    return conf.DatasetsConfig(root_dir=None, name_dataset="treesatai_ts", treesatai_ts=conf.TreeSatAITSConfig(rel_dir=""))


# mirrors the reference's tests/test_mae.py (construction for every fusion mode / option)
@pytest.mark.parametrize("fusion_mode", ["shared", "monotemp", "mod", "group"])
def test_mae_constructs(fusion_mode):
    m = mae_tiny(datasets=treesat(), mask=conf.MaskConfig(), fusion_mode=fusion_mode, inter_depth=0, **ARGS)
    want = {"shared": ["shared"], "monotemp": ["aerial", "s2", "s1_asc", "s1_des"],
            "mod": ["aerial", "s2", "s1_asc", "s1_des"], "group": ["aerial", "s2", "s1"]}[fusion_mode]
    assert list(m.encoder) == want and m.encoder_inter is None
    assert "mask_token.s1_asc" in m.state_dict() and "enc_pos_encoding" not in m.state_dict()


# tests/test_mae.py:42-74 of the reference: test_mae_architecture
@pytest.mark.parametrize("interpolate,type_head,fac_abs_enc,fac_date_enc", itertools.product(
    ["nearest", "bilinear"], ["attentive", "linear"], [1.0, 0.0], [1.0, 0.0]))
def test_mae_architecture_like_reference(interpolate, type_head, fac_abs_enc, fac_date_enc):
    m = mae_tiny(datasets=treesat(), mask=conf.MaskConfig(), interpolate=interpolate, fusion_mode="group", model="mae",
                 inter_depth=0, num_levels=1, type_head=type_head, fac_abs_enc=fac_abs_enc, fac_date_enc=fac_date_enc)
    assert m.interpolate == interpolate and m.fac_date_enc == fac_date_enc
    assert float(m.enc_pos_encoding.abs().max()) == (0.0 if fac_abs_enc == 0.0 else 1.0)


@pytest.mark.parametrize("factory,params_m", [(mae_medium, 176.2), (mae_large, None), (mae_small, None)])
def test_sizes_and_state_dict_keys_match_oracle(factory, params_m):
    ds = conf.DatasetsConfig(name_dataset="flair", flair=conf.FLAIRConfig(filter_inputs=["aerial", "s2"], filter_targets=[]))
    m = factory(datasets=ds, mask=conf.MaskConfig(), fusion_mode="group", inter_depth=3, **ARGS)
    size = {mae_medium: "medium", mae_large: "large", mae_small: "small"}[factory]
    o = om.build_oracle(ds, conf.MaskConfig(), model_size=size, fusion_mode="group", inter_depth=3, **ARGS)
    assert {k: tuple(v.shape) for k, v in m.state_dict().items()} == {k: tuple(v.shape) for k, v in o.state_dict().items()}
    if params_m:
        assert abs(sum(p.numel() for p in m.parameters()) / 1e6 - params_m) < 0.1
    g = m.group_specs
    assert (g["aerial"].L, g["aerial"].k, g["s2"].L, g["s2"].k, m.joint_N) == (1024, 768, 400, 300, 356)


# mirrors the reference's tests/test_model.py (SSLModule construction + its two constructor errors)
@pytest.mark.parametrize("fusion_mode,inter_depth,loss", itertools.product(["mod", "group"], [0, 3], ["l1_norm", "l1"]))
def test_ssl_module_constructs(fusion_mode, inter_depth, loss):
    mod = SSLModule(datasets=treesat(), mask=conf.MaskConfig(), interpolate="nearest", fusion_mode=fusion_mode,
                    inter_depth=inter_depth, model="mae", model_size="tiny", loss=loss)
    assert mod.model.inter_depth == inter_depth and "loss_rec_train" in mod.metrics
    assert any(k.startswith("model.encoder.") for k in mod.state_dict())


# tests/test_model.py:66-100 of the reference: test_ssl_architecture (interpolate x head type x loss, then .setup(stage))
@pytest.mark.parametrize("interpolate,type_head,loss,stage", itertools.product(
    ["nearest", "bilinear"], ["attentive", "linear"], ["l1_norm", "l1"], ["train", "val", "test"]))
def test_ssl_architecture_like_reference(interpolate, type_head, loss, stage):
    mod = SSLModule(datasets=treesat(), mask=conf.MaskConfig(), interpolate=interpolate, fusion_mode="group", inter_depth=0,
                    model="mae", model_size="tiny", type_head=type_head, loss=loss)
    mod.setup(stage=stage)
    head = mod.model.heads["treesat_mlc_thresh"]
    assert hasattr(head, "reduce") == (type_head == "attentive") and head.linear.out_features == 15


@pytest.mark.parametrize("fusion_mode,stage", itertools.product(["shared", "monotemp", "mod", "group"], ["train", "val", "test"]))
def test_ssl_mae_like_reference(fusion_mode, stage):        # tests/test_model.py:8-34
    mod = SSLModule(datasets=treesat(), mask=conf.MaskConfig(), interpolate="nearest", fusion_mode=fusion_mode, inter_depth=0,
                    model="mae", model_size="tiny")
    mod.setup(stage=stage)


def test_ssl_module_errors():
    with pytest.raises(NotImplementedError):
        SSLModule(datasets=treesat(), mask=conf.MaskConfig(), interpolate="nearest", fusion_mode="shared", inter_depth=3,
                  model="mae", model_size="tiny")
    with pytest.raises(ValueError):
        SSLModule(datasets=treesat(), mask=conf.MaskConfig(), interpolate="nearest", fusion_mode="group", inter_depth=0,
                  model="mae", model_size="huge")
    with pytest.raises(ValueError):
        SSLModule(datasets=treesat(), mask=conf.MaskConfig(), interpolate="nearest", fusion_mode="group", inter_depth=0,
                  model="mae", model_size="tiny", loss="l3")
    with pytest.raises(ValueError):      # tests/test_model.py:135-145 (xfail in the reference): unknown model name
        SSLModule(datasets=treesat(), mask=conf.MaskConfig(), interpolate="nearest", fusion_mode="group", inter_depth=0,
                  model="MAE", model_size="tiny")
    with pytest.raises(TypeError):       # tests/test_model.py:147-149: missing constructor arguments
        SSLModule(model="mae", model_size="tiny")


def test_positional_tables_match_oracle():
    assert torch.equal(posemb_sincos_2d(12, 12, 40, 8), ol.posemb_sincos_2d(12, 12, 40, 8))
    tab = posemb_sincos_2d(96, 96, 24, 8)
    for grid in (3, 5, 15, 96):
        assert torch.equal(pool_pos_table(tab, grid), ol.pool_pos_encoding(tab, grid))


@pytest.mark.parametrize("fusion_mode", ["group", "mod", "shared"])
def test_struct_mask_draw_order_matches_oracle(fusion_mode):
    """Same global seed -> the product's host draws equal the oracle's (which equal the reference's, see goldens)."""
    ds = treesat()
    kw = dict(fusion_mode=fusion_mode, inter_depth=0, **ARGS)
    m = mae_tiny(datasets=ds, mask=conf.MaskConfig(), depth=2, **kw)
    o = om.build_oracle(ds, conf.MaskConfig(), model_size="tiny", depth=2, **kw)
    B = 3
    fold = fusion_mode in ("shared", "monotemp")
    for s in m.mod_specs.values():
        s.Beff = B * s.Dates if fold else B
    groups = list(m.group_specs.values())
    for g in groups:
        g.Beff = g.mods[0].Beff
    torch.manual_seed(123)
    got = draw_struct_masks(groups, m.mod_specs)
    noise_got = {g.name: torch.rand(g.Beff, g.L) for g in groups}
    torch.manual_seed(123)
    want = o.draw_struct_masks({g.name: (g.Beff, g.L) for g in groups})
    noise_want = {g.name: torch.rand(g.Beff, g.L) for g in groups}
    for g in groups:
        assert torch.equal(got[g.name], want[g.name][:, :, 0])
        assert torch.equal(noise_got[g.name], noise_want[g.name])
        assert g.k == om.OracleMAE.num_masked(o.mask_ratio[g.name], g.L)


@pytest.mark.parametrize("name", ["bg_aerial_s2", "bg_dem_mod"])
def test_band_groups_specs_and_draws(name):
    """Several band-groups per modality (``maestro/ssl/mim.py:49-57``, ``mae.py:193-210``): one spec per band-group in (g, d)
    order on the group's token axis, and the host draws -- including the per-(sample, band-group) ``mask_bands`` draw --
    equal the oracle's (which equal the reference's recorded draws, see ``tests/test_oracle_golden.py``)."""
    from oracle.gen_golden import build_datasets, case_table
    case = case_table()[name]
    ds = build_datasets(case, conf)
    mask = conf.MaskConfig(**case["mask_kw"])
    kw = dict(fusion_mode=case["fusion"], inter_depth=case["inter_depth"], **ARGS, **case["model_kw"])
    m = mae_tiny(datasets=ds, mask=mask, **kw)
    o = om.build_oracle(ds, mask, model_size="tiny", **kw)
    src = next(n for n, parts in m.src_specs.items() if len(parts) > 1)
    parts = m.src_specs[src]
    assert [p.name for p in parts] == [f"{src}#{i}" for i in range(len(parts))] and all(p.src == src for p in parts)
    assert [p.c0 for p in parts] == [0] + list(torch.tensor([p.C for p in parts]).cumsum(0)[:-1])
    assert all(b.tok_off == a.tok_off + a.n_tok and b.slot == a.slot + 1 for a, b in zip(parts, parts[1:]))
    assert m.mask_token[src].shape[1] == len(parts) and parts[0].p_bands == case["mask_kw"]["mask_bands"]
    B = 4  # noqa: N806
    for s in m.mod_specs.values():
        s.Beff = B
    groups = list(m.group_specs.values())
    for g in groups:
        g.Beff = B
    for seed in (1, 2, 3):
        torch.manual_seed(seed)
        got = draw_struct_masks(groups, m.mod_specs)
        torch.manual_seed(seed)
        want = o.draw_struct_masks({g.name: (g.Beff, g.L) for g in groups})
        for g in groups:
            assert torch.equal(got[g.name], want[g.name][:, :, 0])
    # dates folded into the batch: the band-groups fold with them -> one sequence set per band-group, rows (b, g, d) of the
    # reference group's noise draw
    f = mae_tiny(datasets=ds, mask=mask, **dict(kw, fusion_mode="shared", inter_depth=0))
    fparts = [f.group_specs[p.group] for p in f.src_specs[src]]
    assert [g.name for g in fparts] == [f"{src}#{i}" for i in range(len(parts))]
    assert all(g.draw == src and g.draw_G == len(parts) and g.draw_g == i and g.model == "shared" for i, g in enumerate(fparts))


def test_onecycle_matches_torch_and_lr_rule():
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.AdamW([p], lr=1.0)
    sched = torch.optim.lr_scheduler.OneCycleLR(opt, max_lr=0.37, total_steps=50, pct_start=0.2, cycle_momentum=False,
                                                div_factor=1000, final_div_factor=1e4)
    mine = OneCycle(0.37, 50, 0.2, 1000.0, 1e4)
    for step in range(50):
        assert abs(opt.param_groups[0]["lr"] - mine.lr(step)) < 1e-9
        opt.step()
        if step < 49:
            sched.step()
    assert abs(scaled_lr(3e-5, 32, 1, 1, 8) - 3e-5 * (32 * 8 / 3.0) ** 0.5) < 1e-12


def test_config_loader_cli_grammar():
    cfg = conf.load_experiment(["datasets.name_dataset=flair", "datasets.flair.filter_inputs=[aerial,s2]",
                                "datasets.flair.filter_targets=[]", "model.model_size=medium", "opt_pretrain.batch_size=64",
                                "mask.mask_ratio=0.6", "datasets.flair.s2.num_dates=8"])
    assert list(cfg["datasets"].dataset.inputs) == ["aerial", "s2"] and cfg["datasets"].dataset.s2.num_dates == 8
    assert cfg["model"].model_size == "medium" and cfg["opt_pretrain"].batch_size == 64 and cfg["mask"].mask_ratio == 0.6
    with pytest.raises(ValueError):
        conf.load_experiment(["model.nope=1"])
    with pytest.raises(ValueError):
        conf.FLAIRConfig(ref_input="spot")


def test_checkpoint_round_trip_and_reference_style_keys(tmp_path):
    mod = SSLModule(datasets=treesat(), mask=conf.MaskConfig(mask_ratio=0.6), interpolate="nearest", fusion_mode="group",
                    inter_depth=3, model="mae", model_size="tiny", loss="l1_norm")
    path = tmp_path / "pretrain-epoch=0.ckpt"
    ckpt = mod.checkpoint()
    # a reference checkpoint written with use_ema=True additionally carries the EMA copy
    assert "model.heads.treesat_mlc_thresh.linear.weight" in ckpt["state_dict"]      # heads: same keys as the reference
    ckpt["state_dict"]["ema_model.mask_token.aerial"] = torch.zeros(1, 1, 1, 1, 512)
    torch.save(ckpt, path)
    assert all(k.startswith("model.") for k in mod.checkpoint()["state_dict"])
    new = SSLModule.load_from_checkpoint(path, map_location="cpu", strict=False, datasets=treesat())
    assert new.loss_name == "l1_norm" and new.model.mask_ratio["aerial"] == 0.6 and new.model.inter_depth == 3
    assert not new.loaded_missing and sorted(new.loaded_unexpected) == ["ema_model.mask_token.aerial"]
    for (k, a), (_, b) in zip(mod.model.state_dict().items(), new.model.state_dict().items()):
        assert torch.equal(a, b), k
    with pytest.raises(RuntimeError):
        SSLModule.load_from_checkpoint(path, strict=True, datasets=treesat())


def test_transform_flag_draws_follow_reference_order():
    """Host side of the input staging: three ``rng.choice([True, False])`` draws per sample, in the reference's order
    (maestro/dataset/dataset.py:230-249), packed as bit0 / bit1 / bit2."""
    import numpy as np
    from maestro_amd.train.staging import draw_transform_flags
    from oracle import staging as ost
    flags = draw_transform_flags(np.random.default_rng(123), 16)
    rng = np.random.default_rng(123)
    assert flags.tolist() == [ost.draw_flags(rng) for _ in range(16)]
    assert draw_transform_flags(np.random.default_rng(0), 4, use_transform=False).tolist() == [0, 0, 0, 0]
    a = np.arange(24).reshape(1, 2, 3, 4)   # flips only (a transpose needs square rasters)
    assert np.array_equal(ost.transform_rasters({"r": a}, 3)["r"], a[:, :, ::-1, ::-1])


def test_head_state_dict_keys_match_oracle():
    """Probe / finetune heads: the holder tree has the reference's keys and shapes (the oracle's tree was loaded
    strict=True into the real reference in oracle/gen_golden.py:run_sup_case)."""
    import maestro_amd.conf as conf
    from maestro_amd.ssl import mae as pmae
    from oracle import mae as om
    from oracle.gen_golden import build_datasets, sup_case_table
    for name, case in sup_case_table().items():
        ds = build_datasets(case, conf)
        kw = dict(interpolate="nearest", fusion_mode=case["fusion"], inter_depth=case["inter_depth"], model="mae",
                  num_levels=1, type_head=case["type_head"], fac_abs_enc=1.0, fac_date_enc=1.0, **case["model_kw"])
        ours = getattr(pmae, f"mae_{case['size']}")(datasets=ds, mask=conf.MaskConfig(), **kw).state_dict()
        ref = om.build_oracle(ds, conf.MaskConfig(), model_size=case["size"], **kw).state_dict()
        assert set(ours) == set(ref), (name, set(ours) ^ set(ref))
        assert all(ours[k].shape == ref[k].shape for k in ref), name
        assert any(k.startswith("heads.") for k in ref)


def test_adamw_bias_corrections_are_float32_like_the_kernel_host_side():
    """``mh_adamw`` computes 1 - b1^t and sqrt(1 - b2^t) in float32 (powf / sqrtf); the device-scalar variant gets the same
    bits from ``hip.adamw_bias_corrections`` (no GPU or library needed: libm only)."""
    import numpy as np

    from maestro_amd import hip
    for t in (1, 2, 7, 100, 5000):
        bc1, bc2 = hip.adamw_bias_corrections(0.9, 0.99, t)
        assert bc1 == float(np.float32(1.0) - np.float32(0.9) ** np.float32(t)) or abs(bc1 - (1 - 0.9 ** t)) < 1e-6
        assert abs(bc2 - (1 - 0.99 ** t) ** 0.5) < 1e-6        # float32 arithmetic on purpose (0.99 is not exact in float32)
        assert np.float32(bc1) == bc1 and np.float32(bc2) == bc2      # exactly representable: they cross the ABI as float


def test_lr_rule_counts_micro_batches_like_the_reference():
    """model.py:120-128: lr = base_lr * sqrt(batch * accumulate * nodes * devices / 3)."""
    from maestro_amd.train.optim import scaled_lr
    assert scaled_lr(3e-5, 32, 1, 1, 1) == pytest.approx(3e-5 * (32 / 3) ** 0.5)
    assert scaled_lr(3e-5, 32, 4, 1, 8) == pytest.approx(3e-5 * (32 * 4 * 8 / 3) ** 0.5)
    assert scaled_lr(3e-5, 32, 2, 1, 1) == pytest.approx(scaled_lr(3e-5, 64, 1, 1, 1))


def test_param_store_notices_updates_made_outside_the_engine(monkeypatch):
    """``p.data = view`` does not share ``flat``'s version counter, so the staleness key of the bf16
    shadow must include the parameters' own counters -- a torch optimizer step, ``load_state_dict`` and ``p.add_()`` all
    have to trigger a re-cast, FusedAdamW's raw-pointer update (followed by ``mark_synced``) must not."""
    import torch
    from torch import nn

    from maestro_amd import engine as eng_mod

    casts = []
    monkeypatch.setattr(eng_mod.hip, "cast_bf16", lambda src, dst, n: (casts.append(n), dst.copy_(src.to(dst.dtype))))
    lin = nn.Linear(8, 4)
    store = eng_mod.ParamStore(list(lin.named_parameters()), "cpu")
    assert store.refresh_half() and len(casts) == 1 and not store.refresh_half()
    opt = torch.optim.AdamW(lin.parameters(), lr=0.1)
    lin.weight.grad.fill_(1.0)
    opt.step()                                                   # bumps p._version only
    assert store.refresh_half(), "optimizer step went unnoticed"
    assert torch.equal(store.h(lin.weight).float(), lin.weight.detach().bfloat16().float())
    lin.load_state_dict({k: torch.ones_like(v) for k, v in lin.state_dict().items()})
    assert store.refresh_half(), "load_state_dict went unnoticed"
    with torch.no_grad():
        lin.bias.add_(1.0)
    assert store.refresh_half() and not store.refresh_half()
    store.flat.mul_(2.0)                                         # writes through the flat view bump flat._version only
    assert store.refresh_half()
    store.mark_synced()
    assert not store.refresh_half()
    assert store.fresh and store.grad_all.numel() == store.total + eng_mod.ALIGN and store.extra.numel() == eng_mod.ALIGN


def test_reference_written_checkpoint_loads_without_the_reference(golden_dir, tmp_path):
    """``tests/golden/ref_written.ckpt.gz`` was written by the REFERENCE ``SSLModule`` (tiny, ``use_ema=True``;
    ``oracle/gen_golden.py ckpt``) in Lightning's layout -- its pickled hyper-parameters hold a
    ``maestro.conf.mask.MaskConfig`` instance (``maestro/train/model.py:118``).  It must load here, where ``maestro`` is not
    importable, strictly key for key including ``ema_model.*`` (``maestro/run_experiment.py:66-74``)."""
    import gzip
    import importlib.util
    import zipfile

    import maestro_amd.conf as conf
    from maestro_amd.train.model import SSLModule
    from oracle.gen_golden import CKPT_CASE, CKPT_HP, CKPT_MASK, build_datasets, ckpt_value

    assert importlib.util.find_spec("maestro") is None, "this test must run without the reference on the path"
    path = tmp_path / "ref.ckpt"
    path.write_bytes(gzip.decompress((golden_dir / "ref_written.ckpt.gz").read_bytes()))
    with zipfile.ZipFile(path) as z:
        pkl = z.read(next(n for n in z.namelist() if n.endswith("data.pkl")))
    assert b"maestro.conf.mask" in pkl and b"MaskConfig" in pkl       # the reference's own class path is in the pickle
    ds = build_datasets(CKPT_CASE, conf)
    mod = SSLModule.load_from_checkpoint(str(path), map_location="cpu", strict=True, datasets=ds)
    assert mod.loaded_missing == [] and mod.loaded_unexpected == []
    assert isinstance(mod._mask, conf.MaskConfig) and vars(mod._mask) == vars(conf.MaskConfig(**CKPT_MASK))
    assert mod.loss_name == CKPT_HP["loss"] and mod.ema_model is not None and mod.model.inter_depth == CKPT_HP["inter_depth"]
    sd = {k: v for k, v in mod.state_dict().items() if k != "_anchor"}
    assert sum(k.startswith("ema_model.") for k in sd) == sum(k.startswith("model.") for k in sd) > 100
    for k, v in sd.items():
        assert float(v.flatten()[0]) == float(torch.tensor(ckpt_value(k), dtype=v.dtype)), k
        assert float(v.min()) == float(v.max())
    # the default call of run_experiment.py (strict=False) works as well, and so does dropping the EMA copy
    assert SSLModule.load_from_checkpoint(str(path), strict=False, datasets=ds, use_ema=False).loaded_unexpected != []
    # what this repo writes names the reference's class path only (nothing of maestro_amd is pickled)
    out = tmp_path / "ours.ckpt"
    mod.save_checkpoint(out)
    with zipfile.ZipFile(out) as z:
        pkl = z.read(next(n for n in z.namelist() if n.endswith("data.pkl")))
    assert b"maestro.conf.mask" in pkl and b"maestro_amd" not in pkl
    again = SSLModule.load_from_checkpoint(str(out), strict=True, datasets=ds)
    assert all(torch.equal(a, b) for a, b in zip(again.state_dict().values(), mod.state_dict().values()))


def test_module_log_helpers_and_state_dict_surface():
    """``log_metric`` / ``log_step`` (``maestro/train/base.py:153-187``) forward to ``self.log`` with the reference's arguments;
    the autograd anchor of the engine bridge is neither a parameter nor a state-dict entry (Lightning's own checkpoints then
    carry exactly the reference module's keys)."""
    mod = SSLModule(datasets=treesat(), mask=conf.MaskConfig(), interpolate="nearest", fusion_mode="group", inter_depth=0,
                    model="mae", model_size="tiny")
    seen = []
    mod.log = lambda **k: seen.append(k)
    mod.log_metric("pretrain_loss_rec/val", 1.5)
    mod.log_step("loss_rec", 2.0, "pretrain", "train")
    mod.log_step("loss_rec", 2.0, "pretrain", "val")          # only the train stage logs per step
    assert seen == [dict(name="pretrain_loss_rec/val", value=1.5, on_step=False, on_epoch=True, prog_bar=True, logger=True, sync_dist=True),
                    dict(name="pretrain_loss_rec/step_train", value=2.0, on_step=True, on_epoch=False, prog_bar=True, logger=True,
                         sync_dist=True)]
    assert "_anchor" not in mod.state_dict() and all(n != "_anchor" for n, _ in mod.named_parameters())
    assert mod._anchor.requires_grad and mod._anchor.is_leaf


def test_dropped_engines_retire_their_graphs(monkeypatch):
    """An engine that gets garbage-collected must NOT destroy its hipGraphs on the spot (the cyclic collector can run between two
    launches of another engine's step; destroying graphs then corrupted later replays on ROCm 7.2): they are handed to a module
    list and destroyed at the next safe point, after a device synchronisation (``drain_retired_graphs``)."""
    import gc
    import weakref

    import torch

    from maestro_amd import engine

    class FakeEngine:
        def __init__(self):
            self._graphs = {}
            self.me = self                                   # a reference cycle, like the real engines
            weakref.finalize(self, engine._retire_graphs, self._graphs)

    destroyed = []

    class FakeGraph:
        def __del__(self):
            destroyed.append(1)

    monkeypatch.setattr(engine, "_RETIRED_GRAPHS", [])
    synced = []
    monkeypatch.setattr(torch.cuda, "synchronize", lambda *a, **k: synced.append(1))
    e = FakeEngine()
    e._graphs["forward"] = {"key": 1, "graph": FakeGraph(), "spans": []}
    del e
    gc.collect()
    assert not destroyed and len(engine._RETIRED_GRAPHS) == 1, "the graph must outlive its engine until the next safe point"
    engine.drain_retired_graphs()
    assert destroyed == [1] and synced == [1] and not engine._RETIRED_GRAPHS
    engine.drain_retired_graphs()                            # nothing retired: no synchronisation
    assert synced == [1]


def test_engine_adamw_without_an_engine_is_torch_adamw():
    """``EngineAdamW`` (what ``configure_optimizers`` returns) before any engine owns the parameters: torch's own step, torch's
    state layout; a closure is evaluated with gradients enabled (Lightning's automatic optimisation runs the step inside it)."""
    import io

    from maestro_amd.train.optim import EngineAdamW
    torch.manual_seed(0)
    a = [torch.nn.Parameter(torch.randn(5, 3)), torch.nn.Parameter(torch.randn(7))]
    b = [torch.nn.Parameter(p.detach().clone()) for p in a]
    oa = EngineAdamW(a, lambda: None, lr=0.05, betas=(0.9, 0.99), weight_decay=0.01)
    ob = torch.optim.AdamW(b, lr=0.05, betas=(0.9, 0.99), weight_decay=0.01)
    assert isinstance(oa, torch.optim.AdamW)
    for step in range(3):
        g = [torch.randn_like(p, generator=torch.Generator().manual_seed(10 * step + i)) for i, p in enumerate(a)]
        for p, q, gi in zip(a, b, g):
            p.grad, q.grad = gi.clone(), gi.clone()
        seen = []

        def closure():
            seen.append(torch.is_grad_enabled())
            return torch.tensor(1.5)

        assert float(oa.step(closure)) == 1.5 and seen == [True]
        ob.step()
    for p, q in zip(a, b):
        assert torch.equal(p, q)
    sd = oa.state_dict()
    assert set(sd["state"][0]) == {"step", "exp_avg", "exp_avg_sq"} and float(sd["state"][0]["step"]) == 3.0
    assert sd["state"][0]["step"] is not sd["state"][1]["step"]           # never a shared counter in a checkpoint
    buf = io.BytesIO()
    torch.save(sd, buf)
    buf.seek(0)
    ob2 = torch.optim.AdamW(b, lr=0.05, betas=(0.9, 0.99), weight_decay=0.01)
    ob2.load_state_dict(torch.load(buf, weights_only=False))
    oa.load_state_dict(ob.state_dict())
    assert oa._bound is None
    for p, q in zip(a, b):
        p.grad, q.grad = torch.ones_like(p), torch.ones_like(q)
    oa.step()
    ob2.step()
    for p, q in zip(a, b):
        assert torch.allclose(p, q, rtol=0, atol=1e-7)


def test_engine_adamw_falls_back_when_it_does_not_cover_the_engine():
    """Several parameter groups, or gradients that are not the engine's flat-buffer slices: torch's implementation runs."""
    from types import SimpleNamespace

    from maestro_amd.train.optim import EngineAdamW
    w = [torch.nn.Parameter(torch.ones(4)), torch.nn.Parameter(torch.ones(2))]
    flat_grad = torch.zeros(8)
    store = SimpleNamespace(params=w, offset={id(w[0]): 0, id(w[1]): 4}, total=8, fresh=False,
                            g=lambda p: flat_grad[0:4] if p is w[0] else flat_grad[4:6])
    eng = SimpleNamespace(store=store)
    two_groups = EngineAdamW([{"params": [w[0]]}, {"params": [w[1]], "lr": 0.5}], lambda: eng, lr=0.1)
    assert not two_groups._eligible(eng)
    one = EngineAdamW(w, lambda: eng, lr=0.1)
    w[0].grad, w[1].grad = torch.ones(4), torch.ones(2)                   # ordinary tensors, not the flat buffer's slices
    assert not one._eligible(eng)
    w[0].grad, w[1].grad = flat_grad[0:4], flat_grad[4:6]
    assert one._eligible(eng)
    store.fresh = True                                                    # no backward has written the buffer yet
    assert not one._eligible(eng)
    one.step()                                                            # (torch's path: must simply work)
    assert one._fused is None
