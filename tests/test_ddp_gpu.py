"""Two data-parallel ranks on ONE GPU (gloo rehearsal of the RCCL path): parameters stay identical across ranks and the
applied gradient is the mean over ranks."""

import os
import queue

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu



def _free_port() -> int:
    """A TCP port nobody listens on right now (fixed pid-derived ports collided between tests of one process: EADDRINUSE)."""
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]

def _build(dev):
    import maestro_amd.conf as conf
    from maestro_amd.ssl.mae import mae_tiny
    ds = conf.DatasetsConfig(name_dataset="treesatai_ts", treesatai_ts=conf.TreeSatAITSConfig(
        filter_targets=[], aerial=conf.InputRasterConfig(image_size=60, patch_size=conf.PatchSizeConfig(mae=20), bands=4,
                                                         norm_bands=[1, 3], norm_fac=255.0)))
    torch.manual_seed(0)   # identical initial weights on every rank
    model = mae_tiny(datasets=ds, mask=conf.MaskConfig(), interpolate="nearest", fusion_mode="group", inter_depth=1,
                     model="mae", num_levels=1, depth=2)
    return ds, model


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from maestro_amd.train.trainer import PretrainLoop, synthetic_batch
    dev = torch.device("cuda:0")
    ds, model = _build(dev)
    loop = PretrainLoop(model, 2, dev, total_steps=10, world_size=world, bucket_mb=1)
    batch = synthetic_batch(ds.dataset, 2, dev, seed=rank)   # different tiles per rank
    torch.manual_seed(100 + rank)                           # different masks per rank
    eng = loop.engine
    # step 1 by hand to capture the synchronised gradient
    eng.forward(batch)
    eng.zero_grad()
    loop.sync.begin()
    eng.backward()
    local = None
    scale = loop.sync.finish()
    summed = eng.store.grad.clone()
    gathered = [torch.zeros_like(summed.cpu()) for _ in range(world)]
    dist.all_gather(gathered, summed.cpu())
    same_grad = all(torch.equal(gathered[0], g) for g in gathered)
    for _ in range(3):
        loop.step(batch)
    loop.flush()                                            # the last update is queued for the next forward
    torch.cuda.synchronize()
    flat = eng.store.flat.cpu()
    allp = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(allp, flat)
    out.put((rank, same_grad, scale, all(torch.equal(allp[0], p) for p in allp), bool(torch.isfinite(flat).all()),
             len(loop.sync.launched)))
    dist.destroy_process_group()


def test_two_ranks_stay_in_sync():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    res = []
    for _ in procs:   # fail fast if a rank dies instead of waiting on the queue
        for _ in range(240):
            try:
                res.append(out.get(timeout=1))
                break
            except Exception:  # noqa: BLE001
                assert all(p.exitcode in (None, 0) for p in procs), "a rank crashed"
        else:
            raise AssertionError("ranks did not report within 240 s")
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, same_grad, scale, same_params, finite, buckets in res:
        assert same_grad, f"rank {rank}: all-reduced gradients differ between ranks"
        assert scale == 0.5 and same_params and finite, (rank, scale, same_params, finite)
        assert buckets >= 2, "gradient exchange should be split into several overlapped buckets"


# ------------------------------------------------------------------------------------------------------------------
# The FULL launch plan (PretrainLoop.step: backward segments with the gradient hook, buckets overlapped with the backward,
# GradSync.finish_split + the two-part AdamW under which the head bucket is reduced, hipGraph replay from the second step on)
# for 5 optimizer steps on two gloo ranks sharing the card, against ONE process that runs the concatenated batch with the
# concatenated mask draws: what Lightning's DDP gives the reference (maestro/conf/trainer.py:9-14) -- every rank holds the
# same parameters, and they are the parameters of the large-batch run.
def _build_single_groups():
    import maestro_amd.conf as conf
    from maestro_amd.ssl.mae import mae_tiny
    # single-modality groups only: then every sample masks the same number of tokens per modality and the mean of the ranks'
    # losses IS the loss of the concatenated batch (model.py:241-243 averages over the masked pixels of the whole batch)
    ds = conf.DatasetsConfig(name_dataset="treesatai_ts", treesatai_ts=conf.TreeSatAITSConfig(
        filter_inputs=["aerial", "s2"], filter_targets=[],
        aerial=conf.InputRasterConfig(image_size=60, patch_size=conf.PatchSizeConfig(mae=20), bands=4, norm_bands=[1, 3], norm_fac=255.0)))
    torch.manual_seed(0)
    model = mae_tiny(datasets=ds, mask=conf.MaskConfig(), interpolate="nearest", fusion_mode="group", inter_depth=1,
                     model="mae", num_levels=1, depth=2)
    return ds, model


def _record_draws(eng, log):
    inner = eng.draw_masks

    def draw():
        noise, struct = inner()
        log.append(({g: t.clone() for g, t in noise.items()}, {g: t.clone() for g, t in struct.items()}))
        return noise, struct

    eng.draw_masks = draw


def _plan_worker(rank, world, port, out, bucket_dtype, steps):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from maestro_amd.train.trainer import PretrainLoop, synthetic_batch
    dev = torch.device("cuda:0")
    ds, model = _build_single_groups()
    if rank == 1:      # the ranks start from DIFFERENT weights: PretrainLoop's broadcast must bring rank 0's everywhere
        with torch.no_grad():
            for p in model.parameters():
                p.add_(0.01)
    # (3 MB buckets: the tail of this small model's buffer goes out under the backward, the head stays for finish_split)
    loop = PretrainLoop(model, 2, dev, base_lr=3e-3, total_steps=20, world_size=world, bucket_mb=3,
                        bucket_dtype=torch.bfloat16 if bucket_dtype == "bf16" else None)
    eng, draws, losses, splits = loop.engine, [], [], []
    _record_draws(eng, draws)
    inner_split = loop.sync.finish_split

    def finish_split():
        r = inner_split()
        splits.append(r[1])
        return r

    loop.sync.finish_split = finish_split
    batch = synthetic_batch(ds.dataset, 2, dev, seed=rank)
    torch.manual_seed(100 + rank)
    for _ in range(steps):
        losses.append(float(loop.step(batch).item()))
    torch.cuda.synchronize()
    flat = eng.store.flat.cpu()
    allp = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(allp, flat)
    # (numpy arrays travel through the queue by value; torch tensors would be passed as shared-memory handles that die with this process)
    draws_np = [({g: t.numpy() for g, t in n.items()}, {g: t.numpy() for g, t in s.items()}) for n, s in draws]
    splits = [(s, eng.store.total, sorted(loop.sync.launched)) if s == 0 else s for s in splits]   # (diagnostics when the split is empty)
    out.put((rank, all(torch.equal(allp[0], p) for p in allp), flat.numpy() if rank == 0 else None, draws_np, losses, splits,
             len(eng._graphs), {k: v.cpu().numpy() for k, v in batch.items()}))
    dist.destroy_process_group()


def _rs_ag_worker(rank, world, port, out, steps):
    """``exchange_mode="rs_ag"``: every bucket reduce-scattered, each rank updates its half of every bucket, the fp32 masters
    all-gathered -- and the SAME run under the all-reduce plan (second model, same seeds) for comparison."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from maestro_amd.train.trainer import PretrainLoop, synthetic_batch
    dev = torch.device("cuda:0")
    res = {}
    for mode in ("all_reduce", "rs_ag"):
        ds, model = _build_single_groups()
        loop = PretrainLoop(model, 2, dev, total_steps=50, world_size=world, bucket_mb=1, base_lr=2e-3, exchange_mode=mode)
        batch = synthetic_batch(ds.dataset, 2, dev, seed=rank)
        torch.manual_seed(100 + rank)
        losses = [float(loop.step(batch).item()) for _ in range(steps)]
        torch.cuda.synchronize()
        flat = loop.engine.store.flat.cpu()
        allp = [torch.zeros_like(flat) for _ in range(world)]
        dist.all_gather(allp, flat)
        half_ok = torch.equal(loop.engine.store.half.float().cpu(), flat.bfloat16().float())     # shadows follow the gathered masters
        owned = loop.sync.owned()
        res[mode] = dict(flat=flat, same=all(torch.equal(allp[0], p) for p in allp), losses=losses, half_ok=half_ok,
                         sharded=sum(hi - lo for lo, hi, a, b in owned if (a, b) != (lo, hi)),
                         whole=sum(hi - lo for lo, hi, a, b in owned if (a, b) == (lo, hi)), buckets=len(loop.sync.launched),
                         loss_mean=float(loop.loss_mean.item()))
        if mode == "rs_ag":      # the moments: owner-only until gathered; afterwards equal on both ranks
            from maestro_amd import hip
            try:                 # an ungathered sharded state must not be saved as if it were whole
                loop.opt.state_dict()
                refused = False
            except hip.HipExtensionError:
                refused = True
            try:                 # round 6: the loop's state_dict is NOT a collective any more: ungathered, it refuses as well
                loop.state_dict()
                refused = False
            except hip.HipExtensionError:
                pass
            loop.gather_state()             # the explicit collective (every rank) ...
            sd = loop.state_dict()          # ... then the whole Adam state (copies) + the fp32 masters, on whichever rank asks
            m = loop.opt.m.cpu()
            allm = [torch.zeros_like(m) for _ in range(world)]
            dist.all_gather(allm, m)
            res[mode]["moments_same"] = all(torch.equal(allm[0], x) for x in allm) and float(m.abs().sum()) > 0
            # save -> load into a FRESH loop (same weights) -> both continue: an accumulated step (no hook: the static bucket cuts
            # are the same) and a plain one; parameters must stay equal
            ds2, model2 = _build_single_groups()
            loop2 = PretrainLoop(model2, 2, dev, total_steps=50, world_size=world, bucket_mb=1, base_lr=2e-3, exchange_mode=mode)
            loop2.engine.store.flat.copy_(loop.engine.store.flat)
            from maestro_amd.train.ddp import resync_engine
            resync_engine(loop2.engine)
            loop2.load_state_dict({k: (v if k != "optimizer" else {kk: (vv.clone() if isinstance(vv, torch.Tensor) else vv) for kk, vv in v.items()})
                                   for k, v in sd.items()})
            cont = []
            for lp in (loop, loop2):
                torch.manual_seed(555 + rank)
                lp.step([batch, batch])
                lp.step(batch)
                torch.cuda.synchronize()
                cont.append(lp.engine.store.flat.cpu())
            drift = ((cont[0] - cont[1]).double().norm() / cont[0].double().norm()).item()
            res[mode].update(refused=refused, resumed_rel=drift, state_keys=sorted(sd["optimizer"]))
            del loop2
        del loop
    rel = ((res["rs_ag"]["flat"] - res["all_reduce"]["flat"]).double().norm() / res["all_reduce"]["flat"].double().norm()).item()
    moved = ((res["all_reduce"]["flat"] - res["rs_ag"]["flat"]).abs().max().item())
    out.put((rank, {k: {kk: vv for kk, vv in v.items() if kk != "flat"} for k, v in res.items()}, rel, moved))
    dist.destroy_process_group()


def test_reduce_scatter_all_gather_plan_equals_the_all_reduce_plan():
    """SURVEY §8(e) / VERDICT r03 item 8b: ``GradSync(mode="rs_ag")`` + ``FusedAdamW.step_sharded`` on two gloo ranks sharing the
    card.  Parameters identical on both ranks after 4 steps; equal to the all-reduce plan's (same summands, the collective's
    order of addition aside); most buckets sharded; bf16 shadows rebuilt from the gathered masters; the loss mean rides along."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rs_ag_worker, args=(r, 2, port, out, 4)) for r in range(2)]
    for p in procs:
        p.start()
    try:
        res = [out.get(timeout=600) for _ in procs]
    except queue.Empty:
        for p in procs:
            p.kill()
        pytest.fail("rs_ag workers did not finish")
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, r, rel, moved in res:
        assert r["rs_ag"]["same"] and r["all_reduce"]["same"], (rank, "parameters differ between the ranks")
        assert r["rs_ag"]["half_ok"] and r["rs_ag"]["moments_same"], rank
        assert r["rs_ag"]["refused"], "FusedAdamW.state_dict() handed out sharded, ungathered moments"
        assert r["rs_ag"]["resumed_rel"] < 1e-6, (rank, r["rs_ag"]["resumed_rel"])     # (atomically accumulated bias gradients: not bit-equal)
        assert {"exchange_mode", "world", "span", "t"} <= set(r["rs_ag"]["state_keys"])
        # all of the payload but the short rest of every bucket (< 64 x world elements each) went through the reduce-scatter
        assert r["rs_ag"]["sharded"] > 0 and r["rs_ag"]["whole"] < 128 * r["rs_ag"]["buckets"], (rank, r["rs_ag"])
        assert r["all_reduce"]["sharded"] == 0
        assert rel < 1e-5, (rank, rel, moved)              # same sums of two addends: in practice bit-equal
        assert abs(r["rs_ag"]["loss_mean"] - r["all_reduce"]["loss_mean"]) < 1e-5 * abs(r["all_reduce"]["loss_mean"])
        assert r["rs_ag"]["losses"][-1] < r["rs_ag"]["losses"][0]


@pytest.mark.parametrize("bucket_dtype", ["f32", "bf16"])
def test_full_plan_two_ranks_equal_the_concatenated_batch(bucket_dtype):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    steps = 5
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_plan_worker, args=(r, 2, port, out, bucket_dtype, steps)) for r in range(2)]
    for p in procs:
        p.start()
    res = {}
    for _ in procs:
        for _ in range(120):
            try:
                item = out.get(timeout=1)
                res[item[0]] = item
                break
            except queue.Empty:
                assert all(p.exitcode in (None, 0) for p in procs), "a rank crashed"
        else:
            raise AssertionError("ranks did not report within 120 s")
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank in (0, 1):
        _, same_params, _, draws, losses, splits, n_graphs, _ = res[rank]
        assert same_params, f"rank {rank}: parameters differ between the ranks after {steps} steps of the full plan"
        assert len(draws) == steps and all(l == l for l in losses)
        assert all(isinstance(s, int) and s > 0 for s in splits), f"finish_split must leave the head bucket to the two-part AdamW (splits {splits})"
        assert n_graphs >= 2, "the backward segments should have been captured (hipGraph replay from step 2 on)"
    if bucket_dtype == "bf16":
        return          # bf16 buckets round every gradient before the sum: rank consistency is the claim, not large-batch equality
    # ---- the same 5 steps in ONE process on the concatenated batch with the concatenated draws
    from maestro_amd.train.trainer import PretrainLoop
    dev = torch.device("cuda:0")
    ds, model = _build_single_groups()
    loop = PretrainLoop(model, 4, dev, base_lr=3e-3, total_steps=20, world_size=1)
    eng = loop.engine
    init = eng.store.flat.cpu().clone()                       # = rank 0's seeded initial weights, in the engine's flat order
    log = iter(zip(res[0][3], res[1][3]))

    def draw():
        (n0, s0), (n1, s1) = next(log)
        cat = lambda a, b: torch.cat([torch.from_numpy(a), torch.from_numpy(b)])  # noqa: E731
        return {g: cat(n0[g], n1[g]) for g in n0}, {g: cat(s0[g], s1[g]) for g in s0}

    eng.draw_masks = draw
    batch = {k: torch.cat([torch.from_numpy(res[0][7][k]), torch.from_numpy(res[1][7][k])]).to(dev) for k in res[0][7]}
    single_losses = [float(loop.step(batch).item()) for _ in range(steps)]
    torch.cuda.synchronize()
    flat_single, flat_dp = eng.store.flat.cpu(), torch.from_numpy(res[0][2])
    # the update itself (not the parameters, which are dominated by their initial values) must agree
    upd_single, upd_dp = flat_single - init, flat_dp - init
    rel = ((upd_single - upd_dp).double().norm() / upd_single.double().norm()).item()
    assert upd_single.abs().max().item() > 0 and rel < 2e-2, f"update of the 2-rank run differs from the large-batch run: rel L2 {rel:.3e}"
    for a, b0, b1 in zip(single_losses, res[0][4], res[1][4]):
        assert abs(a - 0.5 * (b0 + b1)) < 2e-3 * abs(a), (a, b0, b1)     # mean of the ranks' losses = the large batch's loss


# ------------------------------------------------------------------------------------------------------------------
# The same exchange on the LIGHTNING surface (SSLModule.training_step -> loss.backward() -> torch optimizer, with
# EngineDDPCallback's hooks called in Lightning's order): overlapped buckets inside the backward (default) against the single
# exchange after it, with two accumulated micro-batches in the last step (d loss = 1/2 from the trainer).
def _lightning_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from types import SimpleNamespace

    import maestro_amd.conf as conf
    from maestro_amd.train.ddp import EngineDDPCallback
    from maestro_amd.train.model import SSLModule
    from maestro_amd.train.trainer import synthetic_batch
    dev = torch.device("cuda:0")
    ds = conf.DatasetsConfig(name_dataset="treesatai_ts", treesatai_ts=conf.TreeSatAITSConfig(
        filter_inputs=["aerial", "s2"], filter_targets=[],
        aerial=conf.InputRasterConfig(image_size=60, patch_size=conf.PatchSizeConfig(mae=20), bands=4, norm_bands=[1, 3], norm_fac=255.0)))
    res = {}
    for overlap in (False, True):
        torch.manual_seed(rank)        # DIFFERENT initial weights per rank: on_fit_start must bring rank 0's everywhere
        mod = SSLModule(datasets=ds, mask=conf.MaskConfig(), interpolate="nearest", fusion_mode="group", inter_depth=1,
                        model="mae", model_size="tiny", loss="l2_norm", use_ema=False)
        mod.trainer = SimpleNamespace(ssl_phase="pretrain", train_dataloader=SimpleNamespace(batch_size=2),
                                      accumulate_grad_batches=1, num_nodes=1, num_devices=world, base_lr=3e-3, wd=0.01, b1=0.9,
                                      b2=0.99, final_factor=1e7, estimated_stepping_batches=20, max_epochs=5)
        cb = EngineDDPCallback(bucket_mb=1, overlap=overlap)
        cb.on_fit_start(mod.trainer, mod)
        opt = mod.configure_optimizers()["optimizer"]
        batches = [synthetic_batch(ds.dataset, 2, dev, seed=10 * rank + i) for i in range(2)]
        torch.manual_seed(100 + rank)  # different masks per rank, the same in both modes
        grads, n_buckets = [], []
        for step, micro in enumerate(([0], [1], [0, 1])):
            opt.zero_grad(set_to_none=True)
            for i in micro:
                cb.on_train_batch_start(mod.trainer, mod, batches[i], step)
                loss = mod.training_step(batches[i], step)["loss"] / len(micro)
                cb.on_before_backward(mod.trainer, mod, loss)
                loss.backward()
                cb.on_after_backward(mod.trainer, mod)
                n_buckets.append(len(cb._sync.launched))
            torch.cuda.synchronize()
            grads.append(mod.model._engine.store.grad.cpu().numpy().copy())
            opt.step()
        torch.cuda.synchronize()
        eng = mod.model._engine
        flat = torch.cat([p.detach().reshape(-1) for p in eng.store.params]).cpu()
        allp = [torch.zeros_like(flat) for _ in range(world)]
        dist.all_gather(allp, flat)
        res[overlap] = (all(torch.equal(allp[0], p) for p in allp), flat.numpy(), grads, n_buckets,
                        sorted(k for k in eng._graphs if k.startswith("bwd")))
    out.put((rank, res))
    dist.destroy_process_group()


def test_lightning_surface_overlapped_exchange_two_ranks():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import numpy as np
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_lightning_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    res = {}
    for _ in procs:
        for _ in range(180):
            try:
                rank, item = out.get(timeout=1)
                res[rank] = item
                break
            except queue.Empty:
                assert all(p.exitcode in (None, 0) for p in procs), "a rank crashed"
        else:
            raise AssertionError("ranks did not report within 180 s")
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank in (0, 1):
        for overlap in (False, True):
            same, flat, grads, n_buckets, graphs = res[rank][overlap]
            assert same, f"rank {rank}, overlap={overlap}: parameters differ between the ranks"
            assert np.isfinite(flat).all() and all(np.abs(g).max() > 0 for g in grads)
        plain, ovl = res[rank][False], res[rank][True]
        assert all(n >= 2 for n in ovl[3]), f"overlap mode should launch several buckets per backward: {ovl[3]}"
        assert any(k.endswith(":h") for k in ovl[4]) and not any(k.endswith(":h") for k in plain[4]), (ovl[4], plain[4])
        for step, (g0, g1) in enumerate(zip(plain[2], ovl[2])):   # (same masks, same weights up to the plans' summation order)
            rel = np.linalg.norm(g0 - g1) / np.linalg.norm(g0)
            assert rel < 1e-3, f"rank {rank} step {step}: overlapped exchange changes the averaged gradient (rel {rel:.2e})"
        upd = np.linalg.norm(plain[1] - ovl[1]) / np.linalg.norm(plain[1])
        assert upd < 1e-3, upd
    for overlap in (False, True):
        assert np.array_equal(res[0][overlap][1], res[1][overlap][1])
