"""Two data-parallel ranks on ONE GPU (gloo rehearsal of the RCCL path): parameters stay identical across ranks and the
applied gradient is the mean over ranks."""

import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


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
    port = 29700 + os.getpid() % 1000
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
