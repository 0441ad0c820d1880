"""Two data-parallel ranks over RCCL (backend "nccl"), one GPU each: skipped unless at least two GPUs are visible, so the
driver's multi-GPU node exercises it while a one-GPU box does not.  Checks what the reference gets from Lightning's DDP wrap
(``maestro/conf/trainer.py:9-14``): rank 0's weights everywhere before the first step, identical parameters after several
steps with per-rank tiles and masks, and the cross-rank mean of the step loss (``maestro/train/logger.py:268-276``) carried in
the first gradient bucket.  Also runs the opt-in bf16 bucket mode and the supervised loop's segmented exchange."""

import os

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

def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    import maestro_amd.conf as conf
    from maestro_amd.ssl.mae import mae_tiny
    from maestro_amd.train.trainer import PretrainLoop, synthetic_batch
    ds = conf.DatasetsConfig(name_dataset="treesatai_ts", treesatai_ts=conf.TreeSatAITSConfig(
        filter_targets=[], aerial=conf.InputRasterConfig(image_size=60, patch_size=conf.PatchSizeConfig(mae=20), bands=4,
                                                         norm_bands=[1, 3], norm_fac=255.0)))
    res = {}
    for mode, bucket_dtype in (("f32", None), ("bf16", torch.bfloat16), ("rs_ag", None)):
        torch.manual_seed(1234 + rank)            # DIFFERENT initial weights per rank: the loop must broadcast rank 0's
        model = mae_tiny(datasets=ds, mask=conf.MaskConfig(), interpolate="nearest", fusion_mode="group", inter_depth=1,
                         model="mae", num_levels=1, depth=2)
        loop = PretrainLoop(model, 2, dev, total_steps=10, world_size=world, bucket_mb=1, bucket_dtype=bucket_dtype,
                            exchange_mode="rs_ag" if mode == "rs_ag" else "all_reduce")
        flat0 = loop.engine.store.flat.clone()
        g0 = [torch.zeros_like(flat0) for _ in range(world)]
        dist.all_gather(g0, flat0)
        batch = synthetic_batch(ds.dataset, 2, dev, seed=rank)
        torch.manual_seed(100 + rank)
        losses = []
        for _ in range(4):
            loss = loop.step(batch)
            both = [torch.zeros(1, device=dev) for _ in range(world)]
            dist.all_gather(both, loss.detach().reshape(1).clone())
            losses.append((float(loop.loss_mean), float(sum(both)) / world))
        loop.flush()
        torch.cuda.synchronize()
        flat = loop.engine.store.flat
        allp = [torch.zeros_like(flat) for _ in range(world)]
        dist.all_gather(allp, flat)
        res[mode] = (all(torch.equal(g0[0], g) for g in g0), all(torch.equal(allp[0], p) for p in allp),
                     bool(torch.isfinite(flat).all()), losses, len(loop.sync.launched))
    out.put((rank, res))
    dist.destroy_process_group()


def test_two_ranks_over_rccl():
    if not torch.cuda.is_available() or torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (one rank per GPU over RCCL)")
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    res = []
    for _ in procs:
        for _ in range(300):
            try:
                res.append(out.get(timeout=1))
                break
            except Exception:  # noqa: BLE001
                assert all(p.exitcode in (None, 0) for p in procs), "a rank crashed"
        else:
            raise AssertionError("ranks did not report within 300 s")
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, per_mode in res:
        for mode, (same_start, same_end, finite, losses, buckets) in per_mode.items():
            assert same_start, f"rank {rank} ({mode}): parameters were not broadcast from rank 0"
            assert same_end and finite, f"rank {rank} ({mode}): parameters diverged between ranks"
            assert buckets >= 2
            for mean_slot, mean_ref in losses:
                assert abs(mean_slot - mean_ref) <= 1e-6 * abs(mean_ref), (mode, mean_slot, mean_ref)


# ------------------------------------------------------------------------------------------------------------------
# RCCL on the ONE-GPU box: the one-rank process group runs the whole data-parallel launch plan -- bucketed
# collectives issued from the gradient hook between the backward segments, hipGraph capture next to RCCL's watchdog thread, the
# two-part AdamW under the last bucket -- in a CHILD process started by torch.distributed.run (never an exec of this process,
# which has initialised the GPU).  With one rank every collective is the identity, so the loss trajectory must equal the plain
# run's; both exchange modes.
def _bench_child(extra, launcher):
    import json
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MAESTRO_WARM_PASSES="0")
    cmd = [sys.executable]
    if launcher:
        port = _free_port()
        cmd += ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                "--master-port", str(port)]
    cmd += [str(root / "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "3", "--batch", "8", "--cpu-seconds", "0",
            "--no-kernel-timing", "--log-losses", *extra]
    r = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (cmd, r.stdout[-1500:], r.stderr[-3000:])
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    return json.loads(line)


def test_one_rank_rccl_rehearsal_of_the_exchange_plan():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    plain = _bench_child([], 0)
    assert plain["config"]["exchange"] == "none" and len(plain["losses"]) == 4 and "comm" not in plain
    for k, mode in enumerate(("all_reduce", "rs_ag"), start=1):
        reh = _bench_child(["--rehearse-exchange", "--exchange-mode", mode], k)
        assert reh["config"]["exchange"] == mode and reh["n_gpus"] == 1
        # the line's self-diagnosis of the exchange (round 6): what the process group really is, what the plan moved, how long the
        # step's main stream waited for collectives
        comm = reh["comm"]
        assert comm["backend"] == "nccl" and comm["world_size"] == 1 and comm["exchange"] == mode and comm["rccl_version"]
        assert comm["buckets_per_step"] >= 1 and comm["mbytes_per_step"] > 0.9 * comm["grad_mbytes"]
        assert comm["exposed_ms_per_step"] is not None and 0.0 <= comm["exposed_ms_per_step"] < 50.0
        for a, b in zip(plain["losses"], reh["losses"]):
            assert a == a and abs(a - b) <= 1e-5 * abs(a), (mode, plain["losses"], reh["losses"])
        assert reh["losses"][-1] < reh["losses"][0]           # ... and it trains
