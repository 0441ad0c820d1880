#!/usr/bin/env python
"""MAE-pretrain tiles/sec on synthetic FLAIR-HUB-shaped batches (BASELINE.json metric), 1..8 MI355X.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One step = host mask draws + forward + masked loss + backward + (N>1: RCCL gradient all-reduce) + fused AdamW on
a resident synthetic batch of B=32 tiles per GPU (weak scaling).  Prints ONE JSON line on rank 0.
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import maestro_amd.conf as conf  # noqa: E402

# train GFLOP per tile (GEMM + attention matmuls, 3x forward) per workload
WORKLOADS = {
    "c2": dict(desc="ViT-B MAE, FLAIR aerial RGB+NIR 512x512x4 monotemporal", size="medium", gflop_tile=243.0,
               ds=lambda: conf.DatasetsConfig(name_dataset="flair", flair=conf.FLAIRConfig(filter_inputs=["aerial"], filter_targets=[]))),
    "c3": dict(desc="ViT-B MAE, FLAIR-HUB-shaped aerial 512x512x4 + Sentinel-2 16x10x10x10 time series", size="medium",
               gflop_tile=330.0,
               ds=lambda: conf.DatasetsConfig(name_dataset="flair", flair=conf.FLAIRConfig(filter_inputs=["aerial", "s2"], filter_targets=[]))),
    "c3p": dict(desc="ViT-B MAE, full FLAIR-HUB (aerial, dem, s2, s1_asc, s1_des)", size="medium", gflop_tile=431.9,
                ds=lambda: conf.DatasetsConfig(name_dataset="flair", flair=conf.FLAIRConfig(filter_targets=[]))),
    "c4": dict(desc="ViT-L MAE, TreeSatAI-TS (aerial, s2, s1_asc, s1_des)", size="large", gflop_tile=262.1,
               ds=lambda: conf.DatasetsConfig(name_dataset="treesatai_ts", treesatai_ts=conf.TreeSatAITSConfig(filter_targets=[]))),
    "c5": dict(desc="ViT-B MAE, S2-NAIP-urban (aerial, spot, s2, s1), bf16 path", size="medium", gflop_tile=301.3,
               ds=lambda: conf.DatasetsConfig(name_dataset="s2_naip", s2_naip=conf.S2NAIPConfig())),
}
# probe / finetune workloads (`--phase probe|finetune`): same inputs, with the dataset's target
SUP_WORKLOADS = {
    "c3": dict(desc="ViT-B, FLAIR-HUB-shaped aerial + Sentinel-2, semantic segmentation (cosia, 15 classes, 512x512)", size="medium",
               ds=lambda: conf.DatasetsConfig(name_dataset="flair", flair=conf.FLAIRConfig(filter_inputs=["aerial", "s2"], filter_targets=["cosia"]))),
    "c4": dict(desc="ViT-L, TreeSatAI-TS (aerial, s2, s1_asc, s1_des), multilabel classification (15 labels)", size="large",
               ds=lambda: conf.DatasetsConfig(name_dataset="treesatai_ts", treesatai_ts=conf.TreeSatAITSConfig())),
}
MFMA_PEAK_TFLOPS = 2500.0  # dense bf16, /opt/skills/guides/MI355X_MICROARCH.md
FP8_PEAK_TFLOPS = 5000.0   # dense fp8 (block-scaled MFMA 16x16x128), same guide
HBM_PEAK_TBS = 8.0          # HBM3E, same guide


def sup_gflop_per_tile(model, phase: str) -> float:
    """Algorithmic GFLOP per tile of a probe / finetune step (GEMM + attention matmuls, SURVEY §8d formula on the FULL
    sequences; finetune = 3 x forward, probe = forward + 2 x heads)."""
    t = next(iter(model.encoder.values()))
    E, M, I = model.embed_dim, model.mlp_dim, t.heads * t.dim_head  # noqa: N806, E741

    def stack(n_tok, depth):
        return depth * (6 * n_tok * E * I + 4 * n_tok * n_tok * I + 2 * n_tok * I * E + 4 * n_tok * E * M)

    groups = list(model.group_specs.values())
    enc = sum(stack(g.L, model.depth - model.inter_depth) for g in groups)
    enc += stack(sum(g.L for g in groups), model.inter_depth) if model.inter_depth else 0
    enc += sum(2 * s.n_tok * s.K * E for s in model.mod_specs.values())
    heads = 0.0
    ds = model.dataset
    for t, c in ds.targets.items():
        head = model.heads[t]
        if c.type_target == "segment":
            G = model.out_grid_size[ds.ref_input]  # noqa: N806
            rows, n = sum(s.D for s in model.mod_specs.values()) * G * G, G * G
            heads += 2 * n * E * head.patch_size ** 2 * c.num_classes
        else:
            rows, n = sum(g.L for g in groups), 1
            heads += 2 * E * c.num_classes
        if hasattr(head, "reduce"):
            heads += 2 * rows * E * 2 * E + 4 * rows * E
    return ((3 * (enc + heads)) if phase == "finetune" else (enc + 3 * heads)) / 1e9


def synthetic_targets(dataset, B: int, device, seed: int = 0) -> dict:  # noqa: N803
    """Deterministic targets of the wire format: rasters int64 [B, 1, 1, H, W] with ~10 % missing_val, multilabel f32 [B, C]."""
    out = {}
    for i, (t, c) in enumerate(dataset.targets.items()):
        g = torch.Generator().manual_seed(4321 + i + 1000 * seed)
        if c.type_target == "segment":
            H = round(dataset.crop_meters / c.resolution_meters)  # noqa: N806
            y = torch.randint(0, c.num_classes, (B, 1, 1, H, H), generator=g)
            y[torch.rand(B, 1, 1, H, H, generator=g) < 0.1] = c.missing_val
        elif c.type_target == "multilabel_classif":
            y = (torch.rand(B, c.num_classes, generator=g) < 0.3).float()
        else:
            y = torch.randint(0, c.num_classes, (B,), generator=g)
        out[t] = y.to(device)
    return out


def build_model(workload: str, phase: str = "pretrain"):
    from maestro_amd.ssl import mae as pmae

    w = (WORKLOADS if phase == "pretrain" else SUP_WORKLOADS)[workload]
    ds = w["ds"]()
    model = getattr(pmae, f"mae_{w['size']}")(datasets=ds, mask=conf.MaskConfig(), interpolate="nearest",
                                              fusion_mode="group", inter_depth=3, model="mae", num_levels=1)
    return ds, model


def host_cores() -> int:
    """Threads for the CPU leg (every host core this process may run on, count stated in the line): the
    affinity mask, limited by the cgroup CPU quota when the box hands this job a share of a larger host (a 256-thread
    pool on a 16-core share only thrashes: the round-2 box reported 256 schedulable CPUs and the leg timed out)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                quota, period = txt[0], float(txt[1])
            else:
                quota, period = txt[0], float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota not in ("max", "-1"):
                n = min(n, max(1, int(float(quota) / period + 0.5)))
        except (OSError, ValueError, IndexError):
            pass
    return max(1, n)


def pick_cpu_threads(limit: int) -> tuple[int, dict]:
    """Thread count for the CPU leg: the candidate (16, 32, 64, ..., all schedulable CPUs) with the highest measured fp32
    GEMM rate on THIS host -- a box that exposes 256 CPUs but schedules the job on a 16-core share runs 15x slower with 256
    threads than with 16 (the round-2 box timed the leg out), so "all cores" has to mean the cores that actually run."""
    forced = os.environ.get("MAESTRO_CPU_THREADS")
    if forced:
        return max(1, min(limit, int(forced))), {}
    cands = sorted({c for c in (8, 16, 32, 64, 128) if c < limit} | {limit})
    a, b = torch.randn(1536, 1536), torch.randn(1536, 1536)
    rates = {}
    for c in cands:
        torch.set_num_threads(c)
        a @ b
        t0 = time.time()
        for _ in range(4):
            a @ b
        rates[c] = round(4 * 2 * 1536 ** 3 / (time.time() - t0) / 1e9, 1)
    best = max(rates, key=rates.get)
    return best, rates


def cpu_model() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline_worker(workload: str, seconds: float, phase: str = "pretrain") -> dict:
    """The oracle (CPU restatement = kind "port") timed on this node's host cores on a bounded sample."""
    from maestro_amd.train.trainer import synthetic_batch
    from oracle import heads as oh
    from oracle import mae as om

    w = (WORKLOADS if phase == "pretrain" else SUP_WORKLOADS)[workload]
    ds = w["ds"]()
    torch.set_float32_matmul_precision("highest")
    cores, rates = pick_cpu_threads(host_cores())
    torch.set_num_threads(cores)
    torch.manual_seed(42)
    model = om.build_oracle(ds, conf.MaskConfig(), model_size=w["size"], interpolate="nearest", fusion_mode="group",
                            inter_depth=3, model="mae", num_levels=1)
    opt = torch.optim.AdamW(model.parameters(), lr=1e-4, betas=(0.9, 0.99), weight_decay=0.01)
    B, WARM = 4, 2  # noqa: N806  (CPU batch 4-8, two warm-up steps, median of the timed ones)
    batch = synthetic_batch(ds.dataset, B, "cpu")
    batch.update(synthetic_targets(ds.dataset, B, "cpu"))
    times = []
    t_end = time.time() + seconds
    for i in range(50 + WARM):
        t0 = time.time()
        if phase == "pretrain":
            loss, _, _ = om.oracle_step(model, batch, "l2_norm")
        else:
            ob, _, _, logits = model({k: v.clone() for k, v in batch.items()}, phase)
            loss = oh.compute_loss_pred(model.dataset, ob, logits)
        opt.zero_grad()
        loss.backward()
        opt.step()
        dt = time.time() - t0
        if i >= WARM:
            times.append(dt)
        if time.time() > t_end and len(times) >= 5:    # >= 5 timed steps, bounded by the seconds budget
            break
    times.sort()
    med = times[len(times) // 2]
    return {"value": round(B / med, 4), "unit": "tiles/s", "cores": cores, "kind": "port", "cpu": cpu_model(),
            "sample": f"{len(times)} timed steps (after {WARM} warm-ups) of the same {workload} {phase} workload at B={B}, fp32 "
                      f"('highest' matmul precision), torch CPU threads={cores}, forward+loss+backward+AdamW, median step "
                      f"{med:.2f} s (min {times[0]:.2f} s); {host_cores()} schedulable CPUs, fp32 GEMM GFLOP/s by thread count {rates}"}


def cpu_baseline(workload: str, seconds: float, phase: str = "pretrain") -> dict:
    """Run the CPU leg in a child process with a hard wall-clock cap so the default bench always finishes in minutes."""
    import subprocess

    cap = 8 * seconds + 90
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-only", "--config", workload, "--cpu-seconds", str(seconds),
           "--phase", phase]
    env = dict(os.environ, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=cap, env=env)
        for line in reversed(r.stdout.strip().splitlines()):
            if line.startswith("{"):
                return json.loads(line)
        return {"value": None, "unit": "tiles/s", "cores": host_cores(), "kind": "port",
                "sample": f"CPU leg failed (rc={r.returncode}): {r.stderr.strip()[-200:]}"}
    except subprocess.TimeoutExpired:
        return {"value": None, "unit": "tiles/s", "cores": host_cores(), "kind": "port",
                "sample": f"CPU leg exceeded its {cap:.0f} s cap on this host"}


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="c3", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=32, help="tiles per GPU (reference default, conf/opt.py:20)")
    ap.add_argument("--loss", default="l2_norm")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp8"],
                    help="GEMM operand type: bf16 (BASELINE metric), fp8 = e4m3 forward GEMMs (BASELINE configs[4]: --config c5)")
    ap.add_argument("--phase", default="pretrain", choices=["pretrain", "probe", "finetune"],
                    help="pretrain = the BASELINE metric; probe / finetune = the supervised branch (configs c3, c4)")
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="budget of the CPU-baseline sample (0 = skip)")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--cpu-baseline-only", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--single-stream", action="store_true", help="profiling aid: no group-parallel streams (clean per-kernel times)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for rehearsals)")
    ap.add_argument("--from-host", action="store_true",
                    help="PCIe-inclusive variant (never the headline value): every step's batch starts in pinned host memory and "
                         "goes through BatchStager (async H2D on a copy stream + on-GPU flips / transposes)")
    ap.add_argument("--overlap-optimizer", action="store_true",
                    help="A/B aid: AdamW of step t inside the forward of step t+1 instead of at the end of the step (pretrain)")
    ap.add_argument("--rehearse-exchange", action="store_true",
                    help="N=1 under torch.distributed.run: create the one-rank RCCL group and run the bucketed exchange plan")
    ap.add_argument("--rehearse-dry", action="store_true",
                    help="the same launch plan (gradient hooks, backward segments, two-part AdamW) without a process group: what the "
                         "plan itself costs, without RCCL's one-rank self-copies")
    ap.add_argument("--exchange-mode", default=None, choices=["all_reduce", "rs_ag"],
                    help="gradient exchange at N > 1: all-reduce per bucket (default) or reduce-scatter -> sharded AdamW -> all-gather")
    ap.add_argument("--log-losses", action="store_true",
                    help="diagnostic: add the loss of every timed step to the JSON line (one tiny device copy per step)")
    ap.add_argument("--shapes", action="store_true", help="print per-shape kernel times to stderr (diagnostic)")
    args = ap.parse_args()
    if args.cpu_baseline_only:
        print(json.dumps(cpu_baseline_worker(args.config, args.cpu_seconds, args.phase)), flush=True)
        return
    if args.phase != "pretrain" and args.config not in SUP_WORKLOADS:
        raise SystemExit(f"--phase {args.phase}: choose --config from {sorted(SUP_WORKLOADS)}")

    import torch.distributed as dist
    from maestro_amd import hip
    from maestro_amd.train.trainer import PretrainLoop, SupervisedLoop, synthetic_batch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        # the line's n_gpus / value are whole-job figures: a launcher that started another number of ranks than --gpus names would
        # make them wrong silently -> refuse (non-zero exit) instead
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch multi-GPU runs with torch.distributed.run (one process per GPU)")
        raise SystemExit(f"--gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    ndev = torch.cuda.device_count()
    if args.backend == "nccl" and world > ndev:
        raise SystemExit(f"{world} ranks need {world} GPUs (found {ndev}); one process per GPU")
    local = local % max(ndev, 1)   # only differs from LOCAL_RANK in single-GPU gloo rehearsals
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1 or args.rehearse_exchange:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.backend)
        if dist.get_world_size() != args.gpus:
            raise SystemExit(f"--gpus {args.gpus} but the process group has {dist.get_world_size()} ranks")

    torch.set_num_threads(min(4, host_cores()))   # host-side torch ops are tiny (mask draws); a 128-thread pool only adds latency
    torch.manual_seed(42)            # identical initial weights on every rank (the loops also broadcast rank 0's)
    ds, model = build_model(args.config, args.phase)
    torch.manual_seed(42 + rank)     # per-rank mask draws from here on
    # Gradient exchange plan at N > 1 (DESIGN §6 "exchange budget"): configurations whose optimizer traffic is a large share of the step
    # (>= 300 M parameters: C3', C4, C5 -- 1.3-3.6 GB of fp32 gradients, AdamW 1.6-4.5 ms per step) take reduce-scatter -> AdamW on the
    # rank's 1 / N chunk -> all-gather (same bytes on the links as the all-reduce, (N - 1) / N of the optimizer pass saved on every
    # rank); the smaller ones (C2, C3: AdamW 0.5-0.9 ms) keep the all-reduce with the AdamW-overlapped tail bucket.  --exchange-mode
    # overrides; fp8 keeps the all-reduce (the e4m3 shadows of foreign chunks would need their scales).
    n_params = sum(p.numel() for p in model.parameters())
    exchange_mode, exchange_why = args.exchange_mode, "--exchange-mode"
    if exchange_mode is None and os.environ.get("MAESTRO_EXCHANGE"):      # flag > environment > rule
        exchange_mode, exchange_why = os.environ["MAESTRO_EXCHANGE"], "MAESTRO_EXCHANGE"
    if exchange_mode is None:
        big = n_params >= 300e6 and args.dtype != "fp8" and not args.overlap_optimizer
        exchange_mode = "rs_ag" if (big and (world > 1 or args.rehearse_exchange or args.rehearse_dry)) else "all_reduce"
        exchange_why = (f"{n_params / 1e6:.0f} M parameters " + (">= 300 M: sharded optimizer pass" if big else "< 300 M (or fp8): all-reduce, tail bucket under AdamW"))
    if args.phase == "pretrain":
        args.exchange_mode = exchange_mode
        loop = PretrainLoop(model, args.batch, dev, loss=args.loss, total_steps=args.steps + args.warmup, world_size=world,
                            exchange=True if (args.rehearse_exchange or args.rehearse_dry) else None,
                            overlap_optimizer=args.overlap_optimizer, dtype=args.dtype, exchange_mode=args.exchange_mode)
    else:
        loop = SupervisedLoop(model, args.batch, dev, phase=args.phase, total_steps=args.steps + args.warmup, world_size=world)
    warm_engine = loop.engine     # start-up passes of the first step (engine.py: warm_passes; MAESTRO_WARM_PASSES): read AFTER the first step
    batch = synthetic_batch(ds.dataset, args.batch, dev, seed=rank)
    batch.update(synthetic_targets(ds.dataset, args.batch, dev, seed=rank))
    if args.single_stream:
        loop.engine.multi_stream = False
    if args.from_host:
        import numpy as np
        from maestro_amd.train.staging import BatchStager, draw_transform_flags
        stager = BatchStager(dev, rasters=list(ds.dataset.inputs))
        host = [{k: v.cpu().pin_memory() for k, v in batch.items()} for _ in range(2)]   # what DataLoader(pin_memory=True) hands over
        rng, turn, inner = np.random.default_rng(7 + rank), [0], loop

        class _HostFed:     # same .step / .engine surface; the staging of step t+1 overlaps the GPU work of step t
            engine = inner.engine

            @staticmethod
            def step(_):
                turn[0] += 1
                return inner.step(stager.stage(host[turn[0] % 2], draw_transform_flags(rng, args.batch)))

            flush = getattr(inner, "flush", staticmethod(lambda: None))
        loop = _HostFed

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    flush = getattr(loop, "flush", lambda: None)   # --overlap-optimizer: the last update is queued for the next forward
    for _ in range(max(args.warmup, 3)):   # >= 3 so the launch segments are captured into hipGraphs before timing
        loop.step(batch)
    flush()
    sync()
    gsync = getattr(loop, "sync", None)
    if gsync is not None:           # exchange statistics of the timed region only (the `comm` object of the line)
        gsync.reset_stats(on=True)
    wait0 = getattr(loop.engine, "host_wait_s", 0.0)
    t0 = time.perf_counter()
    loss_log = []
    marks = []                # one event per step boundary: per-step GPU times (min / median) without any host sync
    for _ in range(args.steps):
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        marks.append(ev)
        loss = loop.step(batch)
        if args.log_losses:
            loss_log.append(loss.detach().clone())
    flush()   # the K-th optimizer update belongs to the timed region: K forwards, K backwards, K AdamW updates
    # host time spent ISSUING the steps (diagnostic: host-bound if ~= elapsed): the time blocked on the mask staging
    # ring's back-pressure (host >= 4 steps ahead of the GPU) is not issue work and is taken out
    t_issue = time.perf_counter() - t0 - (getattr(loop.engine, "host_wait_s", 0.0) - wait0)
    ev = torch.cuda.Event(enable_timing=True)
    ev.record()
    marks.append(ev)
    sync()
    elapsed = time.perf_counter() - t0
    step_seq = [a.elapsed_time(b) for a, b in zip(marks, marks[1:])]
    step_ms = sorted(step_seq)
    loss_val = float(loss.item())   # the engine's loss buffer is static: read it before the roofline leg runs more steps
    # Roofline leg: the same steps once more with HIP events around every MFMA-kernel launch on its stream (event
    # pairs cannot be recorded inside a captured graph, so these steps are launched eagerly; kernels are identical).
    timer = None if args.no_kernel_timing else hip.KernelTimer()
    comm = None
    if gsync is not None:
        # Self-diagnosis of the data-parallel run (no scaling claim is made from it): which backend and how many ranks the process
        # group really has, which exchange plan ran and why, what it moved per step and how long the step's main stream WAITED for
        # collectives (HIP events around GradSync's waits: the exchange time that backward / AdamW did not hide).
        rep = gsync.comm_report()
        gsync.reset_stats(on=False)
        try:
            rccl = ".".join(str(x) for x in torch.cuda.nccl.version())
        except Exception:  # noqa: BLE001
            rccl = None
        comm = {"backend": dist.get_backend() if dist.is_initialized() else None,
                "world_size": dist.get_world_size() if dist.is_initialized() else 1, "rccl_version": rccl,
                "exchange": rep.pop("mode"), "exchange_rule": exchange_why, **rep,
                "grad_mbytes": round(loop.engine.store.total * 4 / 1e6, 1),
                "how": "per optimizer step over the timed region; exposed_ms = HIP events on the main stream around the waits for the "
                       "buckets (all_reduce plan: the tail bucket's wait sits inside the two-part AdamW)"}
        if world > 1:               # the slowest rank's exposure is the job's
            t = torch.tensor([comm["exposed_ms_per_step"] or 0.0], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            comm["exposed_ms_per_step_max_over_ranks"] = round(float(t.item()), 4)
    if timer is not None:
        saved_ms = loop.engine.multi_stream
        loop.engine.multi_stream = False   # one kernel at a time, so each event pair brackets exactly one launch
        hip.set_kernel_timer(timer)
        for _ in range(args.steps):
            loop.step(batch)
        flush()
        hip.set_kernel_timer(None)
        loop.engine.multi_stream = saved_ms
        sync()
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        tiles = args.batch * world * args.steps
        value = tiles / elapsed
        w = dict((WORKLOADS if args.phase == "pretrain" else SUP_WORKLOADS)[args.config])
        if args.phase != "pretrain":
            w["gflop_tile"] = round(sup_gflop_per_tile(model, args.phase), 1)
        out = {
            "metric": f"MAE-{args.phase} tiles/sec", "value": round(value, 2), "unit": "tiles/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3),
            "step_ms": {"min": round(step_ms[0], 3), "median": round(step_ms[len(step_ms) // 2], 3),
                        "max": round(step_ms[-1], 3), "first": round(step_seq[0], 3), "all": [round(x, 2) for x in step_seq],
                        "how": "HIP events on the main stream at every step boundary (this rank); first = the step right after "
                               "the barrier + synchronize (empty GPU queue: the host's issue latency is exposed once)"},
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype,
            "data": "synthetic" if not args.from_host else "synthetic, fed from pinned host memory every step (PCIe-inclusive)",
            "config": {"workload": f"{args.config}: {w['desc']}", "tiles_per_gpu": args.batch,
                       "global_batch": args.batch * world, "loss": args.loss if args.phase == "pretrain" else "loss_pred",
                       "fusion_mode": "group", "inter_depth": 3,
                       "parallelism": f"dp{world}", "params_M": round(loop.engine.store.total / 1e6, 1),
                       "final_loss": round(loss_val, 5), "host_issue_ms_per_step": round(1e3 * t_issue / args.steps, 3),
                       "warm_passes": getattr(warm_engine, "warm_passes_run", 0),     # the passes that actually ran (0 under fp8 / overlap)
                       "exchange": (getattr(loop, "exchange_mode", "all_reduce") if getattr(loop, "sync", None) is not None else "none"),
                       "exchange_rule": exchange_why if getattr(loop, "sync", None) is not None else "one GPU: no exchange"},
            "whole_step": {"train_gflop_per_tile": w["gflop_tile"],
                           "mfma_frac": round(value / world * w["gflop_tile"] / 1e3 / MFMA_PEAK_TFLOPS, 4)},
        }
        if comm is not None:
            out["comm"] = comm
        if args.log_losses:
            out["losses"] = [float(x.item()) for x in loss_log]
        if args.dtype == "fp8":
            out["config"]["precision"] = ("forward GEMMs of the transformer layers: OCP e4m3 operands, scaled MFMA 16x16x128, fp32 "
                                          "accumulate, per-tensor delayed scaling; backward GEMMs and attention bf16; fp32 masters")
        if timer is not None:
            out["roofline"] = timer.roofline(MFMA_PEAK_TFLOPS)
            if out["roofline"]["kernel"] == "gemm_fp8_kernel":     # price an fp8 kernel against the fp8 MFMA peak
                out["roofline"].update(peak=FP8_PEAK_TFLOPS, frac=round(out["roofline"]["achieved"] / FP8_PEAK_TFLOPS, 4))
            elif args.dtype == "fp8" and timer.flops.get("gemm_fp8_kernel"):
                tot = timer.totals()
                ach = timer.flops["gemm_fp8_kernel"] / (tot["gemm_fp8_kernel"] * 1e-3) / 1e12
                out["roofline_fp8_kernel"] = {"bound": "mfma", "kernel": "gemm_fp8_kernel", "achieved": round(ach, 1),
                                              "peak": FP8_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / FP8_PEAK_TFLOPS, 4),
                                              "launches": timer.count["gemm_fp8_kernel"]}
            traffic_file = next((f for f in (os.path.join(ROOT, "profiles", f"r{r:02d}_hbm_traffic.json") for r in (9, 8, 7, 6, 5, 4, 3, 2, 1))
                                 if os.path.exists(f)), "")
            # PMC passes are separate runs (rocprofv3 --pmc) of the DEFAULT workload: the committed summary applies to it only
            if os.path.exists(traffic_file) and args.phase == "pretrain" and args.config == "c3" and args.batch == 32:
                kern = json.load(open(traffic_file))["kernels"].get(out["roofline"]["kernel"])
                if kern:
                    out["roofline"]["traffic"] = kern["hbm_bytes_per_launch"]
                    out["roofline"]["traffic_source"] = (f"profiles/{os.path.basename(traffic_file)} (rocprofv3 --pmc FETCH_SIZE / "
                                                         "WRITE_SIZE passes of this command), collected at commit "
                                                         + str(json.load(open(traffic_file)).get("commit", "unrecorded")))
                whole = json.load(open(traffic_file)).get("hbm_bytes_per_step")
                if whole:   # every kernel's PMC bytes per launch x launches per step: the step's second bound next to mfma_frac
                    out["whole_step"]["hbm_gb_per_step"] = round(whole / 1e9, 1)
                    out["whole_step"]["hbm_frac"] = round(whole / (elapsed / args.steps) / (HBM_PEAK_TBS * 1e12), 4)
            out["roofline"]["measured_over"] = f"{args.steps} eagerly launched single-stream steps right after the timed region"
            out["kernel_times_ms_per_step"] = timer.summary(args.steps)
            # the HBM-bound sub-stages separately, as achieved GB/s on their ALGORITHMIC bytes (same eager leg,
            # HIP events on the launch stream); peak = 8 TB/s (the guide measures 6.3 TB/s for a streaming copy).  Most of these
            # launches move 30-100 MB in 8-25 us: the launch ramp alone (~2 us) caps them well below a long stream's rate.
            out["hbm_substages"] = timer.hbm_substages(HBM_PEAK_TBS * 1e3)
        if timer is not None and args.shapes:
            for ms, kind, shape, n, tf in timer.by_shape(args.steps)[:80]:
                print(f"{ms:8.3f} ms/step {kind:18s} {str(shape):26s} x{n:3d}/step {tf:7.1f} TFLOP/s", file=sys.stderr)
        if world == 1 and args.cpu_seconds > 0:
            out["cpu_baseline"] = cpu_baseline(args.config, args.cpu_seconds, args.phase)
        print(json.dumps(out), flush=True)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
