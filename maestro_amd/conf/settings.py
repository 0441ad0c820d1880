"""Non-dataset config groups (mask / model / opt / trainer / run / data) and the CLI override grammar.

Same field names and defaults as the reference's hydra-zen store entries:
``maestro/conf/mask.py:9-15``, ``model.py:9-18``, ``opt.py:9-50``, ``trainer.py:9-14``,
``run.py:9-27``, ``data.py:9-13``; top-level group list ``maestro/conf/experiment.py:7-32``.
"""

from __future__ import annotations

import ast
from dataclasses import dataclass, field, fields, is_dataclass


@dataclass
class MaskConfig:
    mask_ratio: float = 0.75
    mask_scale: float = 0.0
    mask_mod: float | None = 0.25
    mask_bands: float | None = None
    mask_dates: float | None = 0.25
    mask_loc: float | None = 0.25


@dataclass
class ModelConfig:
    interpolate: str = "nearest"
    fusion_mode: str = "group"
    inter_depth: int = 3
    model: str = "mae"
    model_size: str = "tiny"
    type_head: str = "attentive"
    loss: str = "l1_norm"
    use_date_enc: bool = True
    use_ema: bool = True


@dataclass
class OptConfig:
    b1: float = 0.9
    b2: float = 0.99
    wd: float = 0.01
    accumulate_grad_batches: int = 1


@dataclass
class OptPretrainConfig(OptConfig):
    base_lr: float = 3e-5
    epochs: int = 20
    batch_size: int = 32


@dataclass
class OptProbeConfig(OptConfig):
    base_lr: float = 1e-5
    epochs: int = 10
    batch_size: int = 32


@dataclass
class OptFinetuneConfig(OptConfig):
    base_lr: float = 1e-5
    epochs: int = 20
    batch_size: int = 32
    lw_decay: float | None = None
    final_factor: float = 2
    monitor: str | None = None
    patience: int | None = 5


@dataclass
class TrainerConfig:
    accelerator: str = "auto"
    devices: str = "auto"
    strategy: str = "ddp_find_unused_parameters_true"
    precision: str = "16-mixed"
    num_nodes: int = 1


@dataclass
class DataConfig:
    use_transform: bool = True
    random_dates: bool = True
    random_crop: bool = True
    num_workers: int = 12


@dataclass
class RunConfig:
    exp_dir: str = None
    exp_name: str = None
    exp_uuid: str | None = None
    load_name: str | None = None
    load_phase: str = "pretrain"
    load_uuid: str | None = None
    load_ckpt_path: str | None = None
    fit_name: str | None = None
    fit_phase: str = "pretrain"
    fit_uuid: str | None = None
    fit_ckpt_path: str | None = None
    reproducible: bool = True
    seed: int = 42
    logged_images_per_epoch: int = 5
    use_clearml: bool = False
    clearml_project_name: str = "ssl"
    clearml_tags: list = field(default_factory=lambda: ["multimodal", "hydra"])
    clearml_offline_mode: bool = False


GROUPS = {
    "run": RunConfig,
    "opt_pretrain": OptPretrainConfig,
    "opt_probe": OptProbeConfig,
    "opt_finetune": OptFinetuneConfig,
    "data": DataConfig,
    "mask": MaskConfig,
    "model": ModelConfig,
    "trainer": TrainerConfig,
}


def _parse_value(text: str):
    low = text.lower()
    if low in ("null", "none"):
        return None
    if low in ("true", "false"):
        return low == "true"
    try:
        return ast.literal_eval(text)
    except (ValueError, SyntaxError):
        return text


def load_experiment(overrides: list[str] | None = None) -> dict:
    """Build the experiment config from ``group.field=value`` overrides (``main.py`` CLI grammar).

    Returns a dict with one entry per Hydra group of the reference's ``base_ssl_experiment``
    plus ``datasets`` (a :class:`~maestro_amd.conf.datasets.DatasetsConfig`).  Dataset overrides
    use ``datasets.name_dataset=flair``, ``datasets.flair.filter_inputs=[aerial,s2]`` and
    ``datasets.flair.aerial.image_size=256``-style paths.
    """
    from maestro_amd.conf import datasets as ds

    cfg = {name: cls() for name, cls in GROUPS.items()}
    ds_kwargs: dict = {"root_dir": None, "name_dataset": None}
    ds_sub: dict[str, dict] = {"flair": {}, "treesatai_ts": {}, "pastis_hd": {}, "s2_naip": {}}
    leaf: list[tuple[str, list[str], object]] = []
    for item in overrides or []:
        if "=" not in item:
            raise ValueError(f"Invalid override {item!r}; expected group.field=value")
        path, raw = item.split("=", 1)
        parts = path.split(".")
        # hydra list syntax without quotes: [a,b] -> ["a","b"]
        if raw.startswith("[") and raw.endswith("]") and "'" not in raw and '"' not in raw:
            inner = [s.strip() for s in raw[1:-1].split(",") if s.strip()]
            value = [_parse_value(s) for s in inner]
        else:
            value = _parse_value(raw)
        if parts[0] == "datasets":
            if len(parts) == 2:
                ds_kwargs[parts[1]] = value
            elif len(parts) == 3:
                ds_sub[parts[1]][parts[2]] = value
            else:
                leaf.append((parts[1], parts[2:], value))
            continue
        if parts[0] not in cfg or len(parts) != 2:
            raise ValueError(f"Unknown config path {path!r}")
        names = {f.name for f in fields(cfg[parts[0]])}
        if parts[1] not in names:
            raise ValueError(f"Unknown field {parts[1]!r} in group {parts[0]!r}")
        setattr(cfg[parts[0]], parts[1], value)

    classes = {"flair": ds.FLAIRConfig, "treesatai_ts": ds.TreeSatAITSConfig,
               "pastis_hd": ds.PASTISHDConfig, "s2_naip": ds.S2NAIPConfig}
    built = {}
    for name, cls in classes.items():
        kw = dict(ds_sub[name])
        mods = {}
        for dname, sub, value in leaf:
            if dname != name:
                continue
            proto = cls()
            mod = mods.setdefault(sub[0], getattr(proto, sub[0]))
            tgt = mod
            for p in sub[1:-1]:
                tgt = getattr(tgt, p)
            if not (is_dataclass(tgt) and hasattr(tgt, sub[-1])):
                raise ValueError(f"Unknown dataset field {'.'.join(sub)!r}")
            setattr(tgt, sub[-1], value)
        kw.update(mods)
        built[name] = cls(**kw)
    if ds_kwargs["name_dataset"] is not None:
        cfg["datasets"] = ds.DatasetsConfig(**ds_kwargs, **built)
    else:
        cfg["datasets"] = None
    return cfg
