"""Dataset geometry configs (the reference's Hydra ``datasets`` group, without Hydra).

Field names and defaults follow the reference's dataclasses so that callers can write
``FLAIRConfig(filter_inputs=["aerial", "s2"], filter_targets=[])`` exactly as they do
against the reference:

* raster/input/target leaf configs      -> reference ``maestro/conf/dataset/utils.py:16-76``
* derived attributes (inputs, groups..) -> reference ``maestro/conf/dataset/utils.py:79-155``
* FLAIR / TreeSatAI-TS / PASTIS-HD / S2-NAIP tables
                                         -> reference ``maestro/conf/dataset/{flair,treesatai_ts,pastis_hd,s2_naip}.py``
* ``DatasetsConfig`` selector            -> reference ``maestro/conf/datasets.py:13-41``

Only geometry matters to the MAE hot path: (num_dates, bands, image_size, patch_size,
norm_bands, name_group, name_embed, rescale_elev) per input modality plus ``grid_pos_enc``.
The dataset *readers* are out of scope, so ``dataset_class`` is None here.
"""

from __future__ import annotations

import copy
from dataclasses import dataclass, field
from math import gcd
from typing import Any

ALLOWED_TARGETS = ("classif", "multilabel_classif", "segment")


@dataclass
class PatchSizeConfig:
    mae: int = None  # MISSING in the reference
    dinov2_imagenat: int = 14
    dinov2_sat: int = 16
    dofa: int = 16
    croma: int = 8


@dataclass
class RasterConfig:
    bands: Any = None  # int | list[list[int]]
    norm_bands: list | None = None
    mask_threshold: float = 0.0
    num_dates: int = 1
    norm_fac: float | None = None
    log_scale: bool = False
    rescale_elev: bool = False
    name_embed: str | None = None


@dataclass
class InputConfig:
    image_size: int = None
    patch_size: PatchSizeConfig = None
    name_group: str | None = None


@dataclass
class TargetConfig:
    type_target: str = None
    num_classes: int = None
    missing_val: int = -1

    def __post_init__(self) -> None:
        if self.type_target not in ALLOWED_TARGETS:
            raise ValueError(
                f"Invalid target {self.type_target}.Expected one of {list(ALLOWED_TARGETS)}"
            )


@dataclass
class InputRasterConfig(RasterConfig, InputConfig):
    pass


@dataclass
class TargetRasterConfig(RasterConfig, TargetConfig):
    pass


def _inp(size, patch, bands, **kw):
    return InputRasterConfig(image_size=size, patch_size=PatchSizeConfig(mae=patch), bands=bands, **kw)


class DatasetConfig:
    """Common derived-attribute logic shared by all dataset configs."""

    # subclasses fill these
    _RESOLUTIONS: dict = {}
    _TOTAL_METERS: float = 0.0

    def _finish(self) -> None:
        self.dataset_class = None
        self.total_meters = self._TOTAL_METERS
        self.sizes = {}
        for name_mod, res in self._RESOLUTIONS.items():
            if name_mod not in self.__dict__:
                raise ValueError(f"Invalid modality {name_mod} specified in resolution.")
            mod = getattr(self, name_mod)
            mod.resolution_meters = float(res)
            size = self.total_meters / mod.resolution_meters
            if not float(size).is_integer() and name_mod in self.filter_inputs + self.filter_targets:
                raise ValueError(f"Modality {name_mod}'s resolution does not divide image extent.")
            self.sizes[name_mod] = round(size)
        size_gcd = gcd(*self.sizes.values())
        crop_gcd = self.crop_meters / self.total_meters * size_gcd
        if not float(crop_gcd).is_integer():
            raise ValueError(
                "Crop meters does not correspond to an integer number of pixels."
                f"Use a multiple of {self.total_meters / size_gcd}."
            )
        self.size_gcd, self.crop_gcd = size_gcd, round(crop_gcd)

        self.log_inputs = [m for m in self.log_inputs if m in self.filter_inputs] or self.filter_inputs
        if self.ref_input and self.ref_input not in self.filter_inputs:
            raise ValueError(f"Ref input {self.ref_input} is not selected.")
        self.inputs, self.targets = {}, {}
        for dst, names in ((self.inputs, self.filter_inputs), (self.targets, self.filter_targets)):
            for name_mod in names:
                if name_mod not in self.__dict__:
                    raise ValueError(f"Invalid modality name {name_mod}. Not an attribute.")
                dst[name_mod] = getattr(self, name_mod)
        self.rasters = {
            n: m for n, m in (*self.inputs.items(), *self.targets.items()) if isinstance(m, RasterConfig)
        }
        self.groups = [
            (n, m.name_group if m.name_group is not None else n) for n, m in self.inputs.items()
        ]


def _dataset(name, *, scalars, inputs, targets, resolutions, total_meters):
    """Build a dataset-config class from tables (keeps each dataset to a few lines of data)."""

    def __init__(self, **kw):
        unknown = set(kw) - set(scalars) - set(inputs)
        if unknown:
            raise TypeError(f"{name}: unexpected arguments {sorted(unknown)}")
        for key, default in scalars.items():
            setattr(self, key, copy.deepcopy(kw.get(key, default)))
        for key, factory in inputs.items():
            setattr(self, key, kw[key] if key in kw else factory())
        for key, factory in targets.items():
            setattr(self, key, factory())
        self._finish()

    return type(
        name,
        (DatasetConfig,),
        {"__init__": __init__, "_RESOLUTIONS": resolutions, "_TOTAL_METERS": total_meters,
         "__doc__": f"{name} geometry (see module docstring for the reference file)."},
    )


_S1 = dict(bands=2, norm_bands=[1, 1], num_dates=4, name_group="s1")

FLAIRConfig = _dataset(
    "FLAIRConfig",
    scalars=dict(rel_dir="FLAIR-HUB", csv_dir=None, version=None, val_pretrain=True, filter_percent=None,
                 repeats=1, crop_meters=102.4, grid_pos_enc=160, ref_input="aerial",
                 log_inputs=["aerial", "spot"],
                 filter_inputs=["aerial", "dem", "s2", "s1_asc", "s1_des"], filter_targets=["cosia"]),
    inputs=dict(
        aerial=lambda: _inp(512, 16, [[3, 0, 1, 2]], norm_bands=[1, 3], norm_fac=255.0),
        dem=lambda: _inp(512, 32, 2, norm_fac=1000.0, rescale_elev=True),
        spot=lambda: _inp(64, 4, 4, norm_fac=2000.0),
        s2=lambda: _inp(10, 2, 10, norm_bands=[4, 4, 2], num_dates=16, mask_threshold=0.0, norm_fac=5000.0),
        s1_asc=lambda: _inp(10, 2, norm_fac=5.0, log_scale=True, **_S1),
        s1_des=lambda: _inp(10, 2, norm_fac=5.0, log_scale=True, **_S1),
    ),
    targets=dict(
        cosia=lambda: TargetRasterConfig(type_target="segment", num_classes=15, missing_val=-1, bands=1),
        lpis=lambda: TargetRasterConfig(type_target="segment", num_classes=74, missing_val=-1, bands=1),
    ),
    resolutions=dict(cosia=0.2, lpis=0.2, aerial=0.2, dem=0.2, spot=1.6, s2=10.24, s1_asc=10.24, s1_des=10.24),
    total_meters=102.4,
)

TreeSatAITSConfig = _dataset(
    "TreeSatAITSConfig",
    scalars=dict(rel_dir="TreeSatAI-TS", val_pretrain=True, filter_percent=None, crop_meters=60.0,
                 grid_pos_enc=96, ref_input=None, log_inputs=["aerial"],
                 filter_inputs=["aerial", "s2", "s1_asc", "s1_des"], filter_targets=["treesat_mlc_thresh"]),
    inputs=dict(
        aerial=lambda: _inp(300, 20, 4, norm_bands=[1, 3], norm_fac=255.0),
        s2=lambda: _inp(6, 2, 10, norm_bands=[4, 4, 2], num_dates=16, mask_threshold=0.0, norm_fac=5000.0),
        s1_asc=lambda: _inp(6, 2, norm_fac=5.0, log_scale=True, **_S1),
        s1_des=lambda: _inp(6, 2, norm_fac=5.0, log_scale=True, **_S1),
    ),
    targets=dict(
        treesat_mlc=lambda: TargetConfig(type_target="multilabel_classif", num_classes=15, missing_val=-1),
        treesat_mlc_thresh=lambda: TargetConfig(type_target="multilabel_classif", num_classes=15, missing_val=-1),
    ),
    resolutions=dict(aerial=0.2, s2=10.0, s1_asc=10.0, s1_des=10.0),
    total_meters=60.0,
)

PASTISHDConfig = _dataset(
    "PASTISHDConfig",
    scalars=dict(rel_dir="PASTIS-HD", val_pretrain=True, filter_percent=None, fold=None, repeats=8,
                 crop_meters=160, grid_pos_enc=256, ref_input="s2", log_inputs=["spot"],
                 filter_inputs=["spot", "s2", "s1_asc", "s1_des"], filter_targets=["pastis_seg"]),
    inputs=dict(
        spot=lambda: _inp(160, 16, 3, norm_fac=255.0),
        s2=lambda: _inp(16, 2, 10, norm_bands=[4, 4, 2], num_dates=16, norm_fac=10000.0),
        s1_asc=lambda: _inp(16, 2, [[0, 1]], norm_bands=[1, 1], num_dates=4, norm_fac=20.0, name_group="s1"),
        s1_des=lambda: _inp(16, 2, [[0, 1]], norm_bands=[1, 1], num_dates=4, norm_fac=20.0, name_group="s1"),
    ),
    targets=dict(
        pastis_seg=lambda: TargetRasterConfig(type_target="segment", num_classes=19, missing_val=19, bands=1),
        pastis_mlc=lambda: TargetConfig(type_target="multilabel_classif", num_classes=18),
    ),
    resolutions=dict(pastis_seg=10, spot=1.0, s2=10.0, s1_asc=10.0, s1_des=10.0),
    total_meters=1280 * 1.0,
)

S2NAIPConfig = _dataset(
    "S2NAIPConfig",
    scalars=dict(rel_dir="s2-naip-urban", val_pretrain=True, test_pretrain=True, repeats=5, crop_meters=120,
                 grid_pos_enc=192, ref_input=None, log_inputs=["aerial", "spot"],
                 filter_inputs=["aerial", "spot", "s2", "s1"], filter_targets=[]),
    inputs=dict(
        aerial=lambda: _inp(384, 16, [[3, 0, 1, 2]], norm_bands=[1, 3], norm_fac=255.0),
        spot=lambda: _inp(128, 16, 3, norm_fac=255.0),
        landsat=lambda: _inp(12, 2, 11, num_dates=16, norm_fac=5000.0),
        s2=lambda: _inp(12, 2, 10, norm_bands=[4, 4, 2], num_dates=16, norm_fac=5000.0),
        s1=lambda: _inp(12, 2, 2, norm_bands=[1, 1], num_dates=4, norm_fac=20.0),
    ),
    targets=dict(
        osm_seg=lambda: TargetRasterConfig(type_target="segment", num_classes=6, missing_val=-1),
    ),
    resolutions=dict(osm_seg=1.25, aerial=1.25, spot=1.25, landsat=10.0, s2=10.0, s1=10.0),
    total_meters=512 * 1.25,
)


class DatasetsConfig:
    """Holds the four dataset configs and selects ``.dataset`` by ``name_dataset``."""

    def __init__(self, root_dir=None, name_dataset=None, treesatai_ts=None, pastis_hd=None,
                 flair=None, s2_naip=None) -> None:
        self.root_dir = root_dir
        self.name_dataset = name_dataset
        self.treesatai_ts = treesatai_ts if treesatai_ts is not None else TreeSatAITSConfig()
        self.pastis_hd = pastis_hd if pastis_hd is not None else PASTISHDConfig()
        self.flair = flair if flair is not None else FLAIRConfig()
        self.s2_naip = s2_naip if s2_naip is not None else S2NAIPConfig()
        if name_dataset not in ("treesatai_ts", "pastis_hd", "flair", "s2_naip"):
            raise ValueError(f"Invalid dataset name {name_dataset}. Not an attribute.")
        self.dataset = getattr(self, name_dataset)
        self.dataset_class = self.dataset.dataset_class
