"""Config surface of the MAE pretraining path (reference: ``maestro/conf/**``)."""

from maestro_amd.conf.datasets import (  # noqa: F401
    DatasetConfig,
    DatasetsConfig,
    FLAIRConfig,
    InputRasterConfig,
    PASTISHDConfig,
    PatchSizeConfig,
    RasterConfig,
    S2NAIPConfig,
    TargetConfig,
    TargetRasterConfig,
    TreeSatAITSConfig,
)
from maestro_amd.conf.settings import (  # noqa: F401
    DataConfig,
    MaskConfig,
    ModelConfig,
    OptFinetuneConfig,
    OptPretrainConfig,
    OptProbeConfig,
    RunConfig,
    TrainerConfig,
    load_experiment,
)
