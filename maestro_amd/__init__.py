"""MI355X-native MAE pretraining hot path for MAESTRO (drop-in for maestro/ssl + maestro/train)."""

__version__ = "0.1.0"
