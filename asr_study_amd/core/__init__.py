from . import layers, metrics, ctc_utils, models, optimizers, engine  # noqa: F401
