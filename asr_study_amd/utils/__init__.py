from .hparams import HParams  # noqa: F401
