from . import datasets, losses  # noqa: F401
