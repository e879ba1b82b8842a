from . import losses  # noqa: F401
