from . import abmil, cl, dsmil, rlmil  # noqa: F401
