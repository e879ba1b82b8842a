from . import abmil, cl, clam, dsmil, rlmil  # noqa: F401
