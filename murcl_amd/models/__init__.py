from . import abmil, cl, rlmil  # noqa: F401
