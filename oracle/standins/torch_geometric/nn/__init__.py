from . import inits  # noqa: F401
