from . import inits, norm, glob, conv  # noqa: F401
