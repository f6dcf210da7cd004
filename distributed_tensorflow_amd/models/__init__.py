from .spec import ModelSpec, ArenaLayout, TF_VARIABLE_NAMES  # noqa: F401
