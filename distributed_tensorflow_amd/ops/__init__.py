from . import native, reference  # noqa: F401
