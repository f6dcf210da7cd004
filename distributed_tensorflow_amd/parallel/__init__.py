from . import comm  # noqa: F401
