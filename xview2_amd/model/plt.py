from ..lightning import Model  # noqa: F401
