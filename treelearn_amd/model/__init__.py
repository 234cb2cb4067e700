from .net import TreeLearn  # noqa: F401
