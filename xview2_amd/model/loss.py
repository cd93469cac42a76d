from ..criterion import Loss, compute_loss  # noqa: F401
