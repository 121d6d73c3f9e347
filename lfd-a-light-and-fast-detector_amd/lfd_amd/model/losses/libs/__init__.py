from . import sigmoid_focal_loss_ext  # noqa: F401
