from .focal_loss import *          # noqa: F401,F403
from .iou_loss import *            # noqa: F401,F403
from .cross_entropy_loss import *  # noqa: F401,F403
from .pointwise_loss import *      # noqa: F401,F403
from .bce_losses import *          # noqa: F401,F403
