from .backbone import *   # noqa: F401,F403
from .neck import *       # noqa: F401,F403
from .head import *       # noqa: F401,F403
from .losses import *     # noqa: F401,F403
from .utils import *      # noqa: F401,F403
from .lfd import LFD      # noqa: F401
from .fcos import FCOS, FCOSv1    # noqa: F401
from .lfdv2 import LFDv2  # noqa: F401
