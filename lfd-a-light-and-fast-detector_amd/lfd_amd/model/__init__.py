from .backbone import *   # noqa: F401,F403
from .neck import *       # noqa: F401,F403
from .head import *       # noqa: F401,F403
