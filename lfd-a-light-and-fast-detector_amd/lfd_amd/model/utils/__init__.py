from .nms import *  # noqa: F401,F403
