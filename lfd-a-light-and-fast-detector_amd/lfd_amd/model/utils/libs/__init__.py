from . import nms_ext  # noqa: F401
