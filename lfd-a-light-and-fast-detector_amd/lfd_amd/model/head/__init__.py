from .lfd_head import *
from .fcos_head import *
