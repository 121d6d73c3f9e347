from .lfd_head import *
