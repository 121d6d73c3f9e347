from .simple_neck import *
from .fpn import *
from .simple_fpn import *
