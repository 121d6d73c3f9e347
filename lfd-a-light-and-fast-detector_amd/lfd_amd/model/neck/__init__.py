from .simple_neck import *
