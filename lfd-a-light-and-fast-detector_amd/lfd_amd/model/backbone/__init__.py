from .lfd_resnet import *
