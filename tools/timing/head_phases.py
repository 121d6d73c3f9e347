import sys, os, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'lfd-a-light-and-fast-detector_amd'))
import torch
from lfd_amd import configs, _lib
m = configs.build_model('WIDERFACE_LFD_S'); configs.perturb_weights(m); m.eval().cuda()
x = (torch.rand(8,1080,1920,3, device='cuda')*2-1).half()
with torch.no_grad():
    for _ in range(3): m.forward_resident(x)
torch.cuda.synchronize()
L=_lib.lib(); L.lfd_debug_h2_timing.argtypes=[ctypes.c_void_p]
buf=(ctypes.c_ulonglong*96)(); L.lfd_debug_h2_timing(buf)
for p in range(3):
    v=[buf[p*32+i] for i in range(18)]
    groups=[v[4+i]-v[3+i] for i in range(11)]
    print('pass',p+1,'level-setup',v[1]-v[0],'chunk-setup',v[2]-v[1],'groups',groups,'last-group+',v[16]-v[14],'flush',v[17]-v[16],'total',v[17]-v[0])
