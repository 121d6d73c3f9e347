import sys, os, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'lfd-a-light-and-fast-detector_amd'))
import torch
from lfd_amd import configs, engine, ops
from lfd_amd._lib import check, lib, ptr, stream_ptr
dev = torch.device('cuda', 0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
m = configs.build_model('WIDERFACE_LFD_S'); configs.perturb_weights(m); m.eval().to(dev)
x = (torch.rand(n, 1080, 1920, 3, device=dev) * 2 - 1).half()
meta = torch.tensor([[1920.0, 1080.0, 1.0]] * n).cuda()
with torch.no_grad():
    cls, reg = m.forward_resident(x)
    thr = float(torch.quantile(cls.float().sigmoid().reshape(-1)[:4000000], 1 - 256.0 / 43620))
    plan = engine.get_plan(m, m._backbone, m._neck, m._head, dev)
    st = plan.state_for(n, 1080, 1920)
    desc, _ = m._detect_desc(thr, 0.4, False, 8192)
    out = ops.detect_outputs(desc, n, dev)
    ops.detect_workspace_reset(desc, n, out)
    hs = st.head_groups[0][0]
    d, lv, ab1, ab2 = hs['desc'], hs['levels'], hs['ab1'], hs['ab2']
    z = ops.zero_line(dev); l = lib(); sp = stream_ptr()
    def unf(): check(l.lfd_head_forward_f16(C.byref(d), 3, lv, ptr(ab1), ptr(ab2), None, ptr(st.cls), ptr(st.reg), ptr(z), sp), 'p3')
    def fus():
        check(l.lfd_head_forward_decode_f16(C.byref(d), lv, ptr(ab1), ptr(ab2), None, None, ptr(z), C.byref(desc), ptr(meta), ptr(out.ws), out.ws.numel(), sp), 'p3d')
    def fus_w():
        check(l.lfd_head_forward_decode_f16(C.byref(d), lv, ptr(ab1), ptr(ab2), ptr(st.cls), ptr(st.reg), ptr(z), C.byref(desc), ptr(meta), ptr(out.ws), out.ws.numel(), sp), 'p3dw')
    def post(): ops.detect_from_candidates(desc, n, out)
    def t(f, reps=30):
        f(); torch.cuda.synchronize()
        e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): f()
        e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)*1000/reps
    t0=time.time()
    while time.time()-t0<0.3: unf(); torch.cuda.synchronize()
    def unf_full():
        unf(); ops.detect_batched(desc, st.cls, st.reg, meta, out=out)
    for r in range(3):
        a=t(unf); b=t(lambda:(fus(),post())); u=t(unf_full)
        ops.detect_workspace_reset(desc, n, out); torch.cuda.synchronize()
        f1=t(fus, reps=20); post(); torch.cuda.synchronize()
        print('bs %d: pass3 %.1f us | fused alone %.1f | fused+post %.1f | pass3+detect_batched %.1f' % (n,a,f1,b,u))
    torch.cuda.synchronize(); print('counts', out.counts[:2].tolist())
