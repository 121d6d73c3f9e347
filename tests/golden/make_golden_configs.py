"""tests/golden/make_golden_configs.py -- pins lfd_amd.configs.ARCHS to the REFERENCE's config scripts.

    python tests/golden/make_golden_configs.py          (build container only: reads /root/reference)

For each of the eight shipped model configurations the script takes the source of `prepare_model()` out of the
reference's config file (WIDERFACE_train/WIDERFACE_LFD_{L,M,S,XS}.py, TT100K_train/TT100K_LFD_{L,S}.py,
TrafficLight_train/TL_LFD_{L,S}.py), executes that
function body with RECORDING stand-ins for the classes it instantiates (LFDResNet, SimpleNeck, LFDHead, LFD and the
loss classes), and stores the keyword arguments every constructor received under
known_answers.json['reference_model_configs'].  Nothing of the reference is imported (the config files pull the whole
data pipeline at import time), and no reference source is copied: only the evaluated kwargs are kept.
tests/test_host_logic.py::test_archs_match_reference_configs compares configs.ARCHS against them.
"""
import ast
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))
REF = '/root/reference'
FILES = {
    'WIDERFACE_LFD_L': 'WIDERFACE_train/WIDERFACE_LFD_L.py', 'WIDERFACE_LFD_M': 'WIDERFACE_train/WIDERFACE_LFD_M.py',
    'WIDERFACE_LFD_S': 'WIDERFACE_train/WIDERFACE_LFD_S.py', 'WIDERFACE_LFD_XS': 'WIDERFACE_train/WIDERFACE_LFD_XS.py',
    'TT100K_LFD_L': 'TT100K_train/TT100K_LFD_L.py', 'TT100K_LFD_S': 'TT100K_train/TT100K_LFD_S.py',
    'TL_LFD_L': 'TrafficLight_train/TL_LFD_L.py', 'TL_LFD_S': 'TrafficLight_train/TL_LFD_S.py',
}


class _Sym(list):
    """symbolic per-level list: only its length is a value"""

    def __init__(self, tag, n):
        super().__init__([tag] * n)
        self.tag = tag


class _Rec(object):
    """stands in for an instantiated reference class: remembers its kwargs"""

    def __init__(self, cls, kwargs):
        self.cls, self.kwargs = cls, kwargs

    # what prepare_model() reads back from the objects it built: one entry per tapped level (its len() is used)
    def _levels(self):
        if 'out_indices' in self.kwargs:
            return len(self.kwargs['out_indices'])
        return len(self.kwargs['num_input_strides_list'])

    @property
    def num_output_channels_list(self):
        return _Sym('<%s.num_output_channels_list>' % self.cls, self._levels())

    @property
    def num_output_strides_list(self):
        return _Sym('<%s.num_output_strides_list>' % self.cls, self._levels())


def _recorder(name, log):
    def make(**kwargs):
        r = _Rec(name, kwargs)
        log.append(r)
        return r
    r_type = type(name, (), {})     # so that type(obj).__name__ is the class name (LFDHead gets it that way)

    def make_typed(**kwargs):
        obj = r_type()
        obj.__dict__['_rec'] = _Rec(name, kwargs)
        log.append(obj._rec)
        return obj
    return make if name in ('LFDResNet', 'SimpleNeck', 'LFDHead', 'LFD') else make_typed


def _plain(v):
    if isinstance(v, _Rec):
        return '<%s>' % v.cls
    if hasattr(v, '_rec'):
        return '<%s>' % v._rec.cls
    if isinstance(v, _Sym):
        return v.tag
    if isinstance(v, (list, tuple)):
        return [_plain(e) for e in v]
    if isinstance(v, dict):
        return {k: _plain(e) for k, e in v.items()}
    return v


def extract(path):
    tree = ast.parse(open(path).read())
    fn = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == 'prepare_model']
    assert len(fn) == 1, path
    mod = ast.Module(body=fn, type_ignores=[])
    log = []
    ns = {'config_dict': {}}
    for cls in ('LFDResNet', 'SimpleNeck', 'LFDHead', 'LFD', 'FocalLoss', 'IoULoss', 'CrossEntropyLoss', 'GIoULoss',
                'DIoULoss', 'CIoULoss', 'SmoothL1Loss', 'MSELoss', 'QualityFocalLoss', 'BCEWithLogitsLoss'):
        ns[cls] = _recorder(cls, log)
    exec(compile(mod, path, 'exec'), ns)
    ns['prepare_model']()
    out = {}
    for r in log:
        out[r.cls] = {k: _plain(v) for k, v in r.kwargs.items()}
    out['config_dict'] = {k: _plain(v) for k, v in ns['config_dict'].items() if k != 'model'}
    return out


def main():
    ka_path = os.path.join(HERE, 'known_answers.json')
    ka = json.load(open(ka_path))
    ka['reference_model_configs'] = {name: extract(os.path.join(REF, rel)) for name, rel in FILES.items()}
    json.dump(ka, open(ka_path, 'w'), indent=1)
    for name, d in ka['reference_model_configs'].items():
        bb = d['LFDResNet']
        print(name, bb['stem_mode'], bb['stem_channels'], bb['body_architecture'], bb['body_channels'], bb['out_indices'])


if __name__ == '__main__':
    main()
