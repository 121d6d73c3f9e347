"""oracle/build_ref.py -- TEST INFRASTRUCTURE.

Compiles the reference's OWN CPU NMS extension, unmodified, from the sources where
they lie under /root/reference (lfd/model/utils/build/nms/src/nms_ext.cpp and
cpu/nms_cpu.cpp; the .cu files need THC and cannot be built on any current torch),
into oracle/_ref/ (git-ignored, but shipped to the GPU box by gpurun).  No reference
source is copied into this repository.  The resulting module `lfd_ref_nms_ext`
exposes nms / soft_nms / nms_match exactly as the reference's pybind module does
(nms_ext.cpp:45-49) and is used (a) to pin oracle/lfd_oracle.c and (b) optionally
as the `cpu_baseline.kind == "reference"` NMS leg in bench.py.
"""
import os
import sys

REF = '/root/reference/lfd/model/utils/build/nms/src'
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, '_ref')
NAME = 'lfd_ref_nms_ext'


def built_path():
    import glob
    c = glob.glob(os.path.join(OUT, NAME + '*.so'))
    return c[0] if c else None


def build(verbose=False):
    if not os.path.isdir(REF):
        return built_path()
    os.makedirs(OUT, exist_ok=True)
    p = built_path()
    srcs = [os.path.join(REF, 'nms_ext.cpp'), os.path.join(REF, 'cpu', 'nms_cpu.cpp')]
    if p and all(os.path.getmtime(p) >= os.path.getmtime(s) for s in srcs):
        return p
    from torch.utils.cpp_extension import load
    load(name=NAME, sources=srcs, build_directory=OUT, verbose=verbose,
         extra_cflags=['-O2'])
    return built_path()


def load_ref():
    """Import the prebuilt module (works on the GPU box, where /root/reference is absent)."""
    p = built_path() or build()
    if p is None:
        return None
    import importlib.util
    import torch  # noqa: F401  (the extension links against libtorch)
    spec = importlib.util.spec_from_file_location(NAME, p)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


if __name__ == '__main__':
    print(build(verbose='-v' in sys.argv))
