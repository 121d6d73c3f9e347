"""oracle/ -- CPU restatement of the reference LFD hot path.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package.  The product (lfd-a-light-and-fast-detector_amd/) never does and fails loudly
when its HIP library is missing.

Parity status: PINNED -- against the reference's docstring known-answer vectors and
against outputs of the reference itself (its CPU NMS extension compiled unmodified into
oracle/_ref, and its Python modules imported in the build container to generate
tests/golden/*).  See tests/test_oracle_golden.py and tests/golden/make_golden.py.
"""
from .c_oracle import (nms, batched_nms, multiclass_nms, soft_nms,  # noqa: F401
                       sigmoid_focal_loss_fwd, sigmoid_focal_loss_bwd, iou_loss_fwd,
                       argsort_desc_stable, build as build_c)
