"""lfd_amd -- MI355X-native (gfx950) implementation of the LFD dense-prediction hot path.

Host-side mirror of the reference's operator interface (`LFDResNet`, `SimpleNeck`, `LFDHead`,
`LFD`, `FocalLoss`, `IoULoss`, `CrossEntropyLoss`, `nms`/`batched_nms`/`multiclass_nms`, and the
`nms_ext` / `sigmoid_focal_loss_ext` extension-module surfaces) over the C-ABI library
`liblfd_hip.so` (include/lfd_hip.h) whose kernels are hand-written HIP for CDNA4.
"""
__version__ = '0.1.0'
