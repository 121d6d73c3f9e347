"""Named model configurations = the model sections of the reference's config scripts
(`prepare_model()` in WIDERFACE_train/WIDERFACE_LFD_{L,M,S,XS}.py:76-158,
TT100K_train/TT100K_LFD_{L,S}.py:76-157 and TrafficLight_train/TL_LFD_{L,S}.py:76-150), expressed as plain kwargs so the same dict drives
this package's modules, the oracle (oracle/net_oracle.py `arch`) and the reference modules
(tests/golden/make_golden.py).
"""
import torch

WIDERFACE_RANGES = ((4, 20), (20, 40), (40, 80), (80, 160), (160, 320))      # WIDERFACE_LFD_S.py:134
TT100K_RANGES = ((4, 32), (32, 64), (64, 128), (128, 256))                   # TT100K_LFD_L.py:131


def _wf(stem_mode, stem_channels, body_architecture, body_channels, out_indices):
    return dict(block_mode='faster', stem_mode=stem_mode, stem_channels=stem_channels,
                body_architecture=body_architecture, body_channels=body_channels, out_indices=out_indices,
                num_neck_channels=128, num_classes=1, num_head_channels=128, num_conv_layers=2, conv_kernel_size=1,
                gn_groups=16, share_head_flag=True, merge_path_flag=True, classification_loss_type='FocalLoss',
                regression_loss_type='IoULoss', regression_ranges=WIDERFACE_RANGES, gray_range_factors=(0.9, 1.1),
                range_assign_mode='dist', distance_to_bbox_mode='sigmoid')


def _tt(stem_mode, body_architecture, body_channels, out_indices):
    return dict(block_mode='faster', stem_mode=stem_mode, stem_channels=64, body_architecture=body_architecture,
                body_channels=body_channels, out_indices=out_indices, num_neck_channels=128, num_classes=45,
                num_head_channels=128, num_conv_layers=2, conv_kernel_size=1, gn_groups=16, share_head_flag=True,
                merge_path_flag=False, classification_loss_type='CrossEntropyLoss', regression_loss_type='IoULoss',
                regression_ranges=TT100K_RANGES, gray_range_factors=(0.9, 1.1), range_assign_mode='longer',
                distance_to_bbox_mode='sigmoid')


def _tl(stem_channels, body_architecture, body_channels, out_indices, ranges):
    """TrafficLight_train/TL_LFD_{L,S}.py:76-150: LFD with a norm-free shared head and QualityFocalLoss (loss_weight 2)."""
    return dict(block_mode='faster', stem_mode='fast', stem_channels=stem_channels, body_architecture=body_architecture,
                body_channels=body_channels, out_indices=out_indices, num_neck_channels=128, num_classes=1,
                num_head_channels=128, num_conv_layers=2, conv_kernel_size=1, gn_groups=None, share_head_flag=True,
                merge_path_flag=True, classification_loss_type='QualityFocalLoss', regression_loss_type='IoULoss',
                regression_ranges=ranges, gray_range_factors=(0.9, 1.1), range_assign_mode='dist',
                distance_to_bbox_mode='sigmoid')


ARCHS = {
    'WIDERFACE_LFD_L': _wf('fast', 64, [4, 2, 2, 1, 1], [64, 64, 64, 128, 128], ((0, 3), (1, 1), (2, 1), (3, 0), (4, 0))),
    'WIDERFACE_LFD_M': _wf('fast', 64, [3, 2, 1, 1, 1], [64, 64, 64, 128, 128], ((0, 2), (1, 1), (2, 0), (3, 0), (4, 0))),
    'WIDERFACE_LFD_S': _wf('faster', 64, [4, 2, 2, 3], [64, 64, 64, 128], ((0, 3), (1, 1), (2, 1), (3, 0), (3, 2))),
    'WIDERFACE_LFD_XS': _wf('faster', 32, [4, 2, 2, 3], [64, 64, 64, 64], ((0, 3), (1, 1), (2, 1), (3, 0), (3, 2))),
    'TT100K_LFD_L': _tt('fast', [5, 3, 2, 2], [64, 64, 128, 128], ((0, 4), (1, 2), (2, 1), (3, 1))),
    'TT100K_LFD_S': _tt('faster', [4, 2, 1, 1], [64, 64, 64, 128], ((0, 3), (1, 1), (2, 0), (3, 0))),
    'TL_LFD_L': _tl(64, [5, 3, 2, 2, 2], [64, 64, 128, 128, 128], ((0, 4), (1, 2), (2, 1), (3, 1), (4, 1)),
                    ((4, 32), (32, 64), (64, 128), (128, 256), (256, 512))),
    # 48-channel stem / first stage: the inference engine runs them zero-padded to 64 channels (engine.pad_channels: the real
    # channels are bit-identical to an unpadded evaluation); training goes through PyTorch-ROCm autograd for this config
    'TL_LFD_S': _tl(48, [4, 2, 1, 1, 1], [48, 64, 64, 128, 128], ((0, 3), (1, 1), (2, 0), (3, 0), (4, 0)),
                    ((0, 16), (16, 32), (32, 64), (64, 128), (128, 256))),
}


def build_modules(arch, backbone_cls, neck_cls, head_cls, lfd_cls, focal_cls, iou_cls, ce_cls, seed=666, qfl_cls=None):
    """Instantiates (backbone, neck, head, LFD) from `arch` with the given classes (this package's
    or the reference's -- identical kwargs), under torch.manual_seed(seed) (= the config seed,
    WIDERFACE_LFD_S.py:51)."""
    torch.manual_seed(seed)
    if arch['classification_loss_type'] == 'CrossEntropyLoss':
        cls_loss = ce_cls(reduction='mean', loss_weight=1.0)
    elif arch['classification_loss_type'] == 'QualityFocalLoss':
        cls_loss = qfl_cls(use_sigmoid=True, beta=2.0, reduction='mean', loss_weight=2.0)       # TL_LFD_L.py:79-84
    else:
        cls_loss = focal_cls(use_sigmoid=True, gamma=2.0, alpha=0.25, reduction='mean', loss_weight=1.0)
    reg_loss = iou_cls(eps=1e-6, reduction='mean', loss_weight=1.0)
    bb = backbone_cls(block_mode=arch['block_mode'], stem_mode=arch['stem_mode'], body_mode=None, input_channels=3,
                      stem_channels=arch['stem_channels'], body_architecture=list(arch['body_architecture']),
                      body_channels=list(arch['body_channels']), out_indices=arch['out_indices'], frozen_stages=-1,
                      activation_cfg=dict(type='ReLU', inplace=True), norm_cfg=dict(type='BatchNorm2d'),
                      init_with_weight_file=None, norm_eval=False)
    neck = neck_cls(num_neck_channels=arch['num_neck_channels'], num_input_channels_list=bb.num_output_channels_list,
                    num_input_strides_list=bb.num_output_strides_list, norm_cfg=dict(type='BatchNorm2d'),
                    activation_cfg=dict(type='ReLU', inplace=True))
    head = head_cls(num_classes=arch['num_classes'], num_heads=len(neck.num_output_strides_list),
                    num_input_channels=arch['num_neck_channels'], num_head_channels=arch['num_head_channels'],
                    num_conv_layers=arch['num_conv_layers'], activation_cfg=dict(type='ReLU', inplace=True),
                    norm_cfg=dict(type='GroupNorm', num_groups=arch['gn_groups']) if arch['gn_groups'] else None,
                    share_head_flag=arch['share_head_flag'], merge_path_flag=arch['merge_path_flag'],
                    classification_loss_type=type(cls_loss).__name__, regression_loss_type=type(reg_loss).__name__)
    model = lfd_cls(backbone=bb, neck=neck, head=head, num_classes=arch['num_classes'],
                    regression_ranges=arch['regression_ranges'], gray_range_factors=arch['gray_range_factors'],
                    range_assign_mode=arch['range_assign_mode'], point_strides=neck.num_output_strides_list,
                    classification_loss_func=cls_loss, regression_loss_func=reg_loss,
                    distance_to_bbox_mode=arch['distance_to_bbox_mode'])
    return model


def build_model(name_or_arch, seed=666):
    """This package's LFD for a named configuration."""
    from .model.backbone import LFDResNet
    from .model.head import LFDHead
    from .model.lfd import LFD
    from .model.losses import CrossEntropyLoss, FocalLoss, IoULoss, QualityFocalLoss
    from .model.neck import SimpleNeck
    arch = ARCHS[name_or_arch] if isinstance(name_or_arch, str) else name_or_arch
    return build_modules(arch, LFDResNet, SimpleNeck, LFDHead, LFD, FocalLoss, IoULoss, CrossEntropyLoss, seed, QualityFocalLoss)


def perturb_weights(model, seed=1):
    """Synthetic, non-degenerate weights (there are no downloadable checkpoints offline): fresh
    init has gamma=1, beta=0, mean=0, var=1 and head std 0.01, which would hide BN-fold /
    GroupNorm / Scale bugs and give near-constant logits.  Deterministic given the state_dict key
    order: BN running_mean~N(0,.1), running_var~U(.5,1.5); every norm weight~U(.5,1.5),
    bias~N(0,.1); Scale~U(.8,1.2); head conv weights re-drawn with std 0.1 (0.03 for norm-free heads)."""
    g = torch.Generator().manual_seed(seed)
    sd = model.state_dict()
    seen = set()
    # a norm-free head (TrafficLight configs) has nothing that re-normalises its towers: std 0.1 would give |logits| ~ 100
    head_std = 0.1 if any(k.startswith('_head.') and v.dim() == 1 and k.endswith('.weight') for k, v in sd.items()) else 0.03
    with torch.no_grad():
        for k, v in sd.items():
            if v.data_ptr() in seen:     # shared head: duplicated keys alias one tensor
                continue
            seen.add(v.data_ptr())
            if k.endswith('num_batches_tracked'):
                continue
            if k.endswith('running_mean'):
                v.copy_(torch.randn(v.shape, generator=g) * 0.1)
            elif k.endswith('running_var'):
                v.copy_(torch.rand(v.shape, generator=g) + 0.5)
            elif k.endswith('_scale'):
                v.copy_(torch.rand(v.shape, generator=g) * 0.4 + 0.8)
            elif v.dim() == 1 and k.endswith('.weight'):
                v.copy_(torch.rand(v.shape, generator=g) + 0.5)
            elif v.dim() == 1 and k.endswith('.bias'):
                v.copy_(torch.randn(v.shape, generator=g) * 0.1)
            elif v.dim() == 4 and k.startswith('_head.'):
                v.copy_(torch.randn(v.shape, generator=g) * head_std)
    return model


def synthetic_weights(model, seed=1):
    """Deterministic, non-degenerate weights drawn PER state_dict KEY (generator seeded by crc32(key) ^ seed): two
    implementations of the same architecture -- this package's modules and the reference's -- get identical tensors as
    long as their key names and shapes agree, whatever their construction order.  Used for the sibling meta-architectures
    (FCOS / LFDv2 fixtures, tests/golden/make_golden_siblings.py), whose modules are not built by build_modules.
    conv weights ~ N(0, 1/fan_in) (output convs, < 16 filters: half of that), norm gamma ~ U(.5, 1.5), biases / running
    means ~ N(0, .1), running_var ~ U(.5, 1.5), Scale ~ U(.8, 1.2)."""
    import zlib
    sd = model.state_dict()
    seen = set()
    with torch.no_grad():
        for k in sorted(sd):
            v = sd[k]
            if v.data_ptr() in seen or k.endswith('num_batches_tracked'):
                continue
            seen.add(v.data_ptr())
            g = torch.Generator().manual_seed((zlib.crc32(k.encode()) ^ seed) & 0x7fffffff)
            if k.endswith('running_var'):
                v.copy_(torch.rand(v.shape, generator=g) + 0.5)
            elif k.endswith('_scale'):
                v.copy_(torch.rand(v.shape, generator=g) * 0.4 + 0.8)
            elif v.dim() == 4:
                fan_in = v.shape[1] * v.shape[2] * v.shape[3]
                v.copy_(torch.randn(v.shape, generator=g) * ((0.5 if v.shape[0] < 16 else 1.0) / fan_in ** 0.5))
            elif v.dim() == 1 and k.endswith('.weight'):
                v.copy_(torch.rand(v.shape, generator=g) + 0.5)
            else:
                v.copy_(torch.randn(v.shape, generator=g) * 0.1)
    return model


# ------------------------------------------------------------------ sibling meta-architectures
# No shipped config uses FCOS / LFDv2 / FPN / SimpleFPN; these small compositions exercise them (tests, fixtures, tools).
SIBLING_BACKBONE = dict(block_mode='faster', stem_mode='fast', stem_channels=64, body_architecture=[2, 1, 1],
                        body_channels=[64, 64, 128], out_indices=((0, 1), (1, 0), (2, 0)))

SIBLINGS = {
    # FCOS: FPN with two conv extra levels (ReLU in front), GroupNorm FCOSHead
    'FCOS_FPN': dict(meta='FCOS', backbone=SIBLING_BACKBONE,
                     neck=dict(kind='FPN', num_output_channels=128, num_outputs=5, extra_on_input=False, extra_type='conv',
                               norm_on_lateral=False, relu_on_lateral=False, relu_before_extra=True, norm_cfg=None),
                     head=dict(kind='FCOSHead', num_classes=3, num_head_channels=128, num_layers=2,
                               norm_cfg=dict(type='GroupNorm', num_groups=16)),
                     regress_ranges=((0, 32), (32, 64), (64, 128), (128, 256), (256, 1e8)), pre_nms_bbox_limit=100,
                     post_nms_bbox_limit=20),
    # LFDv2 over SimpleFPN (BN + ReLU laterals, pooled extra level) and an LFDHead of 3x3 convs, separate towers, 64 channels
    'LFDV2_SFPN': dict(meta='LFDv2', backbone=SIBLING_BACKBONE,
                       neck=dict(kind='SimpleFPN', num_output_channels=64, num_outputs=4, extra_on_input=False,
                                 extra_type='pooling', norm_on_lateral=True, relu_on_lateral=True, relu_before_extra=True,
                                 norm_cfg=dict(type='BatchNorm2d'), neighbouring_mode=False),
                       head=dict(kind='LFDHead', num_classes=4, num_head_channels=64, num_conv_layers=2, conv_kernel_size=3,
                                 norm_cfg=dict(type='GroupNorm', num_groups=8), share_head_flag=True, merge_path_flag=False),
                       classification_loss_type='FocalLoss', regression_loss_type='GIoULoss',
                       regression_ranges=((4, 32), (32, 64), (64, 128), (128, 256)), gray_range_factors=(0.9, 1.1),
                       range_assign_mode='sqrt', distance_to_bbox_mode='exp', pre_nms_bbox_limit=150, post_nms_bbox_limit=30),
    # LFDv2 on the modules the fused LFD plan covers (SimpleNeck + 1x1 merged head): softmax scores, 'sigmoid' decode
    'LFDV2_SIMPLE': dict(meta='LFDv2', backbone=SIBLING_BACKBONE,
                         neck=dict(kind='SimpleNeck', num_neck_channels=128),
                         head=dict(kind='LFDHead', num_classes=5, num_head_channels=128, num_conv_layers=2, conv_kernel_size=1,
                                   norm_cfg=dict(type='GroupNorm', num_groups=16), share_head_flag=True, merge_path_flag=True),
                         classification_loss_type='CrossEntropyLoss', regression_loss_type='IoULoss',
                         regression_ranges=((4, 32), (32, 64), (64, 128)), gray_range_factors=(0.9, 1.1),
                         range_assign_mode='longer', distance_to_bbox_mode='sigmoid', pre_nms_bbox_limit=120,
                         post_nms_bbox_limit=25),
    # LFDv2 over SimpleNeck + LFDHeadV1 (per-level output convs outside shared BatchNorm towers): 'exp' decode
    'LFDV2_HEADV1': dict(meta='LFDv2', backbone=SIBLING_BACKBONE,
                         neck=dict(kind='SimpleNeck', num_neck_channels=64),
                         head=dict(kind='LFDHeadV1', num_classes=2, num_head_channels=64, num_conv_layers=2,
                                   norm_cfg=dict(type='BatchNorm2d'), share_head_flag=True, merge_path_flag=True),
                         classification_loss_type='FocalLoss', regression_loss_type='IoULoss',
                         regression_ranges=((4, 32), (32, 64), (64, 128)), gray_range_factors=(0.9, 1.1),
                         range_assign_mode='dist', distance_to_bbox_mode='exp', pre_nms_bbox_limit=200,
                         post_nms_bbox_limit=40),
}


def build_sibling(spec, B, N, H, M, L, seed=1):
    """Instantiate a SIBLINGS entry from module namespaces B (backbone), N (neck), H (head), M (meta-architectures) and L
    (losses) -- this package's or the reference's (identical kwargs) -- and give it synthetic_weights(seed)."""
    spec = SIBLINGS[spec] if isinstance(spec, str) else spec
    bbk = spec['backbone']
    bb = B.LFDResNet(block_mode=bbk['block_mode'], stem_mode=bbk['stem_mode'], body_mode=None, input_channels=3,
                     stem_channels=bbk['stem_channels'], body_architecture=list(bbk['body_architecture']),
                     body_channels=list(bbk['body_channels']), out_indices=bbk['out_indices'], frozen_stages=-1,
                     activation_cfg=dict(type='ReLU', inplace=True), norm_cfg=dict(type='BatchNorm2d'),
                     init_with_weight_file=None, norm_eval=False)
    nk = dict(spec['neck'])
    kind = nk.pop('kind')
    if kind == 'SimpleNeck':
        neck = N.SimpleNeck(num_neck_channels=nk['num_neck_channels'], num_input_channels_list=bb.num_output_channels_list,
                            num_input_strides_list=bb.num_output_strides_list, norm_cfg=dict(type='BatchNorm2d'),
                            activation_cfg=dict(type='ReLU', inplace=True))
        cn = nk['num_neck_channels']
    else:
        neck = getattr(N, kind)(num_input_channels_list=list(bb.num_output_channels_list),
                                num_input_strides_list=list(bb.num_output_strides_list), **nk)
        cn = nk['num_output_channels']
    strides = list(neck.num_output_strides_list)
    hk = dict(spec['head'])
    hkind = hk.pop('kind')
    if spec['meta'] in ('FCOS', 'FCOSv1'):
        head = H.FCOSHead(num_input_channels=cn, num_heads=len(strides), **hk)
        model = getattr(M, spec['meta'])(backbone=bb, neck=neck, head=head, num_classes=hk['num_classes'], regress_ranges=spec['regress_ranges'],
                       point_strides=strides,
                       classification_loss_func=L.FocalLoss(use_sigmoid=True, gamma=2.0, alpha=0.25, loss_weight=1.0),
                       regression_loss_func=L.GIoULoss(loss_weight=1.0),
                       centerness_loss_func=L.BCEWithLogitsLoss(reduction='mean', loss_weight=1.0),
                       classification_threshold=0.05, nms_threshold=0.5, pre_nms_bbox_limit=spec['pre_nms_bbox_limit'],
                       post_nms_bbox_limit=spec['post_nms_bbox_limit'])
    else:
        cls_loss = (L.CrossEntropyLoss(reduction='mean', loss_weight=1.0) if spec['classification_loss_type'] == 'CrossEntropyLoss'
                    else L.FocalLoss(use_sigmoid=True, gamma=2.0, alpha=0.25, reduction='mean', loss_weight=1.0))
        reg_loss = getattr(L, spec['regression_loss_type'])(reduction='mean', loss_weight=1.0)
        head = getattr(H, hkind)(num_input_channels=cn, num_heads=len(strides), activation_cfg=dict(type='ReLU', inplace=True),
                                 classification_loss_type=type(cls_loss).__name__,
                                 regression_loss_type=type(reg_loss).__name__, **hk)
        model = M.LFDv2(backbone=bb, neck=neck, head=head, num_classes=hk['num_classes'],
                        regression_ranges=spec['regression_ranges'], gray_range_factors=spec['gray_range_factors'],
                        range_assign_mode=spec['range_assign_mode'], point_strides=strides, classification_loss_func=cls_loss,
                        regression_loss_func=reg_loss, distance_to_bbox_mode=spec['distance_to_bbox_mode'],
                        classification_threshold=0.05, nms_threshold=0.5, pre_nms_bbox_limit=spec['pre_nms_bbox_limit'],
                        post_nms_bbox_limit=spec['post_nms_bbox_limit'])
    return synthetic_weights(model, seed)


def build_sibling_model(name, seed=1):
    """This package's model for a SIBLINGS entry."""
    from .model import backbone as B, head as H, losses as L, neck as N
    from . import model as M
    return build_sibling(name, B, N, H, M, L, seed)
