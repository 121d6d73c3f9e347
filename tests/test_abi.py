"""The C-ABI library loads and exports every symbol include/lfd_hip.h declares (no compute
calls: this runs without a GPU)."""
import ctypes
import os
import re

from conftest import ROOT
from lfd_amd import _lib


def _header_symbols():
    src = open(os.path.join(ROOT, 'include', 'lfd_hip.h')).read()
    return sorted(set(re.findall(r'LFD_API\s+[\w\s\*]+?\b(lfd_[a-z0-9_]+)\s*\(', src)))


def test_library_exports_every_declared_symbol():
    syms = _header_symbols()
    assert len(syms) >= 20
    lib = ctypes.CDLL(_lib.LIB_PATH)
    missing = [s for s in syms if not hasattr(lib, s)]
    assert not missing, missing


def test_python_binding_covers_the_header():
    assert sorted(_lib.declared_symbols()) == _header_symbols()


def test_abi_version_and_strings():
    l = _lib.lib()
    src = open(os.path.join(ROOT, 'include', 'lfd_hip.h')).read()
    header_version = int(re.search(r'#define LFD_HIP_ABI_VERSION (\d+)', src).group(1))
    assert l.lfd_hip_abi_version() == header_version == _lib.ABI_VERSION == 3
    # the two self-contained extension files check the same number (a stale build selected through LFD_HIP_LIB would index
    # grown structs with the wrong stride)
    for rel in ('model/utils/libs/nms_ext.py', 'model/losses/libs/sigmoid_focal_loss_ext.py'):
        text = open(os.path.join(os.path.dirname(_lib.__file__), rel)).read()
        assert '_ABI_VERSION = %d' % header_version in text and 'lfd_hip_abi_version() != _ABI_VERSION' in text, rel
    assert l.lfd_hip_status_string(0) == b'ok'
    assert l.lfd_hip_status_string(-2) == b'workspace too small'
    assert l.lfd_hip_build_info().startswith(b'gfx950;')


def test_workspace_queries_are_pure_host_functions():
    l = _lib.lib()
    assert l.lfd_nms_workspace_bytes(0) > 0
    assert l.lfd_nms_workspace_bytes(4096) >= 4096 * 64 * 8     # 64x64-tile bitmask rows
    d = _lib.DetectDesc()
    d.num_levels = 1
    d.level_h[0], d.level_w[0], d.level_stride[0] = 4, 4, 8
    d.num_classes = d.num_cls_channels = 1
    d.max_candidates = 16
    assert l.lfd_detect_workspace_bytes(ctypes.byref(d), 2) > 0
    assert l.lfd_detect_ex_workspace_bytes(ctypes.byref(d), 2) >= l.lfd_detect_workspace_bytes(ctypes.byref(d), 2) + 2 * 16 * 4
    assert l.lfd_conv_packed_weight_halfs(64, 64, 3) == 64 * 64 * 9
    hd = _lib.HeadDesc()
    hd.n, hd.num_levels, hd.num_groups = 2, 2, 16
    hd.level_hw[0], hd.level_hw[1] = 130, 64
    assert l.lfd_head_partial_floats(ctypes.byref(hd)) == 2 * (3 + 1) * 16 * 2


def test_invalid_arguments_are_status_codes_not_crashes():
    l = _lib.lib()
    assert l.lfd_nms_f32(None, -1, 0.5, None, None, None, 0, None) == -1
    assert l.lfd_conv2d_nhwc_f16(None, None, None, None, None, None, None, None, None, None) == -1
    assert l.lfd_sigmoid_focal_loss_fwd(None, None, 4, 0, 2.0, 0.25, None, 0, None) == -1
    # round-2 entry points: argument checks happen on the host, before anything touches a device
    import ctypes as C
    one0 = C.c_int(0)
    assert l.lfd_detect_from_candidates(None, 1, None, None, None, None, None, None, 0, None) == -1
    assert l.lfd_detect_workspace_reset(None, 1, None, 0, None) == -1
    assert l.lfd_head_forward_decode_f16(None, None, None, None, None, None, None, None, None, None, 0, None) == -1
    assert l.lfd_groupnorm_finalize_fold(None, None, None, None, 1e-5, None, None, 1, None) == -1
    assert l.lfd_fasterblock_fused_f16(0, 8, 8, None, None, None, None, None, None, None, None) == -1
    assert l.lfd_fasterblock128_fused_f16(1, 8, 8, None, None, None, None, None, None, None) == -1
    assert l.lfd_downblock_fused_f16(1, 8, 8, None, None, None, None, None, None, None, None, None, None) == -1
    assert l.lfd_downblock_fused_f16(0, 8, 8, C.byref(one0), C.byref(one0), C.byref(one0), C.byref(one0), C.byref(one0), C.byref(one0),
                                     C.byref(one0), C.byref(one0), C.byref(one0), None) == -1       # n < 1
    assert l.lfd_conv2d_bn_stats_nhwc_f16(None, None, None, None, None, None, 1e-5, 0.1, None, None, None, 0, None, None) == -1
    assert l.lfd_stem_conv0_train_fwd_bn_stats(None, 1, 8, 8, 64, None, None, 1e-5, 0.1, None, None, None, 0, None, None) == -1
    sg = (_lib.HeadOutSeg * 1)()
    sg[0].channels, sg[0].row0 = 4, 62                                         # rows 62..65 of 64
    assert l.lfd_head_out_split_f16(C.byref(one0), 1, 4, 4, 0, sg, 1, None) == -1
    sg[0].row0 = 0
    assert l.lfd_head_out_split_f16(C.byref(one0), 1, 4, 3, 0, sg, 1, None) == -1          # the level does not fit the point axis
    assert l.lfd_head_out_split_f16(C.byref(one0), 1, 4, 4, 0, sg, 1, None) == -1          # no destination
    assert l.lfd_head_out_grad_f16(C.byref(one0), 1, 4, 4, 0, sg, 1, 1024.0, None, None, 0, None) == -1
    # misaligned NHWC fp16 tensors are refused on the host (16-byte alignment is part of the ABI's conventions)
    buf = (C.c_char * 64)()
    base = C.addressof(buf)
    al, mis = C.c_void_p((base + 15) & ~15), C.c_void_p(((base + 15) & ~15) + 2)
    nn = C.byref(one0)
    assert l.lfd_fasterblock_fused_f16(1, 8, 8, mis, al, nn, nn, nn, nn, nn, None) == -1
    assert l.lfd_downblock_fused_f16(1, 8, 8, al, mis, nn, nn, nn, nn, nn, nn, nn, None) == -1
    assert l.lfd_conv3x3s2_dgrad_nhwc_f16(1, 8, 8, mis, al, nn, None, None) == -1
    assert l.lfd_conv3x3s2_dgrad_nhwc_f16(1, 8, 8, al, al, nn, None, None) == -1             # dy == dx
    # round-3 entry points: the fp32-storage precision mode and the gated update
    assert l.lfd_p32_conv2d_nhwc_f32(None, None, None, None, None, None, None, None) == -1
    assert l.lfd_p32_groupnorm_relu_f32(None, 1, 16, 128, 16, None, None, 1e-5, 1, None, 0, None) == -1
    pd = _lib.P32ConvDesc(1, 8, 8, 48, 64, 3, 1, 1, -1, 0, 0)
    assert l.lfd_p32_conv2d_nhwc_f32(C.byref(pd), C.byref(one0), C.byref(one0), C.byref(one0), C.byref(one0), None, None, None) == -4   # cin % 32
    pd = _lib.P32ConvDesc(1, 8, 8, 64, 64, 5, 1, 1, -1, 0, 0)
    assert l.lfd_p32_conv2d_nhwc_f32(C.byref(pd), C.byref(one0), C.byref(one0), C.byref(one0), C.byref(one0), None, None, None) == -4   # 5x5
    assert l.lfd_p32_conv_packed_weight_halfs(64, 64, 3) == 2 * 2 * 9 * 2 * 2 * 64 * 8
    assert l.lfd_p32_groupnorm_workspace_bytes(8, 16) == 8 * 64 * 16 * 16
    assert l.lfd_sgd_step_f32(C.byref(one0), C.byref(one0), C.byref(one0), 4, 0.1, 0.9, 0.0, 0.0, 0, 0, None, 1, 1, None) == -1   # clip without a norm
    # sibling meta-architecture entry points
    assert l.lfd_detect_batched_ex(None, None, 1, None, None, None, 0, None, None, None, None, None, None, None, 0, None) == -1
    assert l.lfd_upsample_nearest_add_nhwc_f16(None, None, 1, 4, 4, 2, 2, 64, None) == -1
    assert l.lfd_relu_inplace_f16(None, 64, None) == -1
    assert l.lfd_maxpool3x3s2_nhwc_f16(None, None, 1, 4, 4, 64, None) == -1
    assert l.lfd_pack_level_outputs_f32(None, None, 1, 16, 32, 0, 4, 16, 0, 1.0, 0, None) == -1
    one = C.c_int(0)
    assert l.lfd_upsample_nearest_add_nhwc_f16(C.byref(one), C.byref(one), 1, 4, 4, 2, 2, 12, None) == -4      # channels % 8
    assert l.lfd_pack_level_outputs_f32(C.byref(one), C.byref(one), 1, 16, 32, 30, 4, 16, 0, 1.0, 0, None) == -1   # c0 + count > channels
    assert l.lfd_pack_level_outputs_f32(C.byref(one), C.byref(one), 1, 16, 32, 0, 4, 8, 0, 1.0, 0, None) == -1     # rows beyond total_points
    # the fused head-decode pass covers one sigmoid-scored class: anything else is LFD_ERR_UNSUPPORTED (-4), never a wrong answer
    hd = _lib.HeadDesc()
    hd.n, hd.num_levels, hd.num_groups, hd.head_channels = 1, 1, 16, 128
    hd.level_hw[0], hd.level_cin[0], hd.total_points, hd.cls_channels = 16, 64, 16, 46
    hd.final_reg_rows, hd.final_cls_rows = 0, 46
    dd = _lib.DetectDesc()
    dd.num_levels = 1
    dd.level_h[0], dd.level_w[0], dd.level_stride[0] = 4, 4, 8
    dd.num_classes, dd.num_cls_channels, dd.score_mode, dd.max_candidates = 45, 46, 1, 16
    lv = (_lib.HeadLevelPtrs * 1)()
    assert l.lfd_head_forward_decode_f16(C.byref(hd), lv, None, None, None, None, None, C.byref(dd), None, None, 0, None) == -4


def test_ctypes_struct_mirrors_match_the_header_layout(tmp_path):
    """Every struct of include/lfd_hip.h that crosses the ABI by pointer: sizeof and the offset of every field as gcc sees
    them == the ctypes mirrors in lfd_amd/_lib.py (a drifted mirror would silently scramble descriptors)."""
    import ctypes as C
    import subprocess
    from conftest import ROOT
    from lfd_amd import _lib
    pairs = {'lfd_detect_desc_t': _lib.DetectDesc, 'lfd_conv_desc_t': _lib.ConvDesc,
             'lfd_head_desc_t': _lib.HeadDesc, 'lfd_head_level_ptrs_t': _lib.HeadLevelPtrs, 'lfd_assign_desc_t': _lib.AssignDesc,
             'lfd_loss_desc_t': _lib.LossDesc, 'lfd_pack_job_t': _lib.PackJob, 'lfd_detect_ext_t': _lib.DetectExt,
             'lfd_p32_conv_desc_t': _lib.P32ConvDesc, 'lfd_head_out_seg_t': _lib.HeadOutSeg, 'lfd_head_out_level_t': _lib.HeadOutLevel, 'lfd_bn_bwd_level_t': _lib.BnBwdLevel, 'lfd_bn_fwd_level_t': _lib.BnFwdLevel,
             'lfd_wgrad_job_t': _lib.WgradJob, 'lfd_rowsum_job_t': _lib.RowsumJob,
             'lfd_pl_conv_desc_t': _lib.PlConvDesc, 'lfd_pl_level_t': _lib.PlLevel,
             'lfd_pl_head_desc_t': _lib.PlHeadDesc, 'lfd_pl_head_level_t': _lib.PlHeadLevel}
    header = open(os.path.join(ROOT, 'include', 'lfd_hip.h')).read()
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "lfd_hip.h"', 'int main(void) {']
    for cname, mirror in pairs.items():
        assert cname in header, cname
        lines.append('printf("%s %%zu\\n", sizeof(%s));' % (cname, cname))
        for fname, _ in mirror._fields_:
            cfield = 'in' if fname == 'in_' else fname
            lines.append('printf("%s.%s %%zu\\n", offsetof(%s, %s));' % (cname, fname, cname, cfield))
    lines += ['return 0; }']
    src = tmp_path / 'layout.c'
    src.write_text('\n'.join(lines))
    exe = tmp_path / 'layout'
    subprocess.run(['gcc', '-I', os.path.join(ROOT, 'include'), str(src), '-o', str(exe)], check=True, capture_output=True)
    got = dict(l.split() for l in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines())
    for cname, mirror in pairs.items():
        assert int(got[cname]) == C.sizeof(mirror), cname
        for fname, _ in mirror._fields_:
            assert int(got['%s.%s' % (cname, fname)]) == getattr(mirror, fname).offset, (cname, fname)


def test_tuning_knobs_are_explicit_state_and_the_library_reads_no_environment():
    """include/lfd_hip.h lfd_tuning_set / lfd_tuning_get: the only mutable global state of the library; csrc/ contains no getenv
    ("no hidden state"); lfd_amd/_lib.py translates the LFD_* names of rounds 1-3 at load time."""
    import re
    from conftest import ROOT
    from lfd_amd import _lib
    l = _lib.lib()
    header = open(os.path.join(ROOT, 'include', 'lfd_hip.h')).read()
    keys = dict((m.group(1), int(m.group(2))) for m in re.finditer(r'LFD_TUNE_([A-Z0-9_]+) = (\d+)', header))
    count = keys.pop('COUNT')
    assert keys == _lib.TUNE_KEYS and count == len(keys)
    defaults = {'HEAD2': 1, 'H2_CHUNK': 0, 'H2_AGPR': 1, 'H2_A1': 1, 'STEM2X': 1, 'X2_ALN': 1, 'X2_STAGGER': 0, 'BLOCK_ROWS': -1,
                'ROWS_WGS': 0, 'CONV128_SPLITK': 1, 'CONV0_VALU': 0, 'PL_C3': 2, 'PL_HEAD_OUT_REGS': 1, 'PL_HEAD_ROLES': 1, 'PL_STEM': 1}
    for name, key in keys.items():
        if os.environ.get('LFD_' + name) is None:
            assert l.lfd_tuning_get(key) == defaults[name], name
    prev = _lib.tune('ROWS_WGS', 192)
    assert _lib.tune('ROWS_WGS') == 192
    _lib.tune('ROWS_WGS', prev)
    assert l.lfd_tuning_set(count, 1) == -1 and l.lfd_tuning_set(-1, 1) == -1 and l.lfd_tuning_get(99) == 0
    csrc = os.path.join(ROOT, 'lfd-a-light-and-fast-detector_amd', 'csrc')
    for fn in os.listdir(csrc):
        if fn.endswith(('.hip', '.h')):
            code = re.sub(r'//[^\n]*', '', open(os.path.join(csrc, fn)).read())
            assert 'getenv' not in code, fn
