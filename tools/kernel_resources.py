"""Registers, LDS, scratch and occupancy of every kernel in csrc/, as the compiler reports them
(hipcc -Rpass-analysis=kernel-resource-usage, gfx950) -> profiles/<tag>_kernel_resources.txt.  No GPU needed.

    python tools/kernel_resources.py [tag]
"""
import glob
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'lfd-a-light-and-fast-detector_amd', 'csrc')


def demangle(names):
    out = subprocess.run(['c++filt'], input='\n'.join(names), capture_output=True, text=True).stdout.split('\n')
    return [re.sub(r'\(anonymous namespace\)::', '', o) for o in out]


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else 'r03'
    rows = []
    # the flags the library is built with: csrc/Makefile's FLAGS plus the per-file FLAGS_<name> (e.g. -amdgpu-mfma-vgpr-form=1 for the
    # one-wave-per-SIMD kernels: without it the fused stem reports 164 B of scratch that the shipped object does not have)
    mk = open(os.path.join(CSRC, 'Makefile')).read()
    per_file = {m.group(1): m.group(2).split() for m in re.finditer(r'^FLAGS_(\w+)\s*=\s*(.*)$', mk, re.M)}
    for src in sorted(glob.glob(os.path.join(CSRC, '*.hip'))):
        base = os.path.splitext(os.path.basename(src))[0]
        r = subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-ffp-contract=off', '-fno-fast-math',
                            '-DLFD_BUILDING', '-I' + os.path.join(ROOT, 'include')] + per_file.get(base, []) +
                           ['-c', src, '-o', '/dev/null', '-Rpass-analysis=kernel-resource-usage'], capture_output=True, text=True)
        cur = None
        for line in r.stderr.split('\n'):
            m = re.search(r'remark: (?:\[[^\]]*\] )?\s*(Function Name|SGPRs|TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs Spill|VGPRs Spill|LDS Size \[bytes/block\]): (\S+)', line)
            if not m:
                continue
            k, v = m.group(1), m.group(2)
            if k == 'Function Name':
                cur = dict(file=os.path.basename(src), name=v)
                rows.append(cur)
            elif cur is not None:
                cur[k] = v
    names = demangle([r_['name'] for r_ in rows])
    lines = ['# compiler-reported resources of every kernel, compiled with the flags of csrc/Makefile incl. the per-file ones (-Rpass-analysis=kernel-resource-usage; tools/kernel_resources.py)',
             '# file | kernel | VGPRs | AGPRs | SGPRs | VGPR spill | scratch B/lane | LDS B/block (static) | occupancy waves/SIMD']
    for r_, n in zip(rows, names):
        n = re.sub(r'\(.*', '', n) if len(n) > 110 else n
        lines.append('%-16s | %-100s | %4s | %4s | %4s | %3s | %4s | %6s | %s' % (
            r_['file'], n[:100], r_.get('VGPRs', '?'), r_.get('AGPRs', '?'), r_.get('TotalSGPRs', r_.get('SGPRs', '?')), r_.get('VGPRs Spill', '?'),
            r_.get('ScratchSize [bytes/lane]', '?'), r_.get('LDS Size [bytes/block]', '?'), r_.get('Occupancy [waves/SIMD]', '?')))
    out = os.path.join(ROOT, 'profiles', '%s_kernel_resources.txt' % tag)
    open(out, 'w').write('\n'.join(lines) + '\n')
    print(out, len(rows), 'kernels')
    spills = [l for l in lines[2:] if l.split('|')[5].strip() not in ('0', '?') or l.split('|')[6].strip() not in ('0', '?')]
    print('kernels with VGPR spills or scratch:', len(spills))
    for l in spills:
        print(l)


if __name__ == '__main__':
    main()
