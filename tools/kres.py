"""Summarise `-Rpass-analysis=kernel-resource-usage` remarks: one line per kernel (name, VGPR, AGPR, scratch, occupancy, LDS).
   hipcc ... -Rpass-analysis=kernel-resource-usage -c x.hip 2>&1 | python tools/kres.py"""
import re, subprocess, sys
rows, cur = [], None
for line in sys.stdin:
    m = re.search(r'remark: (?:\s*)(Function Name|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]|VGPRs Spill): (\S+)', line)
    if not m:
        if 'error' in line or 'warning' in line:
            print(line.rstrip())
        continue
    k, v = m.groups()
    if k == 'Function Name':
        cur = {'name': v}
        rows.append(cur)
    elif cur is not None:
        cur[k.split(' ')[0] if k != 'VGPRs Spill' else 'spill'] = v
names = [r['name'] for r in rows]
try:
    dem = subprocess.run(['/opt/rocm/lib/llvm/bin/llvm-cxxfilt'] + names, capture_output=True, text=True).stdout.splitlines()
except Exception:
    dem = names
for r, d in zip(rows, dem):
    d = re.sub(r'\(anonymous namespace\)::|pl::|\(.*$', '', d).replace('void ', '')
    print('%-70s v%-4s a%-4s scratch %-4s spill %-3s occ %s lds %s' % (d, r.get('VGPRs'), r.get('AGPRs'), r.get('ScratchSize'), r.get('spill'), r.get('Occupancy'), r.get('LDS')))
