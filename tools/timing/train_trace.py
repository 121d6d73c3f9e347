"""tools/timing/train_trace.py KERNEL_TRACE.csv -- one training iteration (the last complete graph replay in the trace) in launch
order: per kernel class time, and what the small launches cost: time by duration bucket, idle gaps between consecutive kernels,
overlap (kernels of different streams running at the same time)."""
import csv
import re
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in rows), key=lambda t: t[0])
# an iteration ends with k_sgd; take the last complete one
ends = [i for i, e in enumerate(ev) if 'k_sgd' in e[2]]
assert len(ends) >= 2, 'need two k_sgd launches'
it = ev[ends[-2] + 1:ends[-1] + 1]
t0, t1 = it[0][0], max(e[1] for e in it)


def short(n):
    n = re.sub(r'\(anonymous namespace\)::', '', n)
    n = re.sub(r'^void ', '', n)
    m = re.match(r'([\w:]+(<[^(]*>)?)', n)
    return (m.group(1) if m else n)[:70]


print('iteration: %d launches, wall %.3f ms, sum of kernel durations %.3f ms' % (len(it), (t1 - t0) / 1e6, sum(e[1] - e[0] for e in it) / 1e6))
# busy time (union of intervals) and gaps
busy, cur_s, cur_e, gaps = 0, it[0][0], it[0][1], []
for s, e, _ in it[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        gaps.append(s - cur_e)
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print('device busy (union) %.3f ms; %d idle gaps, total %.3f ms, median %.2f us, max %.1f us' % (
    busy / 1e6, len(gaps), sum(gaps) / 1e6, (sorted(gaps)[len(gaps) // 2] / 1e3) if gaps else 0, (max(gaps) / 1e3) if gaps else 0))
buckets = [(0, 4), (4, 6), (6, 8), (8, 12), (12, 20), (20, 50), (50, 100), (100, 1e9)]
print('launches by duration:')
for lo, hi in buckets:
    sel = [e for e in it if lo * 1e3 <= e[1] - e[0] < hi * 1e3]
    print('  %4g - %-6g us: %4d launches, %.3f ms' % (lo, hi if hi < 1e9 else float('inf'), len(sel), sum(e[1] - e[0] for e in sel) / 1e6))
cls = defaultdict(lambda: [0, 0])
for s, e, n in it:
    c = cls[short(n)]
    c[0] += 1
    c[1] += e - s
print('by kernel:')
for n, (k, t) in sorted(cls.items(), key=lambda kv: -kv[1][1]):
    print('  %-70s %4d  %8.1f us  avg %7.1f' % (n, k, t / 1e3, t / 1e3 / k))
if len(sys.argv) > 2:
    items = [(s, e, short(n)) for s, e, n in it]
    if len(sys.argv) > 3 and sys.argv[3]:
        try:
            for r in csv.DictReader(open(sys.argv[3])):
                s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
                if t0 <= s <= t1:
                    items.append((s, e, 'MEMCPY %s %s bytes' % (r.get('Direction', ''), r.get('Bytes', r.get('Size', '?')))))
        except Exception as ex:
            print('memory copy trace unreadable:', ex)
    items.sort()
    print('launch order (start us, duration us, gap to previous end us):')
    prev = None
    for s, e, n in items:
        print('  %9.1f %8.1f %7.1f  %s' % ((s - t0) / 1e3, (e - s) / 1e3, ((s - prev) / 1e3) if prev else 0.0, n))
        prev = e if prev is None else max(prev, e)
