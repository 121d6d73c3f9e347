"""Print the kernel timeline (start offset, gap to the previous kernel's end, duration) of the LAST forward in a rocprofv3
--kernel-trace CSV.  Usage: python tools/timing/timeline.py <kernel_trace.csv> [first-kernel-substring, default k_stem2x]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
key = sys.argv[2] if len(sys.argv) > 2 else 'k_stem2x'
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if key in r['Kernel_Name']]
i0 = idx[-1]
t0 = int(rows[i0]['Start_Timestamp'])
prev_end = t0
for r in rows[i0:]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    name = r['Kernel_Name'].replace('(anonymous namespace)::', '')
    print('%8.1f gap %5.1f dur %6.1f  %s' % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3, name[:90]))
    prev_end = e
print('total %.1f us' % ((prev_end - t0) / 1e3))
