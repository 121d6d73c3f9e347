"""Largest individual kernel launches of the LAST iteration in a rocprofv3 --kernel-trace CSV (iteration = from the last launch of the
kernel named by argv[2] on).  Usage: python tools/timing/top_launches.py <kernel_trace.csv> <first-kernel-substring> [top N]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
key = sys.argv[2]
top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
idx = [i for i, r in enumerate(rows) if key in r['Kernel_Name']]
it = rows[idx[-1]:]
tot = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in it)
print('%d launches, %.1f us of kernel time' % (len(it), tot / 1e3))
for r in sorted(it, key=lambda r: int(r['Start_Timestamp']) - int(r['End_Timestamp']))[:top]:
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    print('%8.1f us  grid %-18s %s' % (d, '%sx%sx%s' % (r.get('Grid_Size_X', '?'), r.get('Grid_Size_Y', '?'), r.get('Grid_Size_Z', '?')),
                                      r['Kernel_Name'].replace('(anonymous namespace)::', '')[:110]))
