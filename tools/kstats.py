"""First rows of a rocprofv3 kernel_stats.csv: python tools/kstats.py <csv> [rows] [name filter]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 15
flt = sys.argv[3] if len(sys.argv) > 3 else ''
for r in rows:
    if flt and flt not in r['Name']:
        continue
    print('%-84s calls %5s avg %9.1f us  %5s %%' % (r['Name'][:84], r['Calls'], float(r['AverageNs']) / 1e3, r['Percentage']))
    n -= 1
    if n <= 0:
        break
