"""Summarise a rocprofv3 rocpd sqlite DB: per-kernel stats + GPU idle fraction in the steady state."""
import sqlite3, sys, collections
db = sys.argv[1]
c = sqlite3.connect(db)
rows = c.execute("select name, start, end from kernels order by start").fetchall()
if not rows: sys.exit('no kernels')
# steady state: last 60 % of dispatches
rows2 = rows[int(len(rows)*0.4):]
agg = collections.defaultdict(lambda: [0, 0.0])
for n, s, e in rows2:
    agg[n][0] += 1; agg[n][1] += (e - s) / 1e3
span = (rows2[-1][2] - rows2[0][1]) / 1e3
busy = sum(v[1] for v in agg.values())
print('steady-state window: %.1f us span, %.1f us kernel time (%.1f%% busy), %d dispatches' % (span, busy, 100*busy/span, len(rows2)))
for n, (cnt, tot) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[2]) if len(sys.argv) > 2 else 25]:
    print('%7.1f us avg  x%5d  %5.1f%%  %s' % (tot/cnt, cnt, 100*tot/busy, n[:110]))
