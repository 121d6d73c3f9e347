"""Summarise FETCH_SIZE / WRITE_SIZE passes into per-launch HBM bytes per kernel class (profiles/pmc_traffic.json).
MI355X_MICROARCH.md: FETCH_SIZE (KB) reads exactly 1/2 of a wide coalesced stream on gfx950 -> doubled; WRITE_SIZE
is uncalibrated (reported as-is)."""
import sqlite3, sys, json, collections, re
out = {}
raw = collections.defaultdict(lambda: collections.defaultdict(list))
for db in sys.argv[1:3]:
    c = sqlite3.connect(db)
    for name, cname, val, did in c.execute("select kernel_name, counter_name, value, dispatch_id from counters_collection"):
        raw[name][(cname, did)].append(val)
def cls(name):
    m = re.search(r'k_conv<(\d+), (\d), (\d), (\d), (true|false), (true|false), (true|false), (true|false)(?:, (?:true|false))?>', name)
    if m:
        cin, ks, s, nct, wreg, tail, res, ds = m.groups()[:8]
        return 'k_conv<cin=%s,k=%s,s=%s,nct=%s%s%s%s>' % (cin, ks, s, nct, ',tail' if tail == 'true' else '', ',res' if res == 'true' else '', ',ds' if ds == 'true' else '')
    m = re.search(r'(k_[a-z0-9_]+)(<[^>]*>)?', name)
    return (m.group(1) + (m.group(2) or '')) if m else None
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for name, d in raw.items():
    k = cls(name)
    if not k: continue
    per = collections.defaultdict(float)
    for (cname, did), vals in d.items():
        per[(cname, did)] += sum(vals)
    for (cname, did), v in per.items():
        agg[k][cname].append(v)
for k, d in sorted(agg.items()):
    f = d.get('FETCH_SIZE', []); w = d.get('WRITE_SIZE', [])
    e = {'launches_sampled': max(len(f), len(w))}
    if f: e['fetch_bytes_per_launch_corrected_x2'] = round(2 * 1024 * sum(f) / len(f))
    if w: e['write_bytes_per_launch_uncalibrated'] = round(1024 * sum(w) / len(w))
    if f and w: e['hbm_bytes_per_launch'] = e['fetch_bytes_per_launch_corrected_x2'] + e['write_bytes_per_launch_uncalibrated']
    out[k] = e
json.dump(out, open(sys.argv[3], 'w'), indent=1, sort_keys=True)
print(json.dumps(out, indent=1, sort_keys=True)[:3000])
