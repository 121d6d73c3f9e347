"""profiles/rNN_precise_pmc_hbm_traffic_raw.json (per kernel name, per launch; tools/pmc_traffic.py) -> profiles/pmc_traffic_precise.json
keyed by the kernel-class names bench.py's precise_breakdown() reports: launch-weighted averages where a class spans several
template instances; the head class = the SUM over its launches of one forward.   python tools/pmc_to_bench_p2.py <raw.json> <out.json>"""
import json, sys
raw = json.load(open(sys.argv[1]))


def wavg(pred):
    ks = [k for k in raw if pred(k) and 'hbm_bytes_per_launch' in raw[k]]
    n = sum(raw[k]['launches_sampled'] for k in ks)
    return round(sum(raw[k]['hbm_bytes_per_launch'] * raw[k]['launches_sampled'] for k in ks) / n) if n else None


stem = [k for k in raw if k.startswith('k_pl_stem2x')]
fwd = sum(raw[k]['launches_sampled'] for k in stem) or None
head = [k for k in raw if k.startswith('k_pl_head') or k.startswith('k_pl_conv_ml')]
out = {
    'whole stem: conv3x3 s2 (3->64) + 1x1 + conv3x3 s2 + 1x1, pair-1 output never in HBM (k_pl_stem2xs: row stream, producer + consumer waves)': wavg(lambda k: k.startswith('k_pl_stem2x')),
    'conv3x3 s1 64->64 (+ residual) (k_pl_c3p)': wavg(lambda k: k.startswith('k_pl_c3')),
    'stage entry: conv3x3 s2 + 1x1 s2 identity branch (k_pl_conv<.,3,2,DS>)': wavg(lambda k: k.startswith('k_pl_conv<64, 3, 2') or k.startswith('k_pl_conv<128, 3, 2')),
    'conv3x3 s1 128->128 (k_pl_conv<128,3,1>)': wavg(lambda k: k.startswith('k_pl_conv<128, 3, 1')),
}
if fwd and head:
    tot = round(sum(raw[k]['hbm_bytes_per_launch'] * raw[k]['launches_sampled'] for k in head if 'hbm_bytes_per_launch' in raw[k]) / fwd)
    out['neck + head 1x1 convs of all pyramid levels, GroupNorm in the consumer (k_pl_head / k_pl_head_out: flat tiles, fp32 intermediates)'] = tot
    out['neck + head 1x1 convs of all pyramid levels, GroupNorm in the consumer (k_pl_conv_ml)'] = tot
out['_per_kernel'] = {k: v.get('hbm_bytes_per_launch') for k, v in raw.items() if k.startswith('k_pl_')}
out['_note'] = ('HBM bytes per launch = 2 x FETCH_SIZE (gfx950 half-count correction, MI355X_MICROARCH.md) + WRITE_SIZE, separate rocprofv3 --pmc '
                'passes of tools/bench_precise.py (eager launches, 8 x 1080p), averaged over the launches of each kernel class; the head '
                'entry is the sum over the head launches of ONE forward; raw per-kernel numbers in ' + sys.argv[1].split('/')[-1])
json.dump(out, open(sys.argv[2], 'w'), indent=1)
print(json.dumps(out, indent=1))
