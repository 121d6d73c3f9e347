"""profiles/r01_pmc_hbm_traffic_raw.json (per kernel class, per launch) -> profiles/pmc_traffic.json keyed by the
kernel-class names bench.py reports (launch-weighted averages where a bench class spans several template instances)."""
import json, sys
import re
raw0 = json.load(open(sys.argv[1]))
raw = {}
for k, v in raw0.items():      # normalise 'k_conv<64, 3, 1, 2, true, false, true, false, false>' -> 'k_conv<cin=64,k=3,s=1,nct=2,res>'
    m = re.match(r'k_conv<(\d+), (\d), (\d), (\d), (true|false), (true|false), (true|false), (true|false)', k)
    if m:
        cin, ks, st, nct, wreg, tail, res, ds = m.groups()
        k = 'k_conv<cin=%s,k=%s,s=%s,nct=%s%s%s%s>' % (cin, ks, st, nct, ',tail' if tail == 'true' else '', ',res' if res == 'true' else '', ',ds' if ds == 'true' else '')
    raw[k] = v
def wavg(keys):
    n = sum(raw[k]['launches_sampled'] for k in keys if k in raw)
    return round(sum(raw[k]['hbm_bytes_per_launch'] * raw[k]['launches_sampled'] for k in keys if k in raw) / n) if n else None
stem = [k for k in raw if k.startswith('k_stem2x')]      # k_stem2x<false> (fp16 frames) / k_stem2x<true> (uint8 frames)
fwd = sum(raw[k]['launches_sampled'] for k in stem) or None
out = {
    'conv3x3_s1_64to64 (k_conv)': wavg(['k_conv<cin=64,k=3,s=1,nct=2>', 'k_conv<cin=64,k=3,s=1,nct=2,res>']),
    'whole faster-stem fused: 3x3s2+1x1+3x3s2+1x1 (k_stem2x)': wavg(stem),
    'fasterblock_fused_2x_conv3x3_s1_64to64 (k_block64_rows on large maps, k_block64 on small ones)': wavg(['k_block64', 'k_block64_rows']),      # tiles on small maps, rows on large ones
    'all conv3x3 s1 64->64 (fused residual blocks k_block64_rows / k_block64 + stand-alone k_conv)': wavg(['k_block64', 'k_block64_rows', 'k_conv<cin=64,k=3,s=1,nct=2>', 'k_conv<cin=64,k=3,s=1,nct=2,res>']),
    'downblock_fused_conv3x3_s2+1x1_s2+conv3x3_s1_64to64 (k_down64)': wavg(['k_down64']),
    'conv3x3_s2_64to64+downsample1x1s2 (k_conv)': wavg(['k_conv<cin=64,k=3,s=2,nct=2,ds>']),
    'conv3x3_s2_64to128+downsample1x1s2 (k_conv)': wavg(['k_conv<cin=64,k=3,s=2,nct=4,ds>']),
    'conv3x3_s1_128to128 (k_conv)': wavg(['k_conv<cin=128,k=3,s=1,nct=4>', 'k_conv<cin=128,k=3,s=1,nct=4,res>']),
    'conv3x3_s1_128to128 small map, split-K (k_conv128_splitk)': wavg(['k_conv128_splitk']),
    'fasterblock128_fused_2x_conv3x3_s1_128to128 small map (k_block128)': wavg(['k_block128']),
}
if fwd:   # the head is one bench "launch" = all k_head / k_gn_finalize launches of a forward
    out['neck+head 3-pass GN recompute (k_head2 x3 + k_gn_finalize x2)'] = round(sum(
        v['hbm_bytes_per_launch'] * v['launches_sampled'] for k, v in raw.items() if k.startswith('k_head') or k == 'k_gn_finalize') / fwd)
out['_note'] = ('HBM bytes per launch = 2 x FETCH_SIZE (gfx950 half-count correction, MI355X_MICROARCH.md) + WRITE_SIZE, separate --pmc '
                'passes of `bench.py --steps 4 --no-graph`, averaged over the launches of each kernel class; raw per-kernel numbers in '
                '' + sys.argv[1].split('/')[-1] + ' (tools/collect_profiles.sh, tools/pmc_traffic.py, tools/pmc_to_bench.py)')
json.dump(out, open(sys.argv[2], 'w'), indent=1)
print(json.dumps(out, indent=1))
