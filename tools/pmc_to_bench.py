"""profiles/r01_pmc_hbm_traffic_raw.json (per kernel class, per launch) -> profiles/pmc_traffic.json keyed by the
kernel-class names bench.py reports (launch-weighted averages where a bench class spans several template instances)."""
import json, sys
raw = json.load(open(sys.argv[1]))
def wavg(keys):
    n = sum(raw[k]['launches_sampled'] for k in keys if k in raw)
    return round(sum(raw[k]['hbm_bytes_per_launch'] * raw[k]['launches_sampled'] for k in keys if k in raw) / n) if n else None
stem = [k for k in raw if k.startswith('k_stem2x')]      # k_stem2x<false> (fp16 frames) / k_stem2x<true> (uint8 frames)
fwd = sum(raw[k]['launches_sampled'] for k in stem) or None
out = {
    'conv3x3_s1_64to64 (k_conv)': wavg(['k_conv<cin=64,k=3,s=1,nct=2>', 'k_conv<cin=64,k=3,s=1,nct=2,res>']),
    'whole faster-stem fused: 3x3s2+1x1+3x3s2+1x1 (k_stem2x)': wavg(stem),
    'conv3x3_s2_64to64+downsample1x1s2 (k_conv)': wavg(['k_conv<cin=64,k=3,s=2,nct=2,ds>']),
    'conv3x3_s2_64to128+downsample1x1s2 (k_conv)': wavg(['k_conv<cin=64,k=3,s=2,nct=4,ds>']),
    'conv3x3_s1_128to128 (k_conv)': wavg(['k_conv<cin=128,k=3,s=1,nct=4>', 'k_conv<cin=128,k=3,s=1,nct=4,res>']),
}
if fwd:   # the head is one bench "launch" = all k_head / k_gn_finalize launches of a forward
    out['neck+head 3-pass GN recompute (k_head2 x3 + k_gn_finalize x2)'] = round(sum(
        v['hbm_bytes_per_launch'] * v['launches_sampled'] for k, v in raw.items() if k.startswith('k_head') or k == 'k_gn_finalize') / fwd)
out['_note'] = ('HBM bytes per launch = 2 x FETCH_SIZE (gfx950 half-count correction, MI355X_MICROARCH.md) + WRITE_SIZE, separate --pmc '
                'passes of `bench.py --steps 4 --no-graph`, averaged over the launches of each kernel class; raw per-kernel numbers in '
                'r01_pmc_hbm_traffic_raw.json (tools/collect_profiles.sh, tools/pmc_traffic.py, tools/pmc_to_bench.py)')
json.dump(out, open(sys.argv[2], 'w'), indent=1)
print(json.dumps(out, indent=1))
