"""Side-by-side per-layer milliseconds of several layer tables (bench.py --layer-table): python tools/lt_diff.py a.json b.json ..."""
import json
import sys

tabs = [json.load(open(p)) for p in sys.argv[1:]]
print('%-22s %-18s' % ('layer', 'shape') + ' '.join('%12s' % p.split('/')[-1][-12:] for p in sys.argv[1:]))
tot = [0.0] * len(tabs)
for i, r in enumerate(tabs[0]['layers']):
    ms = [t['layers'][i]['avg_ms'] for t in tabs]
    for k, m in enumerate(ms):
        tot[k] += m
    print('%-22s %-18s' % (r['layer'], '%s->%d' % (r['in'], r['out'][2])) + ' '.join('%12.3f' % m for m in ms))
print('%-41s' % 'total' + ' '.join('%12.3f' % t for t in tot))
