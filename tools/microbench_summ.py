import json,collections,sys
d=collections.defaultdict(dict)
for l in open(sys.argv[1]):
    r=json.loads(l); d[(r["operands"],r["waves_per_simd"])][r["round"]]=r["cycles_per_mad_at_nominal_clock"]
for k,v in d.items(): print(k, [v[i] for i in sorted(v)])
