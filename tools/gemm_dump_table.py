#!/usr/bin/env python3
"""Group a PDMK_DUMP_GEMM jsonl (one record per GEMM launch of a main step) by (kind, M, N, K, splitk) and print
the shapes by total time."""
import json, sys, collections
rows = [json.loads(l) for l in open(sys.argv[1])]
agg = collections.OrderedDict()
for r in rows:
    shp = r["mnk_sk"]
    key = (tuple(r["kind"]), tuple(tuple(m) for m in shp) if shp and isinstance(shp[0], list) else tuple(shp))
    a = agg.setdefault(key, [0, 0.0, 0.0])
    a[0] += 1; a[1] += r["ms"]; a[2] += r["flops"]
tot = sum(a[1] for a in agg.values())
print(f"total {tot:.2f} ms, {sum(a[2] for a in agg.values())/tot/1e9:.1f} TFLOP/s, {len(rows)} launches")
cum = 0
for key, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[: int(sys.argv[2]) if len(sys.argv) > 2 else 40]:
    cum += a[1]
    print(f"{str(key[0]):28s} {str(key[1])[:110]:34s} n={a[0]:3d} ms={a[1]:7.3f} ({a[1]/a[0]*1e3:7.1f} us each) {a[2]/a[1]/1e9:7.1f} TF/s  cum {cum/tot*100:5.1f}%")
