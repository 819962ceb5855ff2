#!/usr/bin/env python3
"""usage: traffic_probe_sum.py <probe stdout> <fetch_dir> <write_dir>: per variant of tools/traffic_probe.py the HBM bytes per launch."""
import csv, glob, re, sys
variants = []
for l in open(sys.argv[1]):
    if l.startswith("VARIANT"):
        head, kern = l.split(" kernel=", 1)
        v = dict(kv.split("=", 1) for kv in head.split()[1:])
        v["kernel"] = kern.strip()
        variants.append(v)
def load(d):
    rows = []
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        rows += [r for r in csv.DictReader(open(f)) if ("igemm_ring" in r["Kernel_Name"] or "rowblock" in r["Kernel_Name"] or "igemm_kernel" in r["Kernel_Name"] or "conv_halo" in r["Kernel_Name"])]
    rows.sort(key=lambda r: int(r.get("Dispatch_Id", r.get("Dispatch_ID", 0))))
    return rows
fe, wr = load(sys.argv[2]), load(sys.argv[3])
pos = 0
print(f"{'variant':58s} {'read MB':>8s} {'alg':>6s} {'x':>5s} | {'write MB':>8s} {'alg':>6s} {'x':>5s}")
for v in variants:
    n = int(v["reps"])
    f = sum(float(r["Counter_Value"]) for r in fe[pos:pos + n]) / n * 1024 * 2
    w = sum(float(r["Counter_Value"]) for r in wr[pos:pos + n]) / n * 1024
    pos += n
    ar, aw = int(v["alg_read"]), int(v["alg_write"])
    print(f"c{v['cand']:>2s} {v['M']:>6s}x{v['N']:>4s}x{v['K']:>4s} {v['mode']:9s} {v['kernel'][:28]:28s} {f / 1e6:8.1f} {ar / 1e6:6.1f} {f / ar:5.2f} | {w / 1e6:8.1f} {aw / 1e6:6.1f} {w / aw:5.2f}")
