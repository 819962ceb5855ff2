#!/usr/bin/env python3
"""Turns the two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs, each `--kernel-trace --pmc X --output-format
csv`, plan cache pre-filled so that no tuning launches are profiled) of
`python3 bench.py --steps 1 --warmup 1 --no_graph --no_cpu_baseline --no_roofline` into HBM bytes per launch for every GEMM
kernel symbol.  gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE under-reports wide coalesced reads by
2x -> read bytes = 2 * FETCH_SIZE(KiB) * 1024; WRITE_SIZE is exact for 16-byte stores and float atomics.
usage: tools/summarize_pmc.py <fetch_dir> <write_dir> <out.json>"""
import collections, csv, glob, json, re, sys


def symbol(name):
    n = name.replace("(anonymous namespace)::", "").replace("void ", "")
    n = re.sub(r"\(.*$", "", n).strip()
    if "igemm" not in n and "wgrad_ring" not in n and "conv_halo" not in n and "conv_wgrad_halo" not in n and "rowblock" not in n:
        return None
    return n


def load(d):
    files = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    agg = collections.defaultdict(lambda: [0.0, 0])
    for f in files:
        for r in csv.DictReader(open(f)):
            s = symbol(r["Kernel_Name"])
            if s:
                agg[s][0] += float(r["Counter_Value"])
                agg[s][1] += 1
    return agg


fetch, write = load(sys.argv[1]), load(sys.argv[2])
out = {}
for s in fetch:
    f, n = fetch[s]
    w, nw = write.get(s, (0.0, 1))
    rd = 2.0 * f * 1024 / n
    wr = w * 1024 / max(nw, 1)
    out[s] = {"launches_profiled": n, "hbm_read_bytes_per_launch": rd, "hbm_write_bytes_per_launch": wr,
              "hbm_bytes_per_launch": rd + wr}
json.dump(out, open(sys.argv[3], "w"), indent=1)
for s, v in sorted(out.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches_profiled"])[:12]:
    print(f"{s:70s} n={v['launches_profiled']:4d} {v['hbm_bytes_per_launch'] / 1e6:8.1f} MB/launch")
