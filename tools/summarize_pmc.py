#!/usr/bin/env python3
"""Turns the two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs, --kernel-trace only) of
`python bench.py --steps 1 --warmup 1 --no_graph --no_cpu_baseline --no_roofline` into HBM bytes per launch for every
igemm instantiation.  gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE under-reports wide coalesced
reads by 2x -> read bytes = 2 * FETCH_SIZE(KiB) * 1024; WRITE_SIZE is exact for 16-byte stores and float atomics.
usage: tools/summarize_pmc.py <fetch_dir> <write_dir> <out.json>"""
import collections, csv, glob, json, sys

KINDS = {"Li0ELi0ELi0E": "rowk,rowk", "Li1ELi0E": "conv,rowk", "Li2ELi1ELi0E": "colk,colk", "Li2ELi2E": "colk,colk_conv",
         # rocprofv3's demangler garbles <__bf16, 1, 0, c> / <__bf16, 2, 1, 0>:
         "int, E, 0, ": "conv,rowk", "int, EL, int, E, 0>": "colk,colk"}


def kind_of(name):
    if "igemm_kernel" not in name:
        return None
    for pat, kd in KINDS.items():
        if pat in name:
            return kd
    return "other"


def load(d):
    f = glob.glob(d + "/*/*counter_collection.csv")[0]
    agg = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        kd = kind_of(r["Kernel_Name"])
        if kd:
            agg[kd][0] += float(r["Counter_Value"])
            agg[kd][1] += 1
    return agg


fetch, write = load(sys.argv[1]), load(sys.argv[2])
out = {}
for kd in fetch:
    f, n = fetch[kd]
    w, nw = write.get(kd, (0.0, 1))
    rd = 2.0 * f * 1024 / n
    wr = w * 1024 / max(nw, 1)
    out["bf16 igemm<%s>" % kd] = {"launches_profiled": n, "hbm_read_bytes_per_launch": rd, "hbm_write_bytes_per_launch": wr,
                                   "hbm_bytes_per_launch": rd + wr}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out, indent=1))
