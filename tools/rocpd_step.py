#!/usr/bin/env python3
"""Per-kernel time inside ONE steady-state main step of a rocprofv3 trace of bench.py (steps are delimited by the
forward-diffusion kernel that opens each one).  usage: tools/rocpd_step.py <results.db> [step_index] [top_n]"""
import re, sqlite3, sys, collections
con = sqlite3.connect(sys.argv[1])
rows = con.execute("select start, end, name, grid_x, grid_y, workgroup_x from kernels order by start").fetchall()
ad = [i for i, r in enumerate(rows) if "noise_kernel" in r[2]]   # forward-diffusion kernels: TWO per step since round 3 (the
ad = ad[::2]                                                     # teacher graph and the student's forward graph each open with
                                                                 # one, a few us to a few ms apart): every second one opens a step
si = int(sys.argv[2]) if len(sys.argv) > 2 else len(ad) // 2
seg = rows[ad[si]: ad[si + 1]]
def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    m = re.match(r"_ZN12_GLOBAL__N_1\d+([a-z_0-9]+?)I", n)
    if m: n = m.group(1) + n[m.end() - 1:][:34]
    return re.sub(r"\(.*$", "", n)[:64]
agg = collections.OrderedDict()
for s, e, n, gx, gy, wx in seg:
    a = agg.setdefault(short(n), [0, 0])
    a[0] += 1; a[1] += e - s
tot = sum(a[1] for a in agg.values())
print(f"step {si}: {len(seg)} kernels, sum {tot/1e6:.2f} ms, wall {(seg[-1][1]-seg[0][0])/1e6:.2f} ms")
gemm = sum(a[1] for n, a in agg.items() if "igemm" in n)
print(f"igemm total {gemm/1e6:.2f} ms")
for n, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[: int(sys.argv[3]) if len(sys.argv) > 3 else 40]:
    print(f"{n:64s} n={a[0]:4d} {a[1]/1e6:7.3f} ms avg {a[1]/a[0]/1e3:7.1f} us")
