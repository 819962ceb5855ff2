#!/usr/bin/env python3
"""Per-kernel statistics (calls, total/avg/min/max duration, share) from a rocprofv3 rocpd sqlite database
(`rocprofv3 --kernel-trace --stats -d DIR -o NAME -- python3 bench.py ...` writes DIR/NAME_results.db on ROCm 7.2).
usage: tools/rocpd_stats.py <results.db> [out.csv]"""
import re, sqlite3, sys

con = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
rows = con.execute(f"select {name_col}, count(*), sum(end - start), min(end - start), max(end - start) from kernels "
                   f"group by {name_col} order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    n = re.sub(r"\(.*$", "", n)
    return n if len(n) < 110 else n[:107] + "..."


lines = ["Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs"]
for n, c, t, mn, mx in rows:
    lines.append(f"\"{short(n)}\",{c},{t},{t / c:.1f},{100.0 * t / tot:.3f},{mn},{mx}")
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write("\n".join(lines) + "\n")
for l in lines[:45]:
    print(l)
print(f"total kernel time {tot/1e6:.2f} ms over {sum(r[1] for r in rows)} dispatches")
