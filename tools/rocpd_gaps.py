#!/usr/bin/env python3
"""Idle time inside ONE steady-state main step of a rocprofv3 kernel trace of bench.py: the time no kernel at all is running
(union of the kernel intervals of every stream against the step's wall time) and, per stream, the gaps between consecutive
kernels.  usage: tools/rocpd_gaps.py <results.db> [step_index]"""
import sqlite3, sys, collections
con = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
print("columns:", cols)
qcol = next((c for c in ("stream_id", "queue_id", "stream", "queue") if c in cols), None)
rows = con.execute(f"select start, end, name, {qcol or 0} from kernels order by start").fetchall()
ad = [i for i, r in enumerate(rows) if "noise_kernel" in r[2]][::2]
si = int(sys.argv[2]) if len(sys.argv) > 2 else len(ad) // 2
seg = rows[ad[si]: ad[si + 1]]
t0, t1 = seg[0][0], max(r[1] for r in seg)
wall = rows[ad[si + 1]][0] - t0
busy, cur_s, cur_e = 0, None, None
for s, e, _, _ in seg:
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print(f"step {si}: {len(seg)} kernels, wall {wall/1e6:.2f} ms (start of this step's first kernel to the next step's), "
      f"some kernel running {busy/1e6:.2f} ms, nothing running {(wall-busy)/1e6:.2f} ms")
by = collections.defaultdict(list)
for r in seg:
    by[r[3]].append(r)
for q, rs in sorted(by.items(), key=lambda kv: -len(kv[1])):
    gaps = [b[0] - a[1] for a, b in zip(rs, rs[1:]) if b[0] > a[1]]
    small = [g for g in gaps if g < 20000]
    print(f"stream {q}: {len(rs)} kernels, kernel time {sum(r[1]-r[0] for r in rs)/1e6:.2f} ms, {len(small)} gaps < 20 us summing "
          f"{sum(small)/1e6:.2f} ms (median {sorted(small)[len(small)//2]/1e3 if small else 0:.2f} us), "
          f"{len(gaps)-len(small)} longer gaps summing {sum(g for g in gaps if g >= 20000)/1e6:.2f} ms")
    big = sorted(((b[0] - a[1], a[2], b[2]) for a, b in zip(rs, rs[1:]) if b[0] - a[1] >= 20000), reverse=True)[:6]
    for gp, an, bn in big:
        print(f"    gap {gp/1e3:8.1f} us  after {an[:60]:60s}  before {bn[:60]}")
