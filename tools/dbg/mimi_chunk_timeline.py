#!/usr/bin/env python3
"""Per-kernel timeline of ONE 10-frame Mimi chunk decode from a rocprofv3 run of tools/mimi_prof.py:

    cd /tmp && rocprofv3 --kernel-trace --stats -d <dir> -o mimi -- python3 $REPO/tools/mimi_prof.py
    python3 tools/dbg/mimi_chunk_timeline.py <dir>/mimi_results.db

(rocprofv3's default output here is the rocpd SQLite database; the kernel dispatches are joined with their symbols.)"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
rows = list(db.execute(f"select d.start, d.end, s.kernel_name, d.grid_size_x, d.grid_size_y, d.grid_size_z, d.workgroup_size_x "
                       f"from {kd} d join {ks} s on d.kernel_id = s.id order by d.start"))
starts = [i for i, r in enumerate(rows) if "k_rvq" in r[2] and "pick" not in r[2] and r[3] == 10 * r[6]]
i0, i1 = starts[-2], starts[-1]
seg, t0, busy = rows[i0:i1], rows[i0][0], 0.0
agg = {}
for r in seg:
    d = (r[1] - r[0]) / 1e3
    busy += d
    name = r[2].split("(")[0][:44]
    print(f"{(r[0] - t0) / 1e3:8.1f} {d:6.1f}  {name:44s} grid {r[3] // r[6]}x{r[4]}x{r[5]}")
    a = agg.setdefault(name, [0, 0.0]); a[0] += 1; a[1] += d
print(f"{len(seg)} kernels, busy {busy:.1f} us, period {(rows[i1][0] - t0) / 1e3:.1f} us")
for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"   {t:7.1f} us  {n:3d} x  {k}")
