#!/usr/bin/env python3
"""Print the per-dispatch timeline of the last bench step from a rocprofv3 kernel trace
(csv `*_kernel_trace.csv` or rocpd `*_results.db`)."""
import csv, sys, sqlite3
path = sys.argv[1]
rows = []
if path.endswith(".db"):
    db = sqlite3.connect(path)
    q = ("select s.kernel_name, d.start, d.end, d.grid_size_x, d.workgroup_size_x, d.grid_size_y from rocpd_kernel_dispatch d "
         "join rocpd_info_kernel_symbol s on d.kernel_id=s.id order by d.start")
    rows = [(r[0], r[1], r[2], r[3] // max(1, r[4]), r[5]) for r in db.execute(q)]
else:
    for r in csv.DictReader(open(path)):
        rows.append((r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"]),
                     int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])), int(r["Grid_Size_Y"])))
    rows.sort(key=lambda r: r[1])
marks = [i for i, r in enumerate(rows) if "ripple" in r[0] or "key_addr" in r[0]]
# a step starts at the first key-addressing kernel after a score kernel; take the last full step
last = marks[-1]
start = last
while start > 0 and (rows[start][1] - rows[start - 1][2]) < 50_000 and not ("linear" in rows[start - 1][0] and start - 1 < marks[0]):
    start -= 1
    if len([1 for r in rows[start:last] if "ripple" in r[0] or "key_addr" in r[0]]) >= 3 and "expand" in rows[start][0]:
        break
n = int(sys.argv[2]) if len(sys.argv) > 2 else 24
seg = rows[-n:]
t0 = seg[0][1]
tot = 0
for name, s, e, gx, gy in seg:
    print("%9.1f us  dur %8.1f  grid %5d x%-3d %s" % ((s - t0) / 1e3, (e - s) / 1e3, gx, gy, name[:70]))
    tot += e - s
print("sum of durations %.1f us over span %.1f us" % (tot / 1e3, (seg[-1][2] - t0) / 1e3))
