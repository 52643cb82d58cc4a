"""Summarises a rocprofv3 rocpd database (kernel trace) into a small text table for profiles/."""
import sqlite3
import sys

db = sys.argv[1]
c = sqlite3.connect(db)
print("# rocprofv3 --kernel-trace --stats summary of", db.split("/")[-1])
print("%-70s %8s %14s %12s %12s %12s %7s" % ("kernel", "calls", "total_ns", "avg_ns", "min_ns", "max_ns", "pct"))
rows = c.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels group by name order by sum(duration) desc").fetchall()
tot = sum(r[2] for r in rows) or 1
for r in rows:
    print("%-70s %8d %14d %12.0f %12d %12d %6.2f%%" % (r[0][:70], r[1], r[2], r[3], r[4], r[5], 100.0 * r[2] / tot))
print()
print("# per-kernel resources")
for r in c.execute("select distinct name, grid_x, workgroup_x, lds_size, vgpr_count, accum_vgpr_count, sgpr_count, scratch_size from kernels order by name").fetchall():
    print("%-70s grid %7d wg %4d lds %7d vgpr %4d agpr %4d sgpr %4d scratch %d" % (r[0][:70], r[1], r[2], r[3], r[4], r[5], r[6], r[7]))
