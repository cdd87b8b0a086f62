"""per-kernel count / average / minimum duration (us) from a rocprofv3 results database (rocpd sqlite): python tools/debug/rocpd_stats.py file.db [filter]"""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for name, cnt, avg, mn in c.execute("select name, count(*), avg(end-start), min(end-start) from kernels group by name order by sum(end-start) desc limit 25"):
    if flt in name:
        print("%-110s %5d  avg %9.1f  min %9.1f" % (name[:110], cnt, avg / 1e3, mn / 1e3))
