"""Per-kernel durations from a rocprofv3 rocpd database (rocprofv3 --kernel-trace -d DIR -o NAME -> DIR/NAME_results.db)."""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
kd = [t for t in tabs if "kernel_dispatch" in t and "rocpd" in t][0]
ks = [t for t in tabs if "kernel_symbol" in t][0]
q = ("select s.kernel_name, count(*), avg(d.end-d.start)/1e3, min(d.end-d.start)/1e3 from %s d join %s s "
     "on d.kernel_id=s.id group by s.kernel_name order by 2*3 desc" % (kd, ks))
for name, n, avg, lo in c.execute(q):
    print("%-100s n=%4d avg %9.1f us  min %9.1f us" % (name[:100], n, avg, lo))
