"""profiles/ptdb.py <rocprofv3 .db> -- per-kernel statistics and one sweep's timeline from a rocprofv3 kernel trace
(the sqlite output of `rocprofv3 --kernel-trace --stats -o <name>`)."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
rows = list(cur.execute(f"select s.kernel_name, count(*), avg(d.end-d.start), min(d.end-d.start), max(d.end-d.start), "
                        f"sum(d.end-d.start) from {kd} d join {ks} s on d.kernel_id=s.id group by s.kernel_name order by 6 desc"))
print("kernel,calls,avg_us,min_us,max_us,total_ms")
for r in rows[:12]:
    print(f"{r[0]},{r[1]},{r[2] / 1e3:.2f},{r[3] / 1e3:.2f},{r[4] / 1e3:.2f},{r[5] / 1e6:.3f}")
if len(sys.argv) > 2:
    rows = list(cur.execute(f"select d.start, d.end, s.kernel_name from {kd} d join {ks} s on d.kernel_id=s.id order by d.start"))
    mid, prev = len(rows) // 2, None
    for st, en, name in rows[mid:mid + int(sys.argv[2])]:
        print(f"{name[:48]:48s} dur={(en - st) / 1e3:7.1f}us gap_before={((st - prev) / 1e3 if prev else 0):6.1f}us")
        prev = en
