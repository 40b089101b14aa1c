"""Per-kernel durations from a rocprofv3 --kernel-trace result database (rocpd sqlite)."""
import glob, sqlite3, sys
from collections import defaultdict
db = sqlite3.connect(glob.glob(sys.argv[1] + "/**/*.db", recursive=True)[0])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]
ks = [t for t in tabs if "kernel_symbol" in t][0]
acc = defaultdict(list)
for name, a, b in db.execute(f"select s.kernel_name, d.start, d.end from {kd} d join {ks} s on d.kernel_id=s.id order by d.start"):
    acc[name].append((b - a) / 1e6)
for name, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    print("%-70s calls %3d  avg %9.4f ms  min %9.4f  max %9.4f" % (name[:70], len(v), sum(v) / len(v), min(v), max(v)))
