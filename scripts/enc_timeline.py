"""One training-mode encoder forward from a rocprofv3 kernel-trace database (scripts/enc_fwd_only.py under the profiler): every
launch in order with its duration and the gap in front of it, totals per kernel name, and the sum of the gaps.
usage: python scripts/enc_timeline.py <results.db> [pass index from the end, default 2]"""
import re
import sqlite3
import sys
from collections import defaultdict

db = sys.argv[1]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
c = sqlite3.connect(db)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
disp = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
sym = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
scols = [r[1] for r in c.execute("pragma table_info(%s)" % sym)]
name_col = "kernel_name" if "kernel_name" in scols else "display_name"
rows = c.execute("select s.%s, d.start, d.end from %s d join %s s on d.kernel_id = s.id order by d.start" % (name_col, disp, sym)).fetchall()


def short(n):
    n = re.sub(r"^_ZN3xv2\d+", "", n)
    n = re.sub(r"\.kd$", "", n)
    m = re.match(r"([a-z0-9_]+kernel)(I.*?E)?Ev", n)
    if m:
        args = re.findall(r"L[ib](\d+)E", m.group(2) or "")
        return m.group(1) + ("<" + ",".join(args) + ">" if args else "")
    return n[:50]


marks = [i for i, r in enumerate(rows) if "normalize_u8" in r[0]]
a, b = marks[-back - 1], marks[-back]
step = rows[a:b]
t0, t1 = step[0][1], step[-1][2]
print("pass: %d launches, wall %.3f ms, kernel time %.3f ms" % (len(step), (t1 - t0) / 1e6, sum(e - s for _, s, e in step) / 1e6))
gaps = [max(0, step[i + 1][1] - step[i][2]) for i in range(len(step) - 1)]
print("gaps: total %.3f ms, median %.2f us, >3us: %d" % (sum(gaps) / 1e6, sorted(gaps)[len(gaps) // 2] / 1e3, sum(g > 3000 for g in gaps)))
agg = defaultdict(lambda: [0, 0])
for n, s, e in step:
    agg[short(n)][0] += e - s
    agg[short(n)][1] += 1
for n, (t, k) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
    print("  %-58s %4d %8.3f ms  avg %6.1f us" % (n, k, t / 1e6, t / k / 1e3))
print("in order (offset us, gap us, duration us, kernel):")
prev = t0
for n, s, e in step:
    print("  %9.1f %6.1f %7.1f  %s" % ((s - t0) / 1e3, (s - prev) / 1e3, (e - s) / 1e3, short(n)))
    prev = e
