"""Summarise a rocprofv3 rocpd sqlite database (kernel trace) into a per-kernel stats table (markdown/CSV)."""
import re
import sqlite3
import sys


def main(db, out=None, top=40):
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(rocpd_kernel_dispatch)")]
    scols = [r[1] for r in c.execute("pragma table_info(rocpd_info_kernel_symbol)")]
    name_col = "kernel_name" if "kernel_name" in scols else ("display_name" if "display_name" in scols else scols[-1])
    q = ("select s.%s, count(*), sum(d.end-d.start), avg(d.end-d.start), min(d.end-d.start), max(d.end-d.start) "
         "from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id group by s.%s "
         "order by 3 desc" % (name_col, name_col))
    rows = c.execute(q).fetchall()
    total = sum(r[2] for r in rows)
    lines = ["| kernel | calls | total ms | avg us | min us | max us | % |", "|---|---|---|---|---|---|---|"]
    for r in rows[:top]:
        nm = re.sub(r"\(.*", "", r[0])
        nm = nm.replace("xv2::", "").replace("void ", "")
        lines.append("| %s | %d | %.3f | %.1f | %.1f | %.1f | %.1f |" % (nm[:90], r[1], r[2] / 1e6, r[3] / 1e3, r[4] / 1e3,
                                                                    r[5] / 1e3, 100.0 * r[2] / total))
    lines.append("| TOTAL (%d kernels) | %d | %.3f | | | | 100 |" % (len(rows), sum(r[1] for r in rows), total / 1e6))
    txt = "\n".join(lines)
    if out:
        open(out, "w").write(txt + "\n")
    print(txt)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
