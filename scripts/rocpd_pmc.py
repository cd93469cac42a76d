"""Aggregate PMC counters of a rocprofv3 rocpd sqlite database per kernel name (mean per dispatch)."""
import re
import sqlite3
import sys
from collections import defaultdict


def main(db, filt=""):
    c = sqlite3.connect(db)
    pcols = [r[1] for r in c.execute("pragma table_info(rocpd_pmc_event)")]
    icol = [r[1] for r in c.execute("pragma table_info(rocpd_info_pmc)")]
    name_c = "name" if "name" in icol else icol[-1]
    q = ("select s.kernel_name, p.%s, e.value, d.id, d.end-d.start from rocpd_pmc_event e "
         "join rocpd_info_pmc p on e.pmc_id = p.id "
         "join rocpd_kernel_dispatch d on e.event_id = d.event_id "
         "join rocpd_info_kernel_symbol s on d.kernel_id = s.id" % name_c)
    agg = defaultdict(lambda: defaultdict(float))
    cnt = defaultdict(set)
    dur = defaultdict(dict)
    for kn, pn, v, did, dt in c.execute(q):
        kn = re.sub(r"\(.*", "", kn).replace("xv2::", "").replace("void ", "")
        if filt and filt not in kn:
            continue
        agg[kn][pn] += v
        cnt[kn].add(did)
        dur[kn][did] = dt
    for kn in agg:
        n = len(cnt[kn])
        print("== %s  dispatches=%d avg_us=%.1f" % (kn[:100], n, sum(dur[kn].values()) / n / 1e3))
        for pn, v in sorted(agg[kn].items()):
            print("   %-32s %.4g" % (pn, v / n))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "")
