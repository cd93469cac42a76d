"""Timeline analysis of one training step from a rocprofv3 kernel-trace database (rocpd sqlite): per queue/stream busy time,
gaps, overlap between the compute stream and the weight-gradient side stream, and the critical-path kernels.
usage: python scripts/timeline.py <results.db> [step_index]"""
import re
import sqlite3
import sys
from collections import defaultdict


def main(db, which=6):
    c = sqlite3.connect(db)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    disp = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    sym = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    cols = [r[1] for r in c.execute("pragma table_info(%s)" % disp)]
    scols = [r[1] for r in c.execute("pragma table_info(%s)" % sym)]
    name_col = "kernel_name" if "kernel_name" in scols else "display_name"
    qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
    rows = c.execute("select s.%s, d.start, d.end, d.%s from %s d join %s s on d.kernel_id = s.id order by d.start" % (
        name_col, qcol, disp, sym)).fetchall()
    short = lambda n: re.sub(r"\(.*", "", n).replace("xv2::", "").replace("void ", "")[:60]
    # steps are delimited by the AdamW kernel
    marks = [i for i, r in enumerate(rows) if "adamw" in r[0]]
    if len(marks) < which + 2:
        which = max(0, len(marks) - 2)
    a, b = marks[which] + 1, marks[which + 1] + 1
    step = rows[a:b]
    t0, t1 = step[0][1], max(r[2] for r in step)
    print("step %d: %d dispatches, wall %.3f ms" % (which, len(step), (t1 - t0) / 1e6))
    byq = defaultdict(list)
    for n, s, e, q in step:
        byq[q].append((s, e, n))
    for q, L in sorted(byq.items(), key=lambda kv: -len(kv[1])):
        busy = sum(e - s for s, e, _ in L)
        gaps = [L[i + 1][0] - L[i][1] for i in range(len(L) - 1)]
        small = [g for g in gaps if 0 <= g < 20000]
        print("  queue %s: %4d kernels, busy %.3f ms, span %.3f ms, gaps<20us: %d totalling %.3f ms (median %.2f us)" % (
            q, len(L), busy / 1e6, (L[-1][1] - L[0][0]) / 1e6, len(small), sum(small) / 1e6,
            sorted(small)[len(small) // 2] / 1e3 if small else 0))
    # the largest holes of the busiest queue (the compute stream): what ran before and after them
    q0 = max(byq.items(), key=lambda kv: len(kv[1]))[1]
    holes = sorted(((q0[i + 1][0] - q0[i][1], i) for i in range(len(q0) - 1)), reverse=True)[:14]
    print("  largest gaps on the compute queue (us, after -> before):")
    for g, i in holes:
        print("     %8.1f  %-44s -> %s" % (g / 1e3, short(q0[i][2])[:44], short(q0[i + 1][2])[:44]))
    # what the OTHER queues ran inside the largest hole (the step's tail: the compute stream waits for the last weight gradients)
    g0, i0 = holes[0]
    lo, hi = q0[i0][1], q0[i0 + 1][0]
    print("  inside the largest gap (%.1f us), other queues:" % (g0 / 1e3))
    for n, s_, e_, q in step:
        if e_ > lo and s_ < hi and (s_, e_, n) not in (q0[i0], q0[i0 + 1]) and not (s_ == q0[i0][0]):
            print("     %+8.1f .. %+8.1f us  %s" % ((s_ - lo) / 1e3, (e_ - lo) / 1e3, short(n)[:70]))
    import os
    flt = os.environ.get("XV2_TIMELINE_FILTER")
    if flt:      # every launch of the step whose name contains the filter: offset from the step's start, duration, what ran before it on its queue
        print("  launches matching %r:" % flt)
        for q, L in byq.items():
            for i, (s_, e_, n) in enumerate(L):
                if flt in n:
                    print("     %+9.1f us  %7.1f us  after %s" % ((s_ - t0) / 1e3, (e_ - s_) / 1e3, short(L[i - 1][2])[:60] if i else "-"))
    big = sum(g for g, _ in holes if g >= 20000)
    print("  gaps >= 20 us among them: %.3f ms" % (big / 1e6))
    # union busy / overlap
    ev = []
    for n, s, e, q in step:
        ev.append((s, 1))
        ev.append((e, -1))
    ev.sort()
    depth, last, union, multi = 0, t0, 0, 0
    for t, d in ev:
        if depth >= 1:
            union += t - last
        if depth >= 2:
            multi += t - last
        depth += d
        last = t
    print("  any kernel running %.3f ms (idle %.3f ms), >=2 kernels running %.3f ms" % (union / 1e6, (t1 - t0 - union) / 1e6, multi / 1e6))
    if os.environ.get("XV2_TIMELINE_ALL"):      # every kernel name of the step: launches, total and average time
        agg = defaultdict(lambda: [0, 0])
        for n, s_, e_, q in step:
            agg[short(n)][0] += e_ - s_
            agg[short(n)][1] += 1
        print("  all kernels of the step:")
        for n, (t, k) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
            print("     %-64s %4d %8.3f ms  %7.1f us" % (n, k, t / 1e6, t / k / 1e3))
    # phases: forward ends at the loss kernel
    loss_i = next((i for i, r in enumerate(step) if "loss_fwd" in r[0]), None)
    if loss_i is not None:
        tl = step[loss_i][1]
        print("  forward %.3f ms, backward+optimizer %.3f ms" % ((tl - t0) / 1e6, (t1 - tl) / 1e6))
        for label, lo, hi in (("forward", t0, tl), ("backward", tl, t1)):
            agg = defaultdict(lambda: [0, 0])
            for n, s, e, q in step:
                if lo <= s < hi:
                    agg[short(n)][0] += e - s
                    agg[short(n)][1] += 1
            top = sorted(agg.items(), key=lambda kv: -kv[1][0])[:12]
            print("  %s kernel time by name:" % label)
            for n, (t, k) in top:
                print("     %-62s %4d %8.3f ms" % (n, k, t / 1e6))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 6)
