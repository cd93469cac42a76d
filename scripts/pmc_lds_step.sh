#!/bin/bash
# Run ON the GPU box from the repo root: LDS bank-conflict cycles of EVERY kernel of the cfg2 step (one PMC pass) -> gpurun_out/<tag>_lds.txt
TAG=${1:-lds}; shift || true
R=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pl_$TAG
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES -d /tmp/pl_$TAG -o pl -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-encoder-probe --no-other-configs --no-prof --no-split-check "$@" > /dev/null 2>&1
cd $R
python scripts/rocpd_pmc.py $(find /tmp/pl_$TAG -name "*.db" | head -1) > gpurun_out/${TAG}_lds_raw.txt 2>&1
python - <<PY
import re
rows=[]; cur=None
for line in open("gpurun_out/${TAG}_lds_raw.txt"):
    m=re.match(r"== (.*?)  dispatches=(\d+) avg_us=([\d.]+)", line)
    if m: cur={"name":m.group(1)[:70],"n":int(m.group(2)),"us":float(m.group(3))}; rows.append(cur); continue
    m=re.match(r"\s+(\S+)\s+([\d.e+-]+)", line)
    if m and cur is not None: cur[m.group(1)]=float(m.group(2))
rows.sort(key=lambda r:-r["n"]*r["us"])
with open("gpurun_out/${TAG}_lds.txt","w") as f:
    f.write("%-72s %6s %8s %10s %10s %6s\n" % ("kernel","calls","avg us","conflict","lds active","frac"))
    for r in rows[:40]:
        c,a=r.get("SQ_LDS_BANK_CONFLICT",0),r.get("SQ_LDS_IDX_ACTIVE",0)
        f.write("%-72s %6d %8.1f %10.3g %10.3g %6.2f\n" % (r["name"],r["n"],r["us"],c,a,c/a if a else 0))
print(open("gpurun_out/${TAG}_lds.txt").read())
PY
