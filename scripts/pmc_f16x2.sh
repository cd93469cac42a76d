#!/bin/bash
# Run ON the GPU box from the repo root: clock / matrix-pipe duty of the two-plane (F16X2) kernels of the step, one layer each
# (VERDICT r04 item 2): GRBM_GUI_ACTIVE / duration = effective clock, SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x 1024 SIMDs... see
# the table's last column) = duty of the matrix pipe.  Counters in their own passes (no --stats / trace domains next to --pmc).
# -> gpurun_out/r05_pmc_f16x2.md
R=$PWD
OUT=$R/gpurun_out/r05_pmc_f16x2.md
cd /tmp && export TMPDIR=/tmp
echo "| layer / pass | kernel | us | GRBM_GUI_ACTIVE | clock GHz | SQ_VALU_MFMA_BUSY_CYCLES | MFMA duty | SQ_BUSY_CYCLES | SQ_INSTS_MFMA | SQ_INSTS_VALU |" > $OUT
echo "|---|---|---|---|---|---|---|---|---|---|" >> $OUT
for spec in "dec2.c1:fwd:igemm_kernel" "dec2.c1:dgrad:igemm_kernel" "dec2.c1:wgrad:wgrad_alltaps" "dec3.c1:fwd:igemm_kernel" "dec3.c1:wgrad:wgrad_alltaps" \
            "l3.conv1:fwd:sg_conv" "l3.conv3:fwd:sg_conv" "l3.conv2:fwd:sg_conv" "l4.conv1:fwd:sg_conv" "l3.conv3:wgrad:wgrad_tr" "l2.conv1:fwd:sg_conv"; do
  IFS=: read L W F <<< "$spec"
  rm -rf /tmp/pmcF
  XV2_ONE_AMAX=1 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA -d /tmp/pmcF -o f -- python $R/scripts/one_conv.py "$L" $W 8 > /dev/null 2>&1
  python - "$L $W" "$F" >> $OUT <<'PY'
import re, sqlite3, sys, glob
from collections import defaultdict
db = glob.glob("/tmp/pmcF/**/*.db", recursive=True)[0]
c = sqlite3.connect(db)
icol = [r[1] for r in c.execute("pragma table_info(rocpd_info_pmc)")]
name_c = "name" if "name" in icol else icol[-1]
q = ("select s.kernel_name, p.%s, e.value, d.id, d.end-d.start from rocpd_pmc_event e join rocpd_info_pmc p on e.pmc_id = p.id "
     "join rocpd_kernel_dispatch d on e.event_id = d.event_id join rocpd_info_kernel_symbol s on d.kernel_id = s.id" % name_c)
agg, ids, dur = defaultdict(lambda: defaultdict(float)), defaultdict(set), defaultdict(dict)
for kn, pn, v, did, dt in c.execute(q):
    if sys.argv[2] not in kn:
        continue
    kn = re.sub(r"\(.*", "", kn).replace("xv2::", "").replace("void ", "")
    agg[kn][pn] += v; ids[kn].add(did); dur[kn][did] = dt
for kn in agg:
    n = len(ids[kn]); us = sum(dur[kn].values()) / n / 1e3
    g = agg[kn]["GRBM_GUI_ACTIVE"] / n; mf = agg[kn]["SQ_VALU_MFMA_BUSY_CYCLES"] / n
    clock = g / 8 / (us * 1e3)                  # GRBM_GUI_ACTIVE is summed over the 8 XCDs; cycles per ns = GHz
    duty = mf / (g / 8 * 1024) if g else 0.0    # busy cycles summed over 256 CUs x 4 SIMDs
    print("| %s | %s | %.1f | %.4g | %.2f | %.4g | %.3f | %.4g | %.4g | %.4g |" % (
        sys.argv[1], kn[:70], us, g, clock, mf, duty, agg[kn]["SQ_BUSY_CYCLES"] / n, agg[kn]["SQ_INSTS_MFMA"] / n, agg[kn]["SQ_INSTS_VALU"] / n))
PY
done
cat $OUT
