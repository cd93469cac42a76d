#!/bin/bash
# First-run kit for a multi-GPU node (run ON the node, from the repo root): bench.py at 1 / 2 / 4 / 8 ranks, one rank per GPU
# over RCCL, and per run the facts a first run must show - how many ranks really joined, which SyncBatchNorm transport was
# chosen, bus bandwidth of the gradient all-reduce, how much of it hid behind backward.  The scaling efficiency itself is the
# driver's to compute from the `value` fields (SCALE_rNN.json); this prints them side by side.
# usage: scripts/scale.sh [max_gpus] [extra bench args]     XV2_SYNCBN=rccl|auto|oneshot passes through
MAX=${1:-8}; shift || true
export HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out
for N in 1 2 4 8; do
  [ "$N" -gt "$MAX" ] && break
  OUT=gpurun_out/scale_n${N}.json
  if [ "$N" = 1 ]; then
    python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --no-encoder-probe --no-split-check "$@" > $OUT 2> gpurun_out/scale_n${N}.err
  else
    python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((29500 + N)) \
      bench.py --gpus $N --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --no-encoder-probe --no-split-check "$@" > $OUT 2> gpurun_out/scale_n${N}.err
  fi
  python - "$OUT" "$N" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
except Exception as e:
    print("n=%s: no result line (%s) - see gpurun_out/scale_n%s.err" % (sys.argv[2], e, sys.argv[2])); sys.exit(0)
c = d.get("collectives") or {}
print("n_gpus %d  n_ranks_seen %s  %.2f img/s  %.2f ms/step  syncbn: %s  buckets %s | all-reduce %s MB in %s ms = %s GB/s bus | "
      "serialised step %s ms, overlap fraction %s" % (
          d["n_gpus"], d.get("n_ranks_seen"), d["value"], d["ms_per_step"], d["config"].get("syncbn"), d["config"].get("grad_buckets"),
          round(c.get("allreduce_bytes", 0) / 1e6, 1), c.get("allreduce_ms"), c.get("bus_gbs"), c.get("step_ms_collectives_serialised"),
          c.get("overlap_fraction")))
PY
done
python - <<'PY'
import glob, json
v = {}
for f in sorted(glob.glob("gpurun_out/scale_n*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); v[d["n_gpus"]] = d["value"]
    except Exception:
        pass
if 1 in v:
    print("weak scaling vs 1 GPU: " + "  ".join("%d: %.2fx" % (n, v[n] / v[1]) for n in sorted(v)))
PY
