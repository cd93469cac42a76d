"""bench.py's final line: compact, parseable, contract keys (the 25 KB line of round 5 was lost by the
driver's parser).  The round-5 record (profiles/r05_bench_v8.json) is the input: the heaviest line the bench ever produced."""
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("record", ["r05_bench_v8.json", "r05_bench_v1.json", "r04_bench_v9.json"])
def test_compact_line_of_a_full_record_stays_small_and_keeps_the_contract(record):
    import bench
    path = os.path.join(ROOT, "profiles", record)
    full = json.loads(open(path).read().strip().splitlines()[-1])
    full["config"].setdefault("math", None)
    c = bench.compact_line(full, os.path.join(ROOT, "gpurun_out", "bench_detail.json"))
    line = json.dumps(c, separators=(",", ":"))
    assert len(line) < bench.COMPACT_LINE_MAX <= 6144 and "\n" not in line
    for k in ("metric", "value", "unit", "n_gpus", "n_ranks_seen", "steps", "warmup", "ms_per_step", "higher_is_better",
              "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "parity"):
        assert k in c, k
    assert c["value"] == full["value"] and c["ms_per_step"] == full["ms_per_step"]
    assert len(c["config"]["workload"]) <= 200 and set(c["config"]) == {"workload", "global_batch", "parallelism", "syncbn", "math"}
    r = c["roofline"]
    assert r["bound"] in ("hbm", "mfma") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert "per_kernel" not in r and "note" not in r and "math" not in r
    assert c["cpu_baseline"]["kind"] == "port" and c["cpu_baseline"]["cores"] >= 1
    assert c["parity"]["pass"] is True
    assert all("per_kernel" not in e for e in c.get("encoder_forward", []))
    for o in c.get("other_configs", []):
        assert len(o["config"]) <= 40 and 0 < o["roofline"]["mfma"]["frac"] < 1
    assert c["detail"] == "gpurun_out/bench_detail.json"
