#!/usr/bin/env python
"""gpurun_out/parity_rows.jsonl (written by tests/test_model_gpu.py on the GPU box) -> profiles/parity_rNN.md:
per model case the conditioning probe (CPU-fp32 oracle vs its own fp64 run), which gate branch was taken, the HIP
path's error against fp64 and against the CPU-fp32 oracle, label-map mismatches and the gradient-error summary.

    python scripts/parity_table.py gpurun_out/parity_rows.jsonl profiles/parity_r02.md"""
import json
import sys


def fmt(v):
    if v is None:
        return "-"
    if isinstance(v, float):
        return "%.2e" % v if (abs(v) < 1e-2 or abs(v) >= 1e3) and v != 0 else "%.4f" % v
    return str(v)


def main(src, dst):
    rows = {}
    for line in open(src):
        line = line.strip()
        if line:
            r = json.loads(line)
            rows[(r["case"], r.get("batch"), r.get("mode"))] = r      # the last run of a case wins
    cols = [("case", "case"), ("batch", "B"), ("mode", "mode"), ("cond_cpu32_vs_f64", "cond = cpu32 vs f64"),
            ("branch", "gate branch"), ("hip_vs_f64", "hip vs f64"), ("hip_vs_cpu32", "hip vs cpu32"),
            ("argmax_mismatch_outside_ties", "argmax mismatches (outside ties)"),
            ("argmax_mismatch_cpu32_vs_f64", "argmax cpu32 vs f64"), ("loss_hip", "loss hip"),
            ("loss_cpu32", "loss cpu32"), ("grad_median_ratio_hip_over_cpu32", "grad err ratio hip/cpu32 (median)"),
            ("grad_median_err_hip", "grad err hip vs f64 (median)"),
            ("grad_median_err_cpu32", "grad err cpu32 vs f64 (median)"),
            # block-by-block rows (test_blockwise_teacher_forced_parity): oracle blocks fed with the HIP path's own inputs
            ("blocks", "blocks compared"), ("block_max_rel", "worst block (max-abs rel)"),
            ("block_max_rms_rel", "worst block (rms rel)"), ("block_max_rel_at", "at"),
            ("argmax_agreement", "label agreement"), ("loss_rel", "loss rel")]
    out = ["# Model-level parity table (tests/test_model_gpu.py on MI355X)", "",
           "Logits errors are max-abs error / max-abs reference. `cond` is the CPU-fp32 oracle against an fp64 run of the",
           "same oracle (how well-conditioned the case is in fp32 at all); `strict` = plain 1e-3 gate against the CPU-fp32",
           "oracle with exact label maps, taken when cond <= 3e-4; otherwise the HIP path must be within 3 x cond of fp64.",
           "Gradient columns: per-tensor relative L2 error against the fp64 gradients, median over tensors.", "",
           "| " + " | ".join(c[1] for c in cols) + " |", "|" + "---|" * len(cols)]
    for key in sorted(rows, key=lambda k: (str(k[2]), str(k[0]), k[1] or 0)):
        r = rows[key]
        out.append("| " + " | ".join(fmt(r.get(c[0])) for c in cols) + " |")
    for r in rows.values():
        if "grad_global_err_hip_vs_f64" in r:
            out += ["", "%s: whole-gradient error against fp64 - hip %s, cpu32 %s (%d tensors)" % (
                r["case"], fmt(r["grad_global_err_hip_vs_f64"]), fmt(r["grad_global_err_cpu32_vs_f64"]), r["grad_tensors"])]
    open(dst, "w").write("\n".join(out) + "\n")
    print("wrote", dst, len(rows), "rows")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
