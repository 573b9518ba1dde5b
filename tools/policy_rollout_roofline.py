"""Per-kernel roofline of the with-policy rollout from a rocprofv3 kernel_stats.csv (tools/prof_policy_rollout.sh).
Algorithmic bytes / flops per launch at B = 4096 envs, BOTH students per launch (operand row 416 floats,
gates 1024, state 512, heads 64 + 7 + 1):
  policy_features (pair)  in: u8 obs 76 B + h 1 KB + mask 4 B; out: operand row 416 x 4 B               per env and student
  batched GEMM            [2, 4096, 416] x [2, 416, 1024]: A 6.8 MB + B 1.7 MB + C 16.8 MB per student; 3.49 GFLOP per student
  policy_cell_heads (pair) in: gates 4 KB + (h, c) 2 KB + mask; out: (h', c') 2 KB + value / action / log-prob / log-dist 48 B
  mg_step_kernel (pair)   202 B per env-step"""
import csv, json, sys
B = 4096
ALGO = {  # kernel-name substring -> (bytes per launch, flops per launch, what)
    "policy_features_kernel": (2 * B * (76 + 1024 + 4 + 416 * 4), 0, "both students: obs bytes + h*mask in, GEMM operand row out"),
    "policy_cell_heads_kernel": (2 * B * (4096 + 2048 + 4 + 2048 + 48), 0, "both students: gates + state in, state + policy outputs out"),
    "mg_step_kernel": (2 * B * 202, 0, "both students' env copies, 202 B per env-step"),
    "Cijk_": (2 * (B * 416 + 416 * 1024 + B * 1024) * 4, 2 * 2 * B * 416 * 1024, "batched LSTM-input GEMM, both students (rocBLAS / hipBLASLt)"),
}
rows = list(csv.DictReader(open(sys.argv[1])))
out = {"source": sys.argv[1], "time_pair_rollout": json.loads(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].strip().startswith("{") else None,
       "hbm_peak_GBs": 8000.0, "fp32_mfma_peak_TFLOPs": 157.3, "kernels": []}
for r in rows:
    name = r.get("Name") or r.get("KernelName") or ""
    for sub, (nbytes, flops, what) in ALGO.items():
        if sub in name:
            avg_ns = float(r.get("AverageNs") or r.get("Average") or 0)
            calls = int(float(r.get("Calls") or 0))
            if calls < 200 or avg_ns <= 0:                       # the rollout's kernels run >= 256 times per rollout
                continue
            k = {"kernel": name[:110], "calls": calls, "avg_us": avg_ns / 1e3, "algorithmic_bytes_per_launch": nbytes, "what": what,
                 "achieved_GBs": nbytes / avg_ns, "frac_of_hbm_peak": nbytes / avg_ns / 8000.0}
            if flops:
                k["achieved_TFLOPs"] = flops / avg_ns / 1e3
                k["frac_of_fp32_mfma_peak"] = flops / avg_ns / 1e3 / 157.3
            out["kernels"].append(k)
            break
print(json.dumps(out, indent=1))
