"""From a rocprofv3 --kernel-trace CSV of a synchronous bench run: average duration of the N = 768 residual GEMM launches split into
attn.proj (the launch that follows k_vit_attention) and mlp.fc2 (the one that follows the GELU GEMM) -- same kernel, same grid.
python tools/microbench/trace_proj_fc2.py <kernel_trace.csv>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
prev = None
acc = {}
for r in rows:
    n = r["Kernel_Name"]
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if "k_vit_gemm" in n and ("Li2E" in n):          # EPI_RESIDUAL
        key = "proj (after attention)" if prev and "k_vit_attention" in prev else ("fc2 (after fc1)" if prev and "k_vit_gemm" in prev else "other")
        acc.setdefault((key, n[:48]), []).append(d)
    if "pio" in n or "k_vit" in n:
        prev = n
for (k, n), v in sorted(acc.items()):
    v.sort()
    print("%-24s %-50s n=%5d  median %7.1f us  mean %7.1f" % (k, n, len(v), v[len(v) // 2], sum(v) / len(v)))
