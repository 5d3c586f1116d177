#!/bin/bash
for st in 128 20; do for cfg in "8 5" "10 10" "8 5" "10 10" "16 10"; do set -- $cfg
  timeout -k 10 300 python bench.py --steps $st --warmup $([ $st = 20 ] && echo 5 || echo 20) --no-cpu-baseline --no-configs --in-flight $1 --vit-batches $2 > gpurun_out/ab.json 2> gpurun_out/ab.err || { tail -3 gpurun_out/ab.err; continue; }
  python - "$st" "$1" "$2" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/ab.json").read().strip().splitlines()[-1])
print("steps %3s  batches per decode group %2s, per ViT launch %2s: %7.1f captions/s" % (sys.argv[1], sys.argv[2], sys.argv[3], d["value"]))
PY
done; done
