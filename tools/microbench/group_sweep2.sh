#!/bin/bash
# decode-group size (batches) x ViT launch size (batches) at the driver's 20 timed steps and at 128: "steps P VB"
for spec in "20 8 5" "20 5 5" "20 4 4" "20 10 5" "20 6 6" "20 8 5" "128 5 5" "128 8 5"; do
  set -- $spec
  timeout -k 10 300 python bench.py --steps $1 --warmup $([ $1 = 20 ] && echo 5 || echo 16) --no-cpu-baseline --no-configs --in-flight $2 --vit-batches $3 > gpurun_out/ab.json 2> gpurun_out/ab.err || { tail -3 gpurun_out/ab.err; continue; }
  python - "$@" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/ab.json").read().strip().splitlines()[-1])
print("steps %3s  batches per decode group %2s  per ViT launch %s: %7.1f captions/s" % (sys.argv[1], sys.argv[2], sys.argv[3], d["value"]))
PY
done
