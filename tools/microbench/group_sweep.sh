#!/bin/bash
for st in 20 128; do for p in 8 10 8 10 12 16; do
  timeout -k 10 300 python bench.py --steps $st --warmup $([ $st = 20 ] && echo 5 || echo 16) --no-cpu-baseline --no-configs --in-flight $p > gpurun_out/ab.json 2> gpurun_out/ab.err || { tail -3 gpurun_out/ab.err; continue; }
  python - "$st" "$p" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/ab.json").read().strip().splitlines()[-1])
print("steps %3s  batches per decode group %2s: %7.1f captions/s" % (sys.argv[1], sys.argv[2], d["value"]))
PY
done; done
