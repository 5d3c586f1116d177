#!/bin/bash
# A/B: LayerNorm launches present / left out (TIMING ablation, wrong results), default bench at 128 steps, two alternations.
# Needs the diagnostic build:  tools/microbench/build_variant.sh abl -DPIO_ABLATIONS
mkdir -p gpurun_out
export PIO_LIB_PATH=$PWD/tools/microbench/bin/libpio_abl.so
for r in 1 2; do
for v in with_ln skip_ln; do
  if [ $v = skip_ln ]; then export PIO_ABL_SKIP_LN=1; else unset PIO_ABL_SKIP_LN; fi
  timeout -k 10 300 python bench.py --steps 128 --warmup 16 --no-cpu-baseline --no-configs > gpurun_out/ab.json 2> gpurun_out/ab.err || { tail -5 gpurun_out/ab.err; exit 1; }
  python - "$v" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/ab.json").read().strip().splitlines()[-1])
print("%-8s pipelined %7.1f capt/s   forward_sync %7.1f capt/s (%.3f ms)   stages %s" % (sys.argv[1], d["value"], d["forward_sync"]["value"], d["forward_sync"]["ms_per_forward"]["median"],
      {k: round(v["ms_per_step"], 3) for k, v in d["stages"].items()}))
PY
done
done
