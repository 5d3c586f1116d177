#!/bin/bash
# CU partitions between stage 1 (ViT + projection) and the decode chains, through bench.py (128 steps): "STAGE DECODE VB ROLL" per spec
for spec in "0 0 5 704" "0 0 4 0" "224 32 4 0" "208 48 4 0" "224 0 4 0" "192 64 4 0" "0 0 5 704"; do
  set -- $spec
  PIO_STAGE_CUS=$1 PIO_DECODE_CUS=$2 PIO_GEMM_ROLL_MIN_TILES=$4 timeout -k 10 300 python bench.py --steps 128 --warmup 16 --no-cpu-baseline --no-configs --vit-batches $3 > gpurun_out/ab.json 2> gpurun_out/ab.err || { tail -3 gpurun_out/ab.err; continue; }
  python - "$spec" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/ab.json").read().strip().splitlines()[-1])
print("stage CUs / decode CUs / batches per ViT launch / roll threshold = %-16s: %7.1f captions/s" % (sys.argv[1], d["value"]))
PY
done
