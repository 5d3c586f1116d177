#!/bin/bash
# tools/microbench/ab_bench.sh VARIANT...   ("default" = the in-tree library) -> one summary line per variant and mode
for v in "$@"; do
  if [ "$v" = default ]; then unset PIO_LIB_PATH; else export PIO_LIB_PATH=$PWD/tools/microbench/bin/libpio_$v.so; fi
  timeout -k 10 300 python bench.py --steps ${AB_STEPS:-48} --warmup 8 --no-cpu-baseline --no-configs --in-flight ${AB_INFLIGHT:-4} ${AB_EXTRA} > gpurun_out/ab.json 2> gpurun_out/ab.err || { tail -5 gpurun_out/ab.err; exit 1; }
  python - "$v" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/ab.json").read().strip().splitlines()[-1])
print("%-10s pipelined %7.1f capt/s (%.3f ms)  sync %7.1f (%.3f ms)  %s" % (sys.argv[1], d["value"], d["ms_per_step"], d["sync"]["value"], d["sync"]["ms_per_step"],
      {k: round(v["ms_per_step"], 3) for k, v in d["stages"].items()}))
PY
done
