#!/bin/bash
# tools/microbench/ab_env.sh "VARIANT ENV=VAL ..." ...   one bench run per quoted spec (VARIANT = default or a build_variant.sh name)
for spec in "$@"; do
  set -- $spec; v=$1; shift
  if [ "$v" = default ]; then unset PIO_LIB_PATH; else export PIO_LIB_PATH=$PWD/tools/microbench/bin/libpio_$v.so; fi
  env "$@" timeout -k 10 300 python bench.py --steps ${AB_STEPS:-40} --warmup 8 --no-cpu-baseline --no-configs > gpurun_out/ab.json 2> gpurun_out/ab.err || { tail -5 gpurun_out/ab.err; exit 1; }
  python - "$spec" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/ab.json").read().strip().splitlines()[-1])
print("%-44s pipelined %7.1f  sync %7.1f (%.3f ms)  %s" % (sys.argv[1], d["value"], d["forward_sync"]["value"], d["forward_sync"]["ms_per_forward"]["median"],
      {k: round(v["ms_per_step"], 3) for k, v in d["stages"].items() if k.startswith("vit")}))
PY
done
