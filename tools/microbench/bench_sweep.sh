#!/bin/bash
# A/B of bench.py settings on one box (values differ by up to 12 % between boxes, so compare within one call):
#   tools/microbench/bench_sweep.sh  ->  gpurun_out/sweep.log  (one line per setting: captions/s, median ms per group)
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
export PIO_BENCH_NO_160=1 PIO_BENCH_STAT_GROUPS=24 PIO_BENCH_SYNC_STEPS=4
run() {   # label, env assignments..., -- bench args...
  local label=$1; shift
  local envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done
  shift
  env "${envs[@]}" timeout -k 10 240 python3 bench.py --no-cpu-baseline --no-configs "$@" 2> gpurun_out/sweep_err.log | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('%-34s value %7.0f  group median %.2f ms p95 %.2f  gemm %.1f us frac %.3f' % ('$label', d['value'], d['pipelined_groups']['ms_per_group']['median'], d['pipelined_groups']['ms_per_group']['p95'], d['roofline']['avg_launch_us'], d['roofline']['frac']))" >> gpurun_out/sweep.log || return 1
}
: > gpurun_out/sweep.log
run "default (5 per launch, 3 decodes)" -- &&
run "decodes 2" -- --decode-streams 2 &&
run "decodes 4" -- --decode-streams 4 &&
run "decodes 5" -- --decode-streams 5 &&
run "4 per launch" -- --vit-batches 4 &&
run "8 per launch (128 images)" -- --vit-batches 8 &&
run "10 per launch (160 images)" -- --vit-batches 10 &&
run "10 per launch, 10 batches per decode (160 prefixes)" -- --vit-batches 10 --in-flight 10 &&
run "default again" -- &&
run "6 batches per decode" -- --in-flight 6 &&
run "4 batches per decode" -- --in-flight 4 &&
run "no rolling gemm" PIO_GEMM_ROLL_MIN_TILES=1000000 -- &&
run "residual GEMMs on the 128-wide kernel" PIO_GEMM_RRES_MIN_TILES=0 -- &&
run "exact fp32 projection" PIO_PROJECT_EXACT=1 --
cat gpurun_out/sweep.log
