#!/bin/bash
# round 5: two ViT streams side by side on half-chip persistent GEMM grids (the CUs of the two streams out of phase) against one full-chip stream
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
export PIO_BENCH_NO_160=1 PIO_BENCH_STAT_GROUPS=24 PIO_BENCH_SYNC_STEPS=8
: > gpurun_out/r5_two_streams.log
run() {
  local label=$1; shift
  local envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done
  shift
  env "${envs[@]}" timeout -k 10 280 python3 bench.py --no-cpu-baseline --no-configs "$@" 2> gpurun_out/sweep_err.log | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('%-58s value %7.0f  group median %.2f ms  gemm in pipe %.1f us frac %.3f' % ('$label', d['value'], d['pipelined_groups']['ms_per_group']['median'], d['roofline_in_pipeline']['avg_launch_us'], d['roofline_in_pipeline']['frac']))" >> gpurun_out/r5_two_streams.log || { tail -5 gpurun_out/sweep_err.log >> gpurun_out/r5_two_streams.log; return 1; }
}
run "one stage stream (default)" -- &&
run "two stage streams, full-chip grids" -- --stage-streams 2 &&
run "two stage streams, persistent GEMMs on 128 workgroups" PIO_ROLL_MAX_GRID=128 -- --stage-streams 2 &&
run "two stage streams, 128 workgroups, 10 batches per launch" PIO_ROLL_MAX_GRID=128 -- --stage-streams 2 --vit-batches 10 &&
run "one stage stream again" --
cat gpurun_out/r5_two_streams.log
