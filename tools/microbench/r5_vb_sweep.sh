#!/bin/bash
# round 5: images per ViT launch (80 / 128 / 160) with the rolling residual GEMMs, one box:  -> gpurun_out/r5_vb_sweep.log
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
export PIO_BENCH_NO_160=1 PIO_BENCH_STAT_GROUPS=24 PIO_BENCH_SYNC_STEPS=20
: > gpurun_out/r5_vb_sweep.log
run() {
  local label=$1; shift
  local envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done
  shift
  env "${envs[@]}" timeout -k 10 240 python3 bench.py --no-cpu-baseline --no-configs "$@" 2> gpurun_out/sweep_err.log | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('%-40s value %7.0f  sync %6.0f  group median %.2f ms  gemm alone %.1f us frac %.3f | in pipe %.1f us frac %.3f' % ('$label', d['value'], d['forward_sync']['value'], d['pipelined_groups']['ms_per_group']['median'], d['roofline']['avg_launch_us'], d['roofline']['frac'], d['roofline_in_pipeline']['avg_launch_us'], d['roofline_in_pipeline']['frac']))" >> gpurun_out/r5_vb_sweep.log || { tail -5 gpurun_out/sweep_err.log >> gpurun_out/r5_vb_sweep.log; return 1; }
}
run "5 per launch (80 images)" -- --vit-batches 5 &&
run "10 per launch (160 images)" -- --vit-batches 10 &&
run "8 per launch (128 images)" -- --vit-batches 8 &&
run "5 per launch again" -- --vit-batches 5 &&
run "10 per launch again" -- --vit-batches 10 &&
run "10 per launch, 20 steps" -- --vit-batches 10 --steps 20 --warmup 5 &&
run "5 per launch, 20 steps" -- --vit-batches 5 --steps 20 --warmup 5 &&
run "5 per launch, residual on 128 kernel" PIO_GEMM_RRES_MIN_TILES=0 -- --vit-batches 5
cat gpurun_out/r5_vb_sweep.log
