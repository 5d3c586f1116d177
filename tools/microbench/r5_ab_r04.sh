#!/bin/bash
# round 5 against round 4's library (same Python, same box, alternating):  -> gpurun_out/r5_ab_r04.log
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
export PIO_BENCH_NO_160=1 PIO_BENCH_STAT_GROUPS=24 PIO_BENCH_SYNC_STEPS=60
: > gpurun_out/r5_ab_r04.log
run() {
  local label=$1; shift
  local envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done
  shift
  env "${envs[@]}" timeout -k 10 240 python3 bench.py --no-cpu-baseline --no-configs "$@" 2> gpurun_out/sweep_err.log | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
s=d['stages']
print('%-28s value %7.0f  sync %6.0f (%.3f ms)  gemm alone %.1f us frac %.3f | in pipe frac %.3f | sync frac %.3f | stages gemm %.3f attn %.3f ln %.3f proj %.3f dec %.3f' % ('$label', d['value'], d['forward_sync']['value'], d['forward_sync']['ms_per_forward']['median'], d['roofline']['avg_launch_us'], d['roofline']['frac'], d['roofline_in_pipeline']['frac'], d['roofline_sync']['frac'], s['vit_gemm']['ms_per_step'], s['vit_attention']['ms_per_step'], s['vit_layernorm']['ms_per_step'], s['mem_project']['ms_per_step'], s['decode']['ms_per_step']))" >> gpurun_out/r5_ab_r04.log || { tail -5 gpurun_out/sweep_err.log >> gpurun_out/r5_ab_r04.log; return 1; }
}
R04=$PWD/tools/microbench/bin/libpio_r04.so
run "round 4 library" PIO_LIB_PATH=$R04 -- &&
run "round 5" -- &&
run "round 4 library" PIO_LIB_PATH=$R04 -- &&
run "round 5" -- &&
run "round 4 library, 20 steps" PIO_LIB_PATH=$R04 -- --steps 20 --warmup 5 &&
run "round 5, 20 steps" -- --steps 20 --warmup 5
cat gpurun_out/r5_ab_r04.log
