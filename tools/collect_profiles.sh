#!/bin/bash
# Reproduces profiles/ on a 1-GPU MI355X box (run from the repo root; ~2 minutes):
#   tools/collect_profiles.sh r01
# -> profiles/<round>_kernel_stats_sync.csv       rocprofv3 --kernel-trace --stats of `bench.py --in-flight 1`
#    profiles/<round>_kernel_stats_pipelined.csv  ... of the default bench.py run
#    profiles/<round>_vit_gemm_pipelined_by_grid.csv  the same trace, k_vit_gemm launches split by grid size (16 images per launch in the
#                                                 synchronous / profiled regions, 64 in the pipelined timed region)
#    profiles/<round>_bench_line.json             the bench line (with cpu_baseline)
#    profiles/<round>_sq_counters.json            SQ / GRBM counters per kernel of the pipelined run (MFMA utilisation, waits, LDS conflicts)
#    profiles/traffic.json                        HBM bytes per launch from --pmc passes (FETCH_SIZE, WRITE_SIZE, each on the synchronous
#                                                 and on the pipelined run; never
#                                                 combined with other trace domains), summarised by tools/pmc_summary.py
set -e
round=${1:-r01}
R=$(pwd)
out=$R/gpurun_out/profiles_$round
mkdir -p $out/sync $out/pipe $out/pmc_f $out/pmc_w $out/pmc_pf $out/pmc_pw $out/pmc_sq
cd /tmp && export TMPDIR=/tmp
export PIO_BENCH_NO_160=1        # the 160-images-per-launch region of bench.py is profiled by its own run below (its launches would mix into the averages)
mkdir -p $out/pipe160
rocprofv3 --kernel-trace --stats --output-format csv -d $out/sync -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-configs --in-flight 1 > $out/sync/bench.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/pipe -- python3 $R/bench.py --no-cpu-baseline --no-configs > $out/pipe/bench.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/pipe160 -- python3 $R/bench.py --no-cpu-baseline --no-configs --vit-batches 10 > $out/pipe160/bench.log 2>&1
export PIO_BENCH_STAT_GROUPS=4 PIO_BENCH_SYNC_STEPS=4      # counter passes serialise every dispatch: keep the statistics regions short
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/pmc_f -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-configs --in-flight 1 > $out/pmc_f/run.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/pmc_w -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-configs --in-flight 1 > $out/pmc_w/run.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/pmc_pf -- python3 $R/bench.py --steps 8 --warmup 8 --no-cpu-baseline --no-configs > $out/pmc_pf/run.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/pmc_pw -- python3 $R/bench.py --steps 8 --warmup 8 --no-cpu-baseline --no-configs > $out/pmc_pw/run.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $out/pmc_sq -- python3 $R/bench.py --steps 8 --warmup 8 --no-cpu-baseline --no-configs > $out/pmc_sq/run.log 2>&1
unset PIO_BENCH_STAT_GROUPS PIO_BENCH_SYNC_STEPS PIO_BENCH_NO_160
cd $R
python3 tools/pmc_counters.py $(find $out/pmc_sq -name "*counter_collection.csv" | head -1) pio profiles/${round}_sq_counters.json > $out/pmc_sq/summary.txt
python3 tools/pmc_summary.py $(find $out/pmc_f -name "*counter_collection.csv" | head -1) $(find $out/pmc_w -name "*counter_collection.csv" | head -1) profiles/traffic.json $(find $out/pmc_pf -name "*counter_collection.csv" | head -1) $(find $out/pmc_pw -name "*counter_collection.csv" | head -1) profiles/${round}_sq_counters.json | tail -2
find $out -name "*counter_collection.csv" -size +8M -delete
python3 bench.py > $out/bench_line.json 2> $out/bench_line.err
cp $(find $out/sync -name "*kernel_stats.csv" | head -1) profiles/${round}_kernel_stats_sync.csv
cp $(find $out/pipe -name "*kernel_stats.csv" | head -1) profiles/${round}_kernel_stats_pipelined.csv
python3 tools/trace_by_grid.py $(find $out/pipe -name "*kernel_trace.csv" | head -1) k_vit_gemm profiles/${round}_vit_gemm_pipelined_by_grid.csv
python3 tools/trace_by_grid.py $(find $out/pipe160 -name "*kernel_trace.csv" | head -1) k_vit_gemm profiles/${round}_vit_gemm_pipelined_160_by_grid.csv
cp $(find $out/pipe160 -name "*kernel_stats.csv" | head -1) profiles/${round}_kernel_stats_pipelined_160.csv
tail -1 $out/bench_line.json > profiles/${round}_bench_line.json
# gpurun merges at most 64 MiB of gpurun_out/ back: keep the summaries (and the logs), drop the raw traces
mkdir -p $R/gpurun_out/profiles_out
cp profiles/${round}_* profiles/traffic.json $R/gpurun_out/profiles_out/
find $out -name "*.csv" -delete
