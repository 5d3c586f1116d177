#!/bin/bash
# tools/microbench/prof_decode.sh VARIANT...: rocprofv3 kernel stats of decode_probe.py ${AB_N:-16} per library variant -> gpurun_out/prof_dec_<variant>.csv
R=$(pwd)
for v in "$@"; do
  if [ "$v" = default ]; then unset PIO_LIB_PATH; else export PIO_LIB_PATH=$R/tools/microbench/bin/libpio_$v.so; fi
  out=$R/gpurun_out/prof_dec_$v; rm -rf $out; mkdir -p $out
  (cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 $R/tools/microbench/decode_probe.py ${AB_N:-16} > $out/run.log 2>&1)
  cp $(find $out -name "*kernel_stats.csv" | head -1) $R/gpurun_out/prof_dec_$v.csv
  find $out -name "*.csv" -delete
  echo "== $v"; head -14 $R/gpurun_out/prof_dec_$v.csv | cut -c1-150
done
