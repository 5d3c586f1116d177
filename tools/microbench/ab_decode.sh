#!/bin/bash
# tools/microbench/ab_decode.sh VARIANT...  ("default" = the in-tree library): decode_probe (ms per 30-step greedy decode) per variant, ${AB_REPS:-2} alternations
for r in $(seq ${AB_REPS:-2}); do
for v in "$@"; do
  if [ "$v" = default ]; then unset PIO_LIB_PATH; else export PIO_LIB_PATH=$PWD/tools/microbench/bin/libpio_$v.so; fi
  echo "== $v"; timeout -k 10 200 python tools/microbench/decode_probe.py ${AB_N:-16 128} || exit 1
done
done
