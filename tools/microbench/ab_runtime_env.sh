#!/bin/bash
# tools/microbench/ab_runtime_env.sh : the 30-step decode (decode_probe, ms per decode at 16 / 128 prefixes) under HIP runtime settings
# that touch graph launches, packet fences and signals.  One line per setting; "-" = the library's own configuration.
for spec in "-" "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0" "DEBUG_CLR_GRAPH_PACKET_CAPTURE=1" "AMD_OPT_FLUSH=0" "AMD_OPT_FLUSH=1" "ROC_SYSTEM_SCOPE_SIGNAL=0" \
            "DEBUG_HIP_GRAPH_BATCH_SIZE=1024" "DEBUG_HIP_FORCE_GRAPH_QUEUES=1" "ROC_ACTIVE_WAIT_TIMEOUT=100" "HIP_FORCE_DEV_KERNARG=0" "ROC_USE_FGS_KERNARG=0" "-"; do
  if [ "$spec" = "-" ]; then out=$(timeout -k 10 120 python tools/microbench/decode_probe.py 16 128 2>/dev/null | tr '\n' ' ')
  else out=$(env $spec timeout -k 10 120 python tools/microbench/decode_probe.py 16 128 2>/dev/null | tr '\n' ' '); fi
  printf "%-40s %s\n" "$spec" "$out"
done
