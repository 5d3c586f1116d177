# PMC + kernel time of the 32-query bank pass, split-fp16 GEMM2 against the exact fp32 form (PIO_PROJECT_EXACT is read when the bank is set)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3p; mkdir -p $O
C1="SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"
C2="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_BUSY_CYCLES SQ_WAVES"
for v in split exact; do
  unset PIO_PROJECT_EXACT
  [ $v = exact ] && export PIO_PROJECT_EXACT=1
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${v}_t -- python3 $R/tools/microbench/project_run.py > $O/${v}_t.log 2>&1
  timeout -k 10 200 rocprofv3 --pmc $C1 --kernel-trace --output-format csv -d $O/${v}_a -- python3 $R/tools/microbench/project_run.py > $O/${v}_a.log 2>&1
  timeout -k 10 200 rocprofv3 --pmc $C2 --kernel-trace --output-format csv -d $O/${v}_b -- python3 $R/tools/microbench/project_run.py > $O/${v}_b.log 2>&1
done
cd $R
for v in split exact; do
  echo "== $v: kernel time"; grep k_project2 $(find $O/${v}_t -name "*kernel_stats.csv" | head -1) | cut -c1-200
  for p in a b; do f=$(find $O/${v}_$p -name "*counter_collection.csv" | head -1); echo "== $v $p"; [ -n "$f" ] && python3 tools/pmc_counters.py $f k_project $O/${v}_$p.json | grep -A1 "k_project" | grep -v combine | head -6; done
done
find $O -name "*.csv" -delete
