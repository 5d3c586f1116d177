# PMC + kernel time of k_vit_attention at 64 images (tools/microbench/attn_run.py): two rocprofv3 --pmc passes, never combined with trace domains
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/attn_pmc; mkdir -p $O
C1="SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"
C2="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_BUSY_CYCLES SQ_WAVES"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -- python3 $R/tools/microbench/attn_run.py 64 > $O/t.log 2>&1
timeout -k 10 200 rocprofv3 --pmc $C1 --kernel-trace --output-format csv -d $O/a -- python3 $R/tools/microbench/attn_run.py 64 > $O/a.log 2>&1
timeout -k 10 200 rocprofv3 --pmc $C2 --kernel-trace --output-format csv -d $O/b -- python3 $R/tools/microbench/attn_run.py 64 > $O/b.log 2>&1
cd $R
echo "== kernel time"; grep k_vit_attention $(find $O/t -name "*kernel_stats.csv" | head -1) | cut -c1-200
for p in a b; do f=$(find $O/$p -name "*counter_collection.csv" | head -1); [ -n "$f" ] && python3 tools/pmc_counters.py $f k_vit_attention $O/$p.json | grep -A1 "k_vit_attention" | head -4; done
find $O -name "*.csv" -delete
