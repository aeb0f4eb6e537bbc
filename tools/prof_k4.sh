# rocprofv3 passes over the softmax-eta op at the MQAR batch shape (tools/bench_softmax_eta.py 0): kernel trace + two SQ
# counter passes.  Summaries land in gpurun_out/k4prof/.
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/k4prof
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
CMD="python3 $R/tools/bench_softmax_eta.py ${1:-0}"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o run -- $CMD > $O/trace.log 2>&1
rocprofv3 --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_SALU -d $O/pmc1 -o run -- $CMD > $O/pmc1.log 2>&1
rocprofv3 --output-format csv --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE SQ_WAVES -d $O/pmc2 -o run -- $CMD > $O/pmc2.log 2>&1
cd $R
python3 tools/pmc_summary.py trace $O/trace > $O/kernel_stats.csv
python3 tools/pmc_summary.py counters $O/pmc1 $O/pmc2 > $O/pmc_counters.json
find $O -name "*.csv" -size +1M -delete
cat $O/kernel_stats.csv
cat $O/pmc_counters.json
