# rocprofv3 passes over the driver's bench command (default sizes: one fused launch of 524288 matrices per step; only the
# CPU baseline is skipped):
# kernel trace + stats, SQ counters, FETCH_SIZE, WRITE_SIZE.  Summaries land in gpurun_out/benchprof/.
set -e
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/benchprof
mkdir -p $O
CMD="python3 $R/bench.py --no-cpu"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o run -- $CMD > $O/trace.log 2>&1
rocprofv3 --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_SALU -d $O/pmc1 -o run -- $CMD > $O/pmc1.log 2>&1
rocprofv3 --output-format csv --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $O/pmc2 -o run -- $CMD > $O/pmc2.log 2>&1
rocprofv3 --output-format csv --pmc FETCH_SIZE -d $O/pmc3 -o run -- $CMD > $O/pmc3.log 2>&1
rocprofv3 --output-format csv --pmc WRITE_SIZE -d $O/pmc4 -o run -- $CMD > $O/pmc4.log 2>&1
cd $R
python3 tools/pmc_summary.py trace $O/trace > $O/kernel_stats.csv
python3 tools/pmc_summary.py counters $O/pmc1 $O/pmc2 $O/pmc3 $O/pmc4 > $O/pmc_counters.json
grep -h '^{"metric"' $O/trace.log > $O/bench_line.json || true
find $O -name "*.csv" -size +1M -delete
cat $O/kernel_stats.csv
