set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/c5
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/c5/trace -o run -- python3 $R/tools/bench_c5.py > $R/gpurun_out/c5/trace.log 2>&1
rocprofv3 --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_WAIT_ANY SQ_ACTIVE_INST_VALU -d $R/gpurun_out/c5/pmc1 -o run -- python3 $R/tools/bench_c5.py --reps 1 > $R/gpurun_out/c5/pmc1.log 2>&1
rocprofv3 --output-format csv --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $R/gpurun_out/c5/pmc2 -o run -- python3 $R/tools/bench_c5.py --reps 1 > $R/gpurun_out/c5/pmc2.log 2>&1
rocprofv3 --output-format csv --pmc FETCH_SIZE -d $R/gpurun_out/c5/pmc3 -o run -- python3 $R/tools/bench_c5.py --reps 1 > $R/gpurun_out/c5/pmc3.log 2>&1
rocprofv3 --output-format csv --pmc WRITE_SIZE -d $R/gpurun_out/c5/pmc4 -o run -- python3 $R/tools/bench_c5.py --reps 1 > $R/gpurun_out/c5/pmc4.log 2>&1
cd $R
python3 tools/pmc_summary.py trace gpurun_out/c5/trace > gpurun_out/c5/kernel_stats.csv
python3 tools/pmc_summary.py counters gpurun_out/c5/pmc1 gpurun_out/c5/pmc2 gpurun_out/c5/pmc3 gpurun_out/c5/pmc4 > gpurun_out/c5/pmc_counters.json
find gpurun_out/c5 -name "*.csv" -size +1M -delete; find gpurun_out/c5 -name "*.db" -delete
cat gpurun_out/c5/kernel_stats.csv; tail -1 gpurun_out/c5/trace.log; head -c 3000 gpurun_out/c5/pmc_counters.json
