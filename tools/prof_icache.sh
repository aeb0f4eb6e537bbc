# instruction-cache and fetch counters of the n = 64 solver (one fused launch of 131072 matrices), rocprofv3 PMC passes
set -e
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/icache
mkdir -p $O
rocprofv3 -L > $O/avail.txt 2>&1 || true
grep -i -o "SQC_ICACHE[A-Z_]*\|SQ_IFETCH[A-Z_]*\|SQ_INST_LEVEL[A-Z_]*\|SQ_WAIT_INST[A-Z_]*\|SQC_INST[A-Z_]*" $O/avail.txt | sort -u > $O/names.txt || true
cat $O/names.txt
export NMAT=131072 REPS=3 KIND=s5
rocprofv3 --output-format csv --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES -d $O/p1 -o run -- python3 $R/tools/time_c2.py > $O/p1.log 2>&1 || tail -5 $O/p1.log
rocprofv3 --output-format csv --pmc SQ_IFETCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY -d $O/p2 -o run -- python3 $R/tools/time_c2.py > $O/p2.log 2>&1 || tail -5 $O/p2.log
cd $R
python3 tools/pmc_summary.py counters $O/p1 $O/p2 > $O/icache_counters.json || true
find $O -name "*.csv" -size +1M -delete
cat $O/icache_counters.json | head -60
