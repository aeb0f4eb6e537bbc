# per-kernel durations of the two-phase n = 64 solver (rocprofv3 kernel trace of tools/time_c2.py); LIBS as in ab_c2.sh
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
export NMAT=${NMAT:-131072} REPS=3
for lib in ${LIBS:-tree}; do
  if [ $lib = tree ]; then unset EGS_LIB_PATH; else export EGS_LIB_PATH=$R/tools/probe/lib_$lib.so; fi
  for kind in ${KINDS:-s5 randn}; do
    O=$R/gpurun_out/phases_${lib}_$kind
    rm -rf $O; mkdir -p $O
    KIND=$kind rocprofv3 --kernel-trace --stats --output-format csv -d $O -o run -- python3 $R/tools/time_c2.py > $O/log.txt 2>&1
    echo "== $lib $kind"
    python3 $R/tools/pmc_summary.py trace $O | grep -E "eig_real64|hqr_tail|Name" | cut -c1-200
    find $O -name "*.csv" -size +1M -delete
  done
done
