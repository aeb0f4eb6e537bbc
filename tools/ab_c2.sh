#!/bin/bash
# A/B of libraries on the fused n = 64 kernel, benchmark (s5) and Gaussian matrices, interleaved inside one gpurun call
# (the clock drifts between calls).  LIBS = space-separated names under tools/probe/lib_<name>.so; "tree" = the working tree.
export NMAT=${NMAT:-131072} REPS=${REPS:-4}
LIBS=${LIBS:-"tree base"}
for kind in ${KINDS:-s5 randn}; do
  for r in 1 2; do
    for lib in $LIBS; do
      if [ $lib = tree ]; then unset EGS_LIB_PATH; else export EGS_LIB_PATH=$PWD/tools/probe/lib_$lib.so; fi
      KIND=$kind python tools/time_c2.py 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-10s' % '$lib', d['kind'], round(d['ms'],2), 'ms', round(d['spectra_per_s']/1e6,3), 'M/s', d['dlambda_lapack'], d['unconverged'])"
    done
  done
done
