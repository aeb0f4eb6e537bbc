#!/bin/bash
# A/B of run-time switches on the fused n = 64 solver (tools/time_c2.py), interleaved inside one gpurun call:
#   ENVS="- EGS_TAIL_SINGLE=1 EGS_EIG_ONE_PHASE=1" tools/ab_env.sh      ("-" = no switch)
export NMAT=${NMAT:-131072} REPS=${REPS:-4}
for kind in ${KINDS:-s5 randn}; do
  for r in 1 2; do
    for e in ${ENVS:-"-"}; do
      if [ "$e" = "-" ]; then pre=""; else pre="$e"; fi
      env $pre KIND=$kind python tools/time_c2.py 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-22s' % '$e', d['kind'], round(d['ms'],2), 'ms', round(d['spectra_per_s']/1e6,3), 'M/s', d['dlambda_lapack'], d['unconverged'])"
    done
  done
done
