#!/bin/bash
# A/B of the working tree's library against tools/probe/lib_base.so on the fused n = 64 kernel (run under gpurun).
N=${NMAT:-262144}
for r in 1 2; do
  NMAT=$N python tools/check_fused.py > gpurun_out/ab_new$r.json 2>/dev/null
  EGS_LIB_PATH=$PWD/tools/probe/lib_base.so NMAT=$N python tools/check_fused.py > gpurun_out/ab_base$r.json 2>/dev/null
done
python - <<'PY'
import json
for f in ("new1", "base1", "new2", "base2"):
    d = json.load(open("gpurun_out/ab_%s.json" % f))
    print(f, round(d["eig_ms_fused"], 2), round(d["eig_ms_two_kernels"], 2), d["dlambda"], d["info_fused"])
PY
