# rocprofv3 passes over the fused linear-attention eta kernel (tools/run_k3_once.py): kernel trace + two SQ counter passes.
# Summaries land in gpurun_out/k3prof/.
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/k3prof
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
CMD="python3 $R/tools/run_k3_once.py"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o run -- $CMD > $O/trace.log 2>&1
rocprofv3 --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_SALU -d $O/pmc1 -o run -- $CMD > $O/pmc1.log 2>&1
rocprofv3 --output-format csv --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE SQ_WAVES -d $O/pmc2 -o run -- $CMD > $O/pmc2.log 2>&1
rocprofv3 --output-format csv --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_VMEM SQ_INST_CYCLES_VMEM SQ_INSTS_SMEM -d $O/pmc3 -o run -- $CMD > $O/pmc3.log 2>&1 || true
cd $R
python3 tools/pmc_summary.py trace $O/trace > $O/kernel_stats.csv
python3 tools/pmc_summary.py counters $O/pmc1 $O/pmc2 $O/pmc3 > $O/pmc_counters.json || python3 tools/pmc_summary.py counters $O/pmc1 $O/pmc2 > $O/pmc_counters.json
find $O -name "*.csv" -size +1M -delete
cat $O/kernel_stats.csv
python3 - <<'PY'
import json
d=json.load(open("gpurun_out/k3prof/pmc_counters.json"))
for k,c in d.items():
    if "linattn_fused" in k or "eta_finalize" in k:
        n=c["launches"]; gui=c["GRBM_GUI_ACTIVE"]/8/n
        print(k[:60], "cycles/launch %.2fM"%(gui/1e6), "mfma busy %.2f"%(c["SQ_VALU_MFMA_BUSY_CYCLES"]/n/1024/gui), "waves/simd %.2f"%(c["SQ_WAVE_CYCLES"]*4/n/1024/gui), "wait_any %.2f"%(c["SQ_WAIT_ANY"]/c["SQ_WAVE_CYCLES"]), "wait_inst %.2f"%(c["SQ_WAIT_INST_ANY"]/c["SQ_WAVE_CYCLES"]), "wait_lds %.2f"%(c["SQ_WAIT_INST_LDS"]/c["SQ_WAVE_CYCLES"]), "valu/wave", round(c["SQ_INSTS_VALU"]/c["SQ_WAVES"]), "lds/wave", round(c["SQ_INSTS_LDS"]/c["SQ_WAVES"]), "lds active %.2f"%(c["SQ_LDS_IDX_ACTIVE"]/n/256/gui), "bank conflict %.2f"%(c["SQ_LDS_BANK_CONFLICT"]/max(1,c["SQ_LDS_IDX_ACTIVE"])))
        print({kk:vv for kk,vv in c.items()})
PY
