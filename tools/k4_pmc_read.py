import json
d=json.load(open("gpurun_out/k4prof/pmc_counters.json"))
for k,c in d.items():
    if "softmax_nu" in k or "qk_proj" in k:
        n=c["launches"]; gui=c["GRBM_GUI_ACTIVE"]/8/n
        print(k, "cycles/launch %.2fM"%(gui/1e6), "mfma busy %.2f"%(c["SQ_VALU_MFMA_BUSY_CYCLES"]/n/1024/gui), "waves/simd %.2f"%(c["SQ_WAVE_CYCLES"]*4/n/1024/gui), "wait_any %.2f"%(c["SQ_WAIT_ANY"]/c["SQ_WAVE_CYCLES"]), "wait_inst %.2f"%(c["SQ_WAIT_INST_ANY"]/c["SQ_WAVE_CYCLES"]), "valu/wave", round(c["SQ_INSTS_VALU"]/c["SQ_WAVES"]), "lds active %.2f"%(c["SQ_LDS_IDX_ACTIVE"]/n/256/gui))
