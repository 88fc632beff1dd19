mkdir -p gpurun_out/pmc3; export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_MFMA" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  (timeout 60 rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/pmc3/s$i -o p -- python tools/attn_probe.py 2 > gpurun_out/pmc3/s$i.log 2>&1)
done
