cd /tmp && export TMPDIR=/tmp
for c in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_SALU" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT" "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM" "SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS SQ_ACTIVE_INST_MISC"; do
  rm -rf /tmp/pmc; rocprofv3 --pmc $c -d /tmp/pmc -o p -- python /root/repo/scripts/bench_gemm.py > /dev/null 2>&1
  python /root/repo/scripts/rocpd_pmc.py $(ls /tmp/pmc/*/*.db /tmp/pmc/*.db 2>/dev/null | head -1) | grep -E "gemm_glds_kernel<2, 0>|gemm_mfma_kernel<2, true, false, 0>|gemm_mfma_kernel<2, false, true, 1>"
done
