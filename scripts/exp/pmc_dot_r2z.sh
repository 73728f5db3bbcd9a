# PMC passes over the DotInteraction kernels of the final state (one counter group per run, --pmc only).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
: > $R/gpurun_out/r2z_pmc_dot.txt
for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA"; do
  for v in "" "KRS_DOT_BWD_VALU=1"; do
    rm -rf /tmp/pmc; env $v rocprofv3 --pmc $c -d /tmp/pmc -o p -- python $R/scripts/bench_dot.py > /dev/null 2>&1
    python $R/scripts/rocpd_pmc.py $(ls /tmp/pmc/*/*.db /tmp/pmc/*.db 2>/dev/null | head -1) | grep -E "dot_" >> $R/gpurun_out/r2z_pmc_dot.txt
  done
done
sort -u $R/gpurun_out/r2z_pmc_dot.txt | cut -c1-200
