cd /tmp && export TMPDIR=/tmp
for c in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES" "TCP_TCC_READ_REQ_sum TCC_REQ_sum"; do
  rm -rf /tmp/pmc; rocprofv3 --pmc $c -d /tmp/pmc -o p -- python /root/repo/scripts/bench_gemm.py > /dev/null 2>&1
  python /root/repo/scripts/rocpd_pmc.py $(ls /tmp/pmc/*/*.db /tmp/pmc/*.db 2>/dev/null | head -1) | grep -E "gemm_glds256_kernel"
done
