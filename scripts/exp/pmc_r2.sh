# Round-2 PMC passes (one counter group per run, --pmc only: no tracing flags).
#  (a) the FeatureCross GEMMs / elementwise pass through the native harness scripts/exp/gemm_bench (all pipelines)
#  (b) K1 in its two forms through scripts/bench_k1.py
# Output: gpurun_out/r2_pmc_gemm.txt, gpurun_out/r2_pmc_k1.txt (per-kernel averages per counter)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
: > $R/gpurun_out/r2_pmc_gemm.txt
for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA SQ_BUSY_CYCLES"; do
  rm -rf /tmp/pmc; rocprofv3 --pmc $c -d /tmp/pmc -o p -- $R/scripts/exp/gemm_bench 1 > /dev/null 2>&1
  python $R/scripts/rocpd_pmc.py $(ls /tmp/pmc/*/*.db /tmp/pmc/*.db 2>/dev/null | head -1) | grep -E "gemm_|cross_" >> $R/gpurun_out/r2_pmc_gemm.txt
done
: > $R/gpurun_out/r2_pmc_k1.txt
for mode in "" "--multihot"; do
  for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    rm -rf /tmp/pmc; rocprofv3 --pmc $c -d /tmp/pmc -o p -- python $R/scripts/bench_k1.py $mode --iters 5 > /dev/null 2>&1
    python $R/scripts/rocpd_pmc.py $(ls /tmp/pmc/*/*.db /tmp/pmc/*.db 2>/dev/null | head -1) | grep -E "embed_" >> $R/gpurun_out/r2_pmc_k1.txt
  done
done
