# Address-translation counters of the ring GEMMs (one group per pass, --pmc only) under the native harness
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 $R/scripts/exp/gemm_bench.cpp -o /tmp/gemm_bench -I $R/include -L $R/keras_rs_amd -lkrs_hip -Wl,-rpath,$R/keras_rs_amd 2>/dev/null
: > $R/gpurun_out/pmc_gemm_tlb.txt
for c in "TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_sum" "TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum TCP_UTCL1_THRASHING_STALL_sum" "TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_STALL_MULTI_MISS_sum TCP_UTCL1_SERIALIZATION_STALL_sum" "GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE"; do
  rm -rf /tmp/pmc; rocprofv3 --pmc $c -d /tmp/pmc -o p -- /tmp/gemm_bench 1 > /dev/null 2>&1
  echo "counters=[$c]" >> $R/gpurun_out/pmc_gemm_tlb.txt
  python $R/scripts/rocpd_pmc.py $(ls /tmp/pmc/*/*.db /tmp/pmc/*.db 2>/dev/null | head -1) | grep -E "gemm_pp256|cross_bwd" >> $R/gpurun_out/pmc_gemm_tlb.txt
done
cat $R/gpurun_out/pmc_gemm_tlb.txt
