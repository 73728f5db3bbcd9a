# Round 4: native A/B incl. the fused product, then HBM-side counters of every ring GEMM form (separate --pmc passes)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4p; mkdir -p $O
cd $R
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 scripts/exp/gemm_bench.cpp -o /tmp/gemm_bench -I include -L keras_rs_amd -lkrs_hip -Wl,-rpath,$R/keras_rs_amd 2>/dev/null
/tmp/gemm_bench 7 > $O/gemm_ab_with_fused.txt 2>&1
cd /tmp && export TMPDIR=/tmp
: > $O/gemm_pmc.txt
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  rm -rf /tmp/pmc; rocprofv3 --pmc $c -d /tmp/pmc -o p -- /tmp/gemm_bench 1 > /dev/null 2>&1
  echo "counters=[$c]" >> $O/gemm_pmc.txt
  python $R/scripts/rocpd_pmc.py $(ls /tmp/pmc/*/*.db /tmp/pmc/*.db 2>/dev/null | head -1) | grep -E "gemm_pp256|cross_bwd" >> $O/gemm_pmc.txt
done
cat $O/gemm_ab_with_fused.txt | tail -11; cat $O/gemm_pmc.txt
