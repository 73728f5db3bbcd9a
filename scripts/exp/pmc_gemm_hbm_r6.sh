# Round 6: HBM-side counters of the FeatureCross products on the kernels that ship (separate --pmc passes, as the guide prescribes)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6_gemm_hbm; mkdir -p $O
cd $R
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 scripts/exp/gemm_bench.cpp -o /tmp/gemm_bench -I include -L keras_rs_amd -lkrs_hip -Wl,-rpath,$R/keras_rs_amd 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
: > $O/gemm_hbm_pmc.txt
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  rm -rf /tmp/pmc; rocprofv3 --pmc $c -d /tmp/pmc -o p -- /tmp/gemm_bench 1 > /tmp/pmc.log 2>&1
  echo "counters=[$c]" >> $O/gemm_hbm_pmc.txt
  python $R/scripts/rocpd_pmc.py $(ls /tmp/pmc/*/*.db /tmp/pmc/*.db 2>/dev/null | head -1) | grep -E "gemm_pp256|gemm_pp64" >> $O/gemm_hbm_pmc.txt
done
cat $O/gemm_hbm_pmc.txt
