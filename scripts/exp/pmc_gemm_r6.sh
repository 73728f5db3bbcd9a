# Round 6: MFMA duty cycle and effective clock of the six FeatureCross products (verdict r5, next #3): one A/B pass of
# scripts/exp/gemm_bench, then separate rocprofv3 --pmc passes (SQ: MFMA busy / wave cycles / stall buckets; GRBM: active cycles)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6_gemm_pmc; mkdir -p $O
cd $R
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 scripts/exp/gemm_bench.cpp -o /tmp/gemm_bench -I include -L keras_rs_amd -lkrs_hip -Wl,-rpath,$R/keras_rs_amd 2>&1 | tail -3
/tmp/gemm_bench 7 > $O/gemm_ab.txt 2>&1
cd /tmp && export TMPDIR=/tmp
: > $O/gemm_pmc.txt
for c in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS GRBM_GUI_ACTIVE" "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_MFMA SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ACTIVE_CYCLES"; do
  rm -rf /tmp/pmc; rocprofv3 --pmc $c -d /tmp/pmc -o p -- /tmp/gemm_bench 1 > /tmp/pmc.log 2>&1
  echo "counters=[$c]" >> $O/gemm_pmc.txt
  db=$(ls /tmp/pmc/*/*.db /tmp/pmc/*.db 2>/dev/null | head -1)
  if [ -z "$db" ]; then tail -5 /tmp/pmc.log >> $O/gemm_pmc.txt; else python $R/scripts/rocpd_pmc.py $db | grep -E "gemm_pp256|gemm_pp64|cross_bwd|slab_reduce" >> $O/gemm_pmc.txt; fi
done
rm -rf /tmp/kt; rocprofv3 --kernel-trace --stats -d /tmp/kt -o k -- /tmp/gemm_bench 1 > /dev/null 2>&1
python $R/scripts/rocpd_stats.py $(ls /tmp/kt/*/*.db /tmp/kt/*.db 2>/dev/null | head -1) > $O/gemm_kernel_stats.md 2>&1
tail -12 $O/gemm_ab.txt; cat $O/gemm_pmc.txt; head -20 $O/gemm_kernel_stats.md
