# Round 6: HBM traffic of K1 / K2 per launch re-read on the round's final code (separate --pmc passes over scripts/bench_k1.py),
# plus the kernel trace of the same script for the launch durations
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6_k1k2; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
: > $O/k1_k2_pmc.txt
for mode in "--multihot" ""; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc; rocprofv3 --pmc $c -d /tmp/pmc -o p -- python $R/scripts/bench_k1.py $mode --iters 5 > /dev/null 2>&1
    echo "mode=[$mode] counter=$c" >> $O/k1_k2_pmc.txt
    python $R/scripts/rocpd_pmc.py $(ls /tmp/pmc/*/*.db /tmp/pmc/*.db 2>/dev/null | head -1) | grep -E "bag_apply|embed_bag_fwd_vec|embed_gather_hot1|scatter_seg" >> $O/k1_k2_pmc.txt
  done
  rm -rf /tmp/kt; rocprofv3 --kernel-trace --stats -d /tmp/kt -o k -- python $R/scripts/bench_k1.py $mode --iters 20 > /dev/null 2>&1
  KRS_STATS_FULL_NAMES=1 python $R/scripts/rocpd_stats.py $(ls /tmp/kt/*/*.db /tmp/kt/*.db 2>/dev/null | head -1) 12 > $O/k1_k2_kernel_stats${mode// /_}.md
done
cat $O/k1_k2_pmc.txt; head -8 $O/k1_k2_kernel_stats--multihot.md
