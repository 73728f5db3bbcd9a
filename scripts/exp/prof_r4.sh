# Round-4 evidence run (one gpurun call): files land in gpurun_out/$TAG/ and are copied to profiles/ afterwards.
#   usage: bash scripts/exp/prof_r4.sh TAG [full]
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r4z}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
# 1. the default line (metric, roofline, roofline_step, cpu_baseline)
python bench.py > $O/bench_c3.json 2> $O/bench_c3.err
# 2. the same command under the kernel trace
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof; rocprofv3 --kernel-trace --stats -d /tmp/prof -o b -- python $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline > $O/bench_c3_profiled.json 2>/dev/null
python $R/scripts/rocpd_stats.py $(ls /tmp/prof/*/*.db /tmp/prof/*.db 2>/dev/null | head -1) 50 > $O/bench_c3_kernel_stats.md
cd $R
# 3. whole model + host-resident inputs
python bench.py --full-model --host-inputs 2 --no-cpu-baseline > $O/bench_c3_full_model.json 2>/dev/null
# 4. sharded code path at the per-rank batch of N = 8: static exchange with / without prefetch, one-rank RCCL communicator
python bench.py --force-sharded --batch 8192 --no-cpu-baseline > $O/sharded_b8192_static.json 2>/dev/null
python bench.py --force-sharded --batch 8192 --no-cpu-baseline --no-prefetch > $O/sharded_b8192_static_noprefetch.json 2>/dev/null
python bench.py --force-sharded --batch 8192 --no-cpu-baseline --rccl-self 2>/dev/null | grep '^{' > $O/sharded_b8192_rccl_one_rank.json
python bench.py --force-sharded --batch 8192 --no-cpu-baseline --rccl-self --no-prefetch 2>/dev/null | grep '^{' > $O/sharded_b8192_rccl_one_rank_noprefetch.json
# 5. native GEMM A/B
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 scripts/exp/gemm_bench.cpp -o /tmp/gemm_bench -I include -L keras_rs_amd -lkrs_hip -Wl,-rpath,$R/keras_rs_amd 2>/dev/null
/tmp/gemm_bench 7 > $O/gemm_ab.txt 2>&1
if [ "$2" = full ]; then
  cd /tmp; rm -rf /tmp/prof3; rocprofv3 --kernel-trace --stats -d /tmp/prof3 -o b -- python $R/bench.py --full-model --steps 8 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
  python $R/scripts/rocpd_stats.py $(ls /tmp/prof3/*/*.db /tmp/prof3/*.db 2>/dev/null | head -1) 60 > $O/full_model_kernel_stats.md
  cd $R
  python bench.py --force-sharded --batch 8192 --no-cpu-baseline --rccl-self --graph 2>/dev/null | grep '^{' > $O/graph_replay_sharded_b8192_rccl_one_rank.json
  python bench.py --force-sharded --batch 8192 --no-cpu-baseline --exchange exact > $O/sharded_b8192_exact.json 2>/dev/null
  cd /tmp; rm -rf /tmp/prof2; rocprofv3 --kernel-trace --stats -d /tmp/prof2 -o b -- python $R/bench.py --force-sharded --batch 8192 --steps 8 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
  python $R/scripts/rocpd_stats.py $(ls /tmp/prof2/*/*.db /tmp/prof2/*.db 2>/dev/null | head -1) 40 > $O/sharded_b8192_kernel_stats.md
  cd $R
  python bench.py --gpus 2 --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | grep '^{' > $O/bench_gpus2_bare_command.json
  python bench.py --criteo-vocab 1000000 --no-cpu-baseline > $O/bench_c3prime.json 2>/dev/null
  python bench.py --criteo-vocab 40000000 --id-skew 4 --no-cpu-baseline > $O/bench_c5_powerlaw.json 2>/dev/null
  # K1 / K2 standalone + counters (separate --pmc passes)
  python scripts/bench_k1.py --multihot > $O/k1_k2_multihot.txt 2>&1
  python scripts/bench_k1.py > $O/k1_k2_hot1.txt 2>&1
  cd /tmp
  : > $O/k1_k2_pmc.txt
  for mode in "--multihot" ""; do
    for c in FETCH_SIZE WRITE_SIZE; do
      rm -rf /tmp/pmc; rocprofv3 --pmc $c -d /tmp/pmc -o p -- python $R/scripts/bench_k1.py $mode --iters 5 > /dev/null 2>&1
      echo "mode=[$mode] counters=[$c]" >> $O/k1_k2_pmc.txt
      python $R/scripts/rocpd_pmc.py $(ls /tmp/pmc/*/*.db /tmp/pmc/*.db 2>/dev/null | head -1) | grep -E "bag_apply|embed_bag_fwd_vec|embed_gather_hot1|scatter_seg" >> $O/k1_k2_pmc.txt
    done
  done
fi
ls -la $O
