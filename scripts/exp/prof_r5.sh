# Round-5 evidence runs (one gpurun call each).  Files land in gpurun_out/$TAG/ and are copied to profiles/ afterwards.
#   prof_r5.sh TAG suite      GPU test suite + default bench line
#   prof_r5.sh TAG sharded    the per-rank step of an 8-way job on one GPU (B_local = 8192): line + kernel trace
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r5a}
WHAT=${2:-suite}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
if [[ $WHAT == *suite* ]]; then
  timeout 900 python -m pytest tests -m gpu -x -q > $O/tests_gpu.log 2>&1; tail -3 $O/tests_gpu.log
fi
if [[ $WHAT == *bench* ]]; then
  timeout 600 python bench.py > $O/bench_c3.json 2> $O/bench_c3.err; tail -c 600 $O/bench_c3.err
fi
if [[ $WHAT == *sharded* ]]; then
  timeout 300 python bench.py --force-sharded --batch 8192 --no-cpu-baseline > $O/sharded_b8192_static.json 2> $O/sharded_b8192_static.err
  cd /tmp && export TMPDIR=/tmp
  rm -rf /tmp/prof; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof -o b -- python $R/bench.py --force-sharded --batch 8192 --steps 8 --warmup 2 --no-cpu-baseline > $O/sharded_b8192_profiled.json 2>/dev/null
  KRS_STATS_FULL_NAMES=1 python $R/scripts/rocpd_stats.py $(ls /tmp/prof/*/*.db /tmp/prof/*.db 2>/dev/null | head -1) 70 > $O/sharded_b8192_kernel_stats.md
  cd $R
fi
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], round(d["ms_per_step"], 3), d["step_stats"]["median_ms"], d.get("roofline", {}).get("frac"), d.get("host_enqueue_ms_per_step"))
    except Exception as e:
        print(f, "unreadable", e)
PY
if [[ $WHAT == *planab* ]]; then
  # A/B of the owner-side backward plan running ahead on a side stream (KRS_SHARD_PLAN_AHEAD), eager and through the one-rank RCCL communicator + graph leg
  for v in 1 0 1 0; do
    KRS_SHARD_PLAN_AHEAD=$v timeout 300 python bench.py --force-sharded --batch 8192 --no-cpu-baseline --sustained-steps 0 --probe-steps 0 --no-parity > $O/planab_$v.json 2>/dev/null
    python - <<PY
import json
d = json.loads(open("$O/planab_$v.json").read().strip().splitlines()[-1])
print("plan_ahead=$v", round(d["ms_per_step"], 3), d["step_stats"]["median_ms"], "L=1", round(d["also"]["ms_per_step"], 3), "host", round(d["host_enqueue_ms_per_step"], 3))
PY
  done
  timeout 400 python bench.py --force-sharded --rccl-self --batch 8192 --no-cpu-baseline > $O/sharded_b8192_rccl_one_rank.json 2> $O/sharded_b8192_rccl_one_rank.err
  python - <<PY
import json
d = json.loads(open("$O/sharded_b8192_rccl_one_rank.json").read().strip().splitlines()[-1])
print("rccl one rank: value leg", round(d["ms_per_step"], 3), "eager", (d.get("eager_leg") or {}).get("ms_per_step"), "graph", d.get("graph_leg"), "parity", d.get("parity"))
PY
fi
if [[ $WHAT == *newtests* ]]; then
  timeout 1200 python -m pytest tests/test_configs_gpu.py tests/test_bench_multiproc_gpu.py tests/test_full_size_properties_gpu.py tests/test_dense_ops_gpu.py tests/test_embed_bag_bwd_gpu.py "tests/test_layers_gpu.py::test_a_forward_without_a_backward_leaves_no_stale_bookkeeping" tests/test_graph_step_gpu.py tests/test_sharded_gpu.py -q -s -m gpu > $O/newtests.log 2>&1
  grep -E "passed|failed|do not cancel|table [0-9]+:|Error|error" $O/newtests.log | head -60
fi
if [[ $WHAT == *rccl1* ]]; then
  timeout 400 python bench.py --force-sharded --rccl-self --batch 8192 --no-cpu-baseline > $O/sharded_b8192_rccl_one_rank.json 2> $O/sharded_b8192_rccl_one_rank.err
  python - <<PY
import json
d = json.loads(open("$O/sharded_b8192_rccl_one_rank.json").read().strip().splitlines()[-1])
print("rccl one rank: value leg", round(d["ms_per_step"], 3), "eager", (d.get("eager_leg") or {}).get("ms_per_step"), "graph", {k: v for k, v in (d.get("graph_leg") or {}).items() if k != "step_stats"}, "parity ok", d["parity"]["ok"], d["parity"]["update_max_ulp"])
PY
fi
if [[ $WHAT == *tests2* ]]; then
  timeout 1200 python -m pytest tests/test_c5_full_gpu.py tests/test_keras_adapter.py tests/test_configs_gpu.py tests/test_bench_multiproc_gpu.py -q -s -m gpu --durations=8 > $O/tests2.log 2>&1
  grep -E "passed|failed|do not cancel|table [0-9]+:|Error|error|^[0-9.]+s " $O/tests2.log | head -60
fi
if [[ $WHAT == *stacks* ]]; then
  timeout 400 python scripts/prof_step_stacks.py > $O/aten_in_step_c3.md 2> $O/aten_in_step_c3.err; head -30 $O/aten_in_step_c3.md
  timeout 400 python scripts/prof_step_stacks.py --force-sharded --batch 8192 > $O/aten_in_step_sharded_b8192.md 2> $O/aten_in_step_sharded.err; head -40 $O/aten_in_step_sharded_b8192.md
fi
if [[ $WHAT == *k2split* ]]; then
  cd /tmp && export TMPDIR=/tmp
  for sub in all hot100 hot1 mid; do
    python $R/scripts/exp/k2_split.py --subset $sub >> $O/k2_split.txt 2>/dev/null
    for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
      rm -rf /tmp/pmc; timeout 200 rocprofv3 --pmc $c --kernel-trace -d /tmp/pmc -o p -- python $R/scripts/exp/k2_split.py --subset $sub --iters 2 > /dev/null 2>&1
      echo "subset=$sub counters=[$c]" >> $O/k2_split.txt
      python $R/scripts/rocpd_pmc.py $(ls /tmp/pmc/*/*.db /tmp/pmc/*.db 2>/dev/null | head -1) | grep -E "bag_apply_fast_kernel" >> $O/k2_split.txt
    done
  done
  cd $R; cat $O/k2_split.txt
fi
if [[ $WHAT == *trace* ]]; then
  cd /tmp && export TMPDIR=/tmp
  rm -rf /tmp/prof; timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof -o b -- python $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-c2 --sustained-steps 0 > $O/bench_c3_profiled.json 2>/dev/null
  KRS_STATS_FULL_NAMES=1 python $R/scripts/rocpd_stats.py $(ls /tmp/prof/*/*.db /tmp/prof/*.db 2>/dev/null | head -1) 60 > $O/bench_c3_kernel_stats.md
  rm -rf /tmp/prof; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof -o k -- python $R/scripts/bench_k1.py --multihot --iters 20 > $O/k1_k2_multihot.txt 2>/dev/null
  KRS_STATS_FULL_NAMES=1 python $R/scripts/rocpd_stats.py $(ls /tmp/prof/*/*.db /tmp/prof/*.db 2>/dev/null | head -1) 30 > $O/k1_k2_standalone_kernel_stats.md
  rm -rf /tmp/prof; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof -o k -- python $R/scripts/bench_k1.py --iters 20 > $O/k1_k2_hot1.txt 2>/dev/null
  KRS_STATS_FULL_NAMES=1 python $R/scripts/rocpd_stats.py $(ls /tmp/prof/*/*.db /tmp/prof/*.db 2>/dev/null | head -1) 30 >> $O/k1_k2_standalone_kernel_stats.md
  cd $R
  head -12 $O/bench_c3_kernel_stats.md | cut -c1-200
fi
if [[ $WHAT == *full* ]]; then
  timeout 400 python bench.py --full-model --no-cpu-baseline --no-c2 --sustained-steps 0 > $O/bench_c3_full_model.json 2>/dev/null
  timeout 600 python bench.py --criteo-vocab 40000000 --id-skew 4 --no-c2 --sustained-steps 0 > $O/bench_c5_powerlaw.json 2>/dev/null
  python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
fi
if [[ $WHAT == *splitk* ]]; then
  timeout 600 python -m pytest tests/test_dense_ops_gpu.py -q -x -m gpu -k "split_k or ring_gemm or fused_cross" > $O/splitk_tests.log 2>&1; tail -3 $O/splitk_tests.log
  timeout 900 python -m pytest tests/test_layers_gpu.py tests/test_sharded_gpu.py tests/test_mlperf_model_gpu.py tests/test_graph_step_gpu.py -q -x -m gpu > $O/splitk_tests2.log 2>&1; tail -3 $O/splitk_tests2.log
fi
if [[ $WHAT == *plan* ]]; then
  timeout 900 python -m pytest tests/test_embed_bag_bwd_gpu.py tests/test_full_size_properties_gpu.py tests/test_layers_gpu.py -q -x -m gpu > $O/plan_tests.log 2>&1; tail -2 $O/plan_tests.log
  timeout 300 python scripts/bench_k1.py --multihot --iters 10 2>/dev/null | grep -E "plan_variant|k2_plan_us" | cut -c1-200
fi
if [[ $WHAT == *graphstress* ]]; then
  ok=0; bad=0
  for i in ${STRESS_RUNS:-1 2 3 4 5 6 7 8}; do
    if timeout 300 python bench.py --force-sharded --rccl-self --steps 3 --warmup 2 --no-cpu-baseline --batch 8192 --vocab 100000 --sustained-steps 0 --probe-steps 0 > $O/stress_$i.json 2> $O/stress_$i.err; then
      python - <<PY && ok=$((ok+1)) || bad=$((bad+1))
import json, sys
d = json.loads(open("$O/stress_$i.json").read().strip().splitlines()[-1])
g = d.get("graph_leg") or {}
print("run $i", round(d["ms_per_step"], 3), g.get("ok"), g.get("error"))
sys.exit(0 if g.get("ok") else 1)
PY
    else
      bad=$((bad+1)); echo "run $i rc!=0"; grep -m3 -iE "hip error|HIP error|capturing|what\(\)|Error" $O/stress_$i.err | cut -c1-300
    fi
  done
  echo "graph stress: ok=$ok bad=$bad"
fi
if [[ $WHAT == *pmck1* ]]; then
  cd /tmp && export TMPDIR=/tmp
  for mode in "--multihot" ""; do
    for c in FETCH_SIZE WRITE_SIZE; do
      rm -rf /tmp/pmc; timeout 200 rocprofv3 --pmc $c --kernel-trace -d /tmp/pmc -o p -- python $R/scripts/bench_k1.py $mode --iters 5 > /dev/null 2>&1
      echo "mode=[$mode] counters=[$c]" >> $O/k1_k2_pmc.txt
      python $R/scripts/rocpd_pmc.py $(ls /tmp/pmc/*/*.db /tmp/pmc/*.db 2>/dev/null | head -1) | grep -E "bag_apply_fast_kernel|embed_bag_fwd_vec|embed_gather_hot1|scatter_seg|hist_seg" >> $O/k1_k2_pmc.txt
    done
  done
  cd $R; cat $O/k1_k2_pmc.txt
fi
if [[ $WHAT == *virtual8* ]]; then
  timeout 300 python bench.py --force-sharded --virtual-world 8 --no-cpu-baseline > $O/sharded_virtual8.json 2> $O/sharded_virtual8.err
  cd /tmp && export TMPDIR=/tmp
  rm -rf /tmp/prof; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof -o b -- python $R/bench.py --force-sharded --virtual-world 8 --steps 8 --warmup 2 --no-cpu-baseline --sustained-steps 0 > $O/sharded_virtual8_profiled.json 2>/dev/null
  KRS_STATS_FULL_NAMES=1 python $R/scripts/rocpd_stats.py $(ls /tmp/prof/*/*.db /tmp/prof/*.db 2>/dev/null | head -1) 60 > $O/sharded_virtual8_kernel_stats.md
  cd $R
  tail -c 300 $O/sharded_virtual8.err
  python - <<PY
import json
d = json.loads(open("$O/sharded_virtual8.json").read().strip().splitlines()[-1])
print("virtual 8:", round(d["ms_per_step"], 3), d["step_stats"]["median_ms"], "L=1", round(d["also"]["ms_per_step"], 3), "host", round(d["host_enqueue_ms_per_step"], 3), d["exchange"], {k: round(v["ms_per_step"], 3) for k, v in d["phases"].items()}, d.get("overflow_steps"), d.get("invalid"))
PY
fi
