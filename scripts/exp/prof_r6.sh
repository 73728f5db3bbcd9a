# Round-6 evidence runs (one gpurun call each).  Files land in gpurun_out/$TAG/ and are copied to profiles/ afterwards.
#   prof_r6.sh TAG bench      the driver's command (line + side file) and the rocprofv3 kernel table of the same command
#   prof_r6.sh TAG suite      GPU test suite
#   prof_r6.sh TAG full       --full-model line + kernel table
#   prof_r6.sh TAG vw8        rank 0's step of an 8-way job on one GPU (--virtual-world 8): line + kernel table
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r6a}
WHAT=${2:-bench}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
stats() {  # $1 = rocprof dir, $2 = output md
  KRS_STATS_FULL_NAMES=1 python $R/scripts/rocpd_stats.py $(ls $1/*/*.db $1/*.db 2>/dev/null | head -1) 70 > $2
}
if [[ $WHAT == *suite* ]]; then
  timeout 1500 python -m pytest tests -m gpu -q > $O/tests_gpu.log 2>&1; tail -3 $O/tests_gpu.log
fi
if [[ $WHAT == *bench* ]]; then
  ( time timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --detail $O/bench_driver_command_detail.json > $O/bench_driver_command.json 2> $O/bench_driver_command.err ) 2> $O/bench_driver_command.time
  wc -c $O/bench_driver_command.json; tail -3 $O/bench_driver_command.time
  cd /tmp && export TMPDIR=/tmp
  rm -rf /tmp/prof; timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof -o b -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-c2 --no-cpu-baseline --sustained-steps 0 --detail $O/bench_profiled_detail.json > $O/bench_profiled.json 2>/dev/null
  stats /tmp/prof $O/bench_c3_kernel_stats.md
  cd $R
fi
if [[ $WHAT == *full* ]]; then
  timeout 600 python bench.py --steps 30 --warmup 8 --no-c2 --no-cpu-baseline --full-model --detail $O/bench_c3_full_model_detail.json > $O/bench_c3_full_model.json 2>/dev/null
  cd /tmp && export TMPDIR=/tmp
  rm -rf /tmp/prof; timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof -o b -- python $R/scripts/exp/full_steps.py 10 > /dev/null 2>&1
  stats /tmp/prof $O/full_model_kernel_stats.md
  cd $R
fi
if [[ $WHAT == *vw8* ]]; then
  timeout 600 python bench.py --force-sharded --virtual-world 8 --steps 30 --warmup 8 --capacity-settle 16 --detail $O/sharded_virtual_world8_settled_detail.json > $O/sharded_virtual_world8_settled.json 2>/dev/null
  timeout 600 python bench.py --force-sharded --virtual-world 8 --steps 30 --warmup 8 --detail $O/sharded_virtual_world8_detail.json > $O/sharded_virtual_world8.json 2>/dev/null
  cd /tmp && export TMPDIR=/tmp
  rm -rf /tmp/prof; timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof -o b -- python $R/bench.py --force-sharded --virtual-world 8 --steps 20 --warmup 5 --sustained-steps 0 --detail /tmp/vw8_prof_detail.json > /dev/null 2>&1
  stats /tmp/prof $O/sharded_virtual_world8_kernel_stats.md
  cd $R
fi
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/*detail.json")):
    try:
        d = json.load(open(f))
        print(f.split("/")[-1], "ms", round(d["ms_per_step"], 3), "median", round(d["step_stats"]["median_ms"], 3), "K1 frac", d.get("roofline", {}).get("frac"),
              "host", round(d.get("host_enqueue_ms_per_step", 0), 3), "full", (d.get("full_model") or {}).get("ms_per_step"), "wall", d.get("wall_seconds"))
    except Exception as e:
        print(f, "unreadable", e)
PY
