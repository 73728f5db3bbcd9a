# Round-4 closing evidence for the final code (one gpurun call): default bench line, the same command under the kernel
# trace, the whole model, smoke().  Files land in gpurun_out/$TAG/ and are copied to profiles/ afterwards.
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r4f}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
python bench.py > $O/bench_c3.json 2> $O/bench_c3.err
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof; rocprofv3 --kernel-trace --stats -d /tmp/prof -o b -- python $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline > $O/bench_c3_profiled.json 2>/dev/null
python $R/scripts/rocpd_stats.py $(ls /tmp/prof/*/*.db /tmp/prof/*.db 2>/dev/null | head -1) 50 > $O/bench_c3_kernel_stats.md
cd $R
python bench.py --full-model --no-cpu-baseline > $O/bench_c3_full_model.json 2>/dev/null
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1
tail -2 $O/smoke.log
python - <<EOF
import json
for f in ("bench_c3.json", "bench_c3_profiled.json", "bench_c3_full_model.json"):
    try:
        d = json.loads(open("$O/" + f).read().strip().splitlines()[-1])
        print(f, d["ms_per_step"], d["value"], d.get("roofline", {}).get("frac"), (d.get("full_model") or {}).get("ms_per_step"))
    except Exception as e:
        print(f, "unreadable", e)
EOF
