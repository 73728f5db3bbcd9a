# Round-2 final evidence: default bench line, kernel trace of the same step, sharded dry run, C5 on one GPU.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R && (time python bench.py) > gpurun_out/r2z_bench_c3.json 2> gpurun_out/r2z_bench_c3.err
cd /tmp && rm -rf /tmp/prof && rocprofv3 --kernel-trace --stats -d /tmp/prof -o b -- python $R/bench.py --no-cpu-baseline --steps 8 --warmup 2 > $R/gpurun_out/r2z_bench_c3_profiled.json 2>/dev/null
python $R/scripts/rocpd_stats.py $(ls /tmp/prof/*/*.db /tmp/prof/*.db 2>/dev/null | head -1) 45 > $R/gpurun_out/r2z_bench_c3_kernel_stats.md
rm -rf /tmp/prof && rocprofv3 --kernel-trace --stats -d /tmp/prof -o b -- python $R/bench.py --force-sharded --batch 8192 --no-cpu-baseline --steps 8 --warmup 2 > $R/gpurun_out/r2z_sharded_b8192_profiled.json 2>/dev/null
python $R/scripts/rocpd_stats.py $(ls /tmp/prof/*/*.db /tmp/prof/*.db 2>/dev/null | head -1) 60 > $R/gpurun_out/r2z_sharded_b8192_kernel_stats.md
cd $R
python bench.py --force-sharded --batch 8192 --no-cpu-baseline --steps 30 --warmup 5 > gpurun_out/r2z_sharded_b8192.json 2>/dev/null
python bench.py --criteo-vocab 40000000 --id-skew 4 --no-cpu-baseline --steps 10 --warmup 3 > gpurun_out/r2z_bench_c5_powerlaw.json 2>/dev/null
python bench.py --criteo-vocab 1000000 --no-cpu-baseline --steps 10 --warmup 3 > gpurun_out/r2z_bench_c3prime.json 2>/dev/null
python bench.py --rowwise-adagrad --no-cpu-baseline --steps 10 --warmup 3 > gpurun_out/r2z_bench_c3_rowwise.json 2>/dev/null
timeout 100 scripts/exp/gemm_bench 7 > gpurun_out/r2z_gemm_ab.txt 2>&1
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r2z_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, round(d['ms_per_step'],3), round(d['also']['ms_per_step'],3), round(d['roofline']['frac'],4), round(d['also']['roofline']['frac'],4))
    except Exception as e: print(f, 'ERR', e)
PY
