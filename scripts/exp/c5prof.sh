cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/prof && rocprofv3 --kernel-trace --stats -d /tmp/prof -o b -- python $R/bench.py --criteo-vocab 40000000 --id-skew 4 --hotness 1 --no-cpu-baseline --steps 10 --warmup 3 > $R/gpurun_out/r2z_c5_l1_profiled.json 2>/dev/null
python $R/scripts/rocpd_stats.py $(ls /tmp/prof/*/*.db /tmp/prof/*.db 2>/dev/null | head -1) 30 > $R/gpurun_out/r2z_c5_l1_kernel_stats.md
head -24 $R/gpurun_out/r2z_c5_l1_kernel_stats.md | cut -c1-140
python -c "
import json;d=json.loads(open('$R/gpurun_out/r2z_c5_l1_profiled.json').read().strip().splitlines()[-1]);print(d['ms_per_step'],d['also']['ms_per_step'])"
