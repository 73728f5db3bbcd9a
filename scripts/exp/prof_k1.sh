# kernel trace of scripts/bench_k1.py (standalone K1 / K2 plan / K2 apply kernels); $1 = output tag, rest = bench_k1 flags
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$tag; rocprofv3 --kernel-trace --stats -d /tmp/prof_$tag -o b -- python /root/repo/scripts/bench_k1.py "$@" > /root/repo/gpurun_out/prof_k1_$tag.log 2>/dev/null
python /root/repo/scripts/rocpd_stats.py $(ls /tmp/prof_$tag/*/*.db /tmp/prof_$tag/*.db 2>/dev/null | head -1) 40 > /root/repo/gpurun_out/prof_k1_${tag}_stats.md
