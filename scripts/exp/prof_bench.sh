# kernel-trace of the bench command; writes the markdown summary + the JSON lines to gpurun_out/
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof; rocprofv3 --kernel-trace --stats -d /tmp/prof -o b -- python /root/repo/bench.py --steps 8 --warmup 2 > /root/repo/gpurun_out/prof_bench.json 2>/dev/null
python /root/repo/scripts/rocpd_stats.py $(ls /tmp/prof/*/*.db /tmp/prof/*.db 2>/dev/null | head -1) 45 > /root/repo/gpurun_out/prof_bench_stats.md
cd /root/repo && python bench.py --steps 20 --warmup 5 > gpurun_out/bench_default.json 2>/dev/null
