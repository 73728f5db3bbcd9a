# kernel trace of the whole DLRM-DCN-v2 model step (examples/dlrm_dcn_v2.py at the C3 shape)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_full; rocprofv3 --kernel-trace --stats -d /tmp/prof_full -o b -- python /root/repo/scripts/exp/full_steps.py 10 > /root/repo/gpurun_out/prof_full.log 2>&1
python /root/repo/scripts/rocpd_stats.py $(ls /tmp/prof_full/*/*.db /tmp/prof_full/*.db 2>/dev/null | head -1) 60 > /root/repo/gpurun_out/prof_full_stats.md
