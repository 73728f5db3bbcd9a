# HBM traffic of K1 / K2 per launch: separate --pmc passes over scripts/bench_k1.py
cd /tmp && export TMPDIR=/tmp
for mode in "--multihot" ""; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc; rocprofv3 --pmc $c -d /tmp/pmc -o p -- python /root/repo/scripts/bench_k1.py $mode --iters 5 > /dev/null 2>&1
    echo "mode=[$mode]"; python /root/repo/scripts/rocpd_pmc.py $(ls /tmp/pmc/*/*.db /tmp/pmc/*.db 2>/dev/null | head -1) | grep -E "bag_apply_kernel|embed_bag_fwd_vec|embed_gather_hot1"
  done
done
