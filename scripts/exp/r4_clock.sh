# Round 4: is the ring GEMM's wall time the clock?  random vs all-zero operands, and GRBM_GUI_ACTIVE / duration per kernel
# for the product build and its DMA-only / MFMA-only probe builds (scripts/exp/libs/probe1, probe2).
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4a; mkdir -p $O
cd $R
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 scripts/exp/gemm_bench.cpp -o /tmp/gemm_bench -I include -L keras_rs_amd -lkrs_hip 2>/dev/null
export LD_LIBRARY_PATH=$R/keras_rs_amd
{ echo "== random operands"; /tmp/gemm_bench 5 | tail -9; echo "== all-zero operands"; KRS_ZERO=1 /tmp/gemm_bench 5 | tail -9; echo "== random operands again"; /tmp/gemm_bench 5 | tail -9; } > $O/zero_vs_random.txt 2>&1
cd /tmp && export TMPDIR=/tmp
: > $O/clock_pmc.txt
for v in default probe1 probe2; do
  if [ $v = default ]; then L=$R/keras_rs_amd; else L=$R/scripts/exp/libs/$v; fi
  for z in "" 1; do
    rm -rf /tmp/pmc; KRS_ZERO=$z LD_LIBRARY_PATH=$L rocprofv3 --pmc GRBM_GUI_ACTIVE -d /tmp/pmc -o p -- /tmp/gemm_bench 1 > /dev/null 2>&1
    echo "== lib $v  zero=[$z]" >> $O/clock_pmc.txt
    python $R/scripts/rocpd_clock.py $(ls /tmp/pmc/*/*.db /tmp/pmc/*.db 2>/dev/null | head -1) gemm_pp256 >> $O/clock_pmc.txt
  done
done
cat $O/zero_vs_random.txt $O/clock_pmc.txt
