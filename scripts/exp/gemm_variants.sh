# A/B of libkrs_hip.so variants under the native GEMM harness: default build first, then scripts/exp/libs/<name>
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 scripts/exp/gemm_bench.cpp -o /tmp/gemm_bench -I include -L keras_rs_amd -lkrs_hip 2>/dev/null
for v in default "$@" default; do
  if [ $v = default ]; then L=$R/keras_rs_amd; else L=$R/scripts/exp/libs/$v; fi
  echo "== $v"
  LD_LIBRARY_PATH=$L /tmp/gemm_bench 5 | grep -E "^(fwd|dK|dh|dU|dx|layer)" | awk '{print}' | cut -c1-44,64-84
done
