# Development builds of libkrs_hip.so with other -D settings of feature_cross.hip, for scripts/exp/gemm_bench A/B:
#   scripts/exp/build_variants.sh name "-DKRS_PP_A_AUX=2" ...   ->  scripts/exp/libs/<name>/libkrs_hip.so
# (run on the build host; the libs travel to the GPU box with the snapshot; LD_LIBRARY_PATH selects one)
set -e
R=$(cd $(dirname $0)/../.. && pwd)
name=$1; shift
src=${KRS_VARIANT_SRC:-feature_cross}      # which csrc/<src>.hip gets the -D settings
mkdir -p $R/scripts/exp/libs/$name
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -I $R/include "$@" \
  -c $R/keras_rs_amd/csrc/$src.hip -o $R/scripts/exp/libs/$name/$src.o
objs=$(ls $R/keras_rs_amd/build/*.o | grep -v "/$src")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/scripts/exp/libs/$name/libkrs_hip.so $R/scripts/exp/libs/$name/$src.o $objs
echo built $R/scripts/exp/libs/$name/libkrs_hip.so
