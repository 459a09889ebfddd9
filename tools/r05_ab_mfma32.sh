set -e
mkdir -p gpurun_out/r05b
export WARM=100 N=200
( for i in 1 2 3 4; do
  for arm in "-|" "build/m32/libflope_amd_m32.so|" "-|streams=1" "build/m32/libflope_amd_m32.so|streams=1"; do
    lib=${arm%%|*}; opts=${arm#*|}
    if [ "$lib" = "-" ]; then python tools/opt_sweep.py "$opts" 2>&1 | grep poses | awk -v a="$arm" '{print a "  " $0}'
    else FLOPE_AMD_LIB=$GRAFT_REPO_ROOT/$lib python tools/opt_sweep.py "$opts" 2>&1 | grep poses | awk -v a="$arm" '{print a "  " $0}'; fi
  done
done ) > gpurun_out/r05b/ab_mfma32_timing.txt 2>&1
python tools/layer_times.py "streams=1" > gpurun_out/r05b/layers_prod.txt 2>&1
FLOPE_AMD_LIB=$GRAFT_REPO_ROOT/build/m32/libflope_amd_m32.so python tools/layer_times.py "streams=1" > gpurun_out/r05b/layers_m32.txt 2>&1
cat gpurun_out/r05b/ab_mfma32_timing.txt
