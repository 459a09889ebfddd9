# stem A/B: register-weight form (stem_r = 1) against the r02 persistent form (stem_r = 0)
mkdir -p gpurun_out/r05g
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "register_weight_stem or every_stage or cfg1" > gpurun_out/r05g/tests.log 2>&1; tail -4 gpurun_out/r05g/tests.log
python tools/layer_times.py "streams=1,stem_r=1" "streams=1,stem_r=0" 2>&1 | grep -E "stem|TOTAL" > gpurun_out/r05g/layers.txt; cat gpurun_out/r05g/layers.txt
S=512 B=128 python tools/layer_times.py "streams=1,stem_r=1" "streams=1,stem_r=0" 2>&1 | grep -E "stem|TOTAL" > gpurun_out/r05g/layers512.txt; cat gpurun_out/r05g/layers512.txt
WARM=100 N=200 python tools/opt_sweep.py "" "stem_r=0" "" "stem_r=0" "" "stem_r=0" > gpurun_out/r05g/ab.txt 2>&1; cat gpurun_out/r05g/ab.txt
FLOPE_AMD_LIB=$GRAFT_REPO_ROOT/build/dbg/libflope_amd_dbg.so python tools/clock_probe_stem.py 1 2>&1 | grep -v amdgpu > gpurun_out/r05g/stamps.txt; cat gpurun_out/r05g/stamps.txt
