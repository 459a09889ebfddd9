# stem A/B: register-weight form (stem_r = 1, 4 workgroups per CU; build/wpe3: the 3-per-CU build) against the r02 persistent form
mkdir -p gpurun_out/r05g
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "register_weight_stem or every_stage or cfg1" > gpurun_out/r05g/tests.log 2>&1; tail -4 gpurun_out/r05g/tests.log
python tools/layer_times.py "streams=1,stem_r=1" "streams=1,stem_r=0" 2>&1 | grep -E "stem|TOTAL" > gpurun_out/r05g/layers.txt; cat gpurun_out/r05g/layers.txt
FLOPE_AMD_LIB=$GRAFT_REPO_ROOT/build/wpe3/libflope_amd_wpe3.so python tools/layer_times.py "streams=1,stem_r=1" 2>&1 | grep -E "stem|TOTAL" > gpurun_out/r05g/layers_wpe3.txt; cat gpurun_out/r05g/layers_wpe3.txt
S=512 B=128 python tools/layer_times.py "streams=1,stem_r=1" "streams=1,stem_r=0" 2>&1 | grep -E "stem|TOTAL" > gpurun_out/r05g/layers512.txt; cat gpurun_out/r05g/layers512.txt
S=512 B=128 FLOPE_AMD_LIB=$GRAFT_REPO_ROOT/build/wpe3/libflope_amd_wpe3.so python tools/layer_times.py "streams=1,stem_r=1" 2>&1 | grep -E "stem|TOTAL" > gpurun_out/r05g/layers512_wpe3.txt; cat gpurun_out/r05g/layers512_wpe3.txt
WARM=100 N=200 python tools/opt_sweep.py "" "stem_r=0" "" "stem_r=0" "" "stem_r=0" > gpurun_out/r05g/ab.txt 2>&1; cat gpurun_out/r05g/ab.txt
WARM=100 N=200 FLOPE_AMD_LIB=$GRAFT_REPO_ROOT/build/wpe3/libflope_amd_wpe3.so python tools/opt_sweep.py "" "" > gpurun_out/r05g/ab_wpe3.txt 2>&1; cat gpurun_out/r05g/ab_wpe3.txt
