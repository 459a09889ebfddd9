# layer2.1.conv1 / conv2: conv_s1r (s1r = 1) against conv_w4 (s1r = 0)
mkdir -p gpurun_out/r05v
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "register_weight_layer2 or stride2_patch or every_stage or cfg1 or bit_identical" > gpurun_out/r05v/tests.log 2>&1; tail -4 gpurun_out/r05v/tests.log
python tools/layer_times.py "streams=1,s1r=1" "streams=1,s1r=0" 2>&1 | grep -E "layer2|TOTAL" > gpurun_out/r05v/layers.txt; cat gpurun_out/r05v/layers.txt
WARM=100 N=200 python tools/opt_sweep.py "" "s1r=0" "" "s1r=0" "" "s1r=0" > gpurun_out/r05v/ab.txt 2>&1; cat gpurun_out/r05v/ab.txt
