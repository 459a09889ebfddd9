# layer2.0.conv1: conv_s2r (s2r = 1) against conv_mfma<gather> (s2r = 0)
mkdir -p gpurun_out/r05s
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "stride2_patch or every_stage or cfg1 or bit_identical" > gpurun_out/r05s/tests.log 2>&1; tail -4 gpurun_out/r05s/tests.log
python tools/layer_times.py "streams=1,s2r=1" "streams=1,s2r=0" 2>&1 | grep -E "layer2.0.conv1|layer2.0.conv2|TOTAL" > gpurun_out/r05s/layers.txt; cat gpurun_out/r05s/layers.txt
WARM=100 N=200 python tools/opt_sweep.py "" "s2r=0" "" "s2r=0" "" "s2r=0" > gpurun_out/r05s/ab.txt 2>&1; cat gpurun_out/r05s/ab.txt
