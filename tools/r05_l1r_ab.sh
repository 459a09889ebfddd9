# layer 1: conv_l1r (l1r = 1) against conv_r4 (l1r = 0)
mkdir -p gpurun_out/r05w
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "register_weight_layer1 or every_stage or cfg1" > gpurun_out/r05w/tests.log 2>&1; tail -4 gpurun_out/r05w/tests.log
python tools/layer_times.py "streams=1,l1r=1" "streams=1,l1r=0" 2>&1 | grep -E "layer1|TOTAL" > gpurun_out/r05w/layers.txt; cat gpurun_out/r05w/layers.txt
WARM=100 N=200 python tools/opt_sweep.py "" "l1r=0" "" "l1r=0" "" "l1r=0" > gpurun_out/r05w/ab.txt 2>&1; cat gpurun_out/r05w/ab.txt
