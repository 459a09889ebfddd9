mkdir -p gpurun_out/r05k
timeout -k 10 500 python -m pytest tests/test_gpu_yolo.py -m gpu -x -q > gpurun_out/r05k/tests.log 2>&1; tail -5 gpurun_out/r05k/tests.log
python tools/bench_yolo.py > gpurun_out/r05k/yolo_f16.json 2>/dev/null; python tools/bench_yolo.py --dtype f32 > gpurun_out/r05k/yolo_f32.json 2>/dev/null; cat gpurun_out/r05k/yolo_f16.json gpurun_out/r05k/yolo_f32.json
python tools/bench_yolo.py --per-launch 2>/dev/null > gpurun_out/r05k/yolo_per_launch.txt; python tools/bench_yolo.py --dtype f32 --per-launch 2>/dev/null > gpurun_out/r05k/yolo_f32_per_launch.txt
grep -E "chain|total" gpurun_out/r05k/yolo_per_launch.txt gpurun_out/r05k/yolo_f32_per_launch.txt
