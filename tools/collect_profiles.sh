#!/bin/bash
# Evidence run on the GPU box (one gpurun call): rocprofv3 kernel trace + PMC passes of the bench workload, the
# FETCH_SIZE calibration, and the side benches.  Output under gpurun_out/$1 (default r02).
#   /usr/local/graft/bin/gpurun --timeout 1100 -- 'bash tools/collect_profiles.sh r02'
set -o pipefail
TAG=${1:-r05}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export FLOPE_OPTS=profile=1          # ONE stream, but the kernel variants of the production (two-slice) plan: per-launch durations comparable with the in-bench HIP events
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 $ROOT/tools/profile_target.py 20 > $OUT/kt.log 2>&1 || exit 11
echo "kt done"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/pmc_sq -- python3 $ROOT/tools/profile_target.py 4 > $OUT/pmc_sq.log 2>&1 || exit 12
echo "pmc_sq done"
# the schedule bench.py runs: two batch slices on two streams (per-dispatch counters; no trace domain, so nothing perturbs the interleave
# but the counter collection itself) -- VERDICT r3 item 6
FLOPE_OPTS= rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_sq_two_slices -- python3 $ROOT/tools/profile_target.py 4 > $OUT/pmc_sq_two_slices.log 2>&1 || exit 17
echo "pmc_sq two slices done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $ROOT/tools/profile_target.py 4 > $OUT/pmc_fetch.log 2>&1 || exit 13
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $ROOT/tools/profile_target.py 4 > $OUT/pmc_write.log 2>&1 || exit 14
echo "pmc fetch/write done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/calib -- $ROOT/build/fetch_calib > $OUT/calib.log 2>&1 || exit 15
echo "calib done"
unset FLOPE_OPTS
cd $ROOT
python tools/summarize_prof.py $OUT/kt $OUT/pmc_sq $OUT/pmc_fetch $OUT/pmc_write > $OUT/summary.txt
python tools/summarize_prof.py $OUT/kt $OUT/pmc_sq_two_slices > $OUT/summary_two_slices.txt
python tools/summarize_prof.py $OUT/calib > $OUT/calib_summary.txt
python tools/roofline_table.py $OUT > $OUT/roofline.md
python tools/make_traffic_json.py $OUT > $OUT/traffic.json
cp $OUT/traffic.json profiles/$(echo $TAG | sed -E "s/^(r[0-9]+).*/\\1/")_traffic.json      # bench.py reads the committed file (newest round first): same sources, same digest
python bench.py > $OUT/bench.json 2> $OUT/bench.err || exit 16
echo "bench done"
YOLO_DTYPE=f32 python tools/bench_e2e.py yolo 4 16 31 > $OUT/e2e.txt 2>&1
YOLO_DTYPE=f16 python tools/bench_e2e.py yolo > $OUT/e2e_f16_detector.txt 2>&1
python tools/bench_yolo.py > $OUT/yolo_bench.json 2> $OUT/yolo_bench.err
python tools/bench_yolo.py --batch 0 >> $OUT/yolo_bench.json 2>> $OUT/yolo_bench.err
python tools/bench_yolo.py --per-launch 2> /dev/null > $OUT/yolo_per_launch.txt
python tools/bench_yolo.py --dtype f32 > $OUT/yolo_f32_bench.json 2>> $OUT/yolo_bench.err
python tools/bench_yolo.py --dtype f32 --per-launch 2> /dev/null > $OUT/yolo_f32_per_launch.txt
$ROOT/build/launch_floor > $OUT/launch_floor.txt 2>&1
python tools/probe_pipeline.py 2>&1 | grep -v amdgpu.ids > $OUT/pipeline_phases.txt
python tools/latency.py > $OUT/latency.txt 2>&1
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_yolo -- python3 $ROOT/tools/bench_yolo.py --profile-iters 20 > $OUT/kt_yolo.log 2>&1
cd $ROOT
python tools/summarize_prof.py $OUT/kt_yolo > $OUT/yolo_summary.txt
tail -3 $OUT/bench.json | cut -c1-600
cat $OUT/calib_summary.txt
