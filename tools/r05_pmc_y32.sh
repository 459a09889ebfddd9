ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05n
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA --output-format csv -d $OUT/pmc_a -- python3 $ROOT/tools/bench_yolo.py --dtype f32 --profile-iters 3 > $OUT/pmc_a.log 2>&1 || exit 12
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD SQ_WAVES --output-format csv -d $OUT/pmc_b -- python3 $ROOT/tools/bench_yolo.py --dtype f32 --profile-iters 3 > $OUT/pmc_b.log 2>&1 || exit 13
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 $ROOT/tools/bench_yolo.py --dtype f32 --profile-iters 10 > $OUT/kt.log 2>&1
cd $ROOT
python tools/summarize_prof.py $OUT/kt $OUT/pmc_a $OUT/pmc_b > $OUT/summary.txt
grep -A1 "y32m" $OUT/summary.txt | head -60
