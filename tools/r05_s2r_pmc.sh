# PMC counters of the stem kernel, register-weight form (stem_r = 1) against the r02 persistent form (stem_r = 0)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-r05t}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for v in 1 0; do
  export FLOPE_OPTS=profile=1,streams=1,s2r=$v
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --output-format csv -d $OUT/pmc_a$v -- python3 $ROOT/tools/profile_target.py 3 > $OUT/pmc_a$v.log 2>&1 || exit 12
  rocprofv3 --pmc SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/pmc_b$v -- python3 $ROOT/tools/profile_target.py 3 > $OUT/pmc_b$v.log 2>&1 || exit 13
done
cd $ROOT
for v in 1 0; do echo "streams=1,s2r=$v"; python tools/summarize_prof.py $OUT/pmc_a$v $OUT/pmc_b$v | grep -A1 "layer2.0.conv1\|conv_s2r\|gather"; done > $OUT/s2r_pmc.txt
cat $OUT/s2r_pmc.txt
