# LDS bank conflicts and wait shares of the detector's kernels, both dtypes
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05y
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for dt in f16 f32; do
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY --output-format csv -d $OUT/pmc_$dt -- python3 $ROOT/tools/bench_yolo.py --dtype $dt --profile-iters 3 > $OUT/pmc_$dt.log 2>&1 || exit 12
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_$dt -- python3 $ROOT/tools/bench_yolo.py --dtype $dt --profile-iters 10 > $OUT/kt_$dt.log 2>&1
done
cd $ROOT
for dt in f16 f32; do python tools/summarize_prof.py $OUT/kt_$dt $OUT/pmc_$dt > $OUT/summary_$dt.txt; done
python - <<'PY'
import re
for dt in ("f16", "f32"):
    t = open(f"gpurun_out/r05y/summary_{dt}.txt").read().split("\n")
    print("==", dt)
    for i, l in enumerate(t):
        if "SQ_LDS_BANK_CONFLICT" in l:
            name = t[i - 1].split("  ")[0][:60]
            m = dict(re.findall(r"(SQ_\w+)=([\d.e+]+)", l))
            print(f"{name:62s} conflicts {float(m['SQ_LDS_BANK_CONFLICT'])/max(float(m['SQ_LDS_IDX_ACTIVE']),1):6.1%} of LDS cycles; LDS active/busy {float(m['SQ_LDS_IDX_ACTIVE'])/max(float(m['SQ_BUSY_CYCLES']),1):5.2f}; wait_lds/wave {float(m['SQ_WAIT_INST_LDS'])/max(float(m['SQ_WAVE_CYCLES']),1):5.1%}; wait_inst/wave {float(m['SQ_WAIT_INST_ANY'])/max(float(m['SQ_WAVE_CYCLES']),1):4.0%}")
    for l in t[:14]:
        if "kernel" in l or "us" in l or "y" in l[:3]: pass
    print("\n".join(t[1:12]))
PY
