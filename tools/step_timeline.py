"""Timeline of ONE two-stream step from a rocprofv3 kernel trace (tools/profile_target.py with the default 2 slices):
    cd /tmp && rocprofv3 --kernel-trace --output-format csv -d OUT -- python3 tools/profile_target.py 12
    python tools/step_timeline.py OUT
Prints, for the last complete step, every launch's start / end (us from the step's first start), its stream and grid, and how
much of the step has 1 or 2 kernels in flight."""
import csv
import glob
import sys

rows = list(csv.DictReader(open(glob.glob(f"{sys.argv[1]}/**/*kernel_trace.csv", recursive=True)[0])))
ks = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Stream_Id", r.get("Queue_Id", "?")),
              int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"]))) for r in rows), key=lambda t: t[0])
stems = [i for i, k in enumerate(ks) if "stem_pool" in k[2]]
per_step = 2                                   # two slices -> two stem launches per step
first = stems[-2 * per_step]                   # the last-but-one step (complete)
last = stems[-per_step]
step = ks[first:last]
t0 = step[0][0]
end = max(k[1] for k in step)
print(f"step: {len(step)} launches, {(end - t0) / 1e3:.1f} us")
def short(n):
    for key in ("stem_pool", "conv_w4", "conv_r4", "conv_gstag", "conv_mfma", "avgpool", "fc1", "fc2"):
        if key in n:
            return key
    return n[:24]
for s, e, n, q, g in step:
    print(f"{(s - t0) / 1e3:8.1f} {(e - t0) / 1e3:8.1f}  {(e - s) / 1e3:6.1f} us  queue {q:>4s}  {g:5d} wg  {short(n)}")
ev = sorted([(s, 1) for s, *_ in step] + [(e, -1) for _, e, *_ in step])
busy = {0: 0, 1: 0, 2: 0, 3: 0}
cur, prev = 0, t0
for t, d in ev:
    busy[min(cur, 3)] += t - prev
    cur += d; prev = t
print("time with 0 / 1 / 2 / 3+ kernels in flight (us):", " / ".join(f"{busy[i] / 1e3:.1f}" for i in range(4)))
