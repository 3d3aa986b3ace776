#!/bin/bash
# GPU box: per-step durations of the factorisation's launches at P = 4096 (rocprofv3 kernel trace of tools/time_potrf.py)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
PYTHONPATH=$ROOT rocprofv3 --kernel-trace --output-format csv -d $ROOT/gpurun_out/ptrace -- python3 $ROOT/tools/time_potrf.py > /dev/null 2>&1
f=$(find $ROOT/gpurun_out/ptrace -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<"PY"
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
def dur(sub):
    return [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if sub in r["Kernel_Name"]]
d = dur("trail_potf2")
print("fused steps (us):", " ".join("%.0f" % x for x in d[-28:]))
d = dur("potf2_v")
print("standalone diagonal kernels (us):", " ".join("%.1f" % x for x in d[-12:]))
d = dur("oneshot")
print("row panels (us):", " ".join("%.1f" % x for x in d[-32:]))
d = dur("small_kernel")
print("small trailing updates (us):", " ".join("%.1f" % x for x in d[-11:]))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
last = [r for r in rows][-140:]
gaps = [(int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3 for a, b in zip(last, last[1:])]
print("gaps between consecutive launches (us): median %.2f" % sorted(gaps)[len(gaps) // 2])
PY
rm -rf $ROOT/gpurun_out/ptrace
