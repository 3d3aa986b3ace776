ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r3_c2gaps
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
PYTHONPATH=$ROOT rocprofv3 --kernel-trace --output-format csv -d $OUT/t -- python3 $ROOT/tools/time_small_steps.py ${SHAPE:-4096 256} > $OUT/out.txt 2> $OUT/err.txt
f=$(find $OUT/t -name "*kernel_trace.csv" | head -1)
python3 - "$f" > $OUT/gaps.txt <<'PY'
import csv, sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# last 3000 kernels: steady state
rows=rows[-3000:]
prev_end=None
import collections
print('name, dur_us, gap_before_us')
tot_d=tot_g=0
biggaps=[]
for r in rows:
    s,e=int(r['Start_Timestamp']),int(r['End_Timestamp'])
    g=(s-prev_end)/1e3 if prev_end else 0
    tot_d+=(e-s)/1e3; tot_g+=max(g,0)
    prev_end=e
    biggaps.append((r['Kernel_Name'][:60],(e-s)/1e3,g))
for x in biggaps[-70:]: print('%-62s %8.2f %8.2f'%x)
print('sum dur', tot_d, 'sum gaps', tot_g, 'span', (int(rows[-1]['End_Timestamp'])-int(rows[0]['Start_Timestamp']))/1e3)
PY
tail -80 $OUT/gaps.txt
rm -rf $OUT/t
