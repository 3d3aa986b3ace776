ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r3_c2trace
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -- python3 $ROOT/bench.py --workload c2 --steps 200 --warmup 20 --no-cpu-baseline > $OUT/bench_c2_under_rocprof.json 2> $OUT/err.txt
cp $(find $OUT/t -name "*kernel_stats.csv" | head -1) $OUT/c2_kernel_stats.csv
rm -rf $OUT/t
cd $ROOT
python3 bench.py --workload c2 --steps 200 --warmup 20 --no-cpu-baseline > $OUT/bench_c2.json 2>/dev/null
tail -c 600 $OUT/bench_c2.json
