#!/bin/bash
# GPU box: everything profiles/ carries for one round -- bench lines (c4, the 8-GPU shard shape, c2, c3),
# the rocprofv3 kernel trace of the c4 bench, the three PMC passes of the J^T J launch, and the
# tape-model Jacobian timings.  gpurun -- 'bash tools/collect_round_profiles.sh'
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
R=${LSQAMD_ROUND:-r06}
OUT=$ROOT/gpurun_out/prof_$R
mkdir -p $OUT
cd $ROOT
python3 bench.py --steps 20 --warmup 5 > $OUT/${R}_bench_default.json 2> $OUT/bench_c4.err      # the driver's command: c4 headline + config.other_workloads
python3 bench.py --workload shard8192 --steps 40 --warmup 5 --no-cpu-baseline | grep '^{' > $OUT/${R}_bench_shard8192_1gpu.json 2>/dev/null
python3 bench.py --workload c5 | grep '^{' > $OUT/${R}_bench_c5_1gpu.json 2>/dev/null
python3 bench.py --workload c2 --steps 200 --warmup 20 --cpu-seconds 5 > $OUT/${R}_bench_c2_1gpu.json 2>/dev/null
python3 bench.py --workload c3 --steps 40 --warmup 5 --cpu-seconds 10 > $OUT/${R}_bench_c3_1gpu.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_c4 -- python3 $ROOT/bench.py --steps 8 --warmup 2 --no-cpu-baseline > $OUT/${R}_bench_c4_1gpu_under_rocprof.json 2> $OUT/trace_c4.err
cp $(find $OUT/trace_c4 -name "*kernel_stats.csv" | head -1) $OUT/${R}_c4_1gpu_kernel_stats.csv
PYTHONPATH=$ROOT rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_tape -- python3 $ROOT/tools/time_tape.py > $OUT/tape_default.txt 2> $OUT/trace_tape.err
cp $(find $OUT/trace_tape -name "*kernel_stats.csv" | head -1) $OUT/${R}_tape_n65536_kernel_stats.csv
# kernel stats of the other BASELINE configs: c2 (4096, 256), c3 (8192, 1024, one dense block), c5 (128 batched fits of 4096 x 512)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_c2 -- python3 $ROOT/bench.py --workload c2 --steps 200 --warmup 20 --no-cpu-baseline > $OUT/${R}_bench_c2_1gpu_under_rocprof.json 2> $OUT/trace_c2.err
cp $(find $OUT/trace_c2 -name "*kernel_stats.csv" | head -1) $OUT/${R}_c2_1gpu_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_c3 -- python3 $ROOT/bench.py --workload c3 --steps 40 --warmup 5 --no-cpu-baseline > $OUT/${R}_bench_c3_1gpu_under_rocprof.json 2> $OUT/trace_c3.err
cp $(find $OUT/trace_c3 -name "*kernel_stats.csv" | head -1) $OUT/${R}_c3_1gpu_kernel_stats.csv
PYTHONPATH=$ROOT rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_c5 -- python3 $ROOT/tools/run_c5.py nograph > $OUT/${R}_c5_run.txt 2> $OUT/trace_c5.err
cp $(find $OUT/trace_c5 -name "*kernel_stats.csv" | head -1) $OUT/${R}_c5_kernel_stats.csv
PYTHONPATH=$ROOT python3 $ROOT/tools/run_c5.py >> $OUT/${R}_c5_run.txt 2>/dev/null
PYTHONPATH=$ROOT python3 $ROOT/tools/time_small_steps.py > $OUT/${R}_small_steps.txt 2>/dev/null
PYTHONPATH=$ROOT python3 $ROOT/tools/time_small_fit.py > $OUT/${R}_small_fits.txt 2>/dev/null
{ echo "# the same four fits through the general path (LSQAMD_ONE_LAUNCH_FIT=0)"; LSQAMD_ONE_LAUNCH_FIT=0 PYTHONPATH=$ROOT python3 $ROOT/tools/time_small_fit.py 2>/dev/null; } >> $OUT/${R}_small_fits.txt
PYTHONPATH=$ROOT python3 $ROOT/tools/cmp_one_launch.py > $OUT/${R}_one_launch_vs_general.txt 2>/dev/null
PYTHONPATH=$ROOT python3 $ROOT/tools/time_resample.py > $OUT/${R}_resample_small.txt 2>/dev/null
cd $ROOT
{ echo "# the tape COMPILED at lsqamd_set_tape time (jit.hip, hiprtc; the default; this run under rocprofv3)"; grep -E "P =|residual|tape runs" $OUT/tape_default.txt
  echo "# the interpreter kernels of round 2 (LSQAMD_TAPE=i: by terms where the root is a sum, whole-tape reverse sweep otherwise)"; LSQAMD_TAPE=i PYTHONPATH=$ROOT python3 tools/time_tape.py 2>/dev/null | grep -E "P =|residual|tape runs"
} > $OUT/${R}_tape_timing.txt
rm -rf $OUT/trace_c4 $OUT/trace_tape $OUT/trace_c2 $OUT/trace_c3 $OUT/trace_c5
cd $ROOT
LSQAMD_ROUND=$R bash tools/collect_syrk_pmc.sh > $OUT/pmc.log 2>&1
cp gpurun_out/pmc_$R/summary.json $OUT/${R}_syrk_pmc.json
for p in sq fetch write; do cp gpurun_out/pmc_$R/syrk_${p}_pass.csv $OUT/${R}_syrk_pmc_${p}_pass.csv; done
ls -la $OUT
