#!/bin/bash
# GPU box: rocprofv3 --pmc passes (counters only, no runtime/sys tracing; the program goes directly after `--`) over the
# three programs that exercise the kernels the design calls latency-bound: the blocked factorisation alone
# (tools/time_potrf.py: trail_potf2_kernel, potf2_v4_kernel, the one-shot row panel), the config-4 bench (the fused
# whitening kernel) and config 5 without the graph replay (batched J^T J, sum_model_kernel).  Then
# tools/summarize_kernel_pmc.py -> gpurun_out/kpmc_<round>/{potrf,whiten,c5}_pmc.json.
#   gpurun -- 'bash tools/collect_kernel_pmc.sh'
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
R=${LSQAMD_ROUND:-r05}
OUT=$ROOT/gpurun_out/kpmc_$R
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$ROOT
SQ="SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE"
SQ2="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES"
run() { # program-tag, pass-name, counters -- program args...
  local tag=$1 pass=$2 ctr=$3; shift 3
  rocprofv3 --pmc $ctr --output-format csv -d $OUT/$tag/$pass -- "$@" > $OUT/${tag}_$pass.log 2>&1 || echo "pass $tag/$pass failed" >&2
}
TAGS=${LSQAMD_PMC_TAGS:-"potrf whiten c5"}      # (+ c3: the triangular whitening product of one dense 8192-row block; shard: the 8-GPU shard shape)
for tag in $TAGS; do
  case $tag in
    potrf)  prog=(python3 $ROOT/tools/time_potrf.py) ;;
    whiten) prog=(python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline) ;;
    c5)     prog=(python3 $ROOT/tools/run_c5.py nograph) ;;
    c3)     prog=(python3 $ROOT/bench.py --workload c3 --steps 10 --warmup 2 --no-cpu-baseline --two-pass) ;;
    shard)  prog=(python3 $ROOT/bench.py --ndata 8192 --steps 6 --warmup 2 --no-cpu-baseline --whole-fit-maxit 0) ;;
  esac
  run $tag sq "$SQ" "${prog[@]}"
  run $tag sq2 "$SQ2" "${prog[@]}"
  run $tag fetch "FETCH_SIZE TCC_HIT_sum" "${prog[@]}"
  run $tag write "WRITE_SIZE TCC_MISS_sum" "${prog[@]}"
done
cd $ROOT
python3 tools/summarize_kernel_pmc.py $OUT
for tag in $TAGS; do rm -rf $OUT/$tag; done      # raw per-dispatch CSVs: tens of MB
