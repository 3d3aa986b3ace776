#!/bin/bash
# GPU box: three separate rocprofv3 --pmc passes over a short bench run (counters only, no
# runtime/sys tracing), then tools/summarize_pmc.py.  Run from the repo root through gpurun:
#   gpurun -- 'bash tools/collect_syrk_pmc.sh'
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_${LSQAMD_ROUND:-r04}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() { # name, counters...
  local name=$1; shift
  rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/$name.log 2>&1 || true
}
run sq SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY
run fetch FETCH_SIZE TCC_HIT_sum
run write WRITE_SIZE TCC_MISS_sum
cd $ROOT
python3 tools/summarize_pmc.py $OUT
# (the raw per-dispatch CSVs are tens of MB per pass: only the summary and the trimmed J^T J rows travel back)
rm -rf $OUT/sq $OUT/fetch $OUT/write
