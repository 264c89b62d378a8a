#!/bin/bash
# tools/collect_pmc_cells.sh [OUT] [ONLY_REGEX] -- run ON THE GPU BOX (gpurun): rocprofv3 evidence for every
# kernel tools/pmc_cells.py exercises (the one-kernel plans, the BASELINE configs beside the headline,
# the element-wise and composite kernels).  One kernel-trace pass for durations, then each counter
# set in its own --pmc pass (never combined with a trace, as MI355X_MICROARCH.md prescribes).
# Summarise with tools/summarize_pmc_cells.py (runs anywhere).
set -u
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=${1:-$REPO/gpurun_out/pmc_cells}
ONLY=${2:-}
# TRACE_ONLY=1: only the duration pass, into an existing OUT (the counter passes there are kept)
if [ -n "${TRACE_ONLY:-}" ]; then rm -rf "$OUT/trace"; else rm -rf "$OUT"; fi
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
CMD=(python $REPO/tools/pmc_cells.py --reps 3)
[ -n "$ONLY" ] && CMD+=(--only "$ONLY")
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- \
  "${CMD[@]}" --reps 6 --warm-ms 80 > $OUT/manifest.json 2> $OUT/trace.log
[ -n "${TRACE_ONLY:-}" ] && { find $OUT -name '*.db' -delete 2>/dev/null; du -sh $OUT; exit 0; }
run_pmc() {  # name, counters...
  local name=$1; shift
  timeout 600 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -o p -- "${CMD[@]}" > /dev/null 2> $OUT/$name.log
  echo "$name: rc=$? $(find $OUT/$name -name '*counter_collection.csv' | wc -l) csv"
}
run_pmc fetch FETCH_SIZE
run_pmc write WRITE_SIZE
run_pmc sq SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE
run_pmc sq2 SQ_BUSY_CU_CYCLES SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE
run_pmc lds SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE
# keep what the summariser reads small enough to merge back (<= 64 MiB): the per-dispatch CSVs only
find $OUT -name '*.db' -delete 2>/dev/null
du -sh $OUT
