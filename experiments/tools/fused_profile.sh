#!/bin/bash
# tools/fused_profile.sh -- run ON THE GPU BOX: rocprofv3 evidence for fused_pass
# (kernel stats, HBM traffic, L2 hit rate, SQ issue counters) -> gpurun_out/$1
set -u
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/${1:-fusedprof}
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
CMD="python $REPO/tools/one_step.py"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- $CMD > $OUT/trace.log 2>&1
run_pmc() {
  local name=$1; shift
  rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -o p -- $CMD > $OUT/$name.log 2>&1
}
run_pmc fetch FETCH_SIZE
run_pmc write WRITE_SIZE
run_pmc l2 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
run_pmc sq SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY
python $REPO/tools/pmc_sum.py $OUT | tee $OUT/summary.txt
