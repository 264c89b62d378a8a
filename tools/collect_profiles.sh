#!/bin/bash
# tools/collect_profiles.sh -- run ON THE GPU BOX (gpurun): collects the rocprofv3
# evidence behind bench.py's roofline block into gpurun_out/prof/ .
#   1. kernel trace + stats of the default bench command
#   2. PMC passes (each counter set in its own run, --pmc only, as the gfx950 guide
#      prescribes): FETCH_SIZE, WRITE_SIZE, SQ issue counters, LDS counters
#   3. the same FETCH/WRITE passes over tools/ubench's copy kernels (known traffic)
#      to calibrate the counters
# Every pass runs under its own `timeout` (a counter pass that hangs must not eat the GPU budget).
# Summarise afterwards with tools/summarize_profiles.py (runs anywhere).
set -u
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/prof
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export BENCH_SUSTAINED_S=0  # (the 3 s steady-state loop is not part of what the counters describe)
BENCH="python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-probe"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- \
  python $REPO/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-secondary --no-probe > $OUT/bench_under_trace.json 2> $OUT/trace.log
run_pmc() {  # name, counters..., then -- command
  local name=$1; shift
  local ctr=()
  while [ "$1" != "--" ]; do ctr+=("$1"); shift; done; shift
  timeout 300 rocprofv3 --pmc "${ctr[@]}" --output-format csv -d $OUT/$name -o p -- "$@" > /dev/null 2> $OUT/$name.log
}
run_pmc fetch FETCH_SIZE -- $BENCH
run_pmc write WRITE_SIZE -- $BENCH
# (GRBM_GUI_ACTIVE rides along in both SQ passes -- GRBM is its own block -- so that VALU busy
# = SQ_ACTIVE_INST_VALU * 4 / (SIMDs * GRBM_GUI_ACTIVE per XCD) is formed within one pass)
run_pmc sq SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE -- $BENCH
run_pmc sq2 SQ_BUSY_CU_CYCLES SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE -- $BENCH
run_pmc lds SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE -- $BENCH
run_pmc cal_fetch FETCH_SIZE -- $REPO/tools/ubench
run_pmc cal_write WRITE_SIZE -- $REPO/tools/ubench
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*VALU[A-Z_0-9]*\|SQ_BUSY[A-Z_]*\|GRBM_[A-Z_]*" | sort -u > $OUT/available_counters.txt
ls -R $OUT | head -50
