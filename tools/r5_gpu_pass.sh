#!/bin/bash
# One GPU-box pass of round 5: the -m gpu suite under the native-backtrace preload (N runs, crash
# logs kept), the C++ host-call budget, a two-rank dry run of bench.py on one device (both scaling
# modes in one line), the default bench line.
# usage: tools/r5_gpu_pass.sh OUTDIR [suite runs]
set -u
OUT=${1:-gpurun_out/r5b}
RUNS=${2:-12}
mkdir -p "$OUT"
TRACE=$PWD/tools/libabort_trace.so
suite() {  # index
  timeout 600 env LD_PRELOAD="$TRACE" ABORT_TRACE_MAPS="$PWD/$OUT/maps_$1.txt" \
    python -X faulthandler -m pytest tests -m gpu -x -q -p no:cacheprovider > "$OUT/full_$1.log" 2>&1
}
suite 1; rc=$?
echo "suite run 1 rc=$rc $(tail -1 "$OUT/full_1.log")"
crashes=0
if [ $rc -ne 0 ]; then
  tail -60 "$OUT/full_1.log"
  if grep -q "abort_trace: signal" "$OUT/full_1.log"; then crashes=1; mv "$OUT/full_1.log" "$OUT/CRASH_1.log"; fi
fi
if [ $rc -eq 0 ] || [ $crashes -eq 1 ]; then
  for i in $(seq 2 "$RUNS"); do
    suite "$i"; rc=$?
    echo "suite run $i rc=$rc $(tail -1 "$OUT/full_$i.log")"
    if [ $rc -ne 0 ]; then
      crashes=$((crashes + 1)); mv "$OUT/full_$i.log" "$OUT/CRASH_$i.log"
    else
      rm -f "$OUT/maps_$i.txt"; tail -3 "$OUT/full_$i.log" > "$OUT/full_$i.tail"; rm -f "$OUT/full_$i.log"
    fi
  done
  echo "suite: $crashes crashes in $RUNS runs"
fi
timeout 300 tests/cpp/host_call_budget 2000 > "$OUT/host_call_budget.jsonl" 2> "$OUT/host_call_budget.err"
echo "host_call_budget rc=$?"; cat "$OUT/host_call_budget.jsonl"
BENCH_ONE_DEVICE=1 BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 3 --warmup 1 \
  > "$OUT/two_rank_dry_run.json" 2> "$OUT/two_rank_dry_run.err"
echo "two-rank dry run rc=$?"; python - "$OUT/two_rank_dry_run.json" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print({k: d.get(k) for k in ("value", "scaling", "ms_per_step", "rendezvous")}, "weak block:", d.get("weak"))
except Exception as e:
    print("no line:", e)
PY
timeout 900 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"
echo "bench rc=$?"; python - "$OUT/bench.json" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print({k: d.get(k) for k in ("value", "ms_per_step")}, d["roofline"]["frac"], d["roofline"]["transform_frac"])
    print(json.dumps(d.get("composites"), indent=1)[:3000])
    print(json.dumps(d.get("host_path", {}).get("cpp_budget"), indent=1)[:1500])
except Exception as e:
    print("no line:", e)
PY
