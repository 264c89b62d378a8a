#!/bin/bash
# Round 5, GPU pass 3: the suite under the preload in a loop (reports into their own files:
# ABORT_TRACE_LOG), then the A/B tools of the round.
set -u
OUT=${1:-gpurun_out/r5c}
RUNS=${2:-24}
mkdir -p "$OUT"
TRACE=$PWD/tools/libabort_trace.so
crashes=0
for i in $(seq 1 "$RUNS"); do
  timeout 600 env LD_PRELOAD="$TRACE" ABORT_TRACE_LOG="$PWD/$OUT/trace_$i.txt" ABORT_TRACE_MAPS="$PWD/$OUT/maps_$i.txt" \
    python -X faulthandler -m pytest tests -m gpu -x -q -p no:cacheprovider > "$OUT/full_$i.log" 2>&1
  rc=$?
  echo "suite run $i rc=$rc $(tail -1 "$OUT/full_$i.log")"
  if [ $rc -ne 0 ]; then
    crashes=$((crashes + 1)); mv "$OUT/full_$i.log" "$OUT/CRASH_$i.log"
    [ -s "$OUT/trace_$i.txt" ] && cat "$OUT/trace_$i.txt"
    if ! grep -q "abort_trace: signal" "$OUT/trace_$i.txt" 2>/dev/null; then tail -40 "$OUT/CRASH_$i.log"; fi
    if [ $i -eq 1 ] && ! grep -q "abort_trace: signal" "$OUT/trace_$i.txt" 2>/dev/null; then break; fi
  else
    rm -f "$OUT/maps_$i.txt" "$OUT/trace_$i.txt"; tail -3 "$OUT/full_$i.log" > "$OUT/full_$i.tail"; rm -f "$OUT/full_$i.log"
  fi
done
echo "suite: $crashes crashes in $RUNS runs"
timeout 300 python tools/strict8_ab.py > "$OUT/strict8_ab.txt" 2>&1; cat "$OUT/strict8_ab.txt"
timeout 300 python tools/small_batch_ab.py > "$OUT/small_batch_ab.txt" 2>&1; cat "$OUT/small_batch_ab.txt"
timeout 300 python tools/fp64_long_ab.py > "$OUT/fp64_long_ab.txt" 2>&1; cat "$OUT/fp64_long_ab.txt"
timeout 300 python - > "$OUT/dyadic.txt" 2>&1 <<'PY'
import sys
sys.path.insert(0, ".")
import bench, hexl_amd as hx
c = bench.composites(hx)
print({k: c[k] for k in ("dyadic_multiply", "dyadic_multiply_batch")})
PY
cat "$OUT/dyadic.txt"
