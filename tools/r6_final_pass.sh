#!/bin/bash
# Round 6, evidence pass ON THE GPU BOX on the final build: the headline's rocprofv3 trace + counter passes
# (tools/collect_profiles.sh), the counter passes of every other kernel (tools/collect_pmc_cells.sh), the size sweep,
# the default bench line, a soak of the -m gpu suite.  Summarise afterwards (anywhere) with
#   python tools/summarize_profiles.py 6 ; python tools/summarize_pmc_cells.py 6
set -u
OUT=${1:-gpurun_out/r6f}
RUNS=${2:-30}
mkdir -p "$OUT"
timeout 900 bash tools/collect_profiles.sh > "$OUT/collect_profiles.log" 2>&1; tail -3 "$OUT/collect_profiles.log"
timeout 900 bash tools/collect_pmc_cells.sh > "$OUT/collect_pmc_cells.log" 2>&1; tail -3 "$OUT/collect_pmc_cells.log"
timeout 900 python tools/size_sweep.py > "$OUT/size_sweep.md" 2> "$OUT/size_sweep.err"; tail -3 "$OUT/size_sweep.md"
timeout 300 tests/cpp/host_call_budget 2000 > "$OUT/host_call_budget.jsonl" 2> "$OUT/host_call_budget.err"
timeout 900 python tests/bench_composites.py > "$OUT/composites.txt" 2>&1
timeout 900 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "bench rc=$?"
ok=0
for i in $(seq 1 "$RUNS"); do
  if timeout 600 python -m pytest tests -m gpu -x -q > "$OUT/suite_$i.log" 2>&1; then ok=$((ok + 1)); rm -f "$OUT/suite_$i.log"; else echo "suite run $i FAILED"; fi
done
echo "suite soak: $ok of $RUNS runs clean" | tee "$OUT/suite_soak.txt"
