#!/bin/bash
# Round 5 evidence pass ON THE GPU BOX: soak of the suite, the C++ host-call budget, the profile
# collection behind the bench line's counters, the size sweep, the Lazy-family A/B, the bench line.
set -u
OUT=${1:-gpurun_out/r5f}
RUNS=${2:-24}
mkdir -p "$OUT"
tools/suite_soak.sh "$OUT" "$RUNS"
timeout 300 tests/cpp/host_call_budget 2000 > "$OUT/host_call_budget.jsonl" 2> "$OUT/host_call_budget.err"; cat "$OUT/host_call_budget.jsonl"
timeout 1200 tools/collect_profiles.sh > "$OUT/collect.log" 2>&1; tail -5 "$OUT/collect.log"
( cd "$PWD" && timeout 400 python tools/size_sweep.py > "$OUT/size_sweep.md" 2> "$OUT/size_sweep.err" ); tail -3 "$OUT/size_sweep.md"
timeout 300 python tools/lazy_family_ab.py > "$OUT/lazy_family_ab.txt" 2>&1; tail -20 "$OUT/lazy_family_ab.txt"
timeout 900 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "bench rc=$?"
python - "$OUT/bench.json" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print({k: d.get(k) for k in ("value", "ms_per_step")}, d["roofline"]["frac"], d["roofline"]["transform_frac"], d["roofline"]["traffic"])
    print(json.dumps(d["composites"]["key_switch"])[:1500])
    print(json.dumps({k: v for k, v in d["host_path"].items() if k.startswith("N=")})[:1200])
except Exception as e:
    print("no line:", e)
PY
