#!/bin/bash
# Round 5, last evidence pass ON THE GPU BOX (final build): soak of the suite, the C++ host-call
# budget, the sweep columns of the Fp64L range, the DyadicMultiply counter passes, the bench line.
set -u
OUT=${1:-gpurun_out/r5g}
RUNS=${2:-30}
mkdir -p "$OUT"
tools/suite_soak.sh "$OUT" "$RUNS"
timeout 300 tests/cpp/host_call_budget 2000 > "$OUT/host_call_budget.jsonl" 2> "$OUT/host_call_budget.err"; cat "$OUT/host_call_budget.jsonl"
SWEEP_BITS=36,44 SWEEP_LOGN_MAX=16 timeout 300 python tools/size_sweep.py > "$OUT/size_sweep_fp64l.md" 2> "$OUT/size_sweep.err"; cat "$OUT/size_sweep_fp64l.md"
timeout 900 tools/dyadic_pmc.sh "$PWD/$OUT/dyadic_pmc" > "$OUT/dyadic_pmc.txt" 2>&1; tail -4 "$OUT/dyadic_pmc.txt"
timeout 900 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "bench rc=$?"
python - "$OUT/bench.json" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print({k: d.get(k) for k in ("value", "ms_per_step")}, d["roofline"]["frac"], d["roofline"]["transform_frac"], d["roofline"]["traffic"])
    print(json.dumps({k: v for k, v in d["host_path"].items() if k.startswith("N=")})[:1200])
    print(json.dumps(d["composites"]["dyadic_multiply_batch"]))
except Exception as e:
    print("no line:", e)
PY
