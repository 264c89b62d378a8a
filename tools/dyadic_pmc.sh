#!/bin/bash
# Counter passes (each in its own run, --pmc only) over tools/dyadic_probe.py: why the batched
# DyadicMultiply streams below EltwiseMultMod.  Output under OUTDIR; summarise by hand.
set -u
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=${1:-$REPO/gpurun_out/dyadic_pmc}
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o t -- python "$REPO/tools/dyadic_probe.py" > /dev/null 2> "$OUT/trace.log"
for pass in "sq SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE" "fetch FETCH_SIZE" "write WRITE_SIZE"; do
  set -- $pass; name=$1; shift
  timeout 200 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -o p -- python "$REPO/tools/dyadic_probe.py" > /dev/null 2> "$OUT/$name.log"
done
python - "$OUT" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
out = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(out, "*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "dyadic" in k or "eltwise_vec2" in k:
            acc[k.split("(")[0][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = {}
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "dyadic" in r["Name"] or "eltwise_vec2" in r["Name"]:
            dur[r["Name"].split("(")[0][:60]] = float(r["AverageNs"])
for k, c in acc.items():
    m = {n: sum(v) / len(v) for n, v in c.items()}
    ns = dur.get(k)
    gui = m.get("GRBM_GUI_ACTIVE", 0)
    line = {"kernel": k, "avg_us": ns and ns / 1e3,
            "valu_busy": m.get("SQ_ACTIVE_INST_VALU", 0) * 4 / (1024 * gui / 8) if gui else None,
            "valu_insts_per_wave": m.get("SQ_INSTS_VALU", 0) / m["SQ_WAVES"] if m.get("SQ_WAVES") else None,
            "waves_per_simd": m.get("SQ_WAVE_CYCLES", 0) * 4 / (1024 * gui / 8) if gui else None,
            "hbm_bytes": (2 * m.get("FETCH_SIZE", 0) + m.get("WRITE_SIZE", 0)) * 1024,
            "clock_GHz": (gui / 8) / ns if ns and gui else None}
    print(line)
PY
