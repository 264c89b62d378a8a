#!/bin/bash
# tools/pmc_one.sh OUT "python snippet args..." -- run ON THE GPU BOX: the SQ / LDS counter passes and a kernel trace over
# `python tools/pmc_probe.py ARGS` (one transform shape under a tuning setting), for A/B questions about one kernel.
set -u
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$1; shift
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
CMD=(python $REPO/tools/pmc_probe.py "$@")
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- "${CMD[@]}" > $OUT/trace.out 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq -o p -- "${CMD[@]}" > /dev/null 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE --output-format csv -d $OUT/lds -o p -- "${CMD[@]}" > /dev/null 2>&1
python - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
dur = collections.defaultdict(list)
for f in glob.glob(out + "/trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Kernel_Name"].split("(")[0]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for sub in ("sq", "lds"):
    for f in glob.glob(out + f"/{sub}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    if "tile" not in k and "strided" not in k:
        continue
    c = {n: sum(v) / len(v) for n, v in d.items()}
    cyc = c["GRBM_GUI_ACTIVE"] / 8
    ns = sorted(dur[k])[len(dur[k]) // 2] if dur.get(k) else float("nan")
    print(k.replace("hexl_amd::", "")[:70], f"us={ns/1e3:.1f} valu_busy={c['SQ_ACTIVE_INST_VALU']*4/(1024*cyc):.3f} "
          f"waves/simd={c['SQ_WAVE_CYCLES']*4/(1024*cyc):.2f} wait={c['SQ_WAIT_ANY']/c['SQ_WAVE_CYCLES']:.3f} "
          f"stall={c['SQ_WAIT_INST_ANY']/c['SQ_WAVE_CYCLES']:.3f} vmem_busy={c['SQ_ACTIVE_INST_VMEM']*4/(1024*cyc):.3f} "
          f"lds_busy={c['SQ_LDS_IDX_ACTIVE']/(256*cyc):.3f} clock={cyc/ns:.2f}GHz valu/wave={c['SQ_INSTS_VALU']/c['SQ_WAVES']:.0f}")
PY
