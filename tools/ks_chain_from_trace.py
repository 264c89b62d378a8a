"""Reads a rocprofv3 kernel trace (csv) of tools/ks_single_trace.py and prints, for the last launch-by-launch and the
last replayed KeySwitch call, each kernel's duration and the idle gap in front of it.
    python tools/ks_chain_from_trace.py <trace dir>"""
import csv
import glob
import sys

path = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
mac = [i for i, k in enumerate(names) if "ks_mac_kernel" in k]
per = mac[1] - mac[0]   # kernels per call
before = mac[0]         # kernels of a call in front of its multiply-accumulate (the trace starts with a call)
assert before < per, (before, per)
for label, pick in (("launch by launch", mac[24]), ("replayed", mac[-1])):  # 25 calls per mode
    call = rows[pick - before:pick - before + per]
    print(f"== {label}: {len(call)} kernels")
    t_prev = None
    for r in call:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        gap = (s - t_prev) / 1e3 if t_prev is not None else 0.0
        print(f"  gap {gap:6.2f} us | {(e - s) / 1e3:6.2f} us | grid {r.get('Grid_Size_X', '?'):>7s} | {r['Kernel_Name'][:100]}")
        t_prev = e
    print(f"  first start to last end: {(int(call[-1]['End_Timestamp']) - int(call[0]['Start_Timestamp'])) / 1e3:.1f} us")
