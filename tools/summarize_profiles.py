"""Turn gpurun_out/prof/ (tools/collect_profiles.sh) into the committed summaries:

    profiles/rN_kernel_stats.csv   rocprofv3 --stats per-kernel table
    profiles/rN_pmc_summary.md     counters per dispatch, HBM traffic vs algorithmic
    profiles/rN_counters.json      per bench kernel family: HBM bytes per launch, VALU busy,
                                   VALU instructions per wave, shader clock -- with a hash of
                                   the kernel sources they were measured on; read by bench.py
(N = the round, argv[1], default 3)

HBM bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: on gfx950 FETCH_SIZE counts
streaming reads at half their size (calibrated below on copy kernels of known
traffic, as MI355X_MICROARCH.md prescribes), WRITE_SIZE is exact.
"""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "prof")
DST = os.path.join(ROOT, "profiles")
ROUND = sys.argv[1] if len(sys.argv) > 1 else "3"
N, BATCH = 65536, 4096
ALG = 16.0 * N * BATCH


def counters(name):
    acc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(SRC, name, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in acc.items()}


def short(k):
    k = k.replace("hexl_amd::", "")
    return k.split("(")[0]


def family(k):
    fwd = "<true" in k
    if "strided_pass" in k:
        return "ntt_fwd_strided_pass" if fwd else "ntt_inv_strided_pass"
    if "tile_pass" in k:
        return "ntt_fwd_tile_pass_bottom" if fwd else "ntt_inv_tile_pass_bottom"
    return None


def main():
    stats = glob.glob(os.path.join(SRC, "trace", "**", "*kernel_stats.csv"), recursive=True)
    if stats:
        shutil.copy(stats[0], os.path.join(DST, f"r{ROUND}_kernel_stats.csv"))
    fetch, write = counters("fetch"), counters("write")
    calf, calw = counters("cal_fetch"), counters("cal_write")
    sq, lds, sq2 = counters("sq"), counters("lds"), counters("sq2")
    durations = {}  # kernel -> average ns (rocprofv3 --stats of the traced run)
    for f in stats:
        for r in csv.DictReader(open(f)):
            durations[r["Name"]] = float(r["AverageNs"])
    out = [f"# Round-{ROUND} PMC summary (rocprofv3 --pmc, separate passes; averages per dispatch)", "",
           "Collected by `tools/collect_profiles.sh`, summarised by `tools/summarize_profiles.py`.",
           "Command: `rocprofv3 --pmc <counters> -- python bench.py --steps 3 --warmup 1 "
           "--no-cpu-baseline` (N=65536, 55-bit q, batch 4096; default plan = strided_pass<5 "
           "stages> + tile_pass<11 stages on 2048-element tiles>).", "",
           "## HBM traffic calibration (tools/ubench copy kernels: exactly 2 GiB read + 2 GiB "
           "written per launch)", "",
           "| kernel | counter | avg per dispatch (KB) | true KB | factor |", "|---|---|---|---|---|"]
    for cal, cname in ((calf, "FETCH_SIZE"), (calw, "WRITE_SIZE")):
        for k, d in cal.items():
            if "copy" in k and cname in d:
                out.append(f"| {short(k)} | {cname} | {d[cname]:.1f} | 2097152 | "
                           f"{2097152 / d[cname]:.3f} |")
    out += ["", "HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024.", "",
            "## NTT kernels", "",
            "| kernel | FETCH_SIZE (KB) | WRITE_SIZE (KB) | HBM bytes/launch (corrected) | "
            "algorithmic bytes/launch | ratio |", "|---|---|---|---|---|---|"]
    traffic, fam = {}, {}
    for k in fetch:
        if family(k) is None or k not in write:
            continue
        f, w = fetch[k]["FETCH_SIZE"], write[k]["WRITE_SIZE"]
        hbm = (2 * f + w) * 1024
        traffic[short(k)] = hbm
        fam[family(k)] = hbm
        out.append(f"| {short(k)} | {f:.0f} | {w:.0f} | {hbm:.4g} | {ALG:.4g} | {hbm / ALG:.4f} |")
    out += ["", "## SQ / LDS counters", "", "| kernel | counter | avg per dispatch |", "|---|---|---|"]
    for table in (sq, sq2, lds):
        for k, d in table.items():
            if family(k) is None:
                continue
            for c in sorted(d):
                out.append(f"| {short(k)} | {c} | {d[c]:.4g} |")
    # Derived figures.  SQ_ACTIVE_INST_* and SQ_WAVE_CYCLES count quad-cycles summed over waves
    # (MI355X_MICROARCH.md); GRBM_GUI_ACTIVE is summed over the 8 XCDs, so one XCD's busy
    # cycles are a eighth of it.  VALU busy = quad-cycles with a VALU instruction in flight *
    # 4 / (1024 SIMDs * cycles the kernel lasted).
    out += ["", "## Derived per kernel", "",
            "VALU busy = SQ_ACTIVE_INST_VALU * 4 / (1024 SIMDs * GRBM_GUI_ACTIVE / 8 XCDs); shader "
            "clock = (GRBM_GUI_ACTIVE / 8) / kernel duration of the traced run; wave occupancy = "
            "SQ_WAVE_CYCLES * 4 / (1024 SIMDs * cycles) waves per SIMD.", "",
            "| kernel | VALU instr / wave | VALU busy | any-instruction busy | waves / SIMD | "
            "wave-cycles waiting (s_waitcnt, barrier) | shader clock (GHz) | avg ms |",
            "|---|---|---|---|---|---|---|---|"]
    fam_counters = {}
    for k, d in sq.items():
        if not family(k) or not d.get("SQ_WAVES") or not d.get("GRBM_GUI_ACTIVE"):
            continue
        cycles = d["GRBM_GUI_ACTIVE"] / 8.0
        simd_cycles = 1024.0 * cycles
        valu_busy = d["SQ_ACTIVE_INST_VALU"] * 4.0 / simd_cycles
        any_busy = d["SQ_ACTIVE_INST_ANY"] * 4.0 / simd_cycles
        occupancy = d["SQ_WAVE_CYCLES"] * 4.0 / simd_cycles
        waiting = d.get("SQ_WAIT_ANY", 0.0) / d["SQ_WAVE_CYCLES"]
        ns = durations.get(k) or durations.get(k.split("(")[0])
        if ns is None:
            ns = next((v for n, v in durations.items() if short(n) == short(k)), None)
        clock = cycles / ns if ns else None
        per_wave = d["SQ_INSTS_VALU"] / d["SQ_WAVES"]
        out.append(f"| {short(k)} | {per_wave:.0f} | {valu_busy:.3f} | {any_busy:.3f} | {occupancy:.2f} | "
                   f"{waiting:.3f} | {clock if clock is None else round(clock, 3)} | "
                   f"{ns / 1e6 if ns else float('nan'):.3f} |")
        fam_counters[family(k)] = {
            "kernel": short(k), "hbm_bytes_per_launch": fam.get(family(k)),
            "valu_busy": valu_busy, "any_inst_busy": any_busy, "valu_insts_per_wave": per_wave,
            "waves_per_simd": occupancy, "wave_cycles_waiting_frac": waiting,
            "shader_clock_GHz": clock, "traced_avg_ms": ns / 1e6 if ns else None}
    open(os.path.join(DST, f"r{ROUND}_pmc_summary.md"), "w").write("\n".join(out) + "\n")
    sys.path.insert(0, ROOT)
    import bench
    json.dump({"source": f"profiles/r{ROUND}_pmc_summary.md (rocprofv3 --pmc, separate passes; FETCH_SIZE x2 "
               "gfx950 correction; VALU busy = SQ_ACTIVE_INST_VALU * 4 / (1024 * GRBM_GUI_ACTIVE / 8))",
               "kernel_source_sha16": bench.kernel_source_hash(),
               "kernel_sources": list(bench.KERNEL_SOURCES),
               "hbm_bytes_per_launch": traffic, "by_bench_kernel_family": fam_counters,
               "workload": f"N={N}, q=18014398510661633, batch={BATCH} (bench.py default), one launch"},
              open(os.path.join(DST, f"r{ROUND}_counters.json"), "w"), indent=1)
    print("\n".join(out[-8:]))


if __name__ == "__main__":
    sys.exit(main())
