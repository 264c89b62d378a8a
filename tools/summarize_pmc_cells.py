"""Turn gpurun_out/pmc_cells/ (tools/collect_pmc_cells.sh) into committed summaries:

    profiles/rN_pmc_cells.md      one table: every kernel tools/pmc_cells.py exercises, per
                                  (kernel, launch grid): duration of the traced run, algorithmic
                                  bytes and fraction of the 8 TB/s peak, HBM traffic from the
                                  counters and its ratio to the algorithmic bytes, VALU busy, waves
                                  per SIMD, wait fractions, LDS bank-conflict share
    profiles/rN_pmc_cells.json    the same as data, with the hash of the kernel sources it was
                                  measured on; bench.py reads it for its `secondary` blocks

    python tools/summarize_pmc_cells.py ROUND [SRC_DIR]

HBM bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: FETCH_SIZE counts streaming reads at half their
size on gfx950 (MI355X_MICROARCH.md; calibrated on copy kernels of known traffic in
profiles/r5_pmc_summary.md), WRITE_SIZE is exact.  SQ_* cycle counters are quad-cycles summed
over waves, GRBM_GUI_ACTIVE is summed over the 8 XCDs.
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROUND = sys.argv[1] if len(sys.argv) > 1 else "6"
SRC = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "gpurun_out", "pmc_cells")
DST = os.path.join(ROOT, "profiles")
SUFFIX = sys.argv[3] if len(sys.argv) > 3 else "cells"
PEAK = 8.0e12


def short(name):
    name = name.split("(")[0].replace("hexl_amd::", "").replace("void ", "")
    return name


MARK = "eltwise_vec2<AddOp"


def marker_cell(kernel, grid):
    """index of the cell a marker launch announces (tools/pmc_cells.py), or None"""
    if kernel.startswith(MARK) and grid % 256 == 0 and grid // 256 <= 4096 and grid < 4096 * 256:
        return grid // 256 - 1
    return None


def trace():
    out = defaultdict(list)
    meta = {}
    for f in glob.glob(os.path.join(SRC, "trace", "**", "*kernel_trace.csv"), recursive=True):
        cell = None
        for r in sorted(csv.DictReader(open(f)), key=lambda r: int(r["Dispatch_Id"])):
            grid = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
            kernel = short(r["Kernel_Name"])
            mc = marker_cell(kernel, grid)
            if mc is not None:
                cell = mc
                continue
            k = (cell, kernel, grid)
            out[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
            meta[k] = {"workgroup": int(r["Workgroup_Size_X"]), "lds_bytes": int(r["LDS_Block_Size"]),
                       "scratch": int(r["Scratch_Size"])}
    # median of the launches (a warm-up burst of the same call precedes the measured ones: pmc_cells.py --warm-ms)
    return {k: sorted(v)[len(v) // 2] for k, v in out.items()}, meta


def counters(name):
    acc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(SRC, name, "**", "*counter_collection.csv"), recursive=True):
        cell = None
        last_dispatch = None
        for r in sorted(csv.DictReader(open(f)), key=lambda r: int(r["Dispatch_Id"])):
            kernel, grid = short(r["Kernel_Name"]), int(r["Grid_Size"])
            mc = marker_cell(kernel, grid)
            if mc is not None:
                cell = mc
                continue
            acc[(cell, kernel, grid)][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in acc.items()}


def cell_of(kernel, grid, c):
    """(label, algorithmic bytes of ONE launch) of a kernel dispatched inside manifest cell c."""
    if c is None:
        return None, None
    lab = c["label"]
    if c["kind"] == "ntt":
        if kernel.startswith(("tile_pass<", "tile_walk_pass<", "strided_pass<")):
            return lab, 16 * (c["batch"] << c["logn"])  # one read + one write of the batch per pass
        return lab, None
    if c["kind"] == "eltwise":
        if kernel.startswith("eltwise_vec2<"):
            if c["op"] in ("reduce", "reduce41"):
                return lab, c["alg_bytes"]
            return lab, c["alg_bytes"]
        return lab, None
    if c["kind"] == "dyadic":
        return lab, c["alg_bytes"] if kernel.startswith("dyadic_multiply_kernel") else None
    if c["kind"] == "keyswitch":
        n, D, T, C = c["n"], c["D"], c["T"], 2
        K = D + 1
        poly = 8 * n
        if kernel.startswith("ks_mac_kernel"):
            return lab + " -- multiply-accumulate", poly * (T * (D * D + D + (D + 1) * C) + D * C * K)
        if kernel.startswith("ks_round_kernel"):
            return lab + " -- rounding", poly * T * C * (1 + D)
        if kernel.startswith("ks_finish_kernel"):
            return lab + " -- finish", poly * T * C * D * 4
        m = re.match(r"tile_pass_multi<(true|false), (\d+), (\d+), .*, (true|false), (true|false)>$", kernel)
        if m:
            S, TL, epi = int(m.group(2)), int(m.group(3)), m.group(5) == "true"
            polys = (grid // (1 << (TL - (4 if S >= 14 else 3))) << TL) >> S
            if epi:  # the fused tail: last component once per (target, component) + prod in, result in and out
                return lab + " -- fused tail (round | forward NTT | finish)", poly * (polys // D + 3 * polys)
            which = "forward NTT of the operands" if m.group(1) == "true" else "inverse NTT of the targets"
            return lab + " -- " + which, 2 * poly * polys
        if kernel.startswith(("tile_pass<false", "tile_walk_pass<false")):
            return lab + " -- inverse NTT of the last components", 2 * poly * T * C
        if kernel.startswith(("strided_pass", "tile_pass<", "tile_walk_pass")):
            return lab + " -- (a transform)", None
    return lab, None


def main():
    manifest = json.load(open(os.path.join(SRC, "manifest.json")))
    cells = manifest["cells"]
    dur, meta = trace()
    fetch, write, sq, sq2, lds = (counters(n) for n in ("fetch", "write", "sq", "sq2", "lds"))
    rows = []
    by_index = {c.get("index", i): c for i, c in enumerate(cells)}
    for k in sorted(dur, key=lambda k: (-1 if k[0] is None else k[0], k[1], k[2])):
        cell_index, kernel, grid = k
        if kernel.startswith("__amd") or "fill_splitmix" in kernel or cell_index is None:
            continue
        if kernel.startswith(("at::", "void at::")):
            continue
        label, alg = cell_of(kernel, grid, by_index.get(cell_index))
        ns = dur[k]
        row = {"kernel": kernel, "grid": grid, "cell": label, "traced_us": ns / 1e3, "alg_bytes": alg,
               **meta[k]}
        if alg:
            row["frac_of_peak"] = alg / (ns * 1e-9) / PEAK
        if k in fetch and k in write:
            row["hbm_bytes"] = (2 * fetch[k]["FETCH_SIZE"] + write[k]["WRITE_SIZE"]) * 1024
            if alg:
                row["traffic_ratio"] = row["hbm_bytes"] / alg
        d = sq.get(k)
        if d and d.get("GRBM_GUI_ACTIVE") and d.get("SQ_WAVES"):
            cyc = d["GRBM_GUI_ACTIVE"] / 8.0
            simd = 1024.0 * cyc
            row.update(valu_busy=d["SQ_ACTIVE_INST_VALU"] * 4 / simd, any_busy=d["SQ_ACTIVE_INST_ANY"] * 4 / simd,
                       waves_per_simd=d["SQ_WAVE_CYCLES"] * 4 / simd,
                       wait_any_frac=d["SQ_WAIT_ANY"] / d["SQ_WAVE_CYCLES"],
                       wait_inst_frac=d["SQ_WAIT_INST_ANY"] / d["SQ_WAVE_CYCLES"],
                       valu_per_wave=d["SQ_INSTS_VALU"] / d["SQ_WAVES"], waves=d["SQ_WAVES"],
                       clock_GHz=cyc / ns)
        d = lds.get(k)
        if d and d.get("SQ_LDS_IDX_ACTIVE"):
            row.update(lds_bank_conflict=d["SQ_LDS_BANK_CONFLICT"],
                       lds_conflict_share=d["SQ_LDS_BANK_CONFLICT"] / d["SQ_LDS_IDX_ACTIVE"],
                       lds_insts=d.get("SQ_INSTS_LDS"))
            if sq.get(k, {}).get("SQ_WAVE_CYCLES"):
                row["wait_inst_lds_frac"] = d["SQ_WAIT_INST_LDS"] / sq[k]["SQ_WAVE_CYCLES"]
                cyc = d["GRBM_GUI_ACTIVE"] / 8.0
                row["lds_busy"] = d["SQ_LDS_IDX_ACTIVE"] / (256.0 * cyc)  # LDS-array cycles per CU cycle
        elif d is not None:
            row.update(lds_bank_conflict=d.get("SQ_LDS_BANK_CONFLICT", 0.0), lds_conflict_share=0.0)
        d = sq2.get(k)
        if d and sq.get(k, {}).get("SQ_WAVES"):
            w = sq[k]["SQ_WAVES"]
            row.update(salu_per_wave=d["SQ_INSTS_SALU"] / w, smem_per_wave=d["SQ_INSTS_SMEM"] / w,
                       vmem_rd_per_wave=d["SQ_INSTS_VMEM_RD"] / w, vmem_wr_per_wave=d["SQ_INSTS_VMEM_WR"] / w)
        rows.append(row)

    def f(v, spec="{:.3f}"):
        return "—" if v is None else spec.format(v)

    out = [f"# Round-{ROUND} counter profile of every kernel outside the headline configuration", "",
           "`tools/collect_pmc_cells.sh` (one `rocprofv3 --kernel-trace --stats` pass for the durations, then",
           "FETCH_SIZE, WRITE_SIZE, two SQ sets and the LDS set each in its own `--pmc` pass over",
           "`python tools/pmc_cells.py`), summarised by `tools/summarize_pmc_cells.py`. Durations are medians of the",
           "traced run's launches (80 ms of the same call, then 6: the steady state, as `tools/size_sweep.py` measures it;",
           "the counter passes run 3 launches from a cold start -- their `GHz` column is the clock of THOSE launches). `frac` = algorithmic bytes / duration / 8 TB/s; `traffic` = (2·FETCH_SIZE +",
           "WRITE_SIZE)·1024 / algorithmic bytes; VALU busy = SQ_ACTIVE_INST_VALU·4 / (1024 SIMDs · GRBM_GUI_ACTIVE/8);",
           "`wait` = SQ_WAIT_ANY / SQ_WAVE_CYCLES (waves parked at s_waitcnt or a barrier), `stall` = SQ_WAIT_INST_ANY /",
           "SQ_WAVE_CYCLES (issue stalls), `LDS stall` = SQ_WAIT_INST_LDS / SQ_WAVE_CYCLES; `conflict` =",
           "SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE (extra LDS cycles over all LDS-array cycles); `LDS busy` =",
           "SQ_LDS_IDX_ACTIVE / (256 CUs · cycles).", ""]
    hdr = ("| kernel | cell | grid (threads) | µs | frac | traffic | VALU busy | waves/SIMD | wait | stall | LDS stall | "
           "conflict | LDS busy | VALU/wave | GHz |")
    sep = "|" + "---|" * 15
    groups = [("One-kernel transform plans and the headline shape under the cheap policies",
               lambda r: r["kernel"].startswith(("tile_pass<", "strided_pass<", "tile_walk_pass<"))),
              ("Element-wise kernels", lambda r: r["kernel"].startswith(("eltwise", "dyadic"))),
              ("KeySwitch (256 targets per call)", lambda r: "KeySwitch" in (r["cell"] or ""))]
    seen = set()
    for title, pred in groups:
        out += [f"## {title}", "", hdr, sep]
        for r in rows:
            if id(r) in seen or not pred(r):
                continue
            if title.startswith("One-kernel") and (r["cell"] is None or "KeySwitch" in r["cell"]):
                continue
            seen.add(id(r))
            out.append("| `{}` | {} | {} | {:.1f} | {} | {} | {} | {} | {} | {} | {} | {} | {} | {} | {} |".format(
                r["kernel"], r["cell"] or "(inside a composite)", r["grid"], r["traced_us"],
                f(r.get("frac_of_peak")), f(r.get("traffic_ratio"), "{:.4f}"), f(r.get("valu_busy")),
                f(r.get("waves_per_simd"), "{:.2f}"), f(r.get("wait_any_frac")), f(r.get("wait_inst_frac")),
                f(r.get("wait_inst_lds_frac")), f(r.get("lds_conflict_share"), "{:.4f}"), f(r.get("lds_busy")),
                f(r.get("valu_per_wave"), "{:.0f}"), f(r.get("clock_GHz"), "{:.2f}")))
        out.append("")
    os.makedirs(DST, exist_ok=True)
    open(os.path.join(DST, f"r{ROUND}_pmc_{SUFFIX}.md"), "w").write("\n".join(out) + "\n")
    sys.path.insert(0, ROOT)
    import bench
    json.dump({"source": f"profiles/r{ROUND}_pmc_{SUFFIX}.md", "kernel_source_sha16": bench.cell_source_hash(),
               "kernel_sources": list(bench.CELL_SOURCES), "rows": rows},
              open(os.path.join(DST, f"r{ROUND}_pmc_{SUFFIX}.json"), "w"), indent=1)
    print("\n".join(out))


if __name__ == "__main__":
    main()
