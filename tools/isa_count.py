"""Static instruction budget of one kernel (no GPU needed): splits a kernel's gfx950 assembly into
basic blocks and prints, per block and in total, VALU instructions by class (v_mad_u64_u32, other
multiply-class, 64-bit simple, 32-bit simple), LDS and global memory instructions.  The tile and
strided passes are straight-line code behind a few uniform branches, so the executed path is a
set of blocks the caller names (--skip drops blocks by label, e.g. the first-pass conversion).
Usage: python tools/isa_count.py file.s 'tile_pass<true, 11, 0, 11, false, hexl_amd::Lazy, false>' [--skip LBB43_3,...] [--blocks]"""
import re
import subprocess
import sys

MUL = ("v_mad_u64_u32", "v_mul_lo_u32", "v_mul_hi_u32", "v_mad_u32_u24", "v_mul_u32_u24", "v_mad_i32_i24",
       "v_fma_f64", "v_mul_f64", "v_add_f64", "v_rndne_f64")


def kernel_body(txt, want):
    names = re.findall(r"^(_Z\S+):", txt, re.M)
    dem = subprocess.run(["c++filt"], input="\n".join(names), text=True, capture_output=True).stdout.split("\n")
    for mangled, pretty in zip(names, dem):
        if pretty.startswith("void hexl_amd::" + want + "(") or pretty.startswith(want + "("):
            start = txt.index("\n" + mangled + ":")
            end = txt.index(".Lfunc_end", start)
            return pretty, txt[start:end]
    raise SystemExit("kernel not found: " + want)


def classify(op):
    if op.startswith("v_mad_u64_u32"):
        return "mad64"
    if op.startswith(MUL):
        return "mul"
    if op.startswith("v_"):
        return "valu64" if re.search(r"(_u64|_b64|_i64|pk_mov)", op) else "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_")):
        return "vmem"
    if op.startswith("s_load") or op.startswith("s_buffer_load"):
        return "smem"
    if op in ("s_waitcnt", "s_barrier", "s_nop"):
        return op
    return "salu"


def main():
    path, want = sys.argv[1], sys.argv[2]
    skip = set()
    show_blocks = "--blocks" in sys.argv
    if "--skip" in sys.argv:
        skip = set(sys.argv[sys.argv.index("--skip") + 1].split(","))
    pretty, body = kernel_body(open(path).read(), want)
    blocks, cur = [], ["entry", {}]
    hist = {}
    for line in body.split("\n"):
        m = re.match(r"^(\.LBB\d+_\d+):", line) or re.match(r"^; %(bb\.\d+):", line)
        if m:
            blocks.append(cur)
            cur = [m.group(1).lstrip("."), {}]
            continue
        m = re.match(r"^\t([a-z]\w+)", line)
        if not m:
            continue
        c = classify(m.group(1))
        cur[1][c] = cur[1].get(c, 0) + 1
        if cur[0] not in skip and c in ("valu", "valu64", "mul"):
            hist[m.group(1)] = hist.get(m.group(1), 0) + 1
    blocks.append(cur)
    total = {}
    for name, cnt in blocks:
        if show_blocks:
            print(f"  {name:12s} " + " ".join(f"{k}={v}" for k, v in sorted(cnt.items())) + ("   [skipped]" if name in skip else ""))
        if name in skip:
            continue
        for k, v in cnt.items():
            total[k] = total.get(k, 0) + v
    valu = sum(total.get(k, 0) for k in ("mad64", "mul", "valu", "valu64"))
    print(pretty.split("(")[0])
    print(f"  VALU {valu}: mad64 {total.get('mad64', 0)}, other multiply-class {total.get('mul', 0)}, "
          f"64-bit simple {total.get('valu64', 0)}, 32-bit simple {total.get('valu', 0)}; "
          f"LDS {total.get('lds', 0)}, VMEM {total.get('vmem', 0)}, SMEM {total.get('smem', 0)}, "
          f"s_waitcnt {total.get('s_waitcnt', 0)}, s_barrier {total.get('s_barrier', 0)}, s_nop {total.get('s_nop', 0)}")
    print("  non-mad VALU: " + ", ".join(f"{k} {v}" for k, v in sorted(hist.items(), key=lambda kv: -kv[1])))


main()
