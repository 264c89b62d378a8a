"""Static audit of the compiled kernels (no GPU needed): compiles every translation unit of the
device code to gfx950 assembly and reports, per kernel, VGPRs / SGPRs / scratch bytes, and flags what
cost this project measurable time before it was noticed in the ISA:
  * scratch (register spills) above a few bytes,
  * flat_load / flat_store (a pointer the compiler could not place in an address space),
  * tile / strided kernels whose wave-uniform twiddles are not in SGPRs (next_free_sgpr far below
    the single-plan kernels' ~60: the twiddles came through the vector memory path).
Usage: python tools/isa_audit.py [--all]   (default: only flagged kernels)"""
import os
import re
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "hexl_amd", "csrc")
UNITS = [("ntt_kernels.hip", f"-DHEXL_AMD_TU={tu}", f"ntt tu{tu}") for tu in range(8)] + [
    ("eltwise_kernels.hip", None, "eltwise"), ("keyswitch_kernels.hip", None, "keyswitch")]


def asm(unit):
    src, flag, label = unit
    out = os.path.join(tempfile.gettempdir(), f"isa_audit_{label.replace(' ', '_')}.s")
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17",
           f"-I{os.path.join(ROOT, 'include')}", f"-I{CSRC}", "-S", "--cuda-device-only", "-o", out,
           os.path.join(CSRC, src)] + ([flag] if flag else [])
    subprocess.run(cmd, check=True, capture_output=True)
    return label, open(out).read()


def demangle(names):
    try:
        r = subprocess.run(["c++filt"], input="\n".join(names), text=True, capture_output=True)
        return r.stdout.split("\n") if r.returncode == 0 else names
    except OSError:
        return names


show_all = "--all" in sys.argv
flagged = 0
with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 4)) as ex:
    for label, txt in ex.map(asm, UNITS):
        rows = []
        for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", txt, re.S):
            name, body = m.group(1), m.group(2)
            num = lambda k: int(re.search(k + r" (\d+)", body).group(1))  # noqa: E731
            start = txt.index("\n" + name + ":")
            code = txt[start:txt.index("\t.amdhsa_kernel " + name, start)] if "\t.amdhsa_kernel " + name in txt[start:] else ""
            rows.append((name, num("next_free_vgpr"), num("next_free_sgpr"),
                         num("private_segment_fixed_size"), len(re.findall(r"\tflat_(?:load|store)", code))))
        names = demangle([r[0] for r in rows])
        for (name, vg, sg, scratch, flat), pretty in zip(rows, names):
            why = []
            # (a few dwords parked on a cold path are noise: the Fp64 five-stage forward strided
            # pass holds 64 data VGPRs at its 128-VGPR cap and parks five dwords, 20 bytes, across
            # its finish branch; it had 12 bytes and 14 SGPR spills before round 3)
            if scratch > 32:
                why.append(f"scratch {scratch} B")
            if flat:
                why.append(f"{flat} flat accesses")
            experimental = "fused_pass" in name or "mixed_pass" in name
            big_tile = re.search(r"tile_pass(_multi)?<(true|false), 1[1-4],", pretty)
            if big_tile and not experimental and sg < 45 and vg >= 60:
                why.append("uniform values in VGPRs?")
            if why and not experimental:
                flagged += 1
            if show_all or (why and not experimental):
                print(f"[{label}] {pretty.split('(')[0][:90]}: vgpr {vg} sgpr {sg} scratch {scratch}"
                      + (f"  <-- {', '.join(why)}" if why else ""))
print(f"isa_audit: {flagged} kernel(s) flagged")
