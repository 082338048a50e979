#!/usr/bin/env python3
"""Register / LDS / scratch use of every kernel in a HIP source, from the compiler's own metadata.

    python scripts/kernel_resources.py gt4py_amd/csrc/gt4mi.hip [filter]

Compiles the file for gfx950 (device only, to assembly) with the library's flags and prints one line per kernel:
VGPRs, AGPRs, SGPRs, LDS bytes, scratch bytes and the waves per SIMD that register use allows (512 registers
per lane and SIMD on gfx950, unified VGPR + AGPR file)."""
import pathlib
import re
import subprocess
import sys
import tempfile

ROOT = pathlib.Path(__file__).resolve().parent.parent


def demangle(names):
    try:
        out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True).stdout
        return out.splitlines()
    except Exception:
        return names


def main():
    src = pathlib.Path(sys.argv[1])
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    with tempfile.TemporaryDirectory() as tmp:
        asm = pathlib.Path(tmp) / "k.s"
        subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-w",
                        "-mllvm", "-pragma-unroll-threshold=262144",  # as csrc/Makefile
                        f"-I{ROOT / 'include'}", f"-I{ROOT / 'gt4py_amd' / 'csrc'}", "--cuda-device-only", "-S", "-o",
                        str(asm), str(src)], check=True)
        text = asm.read_text()
    rows = []
    for m in re.finditer(r"- \.agpr_count:\s+(\d+).*?\.group_segment_fixed_size:\s+(\d+).*?\.name:\s+(\S+).*?"
                         r"\.private_segment_fixed_size:\s+(\d+).*?\.sgpr_count:\s+(\d+).*?\.vgpr_count:\s+(\d+)", text, re.S):
        ag, lds, name, priv, sg, vg = m.groups()
        rows.append((name, int(vg), int(ag), int(sg), int(lds), int(priv)))
    names = demangle([r[0] for r in rows])
    for dn, (_, vg, ag, sg, lds, priv) in zip(names, rows):
        dn = re.sub(r"\(.*", "", dn).replace("void ", "").replace("gt4mi::", "")
        if flt and flt not in dn:
            continue
        total = -(-vg // 8) * 8  # .vgpr_count already includes the AGPRs (unified file)
        waves = min(8, 512 // max(total, 1))
        print(f"{dn[:100]:100s} vgpr={vg:3d} agpr={ag:3d} sgpr={sg:3d} lds={lds:6d} scratch={priv:4d} waves/simd<={waves}")


if __name__ == "__main__":
    main()
