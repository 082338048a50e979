#!/usr/bin/env python3
"""Registers / LDS / scratch of the kernels hip_codegen generates for a stencil of tests/stencil_zoo.py, from the
compiler's metadata -- no GPU needed (hipcc cross-compiles the generated source to gfx950 assembly).

    GT4MI_CODEGEN_TOP_CACHE=32,163840 python scripts/generated_kernel_resources.py vertical_advection_dycore [-k]

-k keeps the source and the assembly next to each other under /tmp/gt4mi_gen_<name>.{hip,s}."""
import os
import pathlib
import re
import subprocess
import sys

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

import stencil_zoo as zoo  # noqa: E402
from gt4py_amd.cartesian import gtscript  # noqa: E402


def main():
    name = sys.argv[1]
    defn, externals, _, _ = zoo.ZOO[name]
    # building the stencil class generates the source; nothing is compiled by hiprtc or launched before the first call
    obj = gtscript.stencil(backend="hip:mi300", definition=defn, externals=externals, use_kernel_library=False, rebuild=True)
    prog = type(obj)._gt_program_
    src = pathlib.Path(f"/tmp/gt4mi_gen_{name}.hip")
    asm = src.with_suffix(".s")
    src.write_text(prog.source)
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-w",
                    "-include", "hip/hip_runtime.h",  # hiprtc predefines what this header declares
                    "-DGT4MI_UNIT_I_STRIDE=1", "-DGT4MI_NO_ALIAS=1", "--cuda-device-only", "-S", "-o", str(asm), str(src)],
                   check=True)
    text = asm.read_text()
    for m in re.finditer(r"- \.agpr_count:\s+(\d+).*?\.group_segment_fixed_size:\s+(\d+).*?\.name:\s+(\S+).*?"
                         r"\.private_segment_fixed_size:\s+(\d+).*?\.sgpr_count:\s+(\d+).*?\.vgpr_count:\s+(\d+)", text, re.S):
        ag, lds, kname, priv, sg, vg = m.groups()
        print(f"{kname:60s} vgpr={int(vg):3d} (agpr {int(ag):3d}) sgpr={int(sg):3d} lds={int(lds):6d} scratch={int(priv):5d}")
    if "-k" not in sys.argv:
        src.unlink()
        asm.unlink()


if __name__ == "__main__":
    main()
