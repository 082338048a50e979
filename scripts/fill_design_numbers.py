#!/usr/bin/env python3
"""Fill the `@NAME@` placeholders of DESIGN.md section 0 from profiles/<tag>_all_kernels_summary.json (rocprofv3 averages + PMC
traffic) and a bench line: `python scripts/fill_design_numbers.py profiles/r6_all_kernels_summary.json profiles/r6_bench_lap512_n1_final.json`.
Idempotent on a template kept in docs/DESIGN_section0.template (written on the first run)."""
import json
import pathlib
import sys

ROOT = pathlib.Path(__file__).resolve().parent.parent
summary = json.loads(pathlib.Path(sys.argv[1]).read_text())["kernels"]
line = json.loads([ln for ln in pathlib.Path(sys.argv[2]).read_text().splitlines() if ln.startswith("{")][-1])
design = ROOT / "DESIGN.md"
template = ROOT / "docs" / "DESIGN.template.md"
text = template.read_text() if template.exists() else design.read_text()
if not template.exists():
    template.write_text(text)
LUPS = {"lap5_f64_512": 512**3, "laplacian_f64_512x512x128_config1": 512 * 512 * 128, "hdiff_limiter_f32_1024x1024x80": 1024 * 1024 * 80,
        "hdiff_limiter_f32_literal32_1024x1024x80": 1024 * 1024 * 80, "hdiff_limiter_f64_512x1024x80": 512 * 1024 * 80,
        "tridiagonal_f64_1024x1024x160": 1024 * 1024 * 160, "generated_vertical_advection_f64_1024x1024x160": 1024 * 1024 * 160,
        "generated_laplacian_f64_512x512x512": 512**3, "generated_hdiff_limiter_f64_512x1024x80": 512 * 1024 * 80}
KEYS = {"LAP": "lap5_f64_512", "C1": "laplacian_f64_512x512x128_config1", "H32": "hdiff_limiter_f32_1024x1024x80",
        "H32L": "hdiff_limiter_f32_literal32_1024x1024x80", "TRI": "tridiagonal_f64_1024x1024x160", "H64": "hdiff_limiter_f64_512x1024x80",
        "VADV": "generated_vertical_advection_f64_1024x1024x160", "GLAP": "generated_laplacian_f64_512x512x512",
        "GHD": "generated_hdiff_limiter_f64_512x1024x80"}
for tag, name in KEYS.items():
    k = summary[name]
    ms = k["average_ns"] / 1e6
    text = (text.replace(f"@{tag}_MS@", f"{ms:.4f}").replace(f"@{tag}_GLUPS@", f"{LUPS[name] / ms / 1e6:.1f}")
            .replace(f"@{tag}_FRAC@", f"{k['frac_of_hbm_peak']:.3f}").replace(f"@{tag}_TR@", f"{k['traffic_over_algorithmic']:.3f} x"))
tri = (line.get("other_kernels") or {}).get("tridiagonal_f64_1024x1024x160") or {}
sets = tri.get("frac_of_hbm_peak_by_allocation_set") or []
text = text.replace("@TRI_SETS@", " / ".join(f"{v:.3f}" for v in sets) if sets else "n/a")
text = text.replace("@VAL_DEFAULT@", str(line.get("value_default_allocator"))).replace("@VAL_OFF@", str(line.get("value_allocator_off")))
text = text.replace("@VALUE@", str(line.get("value"))).replace("@VALUE_FRAC@", str(line.get("roofline", {}).get("frac")))
design.write_text(text)
left = [w for w in text.split() if w.startswith("@") and w.endswith("@")]
print("placeholders left:", left)
