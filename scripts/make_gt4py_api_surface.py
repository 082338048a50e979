#!/usr/bin/env python3
"""Build-container script (reads /root/reference, which does not travel): the API SURFACE of the reference that the adapter's
gt4py-facing glue touches, as data.

    python scripts/make_gt4py_api_surface.py            # writes tests/golden/gt4py_api_surface.json
    python scripts/make_gt4py_api_surface.py --check

`gt4py_amd.adapter.register_with_gt4py` / `_wrap_for_gt4py` are the ~40 lines that subclass gt4py's `BaseBackend` and `StencilObject`
and read its `StencilBuilder`; gt4py cannot be imported in this image (Python 3.10 < 3.12), so they have never executed
(missing 2).  What CAN be pinned without importing: every name they use exists in the reference with the
arguments they pass.  This script parses the files below with `ast` and records, per module: its functions (with parameter
names), its module-level names and re-exports, and per class its methods (with parameter names), properties and attributes.
Names only -- no source text.  tests/test_adapter.py walks the glue's own source and checks every attribute chain against it.
"""

from __future__ import annotations

import argparse
import ast
import hashlib
import json
import pathlib
import sys

ROOT = pathlib.Path(__file__).resolve().parent.parent
SRC = pathlib.Path("/root/reference/src/gt4py/cartesian")
OUT = ROOT / "tests" / "golden" / "gt4py_api_surface.json"
FILES = ["backend/__init__.py", "backend/base.py", "backend/module_generator.py", "stencil_builder.py", "stencil_object.py",
         "definitions.py", "gtc/gtir_to_oir.py", "gtc/passes/__init__.py", "gtc/passes/oir_pipeline.py", "utils/attrib.py"]


def params(fn: ast.FunctionDef):
    a = fn.args
    return {"positional": [x.arg for x in a.posonlyargs + a.args], "keyword_only": [x.arg for x in a.kwonlyargs],
            "var_positional": a.vararg.arg if a.vararg else None, "var_keyword": a.kwarg.arg if a.kwarg else None}


def targets(stmt):
    if isinstance(stmt, ast.AnnAssign) and isinstance(stmt.target, ast.Name):
        return [stmt.target.id]
    if isinstance(stmt, ast.Assign):
        return [t.id for t in stmt.targets if isinstance(t, ast.Name)]
    return []


def surface(path: pathlib.Path):
    tree = ast.parse(path.read_text())
    mod = {"functions": {}, "names": [], "imports": [], "classes": {}}
    for node in tree.body:
        if isinstance(node, (ast.FunctionDef, ast.AsyncFunctionDef)):
            mod["functions"][node.name] = params(node)
        elif isinstance(node, ast.ClassDef):
            cls = {"bases": [ast.unparse(b) for b in node.bases], "methods": {}, "properties": [], "attributes": []}
            for stmt in node.body:
                if isinstance(stmt, (ast.FunctionDef, ast.AsyncFunctionDef)):
                    decorators = {ast.unparse(d).split(".")[-1].split("(")[0] for d in stmt.decorator_list}
                    if decorators & {"property", "cached_property", "abstractproperty"}:
                        cls["properties"].append(stmt.name)
                    elif "setter" not in decorators:
                        cls["methods"][stmt.name] = params(stmt)
                else:
                    cls["attributes"] += targets(stmt)
            # attributes set on `self` in __init__ are part of the instance's surface too
            for stmt in node.body:
                if isinstance(stmt, ast.FunctionDef) and stmt.name == "__init__":
                    for n in ast.walk(stmt):
                        if isinstance(n, (ast.Assign, ast.AnnAssign)):
                            for t in (n.targets if isinstance(n, ast.Assign) else [n.target]):
                                if isinstance(t, ast.Attribute) and isinstance(t.value, ast.Name) and t.value.id == "self":
                                    cls["attributes"].append(t.attr)
            cls["attributes"] = sorted(set(cls["attributes"]))
            cls["properties"] = sorted(set(cls["properties"]))
            mod["classes"][node.name] = cls
        elif isinstance(node, ast.ImportFrom):
            mod["imports"] += [a.asname or a.name for a in node.names]
        elif isinstance(node, ast.Import):
            mod["imports"] += [(a.asname or a.name).split(".")[0] for a in node.names]
        else:
            mod["names"] += targets(node)
    mod["names"], mod["imports"] = sorted(set(mod["names"])), sorted(set(mod["imports"]))
    return mod


def build():
    return {"generated_by": "scripts/make_gt4py_api_surface.py",
            "sources": {f: hashlib.sha256((SRC / f).read_bytes()).hexdigest() for f in FILES},
            "modules": {f: surface(SRC / f) for f in FILES}}


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--check", action="store_true")
    args = ap.parse_args()
    text = json.dumps(build(), indent=1) + "\n"
    if args.check:
        if not OUT.exists() or OUT.read_text() != text:
            print(f"{OUT} is out of date with {SRC}: run scripts/make_gt4py_api_surface.py", file=sys.stderr)
            return 1
        print(f"{OUT}: up to date")
        return 0
    OUT.write_text(text)
    print(f"wrote {OUT} ({len(text)} bytes)")
    return 0


if __name__ == "__main__":
    sys.exit(main())
