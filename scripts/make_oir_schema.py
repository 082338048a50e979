#!/usr/bin/env python3
"""Build-container script (reads /root/reference, which does not travel): the SCHEMA of the reference's OIR as data.

    python scripts/make_oir_schema.py            # writes tests/golden/oir_schema.json
    python scripts/make_oir_schema.py --check    # exit 1 when the committed file differs from the reference's sources

`gt4py_amd.adapter.oir_to_ir` reads gt4py's OIR by node CLASS NAME and ATTRIBUTE only; gt4py cannot be imported in this image
(Python 3.10 < 3.12), so the hand-built OIR-shaped trees of tests/test_adapter.py were pinned to nothing
(missing 2 / next 7).  This script parses

    /root/reference/src/gt4py/cartesian/gtc/oir.py      (node classes, :28-360)
    /root/reference/src/gt4py/cartesian/gtc/common.py   (their generic bases, enums, offsets, bounds, masks, :54-890)

with `ast` -- no import of gt4py -- and writes, per node class, the names of its fields INCLUDING the inherited ones (annotated
class attributes, bases resolved across the two files), whether a field has a default, the annotation as source text, and per
enum its members and values.  A fixture is data: class, field and member NAMES and the enums' values -- no source text of the
reference beyond those identifiers.  tests/test_adapter.py builds its trees through the schema (a node of an unknown class, an
attribute the class does not have, a missing required field or a string that is no member of the annotated enum is refused) and
checks that the translator reads nothing else.
"""

from __future__ import annotations

import argparse
import ast
import hashlib
import json
import pathlib
import sys

ROOT = pathlib.Path(__file__).resolve().parent.parent
GTC = pathlib.Path("/root/reference/src/gt4py/cartesian/gtc")
OUT = ROOT / "tests" / "golden" / "oir_schema.json"
ENUM_BASES = {"StrEnum", "IntEnum", "Enum"}


def base_name(node: ast.expr) -> str:
    """`common.FieldAccess[Expr, VariableKOffset]` -> 'common.FieldAccess'; `LocNode` -> 'LocNode'; `eve.Node` -> 'eve.Node'."""
    if isinstance(node, ast.Subscript):
        return base_name(node.value)
    if isinstance(node, ast.Attribute):
        return f"{base_name(node.value)}.{node.attr}"
    if isinstance(node, ast.Name):
        return node.id
    return ast.unparse(node)


def classes_of(path: pathlib.Path):
    tree = ast.parse(path.read_text())
    out = {}
    for node in tree.body:
        if not isinstance(node, ast.ClassDef):
            continue
        bases = [base_name(b) for b in node.bases]
        fields, members = {}, {}
        for stmt in node.body:
            if isinstance(stmt, ast.AnnAssign) and isinstance(stmt.target, ast.Name) and not stmt.target.id.startswith("_"):
                ann = ast.unparse(stmt.annotation)
                if ann.startswith("ClassVar") or ann.startswith("typing.ClassVar"):
                    continue
                fields[stmt.target.id] = {"annotation": ann, "required": stmt.value is None}
            elif isinstance(stmt, ast.Assign) and len(stmt.targets) == 1 and isinstance(stmt.targets[0], ast.Name):
                name = stmt.targets[0].id
                if name.isupper() or (name[:1].isupper() and "_" in name):
                    try:
                        members[name] = ast.literal_eval(stmt.value)
                    except Exception:  # enum.auto() and friends: the member exists, its value is the enum's business
                        members[name] = None
        out[node.name] = {"bases": bases, "fields": fields, "members": members, "line": node.lineno}
    return out


def build():
    common, oir = classes_of(GTC / "common.py"), classes_of(GTC / "oir.py")

    def resolve(module: str, name: str, seen=()):
        """All fields of class `name` of `module` ('oir' | 'common'), inherited ones first (a subclass may re-annotate)."""
        table = oir if module == "oir" else common
        if name not in table or (module, name) in seen:
            return {}
        fields = {}
        for b in reversed(table[name]["bases"]):  # (the MRO: an EARLIER base overrides a later one -- collect the later ones first)
            if b.startswith("common."):
                fields.update(resolve("common", b.split(".", 1)[1], seen + ((module, name),)))
            elif "." not in b:  # a class of the same file -- or one imported by name from common (LocNode, AxisBound, ...)
                where = module if b in table else "common"
                fields.update(resolve(where, b, seen + ((module, name),)))
        fields.update(table[name]["fields"])
        return fields

    def is_enum(table, name, seen=()):
        if name not in table or name in seen:
            return False
        return any(b.split(".")[-1] in ENUM_BASES or is_enum(table, b.split(".")[-1], seen + (name,)) for b in table[name]["bases"])

    classes, enums = {}, {}
    for module, table in (("common", common), ("oir", oir)):  # (oir last: its classes shadow the generic bases of the same name)
        for name, info in table.items():
            if is_enum(table, name):
                enums[name] = info["members"]
                continue
            fields = resolve(module, name)
            if not fields and not any(b.split(".")[-1] in ("Node", "LocNode", "GenericNode", "Expr", "Stmt") for b in info["bases"]):
                continue  # exceptions, visitors, helpers: no data model
            classes[name] = {"module": f"gt4py.cartesian.gtc.{module}", "line": info["line"], "bases": info["bases"],
                             "fields": {k: fields[k] for k in sorted(fields)}}
    return {
        "generated_by": "scripts/make_oir_schema.py",
        "sources": {p.name: {"path": str(p), "sha256": hashlib.sha256(p.read_bytes()).hexdigest()} for p in (GTC / "oir.py", GTC / "common.py")},
        "classes": {k: classes[k] for k in sorted(classes)},
        "enums": {k: enums[k] for k in sorted(enums)},
    }


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--check", action="store_true")
    args = ap.parse_args()
    schema = build()
    text = json.dumps(schema, indent=1, sort_keys=False) + "\n"
    if args.check:
        if not OUT.exists() or OUT.read_text() != text:
            print(f"{OUT} is out of date with {GTC}: run scripts/make_oir_schema.py", file=sys.stderr)
            return 1
        print(f"{OUT}: up to date ({len(schema['classes'])} classes, {len(schema['enums'])} enums)")
        return 0
    OUT.write_text(text)
    print(f"wrote {OUT}: {len(schema['classes'])} classes, {len(schema['enums'])} enums")
    return 0


if __name__ == "__main__":
    sys.exit(main())
