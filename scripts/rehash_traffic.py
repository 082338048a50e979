#!/usr/bin/env python3
"""Build container (has .git): stamp a profile summary made on the GPU box with the kernel-source hashes OF THE COMMIT IT RAN ON.

    python scripts/rehash_traffic.py gpurun_out/prof_all_r5 <sha> profiles/r5_all_kernels_summary.json profiles/hbm_traffic.json

The GPU box has no .git and computes `bench.kernel_source_hash` on the snapshot it was sent; this script recomputes the hash from
`git show <sha>:<file>` (so it is the hash of exactly the committed sources the kernels were built from, by the hash function of
the tree that will READ it) and writes the two files under profiles/.  bench.py reports a traffic figure only while the hash of the
sources in its tree equals the stored one."""
import json
import pathlib
import subprocess
import sys

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402


def main() -> int:
    out, sha, summary_to, traffic_to = pathlib.Path(sys.argv[1]), sys.argv[2], pathlib.Path(sys.argv[3]), pathlib.Path(sys.argv[4])

    def read(rel):
        return subprocess.run(["git", "-C", str(ROOT), "show", f"{sha}:{rel}"], check=True, capture_output=True).stdout

    summary = json.loads((out / "all_kernels_summary.json").read_text())
    traffic = json.loads((out / "hbm_traffic.json").read_text())
    for w in bench.KERNEL_SOURCES:
        h = bench.kernel_source_hash(w, read)
        if w in summary["kernels"]:
            summary["kernels"][w]["kernel_source_sha"], summary["kernels"][w]["git_sha"] = h, sha
        if w in traffic:
            traffic[w]["kernel_source_sha"], traffic[w]["git_sha"] = h, sha
            traffic[w]["source"] = str(summary_to)
        now = bench.kernel_source_hash(w)
        print(f"{w:52s} {h} at {sha}; tree now {now}{'' if now == h else '   <-- the sources have changed since: bench.py will report traffic null'}")
    summary["git_sha"] = sha
    summary_to.write_text(json.dumps(summary, indent=1) + "\n")
    traffic_to.write_text(json.dumps(traffic, indent=1) + "\n")
    return 0


if __name__ == "__main__":
    sys.exit(main())
