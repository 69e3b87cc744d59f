"""Similarity of every file under kaolin-wisp_amd/wisp (and csrc) to the same-named file of the reference tree
(difflib ratio over stripped non-empty lines).  Run in the build container only (needs /root/reference)."""
import difflib, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"


def lines(path):
    with open(path, errors="ignore") as f:
        return [l.strip() for l in f if l.strip()]


ref_by_name = {}
for d, _, fs in os.walk(REF):
    for f in fs:
        if f.endswith((".py", ".cu", ".cpp", ".h", ".cuh")):
            ref_by_name.setdefault(os.path.splitext(f)[0], []).append(os.path.join(d, f))

rows = []
for d, _, fs in os.walk(os.path.join(ROOT, "kaolin-wisp_amd")):
    for f in fs:
        if not f.endswith((".py", ".hip", ".h", ".cpp")) or f == "__init__.py":
            continue
        mine = os.path.join(d, f)
        for ref in ref_by_name.get(os.path.splitext(f)[0], []):
            r = difflib.SequenceMatcher(None, lines(mine), lines(ref), autojunk=False).ratio()
            rows.append((r, os.path.relpath(mine, ROOT), os.path.relpath(ref, REF)))
rows.sort(reverse=True)
for r, a, b in rows[: int(sys.argv[1]) if len(sys.argv) > 1 else 15]:
    print(f"{r:.2f}  {a}  <->  {b}")
