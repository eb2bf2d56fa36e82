"""Measured-error baseline of the GPU parity suite.

Every comparison of the GPU tests writes one line `<tag>: field=err field=err ...` to gpurun_out/parity_report.txt
(tests/helpers.py, report()).  This script turns the report of a run on an MI355X box into tests/golden/gpu_parity_baseline.json
(per tag and field: the largest error seen); helpers.report() then fails any later run whose error on a field exceeds
max(3 x that figure, FLOOR) -- a regression guard under EVERY comparison, beside the tolerances the tests state.

    python tests/golden/make_parity_baseline.py [gpurun_out/parity_report.txt] [--merge]

Regenerate it (and look at the diff) whenever a kernel change moves the numbers on purpose.  --merge keeps the larger of the
committed figure and the new one per field (a few comparisons move between boxes: thread-interleaved many-rank runs, termination
states one sweep apart), so that the guard follows the worst green run seen, not the latest.
"""
import json
import os
import re
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
LINE = re.compile(r"^(?P<tag>.+?): (?P<body>(?:[A-Za-z_0-9]+=[0-9.]+e[+-][0-9]+ ?)+)(?:\s+\[.*\])?\s*$")


def parse(line):
    m = LINE.match(line.strip())
    if not m:
        return None
    errs = {}
    for kv in m.group("body").split():
        k, v = kv.split("=")
        errs[k] = float(v)
    return m.group("tag"), errs


def main():
    args = [a for a in sys.argv[1:] if a != "--merge"]
    src = args[0] if args else os.path.join(os.path.dirname(os.path.dirname(HERE)), "gpurun_out", "parity_report.txt")
    out = os.path.join(HERE, "gpu_parity_baseline.json")
    base = {}
    if "--merge" in sys.argv[1:] and os.path.exists(out):
        base = json.load(open(out))
    for ln in open(src):
        p = parse(ln)
        if p is None:
            continue
        tag, errs = p
        slot = base.setdefault(tag, {})
        for k, v in errs.items():
            slot[k] = max(slot.get(k, 0.0), v)
    with open(out, "w") as f:
        json.dump(base, f, indent=0, sort_keys=True)
    print(f"{len(base)} tags -> {out}")


if __name__ == "__main__":
    main()
