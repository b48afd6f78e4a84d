#!/usr/bin/env python3
"""Issue floor of the fused carve kernel from an A/B run of the floor builds (profiles/tools/ab_variants.sh
<out> devprod floorT floorS floorTS ...): rate of the kernel as shipped / rate of the SAME instruction stream with
its tile loads and its stores compiled out (floorTS), for the two workloads whose control flow does not depend on
the loaded data (view dropping off, TSDF); for the default workload only the stores can be taken out (floorS).
Writes counters.json["issue_floor"], stamped with the library build like every other entry.

  summarize_floor.py <issue_floor.txt> <counters.json>"""
import json
import re
import sys

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))
from summarize_pmc import library_build  # noqa: E402


def main():
    rates = {}
    for line in open(sys.argv[1]):
        m = re.match(r"(\w+)\s+default\s+(\d+)\s+\(.*?\)\s+cull0\s+(\d+)\s+tsdf\s+(\d+)", line)
        if m:
            rates.setdefault(m.group(1), []).append(tuple(float(x) for x in m.groups()[1:]))
    avg = {k: [sum(c) / len(c) for c in zip(*v)] for k, v in rates.items()}
    prod = avg.get("devprod") or avg.get("prod")
    ts, st = avg["floorTS"], avg["floorS"]
    entry = {
        "cull0": round(prod[1] / ts[1], 4), "tsdf": round(prod[2] / ts[2], 4),
        "default_without_stores_only": round(prod[0] / st[0], 4),
        "rates_mvoxel_views_per_s": {k: [round(x) for x in v] for k, v in avg.items()},
        "meaning": "kernel as shipped / same instruction stream without tile loads and stores (1.0 = at its issue floor)",
        "source": __import__("os").path.relpath(__import__("os").path.abspath(sys.argv[1]), __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))), "build": library_build(),
    }
    try:
        allc = json.load(open(sys.argv[2]))
    except Exception:
        allc = {}
    allc["issue_floor"] = entry
    json.dump(allc, open(sys.argv[2], "w"), indent=1, sort_keys=True)
    print("issue_floor", entry)


if __name__ == "__main__":
    main()
