#!/usr/bin/env python3
"""Turns the rocprofv3 CSVs written by profile_gpu.sh into one JSON per workload:
per kernel (short name), per-launch AVERAGES of every counter over the dispatches of the run, the
kernel-trace average duration, and for the dominant kernel the HBM bytes per launch
(2 x FETCH_SIZE + WRITE_SIZE, both reported in KiB; the factor 2 is MI355X_MICROARCH.md's gfx950
correction, confirmed for the access widths used here in profiles/r01/fetch_calibration.txt).

  summarize_pmc.py <dir with *_counter_collection.csv> <out.json> [--key K --counters profiles/counters.json]
  summarize_pmc.py <dir> <out.json> --mc-key mc_1024 --counters profiles/counters.json
      (marching cubes: HBM bytes of ALL its kernels per extraction -> counters[mc_1024].hbm_bytes_per_call)
"""
import csv
import glob
import json
import os
import re
import sys


def short(name):
    n = name.replace("(anonymous namespace)::", "")
    n = re.sub(r"^void\s+", "", n)
    n = re.split(r"[<(]", n)[0]
    n = n.split("::")[-1]
    return n.replace("_kernel", "")


def library_build():
    """vcy_version() of the library the counters were collected on (VCY_HIP_LIB or the in-tree build)."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    from vacancy_amd import capi
    return capi.load().vcy_version().decode()


def main():
    d, out = sys.argv[1], sys.argv[2]
    per = {}
    full = {}
    for path in sorted(glob.glob(os.path.join(d, "*_counter_collection.csv"))):
        acc = {}
        dur = {}
        with open(path) as f:
            for row in csv.DictReader(f):
                k = short(row["Kernel_Name"])
                full.setdefault(k, row["Kernel_Name"])
                key = (k, row["Counter_Name"])
                disp = row["Dispatch_Id"]
                acc.setdefault(key, {}).setdefault(disp, 0.0)
                acc[key][disp] += float(row["Counter_Value"])
                if row["Counter_Name"] == "GRBM_GUI_ACTIVE" and row.get("End_Timestamp"):
                    dur.setdefault(k, {})[disp] = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-9
        for (k, cname), disp in acc.items():
            vals = list(disp.values())
            per.setdefault(k, {})[cname] = sum(vals) / len(vals)
            per[k]["dispatches_" + cname] = len(vals)
            if cname == "GRBM_GUI_ACTIVE" and k in dur:
                # shader clock of the launch: busy cycles (summed over the 8 XCDs) / 8 / the SAME dispatch's duration --
                # the median over the dispatches (the kernel-trace run's durations are another run's: a counter pass is slower)
                clocks = sorted(disp[d] / 8.0 / dur[k][d] for d in disp if dur[k].get(d, 0) > 0)
                if clocks:
                    per[k]["shader_clock_hz"] = clocks[len(clocks) // 2]
    stats = glob.glob(os.path.join(d, "*_kernel_stats.csv"))
    if stats:
        with open(stats[0]) as f:
            for row in csv.DictReader(f):
                k = short(row["Name"])
                per.setdefault(k, {})["trace_avg_ns"] = float(row["AverageNs"])
                per[k]["trace_calls"] = int(row["Calls"])
                per[k]["trace_min_ns"] = float(row["MinNs"])
    for k, v in per.items():
        v["kernel"] = full.get(k, k)
        if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
            v["hbm_bytes_per_launch"] = int(2 * v["FETCH_SIZE"] * 1024 + v["WRITE_SIZE"] * 1024)
    json.dump(per, open(out, "w"), indent=1, sort_keys=True)
    print("wrote", out, "kernels:", sorted(per))
    if "--key" in sys.argv:
        key = sys.argv[sys.argv.index("--key") + 1]
        cpath = sys.argv[sys.argv.index("--counters") + 1]
        kernel = sys.argv[sys.argv.index("--kernel") + 1] if "--kernel" in sys.argv else "carve_fused"
        try:
            allc = json.load(open(cpath))
        except Exception:
            allc = {}
        entry = {k: v for k, v in per[kernel].items() if not k.startswith("dispatches_")}
        entry["source"] = os.path.relpath(out, os.path.dirname(os.path.dirname(os.path.abspath(cpath))))
        entry["build"] = library_build()  # bench.py only uses counters of the build it runs
        pre = per.get("footprint_records", {})
        if kernel == "carve_fused" and "hbm_bytes_per_launch" in pre:  # what runs in front of the carve kernel
            entry["prepass_hbm_bytes_per_launch"] = pre["hbm_bytes_per_launch"]
            entry["prepass_trace_avg_ns"] = pre.get("trace_avg_ns")
        allc[key] = entry
        json.dump(allc, open(cpath, "w"), indent=1, sort_keys=True)
        print("updated", cpath, key)
    if "--mc-key" in sys.argv:
        key = sys.argv[sys.argv.index("--mc-key") + 1]
        cpath = sys.argv[sys.argv.index("--counters") + 1]
        entry = mc_entry(per)
        if entry:
            try:
                allc = json.load(open(cpath))
            except Exception:
                allc = {}
            entry["source"] = os.path.relpath(out, os.path.dirname(os.path.dirname(os.path.abspath(cpath))))
            entry["build"] = library_build()
            allc[key] = entry
            json.dump(allc, open(cpath, "w"), indent=1, sort_keys=True)
            print("updated", cpath, key, entry["hbm_bytes_per_call"])


def mc_entry(per):
    """Bytes one extraction moves: every marching-cubes kernel x its launches per extraction (= launches of the cell
    search: mc_sweep, or mc_active on the bit-plane path).  The run holds extractions of both kinds -- bricks outside
    the surface skipped (mc_bits_bricks, the default) and every brick read (mc_bits, "mcskip" 0): the pass over the
    state is counted once per extraction with the per-launch bytes of the kernel of that kind."""
    names = [k for k in per if re.match(r"mc_|scan_chunks|scan_chained|add_chunk_offsets", k) and "hbm_bytes_per_launch" in per[k]]
    calls = per.get("mc_sweep", per.get("mc_active", {})).get("dispatches_FETCH_SIZE", 0)
    if not calls:
        return None
    state_pass = {"mc_bits", "mc_bits_bricks"}
    rest = sum(per[k]["hbm_bytes_per_launch"] * per[k]["dispatches_FETCH_SIZE"] for k in names if k not in state_pass) / calls
    entry = {"extractions_in_run": calls,
             "kernels": {k: {"launches_per_call": per[k]["dispatches_FETCH_SIZE"] / calls,
                             "hbm_bytes_per_launch": per[k]["hbm_bytes_per_launch"]} for k in sorted(names)}}
    if "mc_bits_bricks" in per and "hbm_bytes_per_launch" in per["mc_bits_bricks"]:
        entry["hbm_bytes_per_call"] = int(rest + per["mc_bits_bricks"]["hbm_bytes_per_launch"])
        if "mc_bits" in per and "hbm_bytes_per_launch" in per["mc_bits"]:
            entry["hbm_bytes_per_call_every_brick_read"] = int(rest + per["mc_bits"]["hbm_bytes_per_launch"])
    elif "mc_bits" in per and "hbm_bytes_per_launch" in per["mc_bits"]:
        entry["hbm_bytes_per_call"] = int(rest + per["mc_bits"]["hbm_bytes_per_launch"])
    else:
        entry["hbm_bytes_per_call"] = int(rest)
    return entry


if __name__ == "__main__":
    main()
