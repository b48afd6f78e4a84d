"""Prints the last N dispatches of a rocprofv3 kernel trace CSV: kernel, duration, gap to the previous one, grid."""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
prev = None
for r in rows[-int(sys.argv[2]) if len(sys.argv) > 2 else -14:]:
    n = re.sub(r"\(anonymous namespace\)::|vcy::|void ", "", r["Kernel_Name"])[:34]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%-36s dur %8.1f us gap %8.1f us grid %s" % (n, (e - s) / 1e3, (s - prev) / 1e3 if prev else 0, r["Grid_Size_X"]))
    prev = e
