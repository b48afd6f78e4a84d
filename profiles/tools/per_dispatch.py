"""Durations of every carve_fused_kernel dispatch of a traced bench.py run (rocprofv3 --kernel-trace CSV), in launch order,
and the mean of the timed ones -- what roofline.avg_launch_ms of the bench line (HIP events) must agree with.
usage: python profiles/tools/per_dispatch.py <trace_kernel_trace.csv> [steps=20] [after=7]
(after = dispatches behind the timed region: six for the clock probe, one pair-count launch)"""
import csv, sys
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
after = int(sys.argv[3]) if len(sys.argv) > 3 else 7
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "carve_fused" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows]
timed = d[len(d) - after - steps:len(d) - after]
print("carve_fused_kernel dispatches of the traced run (rocprofv3 --kernel-trace), ms each, in launch order:")
print(" ".join("%.3f" % x for x in d))
print("%d dispatches: %d warm-up (incl. clock settling), %d timed, %d behind the timed region (clock probe steps, pair count)"
      % (len(d), len(d) - after - steps, steps, after))
print("the %d timed ones: mean %.4f ms, min %.3f, max %.3f" % (steps, sum(timed) / len(timed), min(timed), max(timed)))
