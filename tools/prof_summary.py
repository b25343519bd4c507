"""Per-step kernel table from a rocprofv3 --kernel-trace --stats CSV directory (bench_kernel_stats.csv)."""
import csv, sys
d, steps = sys.argv[1], float(sys.argv[2])
rows = list(csv.DictReader(open(d + "/bench_kernel_stats.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("sum of kernel durations: %.1f us/step over %g steps; launches/step %.1f" % (tot / 1e3 / steps, steps, sum(int(r["Calls"]) for r in rows) / steps))
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 30]:
    print("%-84s %6.2f/step %8.1f us/step avg %7.2f" % (r["Name"][:84], int(r["Calls"]) / steps, float(r["TotalDurationNs"]) / 1e3 / steps, float(r["AverageNs"]) / 1e3))
