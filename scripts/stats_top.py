"""Top-N rows of a rocprofv3 *_kernel_stats.csv as a plain table (what gets committed under profiles/)."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("# %s: %d dispatches, %.3f ms GPU kernel time" % (sys.argv[1], sum(int(r["Calls"]) for r in rows), tot / 1e6))
print("  calls   total_ms    avg_us    pct  kernel")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:top]:
    print("%7d %10.3f %9.2f %6.2f  %s" % (int(r["Calls"]), float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3,
                                          100 * float(r["TotalDurationNs"]) / tot, r["Name"][:150]))
