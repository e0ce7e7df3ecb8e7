"""Developer tool: per-kernel averages of a rocprofv3 --stats run (kernel_stats.csv), names shortened."""
import csv
import sys

for r in csv.DictReader(open(sys.argv[1])):
    name = r["Name"].split("(")[0].replace("void ndlqr::", "").replace("ndlqr::", "")
    print("%-44s calls %5s  avg %10.1f us  total %9.3f ms  %5s %%" % (name[:44], r["Calls"], float(r["AverageNs"]) / 1e3,
                                                                     float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
