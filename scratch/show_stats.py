"""print a rocprofv3 kernel_stats.csv as (name, calls per step, avg us, us per step); argv: file [calls-per-step divisor]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
div = float(sys.argv[2]) if len(sys.argv) > 2 else 37.0
tot = 0.0
for r in rows:
    per = float(r['TotalDurationNs']) / 1e3 / div
    tot += per
    print(f"{r['Name'][:118]:118s} n/step={int(r['Calls'])/div:5.1f} avg={float(r['AverageNs'])/1e3:7.1f} us/step={per:7.1f}")
print("total us/step", round(tot, 1))
