import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows:
    n = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    print(n[:100].ljust(100), r["Calls"].rjust(5), ("%.4f" % (float(r["AverageNs"]) / 1e6)).rjust(9), ("%.3f" % (100 * float(r["TotalDurationNs"]) / tot)).rjust(7))
