import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 14]:
    n = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    n = n.split("(")[0][:48]
    print(f"{n:50s} calls {int(r['Calls']):5d}  avg {float(r['AverageNs'])/1e3:10.1f} us  {float(r['Percentage']):6.2f} %")
