import csv, sys, collections
f=sys.argv[1]
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for r in csv.DictReader(open(f)):
    k=r["Kernel_Name"][:60]
    agg[k][r["Counter_Name"]]+=float(r["Counter_Value"])
for k,v in agg.items():
    if "dft" in k or "lean" in k:
        print(k, {a: "%.4g"%b for a,b in v.items()})
