import csv, sys, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:70]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    if "rowquant" not in k: continue
    print(k)
    for c, v in sorted(d.items()):
        v = v[len(v)//4:]
        print("   %-28s %14.0f" % (c, sum(v) / len(v)))
