import csv, glob, sys, collections
d = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(sys.argv[1] + "/*/*counter_collection.csv"):
    for row in csv.DictReader(open(path)):
        if sys.argv[2] in row["Kernel_Name"]:
            d[row["Kernel_Name"][:50]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, cs in d.items():
    print(k)
    for c, v in sorted(cs.items()):
        print("   ", c, round(sum(v) / len(v)), "x", len(v))
