import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in sys.argv[1:]:
    for row in csv.DictReader(open(path)):
        k = row["Kernel_Name"].split("(")[0][:60]
        acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, c in acc.items():
    if "gcl" not in k and "segreduce" not in k and "k_gemm" not in k: continue
    print(k, {n: round(sum(v)/len(v)) for n, v in c.items()})
