"""per-buffer-set means of rocprofv3 --pmc counters for tools/exp_bimodal4.py (80 rollout launches per set and round)"""
import collections, csv, glob, sys
d = sys.argv[1]
f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
vals = collections.defaultdict(lambda: collections.defaultdict(float))
order = []
for r in csv.DictReader(open(f)):
    if "k_lpb" not in r["Kernel_Name"] or ", 4, 64" not in r["Kernel_Name"]:
        continue
    did = int(r["Dispatch_Id"])
    if did not in vals:
        order.append(did)
    vals[did][r["Counter_Name"]] += float(r["Counter_Value"])
order.sort()
names = sorted({k for v in vals.values() for k in v})
print("%d rollout dispatches; counters: %s" % (len(order), names))
for chunk in range(len(order) // 80):
    ids = order[chunk * 80 + 40:chunk * 80 + 80]
    print("round %d set %d: " % (chunk // 6, chunk % 6) + "  ".join("%s %.4g" % (n, sum(vals[i][n] for i in ids) / len(ids)) for n in names))
