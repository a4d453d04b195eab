import csv, glob, os, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)


def short(name):
    if "tile_minimizer_kernel<31, true>" in name: return "tile_min<hpc>"
    if "tile_minimizer_kernel<31, false>" in name: return "tile_min<reg>"
    if "kminmer_kernel" in name: return "kminmer"
    return None


for f in glob.glob(os.path.join(out, "p*", "*", "*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        s = short(r["Kernel_Name"])
        if s: agg[s][r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob(os.path.join(out, "p*", "*", "*kernel_trace.csv")):
    for r in csv.DictReader(open(f)):
        s = short(r["Kernel_Name"])
        if s: dur[s].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
for k in sorted(agg):
    print("==", k, "avg_ms(profiled)", round(sum(dur[k]) / max(len(dur[k]), 1), 3), "launches", len(dur[k]))
    for c in sorted(agg[k]):
        v = agg[k][c]
        print("   %-24s %.5g" % (c, sum(v) / len(v)))
