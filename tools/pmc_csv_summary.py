#!/usr/bin/env python3
"""Mean counter values per (kernel, grid, launch-class) from rocprofv3 --pmc ... --output-format csv runs.
usage: python tools/pmc_csv_summary.py <dir-with-*_counter_collection.csv>... [--period P] [--match substr]
--period P: the traced program launches P different problems round-robin with the same kernel+grid
(tools/ubench/clock_probe pmc: P=2 -> K=4096 / K=16384); dispatches of one kernel are then classed by order mod P."""
import collections, csv, glob, json, os, re, sys

args = [a for a in sys.argv[1:] if not a.startswith("--")]
period = int(sys.argv[sys.argv.index("--period") + 1]) if "--period" in sys.argv else 1
match = sys.argv[sys.argv.index("--match") + 1] if "--match" in sys.argv else ""
if "--period" in sys.argv:
    args.remove(str(period))
if "--match" in sys.argv:
    args.remove(match)
out = collections.OrderedDict()
for d in args:
    for f in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
        per_kernel_order = collections.defaultdict(dict)   # (kernel, grid) -> dispatch id -> order
        rows = list(csv.DictReader(open(f)))
        for r in rows:
            key = (r["Kernel_Name"], int(r["Grid_Size"]))
            did = int(r["Dispatch_Id"])
            if did not in per_kernel_order[key]:
                per_kernel_order[key][did] = len(per_kernel_order[key])
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        dur = collections.defaultdict(list)
        for r in rows:
            if match and match not in r["Kernel_Name"]:
                continue
            key = (r["Kernel_Name"], int(r["Grid_Size"]))
            cls = per_kernel_order[key][int(r["Dispatch_Id"])] % period
            acc[key + (cls,)][r["Counter_Name"]].append(float(r["Counter_Value"]))
            dur[key + (cls,)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        for (name, grid, cls), d2 in acc.items():
            short = re.sub(r"\(.*", "", name).replace("void ", "")
            short = re.sub(r"asq::", "", short)[:90]
            k = f"{short} grid={grid} class={cls}"
            e = out.setdefault(k, {"dispatch_us_under_pmc": round(sum(dur[(name, grid, cls)]) / len(dur[(name, grid, cls)]), 2)})
            for c, v in sorted(d2.items()):
                e[c] = round(sum(v) / len(v), 1)
                e["n"] = len(v)
print(json.dumps(out, indent=1))
