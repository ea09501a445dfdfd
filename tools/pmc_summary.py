"""Dev tool: condense rocprofv3 --pmc output directories into one small JSON (the files kept under profiles/).

usage: pmc_summary.py OUT.json KERNEL_SUBSTRING DIR [DIR ...]
Every DIR is the -d directory of one `rocprofv3 --pmc ... --kernel-trace` pass (separate passes per counter group, as
the MI355X guide prescribes).  Counters are averaged per dispatch of the kernels whose name contains the substring;
FETCH_SIZE gets the gfx950 x2 correction (it tallies 128-B requests at 64 B), WRITE_SIZE is taken as reported.
"""
import csv
import glob
import json
import os
import sys


def main():
    out, needle, dirs = sys.argv[1], sys.argv[2], sys.argv[3:]
    csv.field_size_limit(1 << 30)
    sums, disp, dur = {}, {}, []
    by_grid = {}           # (grid size, kernel template arguments) -> counter -> [sum, dispatches]
    for d in dirs:
        for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            seen = set()
            with open(path, newline="") as f:
                for row in csv.DictReader(f):
                    if needle not in row["Kernel_Name"]:
                        continue
                    c = row["Counter_Name"]
                    sums[c] = sums.get(c, 0.0) + float(row["Counter_Value"])
                    name = row["Kernel_Name"]
                    gkey = f'{row.get("Grid_Size", "?")} {name[name.find("<"):name.find(">") + 1]}'
                    e = by_grid.setdefault(gkey, {}).setdefault(c, [0.0, 0])
                    e[0] += float(row["Counter_Value"]); e[1] += 1
                    disp.setdefault(c, set()).add((path, row["Dispatch_Id"]))
                    if (path, row["Dispatch_Id"]) not in seen:
                        seen.add((path, row["Dispatch_Id"]))
                        dur.append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
    if not sums:
        sys.exit(f"no dispatch of a kernel containing {needle!r} under {dirs}")
    res = {"kernel_contains": needle, "dispatches_per_pass": {c: len(v) for c, v in disp.items()},
           "avg_dur_us_under_pmc": round(sum(dur) / len(dur), 3),
           "per_dispatch": {c: sums[c] / len(disp[c]) for c in sorted(sums)}}
    pd = res["per_dispatch"]
    if "FETCH_SIZE" in pd or "WRITE_SIZE" in pd:
        rd = pd.get("FETCH_SIZE", 0.0) * 1024 * 2          # KB -> B, gfx950 half-count correction
        wr = pd.get("WRITE_SIZE", 0.0) * 1024
        res["hbm_bytes_corrected_per_dispatch"] = {"read": rd, "write": wr, "total": rd + wr}
    # per launch shape (grid size = tiles * 256 threads [* z]): where the traffic above the algorithmic bytes comes from
    res["per_grid_kb"] = {g: {c: round(v[0] / v[1] * (2 if c == "FETCH_SIZE" else 1), 1) for c, v in cs.items()} | {"n": max(v[1] for v in cs.values())}
                          for g, cs in sorted(by_grid.items(), key=lambda kv: -sum(v[0] for v in kv[1].values()))}
    with open(out, "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps(res)[:1500])


if __name__ == "__main__":
    main()
