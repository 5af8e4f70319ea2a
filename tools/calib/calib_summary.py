"""Reads the rocprofv3 outputs of tools/calib/run_calib.sh and prints, per access pattern, known bytes vs FETCH_SIZE / WRITE_SIZE (KiB -> bytes)."""
import glob, json, os, sqlite3, sys, csv

def counters(d):
    out = {}
    dbs = glob.glob(os.path.join(d, "**", "*results.db"), recursive=True)
    if dbs:
        c = sqlite3.connect(dbs[0])
        for k, n, v in c.execute("select kernel_name, counter_name, avg(value) from counters_collection group by kernel_name, counter_name"):
            out[(k.split("(")[0], n)] = v
        return out
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            out[(r["Kernel_Name"].split("(")[0], r["Counter_Name"])] = float(r["Counter_Value"])
    return out

def main():
    o = sys.argv[1]
    known = json.load(open(os.path.join(o, "known_bytes.json")))
    cnt = {}
    cnt.update(counters(os.path.join(o, "fetch"))); cnt.update(counters(os.path.join(o, "write")))
    res = {}
    for k in ("cal_w16", "cal_w2", "cal_w2_blk", "cal_w2_hot", "cal_r16", "cal_r2", "cal_r2_hot"):
        f = cnt.get((k, "FETCH_SIZE")); w = cnt.get((k, "WRITE_SIZE"))
        res[k] = {"known_bytes": known[k], "FETCH_SIZE_bytes": None if f is None else f * 1024, "WRITE_SIZE_bytes": None if w is None else w * 1024}
        main_c = res[k]["WRITE_SIZE_bytes"] if "_w" in k else res[k]["FETCH_SIZE_bytes"]
        res[k]["counter_over_known"] = None if main_c is None else main_c / known[k]
    res["hot_region_bytes"] = known["hot_region_bytes"]
    json.dump(res, open(os.path.join(o, "calib.json"), "w"), indent=1)
    for k, v in res.items():
        print(k, v)

main()
