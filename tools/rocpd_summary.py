#!/usr/bin/env python3
"""Summarise rocprofv3 (rocpd sqlite) outputs of tools/gpu_profile_round.sh into small text/JSON files for profiles/.

usage: rocpd_summary.py <gpurun_out/rXX dir> <profiles prefix>
Writes <prefix>_kernel_stats.csv (the --kernel-trace --stats table), <prefix>_pmc.json (per-launch counter values and
the HBM traffic derived as MI355X_MICROARCH.md §HBM prescribes: FETCH_SIZE and WRITE_SIZE from separate passes, in
KiB, FETCH_SIZE doubled on gfx950) and copies the bench JSON line.
"""
import json
import os
import shutil
import sqlite3
import sys


def main():
    src, prefix = sys.argv[1], sys.argv[2]
    c = sqlite3.connect(os.path.join(src, "trace", "bench_results.db"))
    with open(prefix + "_kernel_stats.csv", "w") as f:
        f.write('"Name","Calls","TotalDurationNs","AverageNs","Percentage"\n')
        for r in c.execute("select name,total_calls,total_duration*1000,average*1000,percentage from top_kernels"):
            f.write('"%s",%d,%d,%d,%.6f\n' % (r[0], r[1], r[2], r[3], r[4]))
    with open(prefix + "_kernel_trace.csv", "w") as f:
        f.write('"Kernel_Name","Start","End","DurationNs","Grid","Workgroup","LDS","Scratch","VGPR","AccumVGPR","SGPR"\n')
        for r in c.execute("select name,start,end,duration,grid_x,workgroup_x,lds_size,scratch_size,vgpr_count,accum_vgpr_count,sgpr_count "
                           "from kernels where name like 'vvcx%' order by start"):
            f.write('"%s",%d,%d,%d,%d,%d,%d,%d,%d,%d,%d\n' % r)
    pmc = {}
    pmc_db = {}
    for sub in sorted(d_ for d_ in os.listdir(src) if d_.startswith("pmc_")):
        p = os.path.join(src, sub, "p_results.db")
        if not os.path.exists(p):
            continue
        d = sqlite3.connect(p)
        for name, val in d.execute("select counter_name, avg(value) from counters_collection where kernel_name like 'vvcx_compress%' group by counter_name"):
            pmc[name] = val
        for name, val in d.execute("select counter_name, avg(value) from counters_collection where kernel_name like 'vvcx_deblock%' group by counter_name"):
            pmc_db[name] = val
    out = {"kernel": "vvcx_compress_kernel_u8", "per_launch_avg": pmc}
    if pmc_db:
        out["deblock_kernel_per_launch_avg"] = pmc_db
        if "FETCH_SIZE" in pmc_db and "WRITE_SIZE" in pmc_db:
            out["deblock_hbm_traffic_bytes_per_launch"] = int(pmc_db["FETCH_SIZE"] * 1024 * 2 + pmc_db["WRITE_SIZE"] * 1024)
    if "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc:
        out["hbm_traffic_bytes_per_launch"] = int(pmc["FETCH_SIZE"] * 1024 * 2 + pmc["WRITE_SIZE"] * 1024)
        out["read_bytes_per_launch"] = int(pmc["FETCH_SIZE"] * 1024 * 2)
        out["write_bytes_per_launch"] = int(pmc["WRITE_SIZE"] * 1024)
        out["note"] = ("FETCH_SIZE/WRITE_SIZE in KiB from separate --pmc passes; FETCH_SIZE doubled (gfx950 counts 128-B requests as 64 B). Calibrated on this "
                       "kernel's access shapes (tools/calib, profiles/r02n_counter_calibration.json): 2 B/lane rows and 32-B partial blocks count exactly; "
                       "WRITE_SIZE counts every global store including rewrites of lines that stay cached (a write-through count: stores issued, scratch "
                       "saves included, not lines evicted); FETCH_SIZE counts L2 misses only (re-reads of a cached region do not appear)")
    for r in c.execute("select average*1000 from top_kernels where name like 'vvcx_compress%'"):
        out["kernel_ms"] = r[0] / 1e6                       # average launch duration of the --kernel-trace run (ms)
    # lane utilisation of the vector ALU: thread-cycles per cycle a wave spends in VALU instructions = active lanes per VALU instruction (64 = every lane works)
    if pmc.get("SQ_ACTIVE_INST_VALU") and pmc.get("SQ_THREAD_CYCLES_VALU") is not None:
        out["valu_active_lanes"] = pmc["SQ_THREAD_CYCLES_VALU"] / pmc["SQ_ACTIVE_INST_VALU"]
        out["valu_lane_utilisation"] = out["valu_active_lanes"] / 64.0
    if pmc.get("SQ_WAVE_CYCLES"):
        for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_LDS"):
            if pmc.get(k) is not None:
                out[k.lower()[3:] + "_per_wave_cycle"] = pmc[k] / pmc["SQ_WAVE_CYCLES"]
    bj = os.path.join(src, "bench.json")
    if not os.path.exists(bj):
        bj = os.path.join(src, "trace.json")
    if os.path.exists(bj):
        line = [l for l in open(bj) if l.startswith("{")][-1]
        out["workload"] = json.loads(line)["config"]["workload"]
        shutil.copy(bj, prefix + "_bench.json")
    json.dump(out, open(prefix + "_pmc.json", "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
