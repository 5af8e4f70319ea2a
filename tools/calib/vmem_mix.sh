#!/bin/bash
# GPU box: memory-instruction mix of the search kernel on one 1080p frame (135 CTU streams) next to its WRITE_SIZE / FETCH_SIZE -> $1/vmem_mix.txt
O=${1:-gpurun_out/vmem}; R=${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p $R/$O; cd /tmp; export TMPDIR=/tmp
export VVCX_TOOLS=0xb5b
timeout -k 10 120 rocprofv3 --pmc SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_FLAT SQ_INSTS_SMEM SQ_INSTS_SALU SQ_INSTS_VALU -d $R/$O/mix -o p -- python3 $R/tools/prof_run.py 1920 1080 > $R/$O/mix.log 2>&1
timeout -k 10 120 rocprofv3 --pmc WRITE_SIZE -d $R/$O/write -o p -- python3 $R/tools/prof_run.py 1920 1080 > $R/$O/write.log 2>&1
timeout -k 10 120 rocprofv3 --pmc FETCH_SIZE -d $R/$O/fetch -o p -- python3 $R/tools/prof_run.py 1920 1080 > $R/$O/fetch.log 2>&1
python3 - <<PY > $R/$O/vmem_mix.txt
import sqlite3, glob
for sub in ("mix", "write", "fetch"):
    for db in glob.glob("$R/$O/" + sub + "/**/*results.db", recursive=True):
        c = sqlite3.connect(db)
        for k, n, v, cnt in c.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection where kernel_name like 'vvcx_compress%' group by kernel_name, counter_name"):
            print(sub, k.split("(")[0], n, v, cnt)
PY
cat $R/$O/vmem_mix.txt
