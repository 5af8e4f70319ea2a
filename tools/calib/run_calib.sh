#!/bin/bash
# GPU box: builds and runs the counter calibration (tools/calib/calib_traffic.hip) under rocprofv3, one --pmc pass per counter -> $1/calib.json
set -e
O=${1:-gpurun_out/calib}; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 $R/tools/calib/calib_traffic.hip -o /tmp/calib_traffic
/tmp/calib_traffic > $R/$O/known_bytes.json
timeout -k 10 120 rocprofv3 --pmc FETCH_SIZE -d $R/$O/fetch -o p -- /tmp/calib_traffic > /dev/null 2> $R/$O/fetch.err
timeout -k 10 120 rocprofv3 --pmc WRITE_SIZE -d $R/$O/write -o p -- /tmp/calib_traffic > /dev/null 2> $R/$O/write.err
timeout -k 10 120 rocprofv3 --kernel-trace --stats -d $R/$O/trace -o p -- /tmp/calib_traffic > /dev/null 2> $R/$O/trace.err
python3 $R/tools/calib/calib_summary.py $R/$O
