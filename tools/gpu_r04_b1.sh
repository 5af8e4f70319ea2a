set -e
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04i; mkdir -p $O; cd $R
PKG=$R/reduce-complexity-for-intra-coding-of-vvc_amd
VVCX_STAMP_PASS=1 VVCX_LIB=$PKG/libvvcx_stamp_pass.so VVCX_TOOLS=0xfff timeout -k 10 200 python tools/prof_run.py 1920 1080 > $O/stamps_pass.txt 2>&1; tail -8 $O/stamps_pass.txt
TAG=r04i LIBS="libvvcx.so libvvcx_isp.so" FRAMES=38 STEPS=2 bash tools/gpu_ab.sh
timeout -k 10 560 python tools/layout_table.py --layouts 1x1+wpp,4x2,15x9 --out gpurun_out/r04i/layout_rest.json 2>&1 | tee $O/layout_rest.log | tail -20
timeout -k 10 300 python tools/bd_rate.py --frames 8 --out gpurun_out/r04i/bdrate.json > $O/bdrate.log 2>&1; tail -6 $O/bdrate.log
