set -e
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04j; mkdir -p $O; cd $R
TAG=r04j LIBS="libvvcx.so libvvcx_nw3.so" FRAMES=57 STEPS=2 bash tools/gpu_ab.sh
timeout -k 10 500 python tools/layout_table.py --qps 22 --layouts 4x2,15x9 --out gpurun_out/r04j/layout_qp22.json 2>&1 | tee $O/layout_qp22.log | tail -6
