# dump the classifier's training rows at the four QPs (device search, tools 0xfff) into gpurun_out/<TAG>/rows_qp<QP>.npz
set -e
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${TAG:-r04f}; mkdir -p $O; cd $R
for qp in 22 27 32 37; do
  timeout -k 10 280 python tools/train_partition_forest.py --qp $qp --device --pictures 12 --max-rows 350000 --dump-rows $O/rows_qp$qp.npz 2>&1 | tail -1
done
ls -la $O
