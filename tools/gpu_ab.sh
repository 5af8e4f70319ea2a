# A/B of library builds on ONE box (gpurun): TAG=r04b LIBS="libvvcx_base.so libvvcx.so" [STEPS=3] [FRAMES=15] [STAMP_LIBS="libvvcx_stamp_isp.so:VVCX_STAMP_ISP ..."] bash tools/gpu_ab.sh
set -e
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
PKG=$R/reduce-complexity-for-intra-coding-of-vvc_amd
O=$R/gpurun_out/${TAG:-r04b}
mkdir -p $O
cd $R
for sl in $STAMP_LIBS; do
  lib=${sl%%:*}; var=${sl##*:}
  env $var=1 VVCX_LIB=$PKG/$lib VVCX_TOOLS=${TOOLS:-0xfff} timeout -k 10 200 python tools/prof_run.py 1920 1080 > $O/stamps_${lib%.so}.txt 2>&1
  tail -12 $O/stamps_${lib%.so}.txt
done
for rep in 1 ${REPS:-}; do
for lib in $LIBS; do
  timeout -k 10 600 python bench.py --frames ${FRAMES:-15} --steps ${STEPS:-3} --warmup 1 --no-cpu-baseline --lib $PKG/$lib $BENCH_ARGS > $O/bench_${lib%.so}_$rep.json 2> $O/bench_${lib%.so}_$rep.err
  python3 -c "import json,sys; d=json.loads([l for l in open('$O/bench_${lib%.so}_$rep.json') if l.startswith('{')][-1]); print('$lib', '$rep', round(d['value'],2), 'CTU/s', round(d['roofline']['kernel_ms'],1), 'ms')"
done
done
