# Round-4 profile of the stream kernel on the GPU box (run through gpurun from the repo root):
#   TAG=r04a [LIB=<alt libvvcx.so>] [STAMPS=1] [PMC=1] [BENCH="--steps 3 --warmup 1"] bash tools/gpu_r04_profile.sh
# stamps: one 1080p frame (135 one-CTU streams) on the diagnostic builds (libvvcx_stamp.so / libvvcx_stamp_dq.so);
# PMC: lane utilisation and issue mix of the default bench workload in separate --pmc passes (no tracing domains beside them).
set -e
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
PKG=$R/reduce-complexity-for-intra-coding-of-vvc_amd
O=$R/gpurun_out/${TAG:-r04a}
mkdir -p $O
cd $R
LIBARG=""
[ -n "$LIB" ] && LIBARG="--lib $R/$LIB"
FR=${FRAMES:-15}
if [ -n "$STAMPS" ]; then
  [ -f $PKG/libvvcx_stamp.so ] && VVCX_LIB=$PKG/libvvcx_stamp.so VVCX_TOOLS=0xfff timeout -k 10 200 python tools/prof_run.py 1920 1080 > $O/stamps_fff.txt 2>&1 && tail -45 $O/stamps_fff.txt
  [ -f $PKG/libvvcx_stamp_dq.so ] && VVCX_STAMP_DQ=1 VVCX_LIB=$PKG/libvvcx_stamp_dq.so VVCX_TOOLS=0xfff timeout -k 10 200 python tools/prof_run.py 1920 1080 > $O/stamps_fff_dq.txt 2>&1 && tail -8 $O/stamps_fff_dq.txt
fi
if [ -n "$BENCH" ]; then
  timeout -k 10 900 python bench.py $BENCH $LIBARG > $O/bench.json 2> $O/bench.err
  tail -c 1500 $O/bench.json
fi
if [ -n "$PMC" ]; then
  cd /tmp
  B="$R/bench.py --frames $FR --steps 2 --warmup 1 --no-cpu-baseline $LIBARG"
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/trace -o bench -- python3 $B > $O/trace.json 2> $O/trace.err
  timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU -d $O/pmc_sq -o p -- python3 $B > $O/pmc_sq.json 2> $O/pmc_sq.err
  timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS -d $O/pmc_mix -o p -- python3 $B > $O/pmc_mix.json 2> $O/pmc_mix.err || echo "pmc_mix pass failed"
  timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_INST_CYCLES_VMEM SQ_WAVES -d $O/pmc_lds -o p -- python3 $B > $O/pmc_lds.json 2> $O/pmc_lds.err || echo "pmc_lds pass failed"
  if [ -n "$PMC_TRAFFIC" ]; then
    timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch -o p -- python3 $B > $O/pmc_fetch.json 2> $O/pmc_fetch.err
    timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write -o p -- python3 $B > $O/pmc_write.json 2> $O/pmc_write.err
  fi
  cd $R
  python3 tools/rocpd_summary.py $O $O/summary > $O/summary.log 2>&1 || tail -5 $O/summary.log
  tail -60 $O/summary.log
fi
