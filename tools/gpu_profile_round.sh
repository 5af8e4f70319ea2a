set -e
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${ROUND_TAG:-r01e}
mkdir -p $O
cd $R
if [ "${PART:-1}" = "1" ]; then
if [ -z "$SKIP_TESTS" ]; then timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -2 $O/pytest.log; fi
cd /tmp
timeout -k 10 420 rocprofv3 --kernel-trace --stats -d $O/trace -o bench -- python3 $R/bench.py --steps 3 --warmup 1 $BENCH_ARGS > $O/bench.json 2> $O/bench.err
tail -1 $O/bench.json
timeout -k 10 240 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch -o p -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline $BENCH_ARGS > $O/pmc_fetch.json 2> $O/pmc_fetch.err
timeout -k 10 240 rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write -o p -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline $BENCH_ARGS > $O/pmc_write.json 2> $O/pmc_write.err
timeout -k 10 240 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU -d $O/pmc_sq -o p -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline $BENCH_ARGS > $O/pmc_sq.json 2> $O/pmc_sq.err
ls -R $O | head -40
fi
if [ "${PART:-2}" = "2" ]; then
# the classifier config, BD-rate against the full search, and the 10-bit 4K configuration (single GPU share of BASELINE config 4)
cd $R
timeout -k 10 300 python bench.py --classifier > $O/bench_classifier.json 2> $O/bench_classifier.err
tail -c 400 $O/bench_classifier.json
[ -n "$SKIP_BDRATE" ] || timeout -k 10 300 python tools/bd_rate.py --frames 8 --out gpurun_out/${ROUND_TAG:-r01e}/bdrate.json > $O/bdrate.log 2>&1
[ -n "$SKIP_BDRATE" ] || tail -2 $O/bdrate.log
timeout -k 10 420 python bench.py --width 3840 --height 2160 --bit-depth 10 --steps 2 --warmup 1 > $O/bench_4k10.json 2> $O/bench_4k10.err
tail -c 600 $O/bench_4k10.json
fi
