# full GPU check of the shipped build (gpurun): the -m gpu suite, smoke(), then the driver's bench command
set -e
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${TAG:-r04e}
mkdir -p $O
cd $R
timeout -k 10 ${TEST_TIMEOUT:-900} python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1 || { tail -30 $O/gpu_tests.log; exit 1; }
tail -3 $O/gpu_tests.log
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 || { tail -20 $O/smoke.log; exit 1; }
tail -1 $O/smoke.log
if [ -n "$BENCH" ]; then
  timeout -k 10 900 python bench.py $BENCH > $O/bench.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
  python3 -c "import json; d=json.loads([l for l in open('$O/bench.json') if l.startswith('{')][-1]); print(round(d['value'],2), 'CTU/s', d['config']['workload'][-90:], d['cpu_baseline']['value'] if 'cpu_baseline' in d else '')"
fi
