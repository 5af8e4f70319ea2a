set -e
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${TAG:-r04k}; mkdir -p $O; cd $R
run() { name=$1; shift; timeout -k 10 400 python bench.py --no-cpu-baseline "$@" > $O/bench_$name.json 2> $O/bench_$name.err; python3 -c "import json; d=json.loads([l for l in open('$O/bench_$name.json') if l.startswith('{')][-1]); print('$name', round(d['value'],2), 'CTU/s', round(d['roofline']['kernel_ms'],1), 'ms')"; }
run qp22 --qp 22 --frames 19 --steps 2 --warmup 1
run qp27 --qp 27 --frames 19 --steps 2 --warmup 1
run qp37 --qp 37 --frames 19 --steps 2 --warmup 1
run qp32_payload --frames 19 --steps 2 --warmup 1 --emit-payload
run qp32_plain --frames 19 --steps 2 --warmup 1
run classifier --classifier --frames 19 --steps 2 --warmup 1
run 4k10 --width 3840 --height 2160 --bit-depth 10 --frames 5 --steps 2 --warmup 1
run 1080p_tiles_1x1_wpp_114f --tiles 1x1 --wpp --frames 114 --steps 1 --warmup 0
