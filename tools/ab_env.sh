# usage: ab_env.sh VAR v1 v2 ...   -- bench.py --no-cpu once per value of VAR
var=$1; shift
for v in "$@"; do env $var=$v python bench.py --no-cpu 2>/dev/null > gpurun_out/bench_${var}_$v.json; done
