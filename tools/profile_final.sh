#!/bin/bash
# Run ON THE GPU BOX (through gpurun): everything the round's bench line quotes, for <tag>:
#   gpurun_out/prof_<tag>/            tools/profile.sh passes of the 8-spp run (kernel trace + PMC)
#   gpurun_out/prof_<tag>_default/    kernel-trace stats and FETCH_SIZE / WRITE_SIZE passes of the DEFAULT bench command
#   gpurun_out/bench_<tag>.json       the default bench line (with the CPU baseline)
set -u
TAG=${1:-r1}
tools/profile.sh $TAG || exit 1
OUT=$PWD/gpurun_out/prof_${TAG}_default; mkdir -p $OUT; export TMPDIR=/tmp
run() { local name=$1; shift
  timeout -s KILL 300 rocprofv3 "$@" --output-format csv -d $OUT/$name -o $name -- python3 bench.py --no-cpu-baseline > $OUT/$name.log 2>&1
  local rc=$?; echo "default/$name rc=$rc"; return $rc; }
run trace --kernel-trace --stats || exit 1
run fetch --pmc FETCH_SIZE || exit 1
run write --pmc WRITE_SIZE || exit 1
timeout -k 10 400 python -u bench.py > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err || exit 1
tail -c 600 gpurun_out/bench_$TAG.json
