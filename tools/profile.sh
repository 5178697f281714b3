#!/bin/bash
# Run ON THE GPU BOX (through gpurun): kernel-trace stats + separate PMC passes of a short bench run.
# usage: tools/profile.sh <tag> [bench args...]; results under gpurun_out/prof_<tag>/
set -u
TAG=${1:-r1}; shift || true
# --no-pmc --pmc-child are always appended: bench.py must not start rocprofv3 passes of its own under this profiler, and runs exactly one step
ARGS=${@:-"--steps 1 --warmup 0 --spp-per-gpu 8 --no-cpu-baseline"}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
echo "$ARGS" > $OUT/args.txt
export TMPDIR=/tmp
run() { # name, rocprof flags...
  local name=$1; shift
  timeout -s KILL 240 rocprofv3 "$@" --output-format csv -d $OUT/$name -o $name -- python3 bench.py $ARGS --no-pmc --pmc-child > $OUT/$name.log 2>&1
  local rc=$?; echo "$name rc=$rc"; return $rc
}
run trace --kernel-trace --stats || exit 1
run sq --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE || exit 1
run tcc --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum || exit 1
run fetch --pmc FETCH_SIZE || exit 1
run write --pmc WRITE_SIZE || exit 1
run sq2 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_INSTS_FLAT || echo "sq2 failed (optional)"
run lds --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_LDS_ADDR_CONFLICT || echo "lds failed (optional)"
find $OUT -name "*.csv" | head -50
