#!/bin/bash
# Run ON THE GPU BOX (through gpurun): texture-addresser (TA) and vector-L1 (TCP) counters of one step of a bench workload, one
# rocprofv3 --pmc pass per counter group (at most 2 TA_ and 4 TCP_ counters per pass).  usage: tools/profile_ta.sh <tag> [bench args...]
# Results under gpurun_out/prof_<tag>/{ta1,ta2,ta3,tcp1,tcp2}; tools/summarise_ta.py <tag> prints the per-kernel ratios.
set -u
TAG=${1:-default}; shift || true
ARGS=${@:-"--steps 1 --warmup 0 --no-cpu-baseline"}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
run() { # name, counters...
  local name=$1; shift
  timeout -s KILL 240 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -o $name -- python3 bench.py $ARGS --no-pmc --pmc-child > $OUT/$name.log 2>&1
  local rc=$?; echo "$name rc=$rc"; return $rc
}
run ta1 TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum GRBM_GUI_ACTIVE || exit 1
run ta2 TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum || exit 1
run ta3 TA_TOTAL_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum || exit 1
run tcp1 TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum || exit 1
run tcp2 TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TA_TCP_STATE_READ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum || exit 1
