#!/bin/bash
# Run ON THE GPU BOX: L1 TLB (UTCL1) requests / misses and the L1-miss latency sum of one step of the default bench workload.
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_utcl; mkdir -p $OUT
timeout -s KILL 240 rocprofv3 --pmc TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_TCC_READ_REQ_LATENCY_sum --output-format csv -d $OUT/u -o u -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-pmc --pmc-child > $OUT/u.log 2>&1; echo rc=$?
