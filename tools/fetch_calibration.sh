#!/bin/bash
# Run ON THE GPU BOX (through gpurun): FETCH_SIZE calibration for per-lane 16-byte gathers (tools/mb/mb_fetch.hip).
# Leaves gpurun_out/fetch_cal/{plain.txt, pmc/**/fetch_counter_collection.csv}; tools/fetch_calibration.py turns them into
# profiles/round2_fetch_calibration.json.
set -u
OUT=$PWD/gpurun_out/fetch_cal; mkdir -p $OUT; export TMPDIR=/tmp
timeout -k 10 120 tools/mb/mb_fetch > $OUT/plain.txt 2>&1 || exit 1
cat $OUT/plain.txt
cd /tmp && timeout -s KILL 180 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc -o fetch -- $OLDPWD/tools/mb/mb_fetch > $OUT/pmc.txt 2>&1
echo "pmc rc=$?"
