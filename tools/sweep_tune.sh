#!/bin/bash
# GPU box: sweep of the persistent extend kernel's RT_TUNE word (refill threshold << 16 | cost(internal) << 8 | cost(leaf)) on the default bench
export TMPDIR=/tmp
for T in 0x080101 0x040101 0x060101 0x0c0101 0x100101 0x080201 0x080102 0x080302 0x080203 0x080403 0x080304 0x0c0302 0x060203; do
  RT_TUNE=$T timeout -k 10 200 python -u bench.py --no-pmc --no-cpu-baseline --steps 2 --warmup 1 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$T', round(d['value'],1), round(d['ms_per_step'],2), round(d['kernel_ms_per_step']['extend_ms'],2))" || exit 1
done
