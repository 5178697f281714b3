#!/bin/bash
# Run ON THE GPU BOX: A/B of library builds / env settings on the default bench.  usage: tools/ab_lib.sh "<env assignments>" ...
for cfg in "$@"; do
  echo "== $cfg"
  env $cfg timeout -k 10 300 python -u bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), round(d['ms_per_step'],1), {k:round(v,1) for k,v in d['kernel_ms_per_step'].items()})" || exit 1
done
