#!/bin/bash
# Run ON THE GPU BOX: A/B of library builds / env settings / bench args.  Each argument: "label|ENV=.. ENV=..|extra bench args".
# Prints one line per variant and appends it to gpurun_out/ab3.txt.  Stops at the first failure (no retries).
BASE=${AB_BASE:-"--steps 2 --warmup 1 --no-cpu-baseline --no-pmc"}
for spec in "$@"; do
  IFS='|' read -r label envs extra <<< "$spec"
  out=gpurun_out/ab3_$(echo "$label" | tr -c 'A-Za-z0-9_\n' '_')
  env $envs timeout -k 10 300 python -u bench.py $BASE $extra > $out.json 2> $out.err || { echo "variant '$label' failed"; tail -5 $out.err; exit 1; }
  python - "$label" $out.json <<'PY' | tee -a gpurun_out/ab3.txt
import json,sys
d=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1]); k=d["kernel_ms_per_step"]; r=d.get("roofline",{})
print(f"{sys.argv[1]:34s} {d['value']:8.1f} Mrays/s  step {d['ms_per_step']:8.2f} ms | extend {k['extend_ms']:8.2f} shade {k['shade_ms']:7.2f} gen {k['generate_ms']:6.2f} acc {k['resolve_ms']:5.2f} | l1 {r.get('l1_lane_access_bound',{}).get('frac',0):.3f}")
PY
  grep -h "^\[stamps\]" $out.err | tee -a gpurun_out/ab3.txt
done
