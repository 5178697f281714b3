#!/bin/bash
# A/B runs of bench.py on the GPU box: tools/ab.sh "ENV1=.. ENV2=.." "ENV.." ...   (each arg = one variant's env)
ARGS=${AB_ARGS:-"--steps 2 --warmup 1 --spp-per-gpu 16 --no-cpu-baseline"}
i=0
for v in "$@"; do
  i=$((i+1))
  env $v timeout -k 10 200 python bench.py $ARGS > gpurun_out/ab_$i.json 2>gpurun_out/ab_$i.err || { echo "variant '$v' failed"; tail -3 gpurun_out/ab_$i.err; exit 1; }
  python - "$v" gpurun_out/ab_$i.json <<'PY'
import json,sys
d=json.load(open(sys.argv[2])); k=d["kernel_ms_per_step"]
print(f"{sys.argv[1]:40s} {d['value']:8.1f} Mrays/s  step {d['ms_per_step']:7.1f} ms  extend {k['extend_ms']:7.1f} shade {k['shade_ms']:5.1f} gen {k['generate_ms']:5.1f}")
PY
done
