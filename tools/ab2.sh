#!/bin/bash
# like ab.sh but each arg is a full set of bench args; prints visits/s too
i=0
for v in "$@"; do
  i=$((i+1))
  timeout -k 10 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline $v > gpurun_out/ab2_$i.json 2>gpurun_out/ab2_$i.err || { echo "variant '$v' failed"; tail -3 gpurun_out/ab2_$i.err; exit 1; }
  python - "$v" gpurun_out/ab2_$i.json <<'PY'
import json,sys
d=json.load(open(sys.argv[2])); k=d["kernel_ms_per_step"]; r=d["roofline"]
seg=d["rays_per_step"]; 
print(f"{sys.argv[1]:44s} {d['value']:8.1f} Mrays/s step {d['ms_per_step']:7.1f} ms extend {k['extend_ms']:7.1f} shade {k['shade_ms']:5.1f} gen {k['generate_ms']:5.1f} | box/seg {r['box_tests_per_segment']:.1f} prim/seg {r['prim_tests_per_segment']:.2f} Gbox/s {r['box_tests_per_segment']*seg/k['extend_ms']/1e6:.2f} algGB/s {r['achieved']:.0f}")
PY
done
