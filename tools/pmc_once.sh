#!/bin/bash
# one PMC pass on the GPU box: tools/pmc_once.sh <tag> "<counters>" [bench args]
TAG=$1; CTR=$2; shift 2
ARGS=${@:-"--steps 1 --warmup 0 --spp-per-gpu 8 --no-cpu-baseline"}
OUT=$PWD/gpurun_out/pmc_$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout -s KILL 240 rocprofv3 --pmc $CTR --output-format csv -d $OUT -o p -- python3 bench.py $ARGS > $OUT/log.txt 2>&1; echo "rc=$?"
python3 - $OUT <<'PY'
import csv,glob,sys,collections
f=glob.glob(sys.argv[1]+"/**/*counter_collection.csv",recursive=True)
if not f: print("no csv"); print(open(sys.argv[1]+"/log.txt").read()[-1500:]); sys.exit(0)
t=collections.defaultdict(lambda: collections.defaultdict(float)); dur=collections.defaultdict(float); seen=set()
for r in csv.DictReader(open(f[0])):
    k=r["Kernel_Name"].split("(")[0]; t[k][r["Counter_Name"]]+=float(r["Counter_Value"])
    if (r["Dispatch_Id"]) not in seen: seen.add(r["Dispatch_Id"]); dur[k]+=float(r["End_Timestamp"])-float(r["Start_Timestamp"])
for k,c in t.items():
    if k.startswith("k_"): print(k, "dur_ms=%.2f"%(dur[k]/1e6), {n:"%.4g"%v for n,v in c.items()})
PY
