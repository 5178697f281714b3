#!/usr/bin/env python3
"""Summarise gpurun_out/prof_<tag>/ (rocprofv3 csv output) into profiles/<tag>_summary.md."""
import csv, glob, os, sys, collections
tag = sys.argv[1] if len(sys.argv) > 1 else "r1"
src = f"gpurun_out/prof_{tag}"
out = [f"# rocprofv3 summary `{tag}`\n", "Command per pass: `rocprofv3 <flags> --output-format csv -- python3 bench.py " + (open(f"{src}/args.txt").read().strip() if os.path.exists(f"{src}/args.txt") else "--steps 1 --warmup 0 --spp-per-gpu 8 --no-cpu-baseline") + "`\n"]
def find(pat):
    r = glob.glob(f"{src}/**/{pat}", recursive=True)
    return r[0] if r else None
f = find("*kernel_stats.csv")
if f:
    out.append("## --kernel-trace --stats\n\n| kernel | calls | total ms | avg us | min us | max us | % |\n|---|---|---|---|---|---|---|")
    for r in csv.DictReader(open(f)):
        out.append(f"| {r['Name'][:60]} | {r['Calls']} | {float(r['TotalDurationNs'])/1e6:.2f} | {float(r['AverageNs'])/1e3:.1f} | {float(r['MinNs'])/1e3:.1f} | {float(r['MaxNs'])/1e3:.1f} | {float(r['Percentage']):.1f} |")
    out.append("")
tot = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
for name in ("sq", "sq2", "tcc", "fetch", "write"):
    f = find(f"{name}_counter_collection.csv")
    if not f: continue
    seen = set()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if name == "sq" and r["Counter_Name"] == "SQ_WAVES": calls[k] += 1
out.append("## PMC (summed over all dispatches of the run; separate passes)\n")
for k, c in tot.items():
    out.append(f"### {k[:70]}  ({calls[k]} dispatches)\n")
    for n, v in sorted(c.items()): out.append(f"- {n}: {v:.6g}")
    d = []
    if c.get("SQ_ACTIVE_INST_VALU") and c.get("SQ_THREAD_CYCLES_VALU"): d.append(f"VALU lane utilisation = THREAD_CYCLES_VALU/(ACTIVE_INST_VALU*64) = {c['SQ_THREAD_CYCLES_VALU']/(c['SQ_ACTIVE_INST_VALU']*64):.3f}")
    if c.get("SQ_WAVE_CYCLES"): 
        for n in ("SQ_WAIT_ANY","SQ_WAIT_INST_ANY","SQ_ACTIVE_INST_ANY","SQ_ACTIVE_INST_VALU"):
            if c.get(n): d.append(f"{n}/WAVE_CYCLES = {c[n]/c['SQ_WAVE_CYCLES']:.3f}")
    if c.get("TCC_HIT_sum") is not None and (c.get("TCC_HIT_sum",0)+c.get("TCC_MISS_sum",0))>0: d.append(f"L2 hit rate = {c['TCC_HIT_sum']/(c['TCC_HIT_sum']+c['TCC_MISS_sum']):.3f}")
    if c.get("FETCH_SIZE") is not None: d.append(f"FETCH_SIZE = {c['FETCH_SIZE']/1e6:.3f} GB as reported (KB units); x2 gfx950 correction for wide streaming reads = {2*c['FETCH_SIZE']/1e6:.3f} GB")
    if c.get("WRITE_SIZE") is not None: d.append(f"WRITE_SIZE = {c['WRITE_SIZE']/1e6:.3f} GB")
    out += [f"- **{x}**" for x in d]; out.append("")
os.makedirs("profiles", exist_ok=True)
open(f"profiles/{tag}_summary.md", "w").write("\n".join(out) + "\n")
print("\n".join(out))
