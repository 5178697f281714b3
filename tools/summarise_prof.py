#!/usr/bin/env python3
"""Summarise gpurun_out/prof_<tag>/ (tools/profile.sh: rocprofv3 csv output, one pass per counter group) into
profiles/<out>_summary.md: per kernel, the trace durations, the raw counter sums and the ratios DESIGN.md quotes.
usage: tools/summarise_prof.py <tag> [out name, default <tag>]

Derived figures (MI355X_MICROARCH.md for the peaks and the counter units):
  VALU lane utilisation  = SQ_THREAD_CYCLES_VALU / (SQ_ACTIVE_INST_VALU * 64)
  VALU useful fraction   = SQ_INSTS_VALU * 64 * lane utilisation / kernel seconds / (256 CU * 64 lanes * 2.4 GHz)
  VALU issue fraction    = SQ_INSTS_VALU * 64 / kernel seconds / the same peak
  fabric GB/s            = (FETCH_SIZE * cal + WRITE_SIZE) KiB / kernel seconds, against 8 TB/s; cal: gathers are counted at size,
                           coalesced streams at half (profiles/round2_fetch_calibration.json) - both readings are printed
Kernel seconds are those of the un-counted --kernel-trace pass (counter passes serialise dispatches and run slower)."""
import collections
import csv
import glob
import json
import os
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r1"
outname = sys.argv[2] if len(sys.argv) > 2 else tag
src = f"gpurun_out/prof_{tag}"
args = open(f"{src}/args.txt").read().strip() if os.path.exists(f"{src}/args.txt") else "?"
PEAK_LANE_OPS = 256 * 64 * 2.4e9


def find(pat):
    r = glob.glob(f"{src}/**/{pat}", recursive=True)
    return r[0] if r else None


def short(name):
    return name.split("(")[0].replace("void ", "").strip()


out = [f"# rocprofv3 summary `{outname}`\n", f"Command per pass: `rocprofv3 <flags> --output-format csv -- python3 bench.py {args}`\n"]
bench = None
if os.path.exists(f"{src}/trace.log"):
    lines = [l for l in open(f"{src}/trace.log") if l.startswith('{"metric"')]
    if lines:
        bench = json.loads(lines[-1])
        out.append(f"Bench line of the traced pass: **{bench['value']:.1f} Mrays/s**, {bench['ms_per_step']:.2f} ms/step, workload: {bench['config']['workload']}; "
                   f"HIP-event kernel ms per step: " + ", ".join(f"{k} {v:.2f}" for k, v in bench["kernel_ms_per_step"].items()) + "\n")
secs = {}
f = find("*kernel_stats.csv")
if f:
    out.append("## --kernel-trace --stats\n\n| kernel | calls | total ms | avg us | min us | max us | % |\n|---|---|---|---|---|---|---|")
    for r in csv.DictReader(open(f)):
        out.append(f"| {r['Name'][:70]} | {r['Calls']} | {float(r['TotalDurationNs'])/1e6:.3f} | {float(r['AverageNs'])/1e3:.1f} | {float(r['MinNs'])/1e3:.1f} | {float(r['MaxNs'])/1e3:.1f} | {float(r['Percentage']):.1f} |")
        secs[short(r["Name"])] = float(r["TotalDurationNs"]) / 1e9
    out.append("")
tot = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
for name in ("sq", "sq2", "tcc", "fetch", "write"):
    f = find(f"{name}_counter_collection.csv")
    if not f:
        continue
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if name == "sq" and r["Counter_Name"] == "SQ_WAVES":
            calls[k] += 1
out.append("## PMC (summed over all dispatches of the run; one pass per counter group)\n")
for k, c in tot.items():
    if not k.startswith("k_"):
        continue
    t = secs.get(k)
    out.append(f"### {k[:80]}  ({calls[k]} dispatches" + (f", {t*1e3:.3f} ms in the trace pass)" if t else ")") + "\n")
    out.append("raw: " + ", ".join(f"{n} {v:.6g}" for n, v in sorted(c.items())) + "\n")
    d = []
    if c.get("SQ_ACTIVE_INST_VALU") and c.get("SQ_THREAD_CYCLES_VALU"):
        util = c["SQ_THREAD_CYCLES_VALU"] / (c["SQ_ACTIVE_INST_VALU"] * 64)
        d.append(f"VALU lane utilisation = {util:.3f}")
        if t and c.get("SQ_INSTS_VALU"):
            d.append(f"VALU useful fraction of the f64-rate peak (3.93e13 lane-op/s) = {c['SQ_INSTS_VALU']*64*util/t/PEAK_LANE_OPS:.3f}; issue fraction = {c['SQ_INSTS_VALU']*64/t/PEAK_LANE_OPS:.3f}")
    if c.get("SQ_WAVE_CYCLES"):
        for n in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU"):
            if c.get(n):
                d.append(f"{n}/WAVE_CYCLES = {c[n]/c['SQ_WAVE_CYCLES']:.3f}")
    if c.get("GRBM_GUI_ACTIVE") and c.get("SQ_WAVE_CYCLES"):
        d.append(f"mean resident waves per SIMD = {c['SQ_WAVE_CYCLES']*4.0/(c['GRBM_GUI_ACTIVE']/8.0)/1024.0:.2f} (SQ_WAVE_CYCLES counts quad-cycles; GRBM_GUI_ACTIVE is summed over 8 XCDs)")
    if c.get("TCC_HIT_sum") is not None and (c.get("TCC_HIT_sum", 0) + c.get("TCC_MISS_sum", 0)) > 0:
        d.append(f"L2 hit rate = {c['TCC_HIT_sum']/(c['TCC_HIT_sum']+c['TCC_MISS_sum']):.3f}")
    if c.get("FETCH_SIZE") is not None:
        fb = c["FETCH_SIZE"] * 1024.0; wb = c.get("WRITE_SIZE", 0.0) * 1024.0
        line = f"FETCH_SIZE = {fb/1e9:.3f} GB as reported (gathers are counted at size), {2*fb/1e9:.3f} GB if all of it were coalesced streams (x2); WRITE_SIZE = {wb/1e9:.3f} GB"
        if t:
            line += f"; fabric rate = {(fb+wb)/t/1e9:.0f} .. {(2*fb+wb)/t/1e9:.0f} GB/s = {(fb+wb)/t/8e12:.2f} .. {(2*fb+wb)/t/8e12:.2f} of the 8 TB/s HBM peak"
        d.append(line)
    out += [f"- **{x}**" for x in d]; out.append("")
os.makedirs("profiles", exist_ok=True)
open(f"profiles/{outname}_summary.md", "w").write("\n".join(out) + "\n")
print("\n".join(out))
