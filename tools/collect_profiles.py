#!/usr/bin/env python3
"""gpurun_out/prof_<tag>{,_default}/ + gpurun_out/bench_<tag>.json (tools/profile_final.sh) -> the files under profiles/ that
bench.py and DESIGN.md quote: round1_<tag>_summary.md, round1_traffic.json, round1_<tag>_default_kernel_stats.{csv,md},
round1_<tag>_bench.json.  usage: tools/collect_profiles.py <tag>"""
import csv, glob, json, os, shutil, subprocess, sys
tag = sys.argv[1]
K = "k_extend_bvh2<1, 1, false>"
subprocess.check_call([sys.executable, "tools/summarise_prof.py", tag], stdout=subprocess.DEVNULL)
os.replace(f"profiles/{tag}_summary.md", f"profiles/round1_{tag}_summary.md")
subprocess.check_call([sys.executable, "tools/pmc_to_json.py", f"profiles/round1_{tag}_summary.md", f"gpurun_out/prof_{tag}/trace.log", K], stdout=subprocess.DEVNULL)
d = f"gpurun_out/prof_{tag}_default"
# kernel-trace stats of the default command
src = glob.glob(f"{d}/trace/**/*kernel_stats.csv", recursive=True)[0]
shutil.copy(src, f"profiles/round1_{tag}_default_kernel_stats.csv")
bench = json.loads([l for l in open(f"{d}/trace.log") if l.startswith('{"metric"')][-1])
rows = list(csv.DictReader(open(src)))
md = [f"# `rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline` ({tag})\n",
      f"bench line of the same run: {bench['value']:.1f} Mrays/s, {bench['ms_per_step']:.1f} ms/step; HIP-event extend avg launch "
      f"{bench['roofline']['avg_launch_ms']:.3f} ms over {bench['roofline']['launches']} launches in the timed steps\n",
      "| kernel | calls | total ms | avg us | min us | max us | % |", "|---|---|---|---|---|---|---|"]
for r in rows:
    md.append(f"| {r['Name'][:70]} | {r['Calls']} | {float(r['TotalDurationNs'])/1e6:.2f} | {float(r['AverageNs'])/1e3:.1f} | {float(r['MinNs'])/1e3:.1f} | {float(r['MaxNs'])/1e3:.1f} | {float(r['Percentage']):.1f} |")
ext = [r for r in rows if r["Name"].startswith("void k_extend_bvh2<1, 1, false>") or K in r["Name"]][0]
md.append(f"\nextend: rocprof average {float(ext['AverageNs'])/1e6:.3f} ms per launch vs {bench['roofline']['avg_launch_ms']:.3f} ms from HIP events in the same run.")
open(f"profiles/round1_{tag}_default_kernel_stats.md", "w").write("\n".join(md) + "\n")
# FETCH_SIZE / WRITE_SIZE of the default command, extend dispatches only
def total(name, counter):
    f = glob.glob(f"{d}/{name}/**/*counter_collection.csv", recursive=True)[0]
    tot, n = 0.0, 0
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter and "k_extend_bvh2" in r["Kernel_Name"]:
            tot += float(r["Counter_Value"]); n += 1
    return tot, n
fs, nd = total("fetch", "FETCH_SIZE"); ws, _ = total("write", "WRITE_SIZE")
fb = json.loads([l for l in open(f"{d}/fetch.log") if l.startswith('{"metric"')][-1])
steps_total = fb["steps"] + fb["warmup"]; seg = fb["rays_per_step"] * steps_total
t = json.load(open("profiles/round1_traffic.json"))
t["default_command_passes"] = {
    "command": f"rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (one pass each) -- python3 bench.py --no-cpu-baseline ({steps_total} steps of {fb['config']['workload'].split(';')[4].strip()}, {nd} extend dispatches)",
    "FETCH_SIZE_KB_sum": fs, "WRITE_SIZE_KB_sum": ws, "extend_dispatches": nd,
    "fetch_bytes_per_launch_as_reported": fs * 1e3 / nd, "fetch_bytes_per_launch_corrected_x2": 2 * fs * 1e3 / nd, "write_bytes_per_launch": ws * 1e3 / nd,
    "fetch_bytes_per_segment_as_reported": fs * 1e3 / seg, "write_bytes_per_segment": ws * 1e3 / seg}
json.dump(t, open("profiles/round1_traffic.json", "w"), indent=1)
line = [l for l in open(f"gpurun_out/bench_{tag}.json") if l.startswith('{"metric"')][-1]
open(f"profiles/round1_{tag}_bench.json", "w").write(line)
print(open(f"profiles/round1_{tag}_default_kernel_stats.md").read()); print(json.dumps(t["default_command_passes"], indent=1))
