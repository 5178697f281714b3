#!/usr/bin/env python3
"""Per-kernel sums of the TA / TCP counter passes of tools/profile_ta.sh and the ratios that say how busy the vector-memory address
path was.  usage: python tools/summarise_ta.py <tag> [out.md]"""
import collections, csv, glob, os, sys
tag = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set)
for f in glob.glob(f"gpurun_out/prof_{tag}/*/*_counter_collection.csv"):
    if os.path.basename(os.path.dirname(f)) not in ("ta1", "ta2", "ta3", "tcp1", "tcp2"): continue
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); disp[k].add(r["Dispatch_Id"])
lines = [f"# TA / TCP counters `{tag}` (one bench step per pass; sums over all dispatches of a kernel)", ""]
for k, v in sorted(acc.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0)):
    g = v.get("GRBM_GUI_ACTIVE", 0)
    if g < 1e6: continue
    cyc = g / 8.0                      # summed over 8 XCDs
    lines.append(f"## {k}  ({cyc / 2.4e6:.1f} ms at 2.4 GHz)")
    lines.append("")
    lines.append("raw: " + ", ".join(f"{n} {x:.6g}" for n, x in sorted(v.items())))
    lines.append("")
    ta = v.get("TA_TA_BUSY_sum", 0) / (256 * cyc)
    lines.append(f"- **TA busy = {ta:.3f}** of the cycles (TA_TA_BUSY summed over 256 TAs / (256 x GRBM_GUI_ACTIVE / 8))")
    if v.get("TA_TOTAL_WAVEFRONTS_sum"):
        lines.append(f"- TA busy cycles per wavefront instruction = {v['TA_TA_BUSY_sum'] / v['TA_TOTAL_WAVEFRONTS_sum']:.1f}; flat/global read wavefronts {v.get('TA_FLAT_READ_WAVEFRONTS_sum', 0):.4g} of {v['TA_TOTAL_WAVEFRONTS_sum']:.4g}")
    lines.append(f"- TA address stalled by TC = {v.get('TA_ADDR_STALLED_BY_TC_CYCLES_sum', 0) / (256 * cyc):.3f}, by TD = {v.get('TA_ADDR_STALLED_BY_TD_CYCLES_sum', 0) / (256 * cyc):.3f}, data stalled by TC = {v.get('TA_DATA_STALLED_BY_TC_CYCLES_sum', 0) / (256 * cyc):.3f} of the cycles")
    lines.append(f"- TCP: gate_en1 (busy) = {v.get('TCP_GATE_EN1_sum', 0) / (256 * cyc):.3f}, gate_en2 = {v.get('TCP_GATE_EN2_sum', 0) / (256 * cyc):.3f}, pending-stall = {v.get('TCP_PENDING_STALL_CYCLES_sum', 0) / (256 * cyc):.3f}, "
                 f"TCP->TA data stall = {v.get('TCP_TCP_TA_DATA_STALL_CYCLES_sum', 0) / (256 * cyc):.3f}, read tag-conflict stall = {v.get('TCP_READ_TAGCONFLICT_STALL_CYCLES_sum', 0) / (256 * cyc):.3f} of the cycles")
    if v.get("TCP_TOTAL_CACHE_ACCESSES_sum"):
        lines.append(f"- TCP cache accesses = {v['TCP_TOTAL_CACHE_ACCESSES_sum']:.4g} = {v['TCP_TOTAL_CACHE_ACCESSES_sum'] / (256 * cyc):.3f} per clock and CU; TA->TCP state reads {v.get('TCP_TA_TCP_STATE_READ_sum', 0):.4g}; "
                     f"read requests to L2 = {v.get('TCP_TCC_READ_REQ_sum', 0):.4g} ({v.get('TCP_TCC_READ_REQ_sum', 0) / v['TCP_TOTAL_CACHE_ACCESSES_sum']:.3f} per cache access)")
    lines.append("")
out = "\n".join(lines)
print(out)
if len(sys.argv) > 2: open(sys.argv[2], "w").write(out + "\n")
