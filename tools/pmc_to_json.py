#!/usr/bin/env python3
"""profiles/<summary>.md (tools/summarise_prof.py) + the profiled run's bench line -> profiles/round1_traffic.json,
the PMC-derived figures bench.py quotes beside its live numbers.  usage: tools/pmc_to_json.py <summary.md> <trace.log> <kernel substring>"""
import json, re, sys
summary, tracelog, kernel = sys.argv[1], sys.argv[2], sys.argv[3]
txt = open(summary).read()
sec = txt[txt.index("### " + [l for l in txt.splitlines() if l.startswith("### ") and kernel in l][0][4:]):]
sec = sec[:sec.index("\n### ", 5)] if "\n### " in sec[5:] else sec
g = lambda name: float(re.search(r"- %s: ([0-9.e+]+)" % name, sec).group(1))
bench = json.loads([l for l in open(tracelog) if l.startswith('{"metric"')][-1])
seg = int(bench["rays_per_step"])
trace_ms = float(re.search(r"\| [^|]*%s[^|]*\| (\d+) \| ([0-9.]+) \|" % re.escape(kernel), txt).group(2))
d = {"source": f"rocprofv3 --pmc passes (separate runs) of `bench.py --steps 1 --warmup 0 --spp-per-gpu 8`, kernel {kernel}; {summary}",
     "kernel": kernel, "ray_segments": seg, "FETCH_SIZE_KB": g("FETCH_SIZE"), "WRITE_SIZE_KB": g("WRITE_SIZE"),
     "fetch_bytes_per_segment_as_reported": g("FETCH_SIZE") * 1e3 / seg, "fetch_bytes_per_segment_corrected_x2": 2 * g("FETCH_SIZE") * 1e3 / seg,
     "write_bytes_per_segment": g("WRITE_SIZE") * 1e3 / seg, "l2_hit_rate": g("TCC_HIT_sum") / g("TCC_REQ_sum"),
     "valu_busy_per_simd": 4 * g("SQ_ACTIVE_INST_VALU") / g("SQ_WAVE_CYCLES"), "valu_lane_utilisation": g("SQ_THREAD_CYCLES_VALU") / (g("SQ_ACTIVE_INST_VALU") * 64),
     "wave_wait_fraction": g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES"),
     "extend_ms_rocprof_kernel_trace": trace_ms, "extend_ms_hip_events_same_run": bench["kernel_ms_per_step"]["extend_ms"],
     "note": "gfx950 FETCH_SIZE tallies 128-B requests at 64 B (MI355X_MICROARCH.md, HBM section): the x2 value is the upper estimate; the access "
             "pattern is 16 B/lane gathers of 64-B pairs; Infinity-Cache hits count as fetches. valu_busy_per_simd = 4 waves/SIMD x SQ_ACTIVE_INST_VALU/SQ_WAVE_CYCLES."}
json.dump(d, open("profiles/round1_traffic.json", "w"), indent=1)
print(json.dumps(d, indent=1))
