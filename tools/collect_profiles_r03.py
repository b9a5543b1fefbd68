"""Copies the round's evidence from gpurun_out/profile_r03/ (tools/profile_round3.sh) into profiles/ and derives
profiles/r03_counters.json — the per-step counter sums bench.py puts into its roofline block."""
import csv, glob, json, os, re, shutil, sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft   # (the hash of the library's sources the counters belong to)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "profile_r03")
DST = os.path.join(ROOT, "profiles")
for f in glob.glob(os.path.join(SRC, "bench_*.json")):
    shutil.copy(f, os.path.join(DST, "r03_" + os.path.basename(f)))
for f in glob.glob(os.path.join(SRC, "kernel_stats_*.csv")):
    shutil.copy(f, os.path.join(DST, "r03_" + os.path.basename(f)))


def pmc(cfg):
    """kernel -> {counter: (mean per launch, launches seen)}"""
    out = {}
    cur = None
    for line in open(os.path.join(SRC, f"pmc_{cfg}", "summary.txt")):
        if not line.startswith(" "):
            cur = line.strip().split("(")[0]
            out[cur] = {}
        else:
            m = re.match(r"\s+(\S+)\s+mean\s+([\d.]+)\s+n=(\d+)", line)
            if m:
                out[cur][m.group(1)] = (float(m.group(2)), int(m.group(3)))
    shutil.copy(os.path.join(SRC, f"pmc_{cfg}", "summary.txt"), os.path.join(DST, f"r03_pmc_{cfg}_summary.txt"))
    return out


def per_frame(cfg):
    """launches per steady-state frame: calls / frames, frames = launches of the resolve step (one per frame); kernels of
    the first frame's schedule alone (before the host knows how little a frame shows) round to zero"""
    calls = {}
    for r in csv.DictReader(open(os.path.join(SRC, f"kernel_stats_{cfg}.csv"))):
        calls[r["Name"].split("(")[0]] = int(r["Calls"])
    frames = max(v for k, v in calls.items() if "k_wf_resolve" in k)
    return {k: float(round(v / frames)) for k, v in calls.items()}


counters = {"_note": "per-step (= per-frame) sums over the kernels of one frame, from rocprofv3 --kernel-trace --pmc passes (tools/pmc2.sh, "
                     "one counter group per run; per-launch means x launches per frame); FETCH_SIZE doubled on gfx950 (64 B counted per "
                     "128-B read request: MI355X_MICROARCH.md HBM section) EXCEPT for the trace kernels' gather of 32-byte ray records at "
                     "random slots, which the guide leaves uncalibrated and tools/ubench/gather32.hip calibrated (profiles/"
                     "r03_gather_calibration.txt: FETCH_SIZE = 64 B per 32-byte record read, i.e. one sub-line request per record, counted "
                     "once: not doubled); WRITE_SIZE taken as reported; KB -> bytes x 1024"}
GATHER_KERNELS = ("k_wf_trace_packet", "k_wf_trace_lane")   # FETCH_SIZE counts their requests whole (see _note)
CFGS = ("cfg2", "cfg2b", "cfg3", "cfg4", "cfg5")
for cfg in CFGS:
    p = pmc(cfg)
    kernels = [k for k in p if any(t in k for t in ("k_primary_p2", "k_wf_", "k_bin_", "k_frame_setup"))]
    if cfg in ("cfg2", "cfg2b"):   # the reference frame: one launch of each per frame
        per = {k: 1.0 for k in kernels}
    else:
        per = per_frame(cfg)
    tot = {"SQ_ACTIVE_INST_VALU": 0.0, "SQ_INSTS_VALU": 0.0, "FETCH_SIZE_KB": 0.0, "WRITE_SIZE_KB": 0.0, "FETCH_BYTES": 0.0}
    detail = {}
    for k in kernels:
        n = per.get(k, 0.0)
        c = p[k]
        d = {"launches_per_frame": round(n, 3)}
        for name, key in (("SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_VALU"), ("SQ_INSTS_VALU", "SQ_INSTS_VALU"), ("FETCH_SIZE", "FETCH_SIZE_KB"),
                          ("WRITE_SIZE", "WRITE_SIZE_KB")):
            if name in c:
                d[key] = c[name][0]
                tot[key] += c[name][0] * n
                if name == "FETCH_SIZE":
                    tot["FETCH_BYTES"] += c[name][0] * n * 1024.0 * (1.0 if any(g in k for g in GATHER_KERNELS) else 2.0)
        detail[k] = d
    traffic = int(tot["FETCH_BYTES"] + tot["WRITE_SIZE_KB"] * 1024)
    dominant = max(detail, key=lambda k: detail[k].get("SQ_ACTIVE_INST_VALU", 0.0) * detail[k]["launches_per_frame"]) if detail else None
    # lanes the vector instructions had live: SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU, where the thr pass ran (both counters of
    # the SAME pass; 64 = every lane of every instruction)
    lanes = {}
    for k in kernels:
        c = p[k]
        if "SQ_THREAD_CYCLES_VALU" in c and c.get("SQ_ACTIVE_INST_VALU", (0, 0))[0] > 0:
            lanes[k] = round(c["SQ_THREAD_CYCLES_VALU"][0] / c["SQ_ACTIVE_INST_VALU"][0], 1)
    counters[cfg] = {"source": f"profiles/r03_pmc_{cfg}_summary.txt", "frames_in_flight": 2 if cfg in ("cfg2", "cfg2b") else 3, "dominant_kernel": dominant,
                     **{k: round(v, 1) for k, v in tot.items()}, "traffic_bytes_per_step": traffic,
                     "valu_lanes_live_of_64": lanes or None, "kernels": detail}
counters["_csrc_tree"] = graft.load_package().csrc_tree()   # bench.py flags its roofline block when the library has changed since
json.dump(counters, open(os.path.join(DST, "r03_counters.json"), "w"), indent=1)
for f in glob.glob(os.path.join(SRC, "*.txt")) :
    if os.path.basename(f).startswith(("frame_graph", "dist_probe")):
        shutil.copy(f, os.path.join(DST, "r03_" + os.path.basename(f)))
for cfg in CFGS:
    c = counters[cfg]
    print(cfg, {k: v for k, v in c.items() if k != "kernels"})
