#!/usr/bin/env python3
"""Condenses one workload's passes of scripts/profile_all.sh into gpurun_out/prof_<tag>/<workload>_counters.json (+ _kernel_stats.csv) and updates
gpurun_out/prof_<tag>/trace_kernel_hbm_traffic.json.  Counters are averaged per launch of each kernel variant; the step counts the per-step figures
divide by come from the bench line of the same command (kr_stats: steps_total, steps_strict_side, rk45 evaluated trial steps)."""
import collections, csv, glob, json, os, subprocess, sys

tag, wl, cmd = sys.argv[1:4]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", f"prof_{tag}", wl)


def variant_of(name):
    for key in ("trace_multi_kernel", "trace_kernel"):
        if key in name:
            return name[name.index(key):name.index(">", name.index(key)) + 1] if ">" in name else name[:80]
    return None


def bench_line(path):
    try:
        return json.loads([l for l in open(path) if l.startswith("{")][-1])
    except Exception:
        return None


commit = os.environ.get("KR_TREE_COMMIT") or subprocess.run(["git", "-C", root, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() or "unknown"
res = {"tag": tag, "workload": wl, "command": f"rocprofv3 --kernel-trace --stats | --pmc <one group per pass> -- {cmd}   (scripts/profile_all.sh)", "kernels": {}}
# durations (unprofiled-counter pass: kernel trace only)
stats = glob.glob(f"{src}/trace/*/*_kernel_stats.csv")
dur = {}
if stats:
    rows = list(csv.DictReader(open(stats[0])))
    with open(os.path.join(os.path.dirname(src), f"{wl}_kernel_stats.csv"), "w") as f:
        f.write(open(stats[0]).read())
    for r in rows:
        v = variant_of(r["Name"])
        if v:
            dur[v] = {"calls": int(r["Calls"]), "avg_ms": float(r["AverageNs"]) / 1e6, "min_ms": float(r["MinNs"]) / 1e6, "max_ms": float(r["MaxNs"]) / 1e6}
    res["top_kernels_by_total_time"] = [{"name": r["Name"][:100], "calls": int(r["Calls"]), "avg_ms": float(r["AverageNs"]) / 1e6, "pct": float(r["Percentage"])} for r in rows[:8]]
per = collections.defaultdict(lambda: collections.defaultdict(list))
meta = {}
for d in sorted(glob.glob(f"{src}/pmc_*")):
    for f in glob.glob(f"{d}/*/*_counter_collection.csv"):
        for row in csv.DictReader(open(f)):
            v = variant_of(row["Kernel_Name"])
            if not v:
                continue
            per[v][row["Counter_Name"]].append(float(row["Counter_Value"]))
            meta[v] = {"grid": int(row["Grid_Size"]), "workgroup": int(row["Workgroup_Size"]), "vgpr": int(row["VGPR_Count"]), "accum_vgpr": int(row.get("Accum_VGPR_Count", 0) or 0),
                       "sgpr": int(row["SGPR_Count"]), "scratch": int(row["Scratch_Size"])}
b = bench_line(f"{src}/bench_trace.json") or {}
roof = b.get("roofline", {})
steps_total = b.get("rk_steps_per_launch", 0)
units_total = roof.get("work_units_per_launch", steps_total)
side = b.get("launch_split", {})
res["bench_line_of_the_trace_pass"] = {k: b.get(k) for k in ("value", "ms_per_step", "rk_steps_per_sec", "rk_steps_per_launch", "rk45")}
res["bench_line_of_the_trace_pass"]["roofline"] = {k: roof.get(k) for k in ("frac", "achieved", "avg_kernel_ms", "split_launch_ms", "longest_ray", "work_units_per_launch")}
res["bench_line_of_the_trace_pass"]["launch_split"] = side
for v, cs in per.items():
    c = {k: sum(x) / len(x) for k, x in cs.items()}
    hog = v.rstrip(">").split(",")[-2].strip() == "true" if v.count(",") >= 5 else False        # <T, METHOD, USE_DEST, FAST, HOG, REFILL>
    units = side.get("work_units_strict_side") if hog else (units_total - (side.get("work_units_strict_side") or 0)) if side else units_total
    # a workload that launches a variant several times per pass (the returning-radiation groups): the counters are per-launch averages, the work units per pass
    passes = 3                                                   # --warmup 1 --steps 2
    launches_per_pass = max(1, round((dur.get(v) or {}).get("calls", passes) / passes))
    if units:
        units = units / launches_per_pass
    k = {"dispatch": meta[v], "duration": dur.get(v), "counters_per_launch": c, "is_strict_side_launch": hog, "launches_per_pass": launches_per_pass, "work_units_per_launch": units}
    der = {}
    if units:
        for name, key in (("valu_wave_instructions_per_wave_step(x64 lanes / work units)", "SQ_INSTS_VALU"), ("salu_per_wave_step", "SQ_INSTS_SALU"), ("branches_per_wave_step", "SQ_INSTS_BRANCH"),
                          ("fp64_fma_per_wave_step", "SQ_INSTS_VALU_FMA_F64"), ("fp64_mul_per_wave_step", "SQ_INSTS_VALU_MUL_F64"), ("fp64_add_per_wave_step", "SQ_INSTS_VALU_ADD_F64"),
                          ("fp64_quarter_rate_per_wave_step", "SQ_INSTS_VALU_TRANS_F64"), ("valu_cvt_per_wave_step", "SQ_INSTS_VALU_CVT"), ("valu_int32_per_wave_step", "SQ_INSTS_VALU_INT32"),
                          ("valu_int64_per_wave_step", "SQ_INSTS_VALU_INT64"), ("smem_per_wave_step", "SQ_INSTS_SMEM"), ("vmem_per_wave_step", "SQ_INSTS_VMEM")):
            if key in c:
                der[name] = 64.0 * c[key] / units
    if "SQ_THREAD_CYCLES_VALU" in c and "SQ_ACTIVE_INST_VALU" in c:
        # (the two come from different passes of the same deterministic workload)
        der["lane_occupancy_pct(SQ_THREAD_CYCLES_VALU / (64 SQ_ACTIVE_INST_VALU))"] = 100.0 * c["SQ_THREAD_CYCLES_VALU"] / (64.0 * c["SQ_ACTIVE_INST_VALU"])
    if "SQ_ACTIVE_INST_VALU" in c:
        der["valu_busy_ms_per_simd(SQ_ACTIVE_INST_VALU x 4 cycles / 1024 SIMDs / 2.4 GHz)"] = c["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / 2.4e6
        if dur.get(v):
            der["valu_busy_pct_of_the_launch"] = 100.0 * der["valu_busy_ms_per_simd(SQ_ACTIVE_INST_VALU x 4 cycles / 1024 SIMDs / 2.4 GHz)"] / dur[v]["avg_ms"]
    if all(x in c for x in ("SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_TRANS_F64")):
        der["executed_fp64_flop_per_launch(64 lanes x (2 FMA + MUL + ADD + TRANS))"] = 64.0 * (2 * c["SQ_INSTS_VALU_FMA_F64"] + c["SQ_INSTS_VALU_MUL_F64"] + c["SQ_INSTS_VALU_ADD_F64"] + c["SQ_INSTS_VALU_TRANS_F64"])
    if "SQ_WAVE_CYCLES" in c and "SQ_WAVES" in c and c["SQ_WAVES"]:
        der["wave_cycles_per_wave"] = c["SQ_WAVE_CYCLES"] / c["SQ_WAVES"]
    if "FETCH_SIZE" in c:
        der["hbm_read_bytes(FETCH_SIZE KB x 1024; gfx950: x2 for wide coalesced 16-B/lane reads, uncalibrated for this strided AoS pattern -- both shown)"] = [c["FETCH_SIZE"] * 1024, 2 * c["FETCH_SIZE"] * 1024]
    if "WRITE_SIZE" in c:
        der["hbm_write_bytes(WRITE_SIZE KB x 1024)"] = c["WRITE_SIZE"] * 1024
    k["derived"] = der
    res["kernels"][v] = k
res["built_from_commit"] = commit
out = os.path.join(os.path.dirname(src), f"{wl}_counters.json")
json.dump(res, open(out, "w"), indent=1)
# HBM traffic of the trace kernels per pass (what bench.py's roofline.traffic_from_profile quotes)
tfile = os.path.join(os.path.dirname(src), "trace_kernel_hbm_traffic.json")
t = json.load(open(tfile)) if os.path.exists(tfile) else {}
# (a round's workloads may be profiled in several gpurun calls, each on a fresh box: entries of the committed file that belong to the same tree are kept)
committed = os.path.join(root, "profiles", "trace_kernel_hbm_traffic.json")
if os.path.exists(committed):
    old = json.load(open(committed))
    if commit in str(old.get("_source", "")):
        t = {**old, **t}
rd = sum(k["counters_per_launch"].get("FETCH_SIZE", 0) * k.get("launches_per_pass", 1) for k in res["kernels"].values()) * 1024
wr = sum(k["counters_per_launch"].get("WRITE_SIZE", 0) * k.get("launches_per_pass", 1) for k in res["kernels"].values()) * 1024
key = {"rk4": "emissivity_rk4", "rk45": "emissivity_rk45", "euler": "emissivity_euler", "imageplane": "imageplane_rk4", "return_radiation": "return_radiation_euler"}[wl]
t[key] = {"read_bytes_raw(FETCH_SIZE x 1024)": rd, "read_bytes_x2(gfx950 correction for wide coalesced reads)": 2 * rd, "write_bytes(WRITE_SIZE x 1024)": wr,
          "algorithmic_bytes(288 B x rays traced)": 288 * (b.get("config", {}).get("rays_per_gpu") or 0)}
t["_source"] = f"scripts/profile_all.sh {tag} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in passes of their own; all trace kernels of one pass summed); tree {commit}"
json.dump(t, open(tfile, "w"), indent=1)
print(json.dumps({v: {"ms": (k["duration"] or {}).get("avg_ms"), **{a: round(x, 2) for a, x in k["derived"].items() if isinstance(x, float)}} for v, k in res["kernels"].items()}, indent=1))
