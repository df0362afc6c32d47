#!/usr/bin/env python3
"""Condenses a gpurun_out/prof_<tag>/ directory (scripts/profile_round.sh) into profiles/<tag>_*.{csv,json}."""
import collections, csv, glob, json, os, shutil, sys
tag = sys.argv[1]
src = f"gpurun_out/prof_{tag}"
os.makedirs("profiles", exist_ok=True)
def newest(pattern):
    """gpurun merges every call's output into gpurun_out/: earlier runs of the same script leave their files (other PIDs) next to the new ones"""
    fs = glob.glob(pattern)
    return [max(fs, key=os.path.getmtime)] if fs else []


ks = newest(f"{src}/trace/*/*_kernel_stats.csv")
if ks:
    shutil.copy(ks[0], f"profiles/{tag}_kernel_stats.csv")
summary = {"tag": tag, "command": "rocprofv3 --kernel-trace --stats / --pmc <group> -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline (scripts/profile_round.sh)", "counters_per_trace_kernel_launch": {}}
for d in sorted(glob.glob(f"{src}/pmc_*")):
    fs = newest(f"{d}/*/*_counter_collection.csv")
    if not fs: continue
    agg = collections.defaultdict(list); per = collections.defaultdict(lambda: collections.defaultdict(list)); meta = {}
    for row in csv.DictReader(open(fs[0])):
        if "trace_kernel" in row["Kernel_Name"]:
            # hybrid mode runs two flavours per pass (fast main launch + strict register-hog side launch): sum them per pass
            # for the whole-pass figures, and keep them apart under "per_kernel_variant"
            agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
            name = row["Kernel_Name"]
            variant = name[name.index("trace_kernel"):name.index(">") + 1] if ">" in name else name[:60]
            per[variant][row["Counter_Name"]].append(float(row["Counter_Value"]))
            meta[variant] = {"grid": int(row["Grid_Size"]), "workgroup": int(row["Workgroup_Size"]), "vgpr": int(row["VGPR_Count"]), "accum_vgpr": int(row.get("Accum_VGPR_Count", 0) or 0), "sgpr": int(row["SGPR_Count"]), "lds": int(row["LDS_Block_Size"]), "scratch": int(row["Scratch_Size"])}
    nvar = max(1, len(per))
    for k, v in agg.items():
        summary["counters_per_trace_kernel_launch"][k] = nvar * sum(v) / len(v)      # per pass = sum over the variants of one pass
    for variant, d in per.items():
        summary.setdefault("per_kernel_variant", {}).setdefault(variant, {}).update({k: sum(v) / len(v) for k, v in d.items()})
        summary["per_kernel_variant"][variant]["dispatch"] = meta[variant]
for f in sorted(glob.glob(f"{src}/bench_*.json")):
    try:
        b = json.loads(open(f).read().strip().splitlines()[-1])
        summary.setdefault("bench_under_profiler", {})[os.path.basename(f)] = {"value_rays_per_s": b["value"], "rk_steps_per_sec": b["rk_steps_per_sec"], "avg_kernel_ms": b["roofline"]["avg_kernel_ms"], "rk_steps_per_launch": b["rk_steps_per_launch"]}
    except Exception:
        pass
c = summary["counters_per_trace_kernel_launch"]
der = {}
if "GRBM_GUI_ACTIVE" in c:
    der["gpu_cycles_per_launch(GRBM_GUI_ACTIVE/8 XCDs)"] = c["GRBM_GUI_ACTIVE"] / 8
    if "SQ_ACTIVE_INST_VALU" in c:
        der["VALUBusy_pct(gfx94x formula: 100*SQ_ACTIVE_INST_VALU*4/1024 SIMDs/(GRBM_GUI_ACTIVE/8))"] = 100 * c["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / (c["GRBM_GUI_ACTIVE"] / 8)
if "SQ_THREAD_CYCLES_VALU" in c and "SQ_ACTIVE_INST_VALU" in c:
    der["VALUUtilization_pct(100*SQ_THREAD_CYCLES_VALU/(SQ_ACTIVE_INST_VALU*64))"] = 100 * c["SQ_THREAD_CYCLES_VALU"] / (c["SQ_ACTIVE_INST_VALU"] * 64)
if all(k in c for k in ("SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_TRANS_F64")):
    der["executed_fp64_flop_per_launch(64 lanes x (2 FMA + MUL + ADD + TRANS))"] = 64 * (2 * c["SQ_INSTS_VALU_FMA_F64"] + c["SQ_INSTS_VALU_MUL_F64"] + c["SQ_INSTS_VALU_ADD_F64"] + c["SQ_INSTS_VALU_TRANS_F64"])
if "FETCH_SIZE" in c: der["hbm_read_bytes(FETCH_SIZE KB x 1024; uncalibrated for 16-B/lane AoS reads, guide says x2 for wide coalesced)"] = c["FETCH_SIZE"] * 1024
if "WRITE_SIZE" in c: der["hbm_write_bytes(WRITE_SIZE KB x 1024)"] = c["WRITE_SIZE"] * 1024
summary["derived"] = der
json.dump(summary, open(f"profiles/{tag}_counters.json", "w"), indent=1)
print(json.dumps(summary, indent=1))
