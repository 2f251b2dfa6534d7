"""Copy the rocprofv3 summaries of scripts/profile_round.sh from gpurun_out/ into profiles/ and rebuild
profiles/traffic.json (HBM bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) KB, per the gfx950 guide)."""
import csv, glob, json, os, shutil, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
traffic = []
for cfg in ("cartpole", "quadrotor", "rocket_soc", "rocket_soc_workspace_kept", "rocket_soc_check_live", "cartpole_check_live"):
    family = "rocket_soc" if cfg.startswith("rocket_soc") else cfg.split("_")[0]
    ks = sorted(glob.glob(f"{root}/gpurun_out/prof_{tag}_{cfg}/*/*_kernel_stats.csv"), key=os.path.getmtime)
    if not ks:
        continue
    shutil.copy(ks[-1], f"{root}/profiles/{tag}_{cfg}_kernel_stats.csv")
    vals = {}
    for kind, ctr in (("F", "FETCH_SIZE"), ("W", "WRITE_SIZE")):
        f = sorted(glob.glob(f"{root}/gpurun_out/pmc{kind}_{tag}_{cfg}/*/*_counter_collection.csv"), key=os.path.getmtime)[-1]
        rows = [r for r in csv.DictReader(open(f)) if "admm" in r["Kernel_Name"]]
        vals[ctr] = sum(float(r["Counter_Value"]) for r in rows) / len(rows)
        kname = rows[0]["Kernel_Name"]
        with open(f"{root}/profiles/{tag}_{cfg}_pmc_{ctr.lower()}.csv", "w") as g:
            w = csv.writer(g)
            w.writerow(["Kernel_Name", "Grid_Size", "Workgroup_Size", "LDS_Block_Size", "VGPR_Count", "Accum_VGPR_Count", "Counter_Name", "Counter_Value"])
            for r in rows:
                w.writerow([r[k] for k in ("Kernel_Name", "Grid_Size", "Workgroup_Size", "LDS_Block_Size", "VGPR_Count", "Accum_VGPR_Count", "Counter_Name", "Counter_Value")])
    # SQ counters: one row per dispatch and counter; average over the dispatches of the ADMM kernel
    sq = {}
    for d in sorted(glob.glob(f"{root}/gpurun_out/pmcS_{tag}_{cfg}_SQ*") + glob.glob(f"{root}/gpurun_out/pmcS_{tag}_{cfg}_MFMA")):   # (not the passes of `cfg`_<pattern>)
        fs = sorted(glob.glob(f"{d}/*/*_counter_collection.csv"), key=os.path.getmtime)
        if not fs:
            continue
        acc = {}
        for r in csv.DictReader(open(fs[-1])):
            if "admm" in r["Kernel_Name"]:
                acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
        for k, v in acc.items():
            sq[k] = sum(v) / len(v)
    if sq:
        if "SQ_INSTS_VALU" in sq and "SQ_BUSY_CYCLES" in sq and "SQ_WAVES" in sq:
            # SQ_BUSY_CYCLES is summed over the 32 shader engines' SQs; a wave64 VALU instruction holds its SIMD 4 cycles
            busy = sq["SQ_BUSY_CYCLES"] / 32.0
            sq["derived_valu_issue_utilisation"] = sq["SQ_INSTS_VALU"] * 4.0 / (busy * 1024.0)
            sq["derived_note"] = "SQ_INSTS_VALU x 4 cycles / (kernel busy cycles x 1024 SIMDs)"
        if "SQ_INSTS_MFMA" in sq and "SQ_BUSY_CYCLES" in sq:
            # v_mfma_f64_16x16x4f64: 2 048 FLOP at 32 FLOP/clk/SIMD = 64 cycles of the SIMD's matrix core each
            sq["derived_mfma_issue_utilisation"] = sq["SQ_INSTS_MFMA"] * 64.0 / (sq["SQ_BUSY_CYCLES"] / 32.0 * 1024.0)
    log = open(f"{root}/gpurun_out/prof_{tag}_{cfg}.log").read()
    line = [l for l in log.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    if sq:
        sq["family"], sq["kernel"], sq["batch"], sq["pattern"] = family, d["config"]["kernel"], d["config"]["batch_per_gpu"], d["config"].get("pattern", "cold")
        sq["library_sha256"] = d.get("library_sha256")
        json.dump(sq, open(f"{root}/profiles/{tag}_{cfg}_sq_counters.json", "w"), indent=1)
    traffic.append({"tag": tag, "library_sha256": d.get("library_sha256"), "family": family, "pattern": d["config"].get("pattern", "cold"), "precision": 0, "batch": d["config"]["batch_per_gpu"], "kernel": d["config"]["kernel"],
                    "FETCH_SIZE_KB": vals["FETCH_SIZE"], "WRITE_SIZE_KB": vals["WRITE_SIZE"],
                    "hbm_bytes_per_launch": (2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0,
                    "algorithmic_bytes_per_launch": d["roofline"]["algorithmic_bytes_per_launch"],
                    "rocprof_kernel": kname,
                    "note": "rocprofv3 --pmc, separate passes; FETCH_SIZE doubled (gfx950 reports half of a coalesced read stream)"})
json.dump(traffic, open(f"{root}/profiles/traffic.json", "w"), indent=1)
print(json.dumps(traffic, indent=1))
