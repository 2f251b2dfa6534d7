#!/usr/bin/env python3
"""Summarise gpurun_out/prof_stream_<tag>/ (rocprofv3 sqlite outputs of scripts/profile_stream.sh) into
profiles/<tag>_stream_config4.json: per-launch averages of the stream kernel's duration and counters."""
import glob
import json
import os
import sqlite3
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", f"prof_stream_{tag}")
out = {"command": "python3 bench.py --config rocket_soc --steps 5 --warmup 1 --no-cpu-baseline", "counters": {}}


def tables(cur, key):
    return [r[0] for r in cur.execute("select name from sqlite_master where type='table'") if key in r[0]]


for db in sorted(glob.glob(os.path.join(src, "*", "*_results.db"))):
    cur = sqlite3.connect(db).cursor()
    kd, ks = tables(cur, "kernel_dispatch")[0], tables(cur, "kernel_symbol")[0]
    pmc, info = tables(cur, "pmc_event"), tables(cur, "info_pmc")
    if "trace" in db:
        q = f"select s.kernel_name, count(*), avg(d.end - d.start) from {kd} d join {ks} s on d.kernel_id = s.id group by s.kernel_name"
        for name, n, avg in cur.execute(q):
            if "streamg" in name:
                out["kernel"] = name.split("(")[0]
                out["launches"] = n
                out["avg_duration_us"] = avg / 1e3
        continue
    # counters come as one row per (dispatch, hardware instance): sum the instances, average the dispatches
    q = (f"select i.name, d.dispatch_id, sum(e.value) from {pmc[0]} e join {info[0]} i on e.pmc_id = i.id "
         f"join {kd} d on e.event_id = d.event_id join {ks} s on d.kernel_id = s.id "
         f"where s.kernel_name like '%streamg%' group by i.name, d.dispatch_id")
    acc = {}
    for name, _, v in cur.execute(q):
        acc.setdefault(name, []).append(v)
    for name, vs in acc.items():
        out["counters"][name] = sum(vs) / len(vs)
c = out["counters"]
if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
    # KB units; FETCH_SIZE doubled on gfx950 (MI355X_MICROARCH.md, HBM / rocprofv3 section)
    out["hbm_bytes_per_launch"] = (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0
    if "avg_duration_us" in out:
        out["hbm_GBps"] = out["hbm_bytes_per_launch"] / (out["avg_duration_us"] * 1e-6) / 1e9
dst = os.path.join(root, "profiles", f"{tag}_stream_config4.json")
json.dump(out, open(dst, "w"), indent=1)
print(open(dst).read())
