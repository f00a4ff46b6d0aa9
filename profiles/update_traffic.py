#!/usr/bin/env python3
"""Folds a round's PMC summaries (profiles/run_profiles.sh rNN -> gpurun_out/keep/rNN_*) into profiles/traffic.json, the
file bench.py reads `roofline.traffic` and `roofline.sort_phase.pmc_bytes` from.  usage: python profiles/update_traffic.py r04"""
import json
import os
import shutil
import sys

R = sys.argv[1]
HERE = os.path.dirname(os.path.abspath(__file__))
KEEP = os.path.join(os.path.dirname(HERE), "gpurun_out", "keep")
for f in os.listdir(KEEP):
    if f.startswith(R + "_"):
        shutil.copy(os.path.join(KEEP, f), os.path.join(HERE, f))
pmc = json.load(open(os.path.join(HERE, R + "_pmc_kslam.json")))
t = json.load(open(os.path.join(HERE, "traffic.json")))
sc = pmc["k_scatter<4>"]
fetch, write = sc["FETCH_SIZE"]["mean"], sc["WRITE_SIZE"]["mean"]
prev = {k: t[k] for k in ("source", "method", "fetch_kb_per_launch_raw", "write_kb_per_launch_raw", "k_scatter_bytes_per_launch",
                          "algorithmic_bytes_per_launch") if k in t}
t.setdefault("round3", prev)
t["source"] = "profiles/%s_pmc_kslam.json (rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes, bench.py --steps 1 --warmup 0)" % R
t["fetch_kb_per_launch_raw"], t["write_kb_per_launch_raw"] = fetch, write
t["write_kb_per_launch_raw_max"], t["write_kb_per_launch_raw_min"] = sc["WRITE_SIZE"]["max"], sc["WRITE_SIZE"]["min"]
t["k_scatter_bytes_per_launch"] = int((2 * fetch + write) * 1024)
sp = json.load(open(os.path.join(HERE, R + "_sort_phase_pmc.json")))
t["sort_phase"] = {"source": "profiles/%s_sort_phase_pmc.json" % R, "bytes_per_call": sp["bytes_per_call"],
                   "dispatches_per_call": sp["dispatches_per_call"], "method": sp["method"],
                   "formula_bytes_per_call": "n_sorted x 16 x (2 x passes + 1), SURVEY.md 8d: 25289871 x 16 x 7 = 2832465552"}
json.dump(t, open(os.path.join(HERE, "traffic.json"), "w"), indent=1)
print("traffic.json: k_scatter<4> %d bytes per launch, sort phase %d bytes per call" % (t["k_scatter_bytes_per_launch"], sp["bytes_per_call"]))
