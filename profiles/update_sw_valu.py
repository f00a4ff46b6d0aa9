#!/usr/bin/env python3
"""profiles/rNN_valu.json (tools/pmc_valu.sh rNN) -> profiles/sw_valu.json, the file bench.py reads the SW phase's VALU
instruction count from (`roofline_valu`).  usage: python profiles/update_sw_valu.py r04"""
import json
import os
import sys

R = sys.argv[1]
HERE = os.path.dirname(os.path.abspath(__file__))
d = json.load(open(os.path.join(HERE, R + "_valu.json")))
out = {"source": "profiles/%s_valu.json (tools/pmc_valu.sh %s: rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS, "
                 "bench.py --steps 1 --warmup 0)" % (R, R),
       "sw_phase_per_align": d["sw_phase_per_align"],
       "per_kernel_per_align": {k: {"valu_wave_instr": v["valu_wave_instr"] / max(1, v.get("dispatches", 1)), "frac_of_peak": v["frac_of_peak"]}
                                for k, v in sorted(d["kernels"].items())}}
json.dump(out, open(os.path.join(HERE, "sw_valu.json"), "w"), indent=1)
print("sw_valu.json: SW phase %.4g VALU wave-instructions per alignment call" % out["sw_phase_per_align"]["valu_wave_instr"])
