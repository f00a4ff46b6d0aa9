# SQ instruction counters of the SW kernels for one bench step -> gpurun_out/keep/<tag>_valu.json
TAG=${1:-r01f}
REPO="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"; cd /tmp && export TMPDIR=/tmp; cd "$REPO" || exit 1; rm -rf /tmp/prof4; mkdir -p /tmp/prof4 gpurun_out/keep
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d /tmp/prof4 -o x -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-full-pipeline > /tmp/o4 2> /tmp/e4
rm -rf /tmp/prof5; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof5 -o x -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-full-pipeline > /tmp/o5 2> /tmp/e5
python3 - "$TAG" <<'PY'
import csv, glob, json, sys
tag = sys.argv[1]
def clean(n): return n.replace('(anonymous namespace)::', '').replace('kslam_api::', '').replace('kslam::', '').replace('void ', '').split('(')[0]
agg = {}
ndisp = {}
for f in glob.glob('/tmp/prof4/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = clean(r['Kernel_Name'])
        if not k.startswith(('k_sw', 'k_cigar_systolic', 'k_banded', 'k_extract_filter', 'k_join_fill')): continue
        agg.setdefault(k, {}).setdefault(r['Counter_Name'], 0.0)
        agg[k][r['Counter_Name']] += float(r['Counter_Value'])
        if r['Counter_Name'] == 'SQ_WAVES':
            ndisp[k] = ndisp.get(k, 0) + 1
dur = {}
for r in csv.DictReader(open(glob.glob('/tmp/prof5/**/*kernel_stats.csv', recursive=True)[0])):
    dur[clean(r['Name'])] = float(r['TotalDurationNs'])
out = {"source": "rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS and, in a separate run, --kernel-trace --stats; bench.py --steps 1 --warmup 0",
       "peak_valu_wave_instr_per_s": 256 * 4 * 2.4e9 / 2,
       "peak_note": "256 CUs x 4 SIMD-32 x 2.4 GHz, a wave64 VALU instruction issues over 2 cycles (MI355X_MICROARCH.md).  "
                    "tools/valu_peak.hip measures what the chip sustains per instruction kind: ~1.08 T/s for the 2-cycle "
                    "kinds (32-bit add/sub/and/or/xor/mov/lshr with VGPR or literal sources) and ~0.575 T/s for the 4-cycle "
                    "kinds (v_max/min_i32, every VOP3 / DPP / SDWA encoding, any SGPR source, v_max_f64): a kernel made mostly "
                    "of the latter tops out near 0.6 T/s = 0.5 of this nominal peak",
       "measured_rate_2cycle_kinds": 1.084e12, "measured_rate_4cycle_kinds": 0.575e12,
       "kernels": {}}
tot_i = tot_t = 0
for k, c in sorted(agg.items()):
    t = dur.get(k)
    if not t: continue
    rate = c.get('SQ_INSTS_VALU', 0) / (t * 1e-9)
    out["kernels"][k] = {"valu_wave_instr": c.get('SQ_INSTS_VALU'), "salu": c.get('SQ_INSTS_SALU'), "lds": c.get('SQ_INSTS_LDS'),
                         "waves": c.get('SQ_WAVES'), "dispatches": ndisp.get(k), "duration_ms": t / 1e6, "valu_wave_instr_per_s": rate,
                         "frac_of_peak": rate / out["peak_valu_wave_instr_per_s"]}
    if k.startswith('k_sw_band'): tot_i += c.get('SQ_INSTS_VALU', 0); tot_t += t
out["sw_band_total"] = {"valu_wave_instr": tot_i, "duration_ms": tot_t / 1e6, "valu_wave_instr_per_s": tot_i / (tot_t * 1e-9),
                        "frac_of_peak": tot_i / (tot_t * 1e-9) / out["peak_valu_wave_instr_per_s"]}
# the SW phase of ONE alignment call: every k_sw* kernel, instructions per dispatch x dispatches per call (1 each)
sw = {k: v for k, v in out["kernels"].items() if k.startswith('k_sw')}
out["sw_phase_per_align"] = {
    "valu_wave_instr": sum(v["valu_wave_instr"] / max(v["dispatches"] or 1, 1) for v in sw.values()),
    "kernels": sorted(sw), "note": "sum over the k_sw* kernels of SQ_INSTS_VALU per dispatch (each runs once per alignment call of the bench workload); bench.py divides it by the ms_sw it measures live"}
json.dump(out, open('gpurun_out/keep/%s_valu.json' % tag, 'w'), indent=1)
for k, v in out["kernels"].items(): print(k.ljust(36), "%.0f G/s  %.2f of peak  (%.2f ms)" % (v["valu_wave_instr_per_s"] / 1e9, v["frac_of_peak"], v["duration_ms"]))
print("sw band total", out["sw_band_total"])
PY
