#!/usr/bin/env python3
"""Where the SW phase's time goes per tier, and who is in each tier (explain 250 bp).

  python tools/sw_tiers.py           READ_LEN=250 PAIRS=1000000 by default (BASELINE configs[4]'s shape, one 1 M-pair batch)

For every candidate of the batch: is its final alignment the whole read on one diagonal (and with how many mismatches), and
which band does the CERTIFICATE demand for its final score (sw.hip certificate_amin / band_holds restated in numpy: the
narrowest tier that is certain to hold every optimal alignment)?  Cross-tabulated; the tier sizes the library actually ran
(KSLAM_DEBUG=1 lines of a child process) and, when profiles/<R>_kernel_stats_250bp.csv exists (rocprofv3 --kernel-trace
--stats of `bench.py --config 4 --pairs 1000000 --no-e2e`), the time of each tier's kernel beside it."""
import csv
import importlib
import json
import os
import re
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def certificate_half_width(score, L, W, match=2, gap_open=5, gap_extend=2):
    """the smallest h such that a band of diagonals [d0 - h, d0 + h - 1] passes band_holds for this score (d0 = 0)"""
    Lm = np.minimum(L, W)
    m00 = (score + match - 1) // match
    amin = m00.copy()
    room = Lm * match - score - gap_open
    g = np.where(gap_extend < match, np.minimum(room // gap_extend + 1, 2047), 1)
    m0 = (score + gap_open + (g - 1) * gap_extend + match - 1) // match
    amin = np.where(room >= 0, np.minimum(amin, m0 - g), amin)
    return np.maximum(L - amin, W - amin + 1)


def child():
    import torch
    import __graft_entry__ as entry
    K = entry.load_package()
    W = importlib.import_module("kslam_amd.workload")
    pairs, L = int(os.environ.get("PAIRS", "1000000")), int(os.environ.get("READ_LEN", "250"))
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev)
    gen.manual_seed(1)
    db, offs = W.make_database(dev, gen, 250, 5, 4_000_000)
    gen.manual_seed(2)
    reads = W.make_reads(dev, gen, db, offs, pairs, read_len=L)
    c = K.Context()
    c.set_index_device(len(offs) - 1, db.data_ptr(), offs)
    c.load_reads_device(reads.shape[0], reads.reshape(-1).data_ptr(), np.arange(reads.shape[0] + 1, dtype=np.uint64) * np.uint64(L))
    c.align_resident()
    n_out, n_cig = c.align_resident()
    ov, cg = c.fetch_results(n_out, n_cig)
    tm = c.timings()
    n = len(ov)
    one_op = ov["cigar_len"] == 1
    whole = np.zeros(n, dtype=bool)
    idx = np.flatnonzero(one_op)
    whole[idx] = (cg[ov["cigar_off"][idx]] == ((L << 4) | 0)) & (ov["query_begin"][idx] == 0) & (ov["query_end"][idx] == L - 1)
    ungapped = one_op                                             # <n>M: one diagonal, possibly clipped
    score = ov["score"].astype(np.int64)
    glen = np.diff(offs.astype(np.int64))[ov["entry"]]
    s0 = np.maximum(ov["rel"].astype(np.int64), 0)
    Wn = np.minimum(L, glen - s0)
    half = certificate_half_width(score, np.full(n, L, dtype=np.int64), Wn)
    tiers = [16, 32, 48, 64, 96] if L <= 160 else [16, 32, 48, 64, 96, 128]
    need = np.full(n, len(tiers), dtype=np.int64)                 # len(tiers): no band certifies -> full matrix
    for k in range(len(tiers) - 1, -1, -1):
        need[2 * half <= tiers[k]] = k
    names = [str(t) for t in tiers] + ["full"]
    table = {}
    for k, nm in enumerate(names):
        sel = need == k
        m = np.where(whole & sel, (2 * L - score) // 5, -1)
        table[nm] = {"candidates": int(sel.sum()), "final_alignment_ungapped": int((ungapped & sel).sum()),
                     "whole_read_on_one_diagonal": int((whole & sel).sum()),
                     "median_mismatches_of_those": int(np.median(m[m >= 0])) if (m >= 0).any() else None,
                     "gapped": int((~ungapped & sel).sum())}
    print(json.dumps({"read_len": L, "pairs": pairs, "candidates": int(n), "ms_sw": round(tm["ms_sw"], 3), "ms_total": round(tm["ms_total"], 3),
                      "tier_the_certificate_demands_for_the_final_score": table,
                      "whole_read_on_one_diagonal": int(whole.sum()), "ungapped": int(ungapped.sum())}))


def main():
    if os.environ.get("KSLAM_SW_TIERS_CHILD"):
        return child()
    env = dict(os.environ, KSLAM_SW_TIERS_CHILD="1", KSLAM_DEBUG="1")
    env.setdefault("READ_LEN", "250")
    r = subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stderr[-3000:])
        raise SystemExit(1)
    out = json.loads([x for x in r.stdout.splitlines() if x.startswith("{")][-1])
    planned, ran, full = None, {}, None
    for line in r.stderr.splitlines():            # the LAST alignment call's lines win
        m = re.match(r"\[kslam\] SW planned: (.*)", line)
        if m:
            planned, ran = [int(x) for x in m.group(1).split("/")], {}
        m = re.match(r"\[kslam\] SW tier (\d+) \((\d+) diagonals\): (\d+) candidates", line)
        if m:
            ran[m.group(2)] = ran.get(m.group(2), 0) + int(m.group(3))
        m = re.match(r"\[kslam\] SW full matrix: (\d+) candidates", line)
        if m:
            full = int(m.group(1))
    out["planned_by_k_sw_plan_per_chunk_last"] = planned
    out["ran_per_tier"] = ran
    out["ran_full_matrix"] = full
    stats = os.environ.get("KERNEL_STATS")
    if stats and os.path.exists(stats):
        per = {}
        for row in csv.DictReader(open(stats)):
            nm = row["Name"]
            if "k_sw" in nm:
                per[nm.replace("(anonymous namespace)::", "").replace("void ", "").replace("kslam::", "").split("(")[0]] = {
                    "calls": int(row["Calls"]), "avg_ms": round(float(row["AverageNs"]) / 1e6, 3), "total_ms": round(float(row["TotalDurationNs"]) / 1e6, 3)}
        out["kernel_stats"] = per
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
