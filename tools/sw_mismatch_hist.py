#!/usr/bin/env python3
"""Measure before building: how many candidates of the configs[1] batch would a closed-form tier for
seed diagonals with m mismatches take?  From the results of one alignment call: rows whose alignment is the whole read on
one diagonal (CIGAR = <L>M, query 0..L-1) have score = match * L - (match + mismatch) * m, which gives m; everything else
is gapped, clipped or truncated.  Prints the histogram; KSLAM_DEBUG=1 in the environment adds the library's tier counts."""
import importlib
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402


def main():
    K = entry.load_package()
    W = importlib.import_module("kslam_amd.workload")
    pairs = int(os.environ.get("PAIRS", "1000000"))
    L = int(os.environ.get("READ_LEN", "150"))
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev)
    gen.manual_seed(1)
    db, offs = W.make_database(dev, gen, 250, 5, 4_000_000)
    gen.manual_seed(2)
    reads = W.make_reads(dev, gen, db, offs, pairs, read_len=L)
    c = K.Context()
    c.set_index_device(len(offs) - 1, db.data_ptr(), offs)
    c.load_reads_device(reads.shape[0], reads.reshape(-1).data_ptr(), np.arange(reads.shape[0] + 1, dtype=np.uint64) * np.uint64(L))
    n_out, n_cig = c.align_resident()
    ov, cg = c.fetch_results(n_out, n_cig)
    tm = c.timings()
    n = len(ov)
    first = cg[ov["cigar_off"][ov["cigar_len"] > 0]]
    one_op = (ov["cigar_len"] == 1)
    whole = np.zeros(n, dtype=bool)
    idx = np.flatnonzero(one_op)
    whole[idx] = (cg[ov["cigar_off"][idx]] == ((L << 4) | 0)) & (ov["query_begin"][idx] == 0) & (ov["query_end"][idx] == L - 1)
    m = (2 * L - ov["score"][whole].astype(np.int64))
    assert (m % 5 == 0).all()
    m //= 5
    hist = np.bincount(m, minlength=12)
    out = {"candidates": int(n), "whole_read_on_one_diagonal": int(whole.sum()),
           "by_mismatches": {str(k): int(v) for k, v in enumerate(hist[:12])},
           "fraction_by_mismatches": {str(k): round(float(v) / n, 4) for k, v in enumerate(hist[:12])},
           "gapped_or_clipped": int(n - whole.sum()), "ms_sw": round(tm["ms_sw"], 3), "ms_total": round(tm["ms_total"], 3)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
