"""The bench workload generator's ground truth and the truth checker (k-slam_amd/workload.py),
validated on the CPU against the oracle: what check_against_truth demands of a result set must hold
for the oracle's own output on the same reads, and must FAIL when a planted overlap is removed."""
import importlib

import numpy as np
import pytest


def _pack(kslam, al):
    ov = np.zeros(len(al), dtype=kslam.OVERLAP_DT)
    for f in ("read", "entry", "rel", "revcomp", "score", "ref_begin", "ref_end", "query_begin",
              "query_end", "cigar_len", "cigar_off"):
        ov[f] = al[f]
    return ov


@pytest.mark.parametrize("read_len", [150, 250])
def test_truth_checker_accepts_the_oracle_and_rejects_damage(kslam, oracle, read_len):
    import torch
    W = importlib.import_module("kslam_amd.workload")
    dev = torch.device("cpu")
    gen = torch.Generator(device=dev)
    gen.manual_seed(5)
    db, offs = W.make_database(dev, gen, 3, 3, 40000)
    gen.manual_seed(6)
    reads, truth = W.make_reads(dev, gen, db, offs, 1500, read_len=read_len, with_truth=True)
    gen.manual_seed(6)
    again = W.make_reads(dev, gen, db, offs, 1500, read_len=read_len)
    assert torch.equal(reads, again)                       # same draws with and without truth
    rl = [reads[i].numpy().tobytes() for i in range(reads.shape[0])]
    gl = [db[int(offs[i]):int(offs[i + 1])].numpy().tobytes() for i in range(len(offs) - 1)]
    al, cg, _ = oracle.align_to_database(rl, gl)
    ov = _pack(kslam, al)
    t_ov = torch.from_numpy(ov.view(np.uint8).copy())
    t_cg = torch.from_numpy(cg.view(np.int32).copy())
    res = W.check_against_truth(t_ov, t_cg, truth, read_len)
    assert res["ok"], res
    assert res["planted_expected"] > 2000 and res["error_free_reads"] > 100
    # damage 1: drop one planted overlap
    want = int(torch.nonzero(truth["seed_ok"])[7])
    hit = np.nonzero((ov["read"] == want) & (ov["entry"] == int(truth["entry"][want])) &
                     (ov["rel"] == int(truth["rel"][want])))[0]
    assert len(hit) >= 1
    keep = np.ones(len(ov), dtype=bool)
    keep[hit] = False
    res = W.check_against_truth(torch.from_numpy(ov[keep].view(np.uint8).copy()), t_cg, truth, read_len)
    assert not res["ok"] and res["planted_missing"] == 1
    # damage 2: a wrong strand, a wrong score, a swapped pair of rows
    bad = ov.copy(); bad["revcomp"][hit[0]] ^= 1
    assert not W.check_against_truth(torch.from_numpy(bad.view(np.uint8).copy()), t_cg, truth, read_len)["ok"]
    bad = ov.copy(); bad["score"][hit[0]] = 7
    assert not W.check_against_truth(torch.from_numpy(bad.view(np.uint8).copy()), t_cg, truth, read_len)["ok"]
    bad = ov.copy(); bad[[0, len(bad) - 1]] = bad[[len(bad) - 1, 0]]
    assert W.check_against_truth(torch.from_numpy(bad.view(np.uint8).copy()), t_cg, truth, read_len)["unsorted_neighbours"] > 0
