"""BASELINE configs[1] at its full size against the oracle, record for record: 1 M x 2 x 150 bp read pairs vs the 1 250-genome
5 Gb bacterial database -- ALL 8.15 M alignments of the batch (read, entry, rel, revComp, score, the four coordinates, CIGAR
length and offset, every CIGAR word), not a sub-database.  The oracle (pinned to the reference compiled in place,
tests/test_oracle.py) runs on every CPU the job may use: about 30 s on 16."""
import importlib
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
PAIRS = int(os.environ.get("KSLAM_TEST_CONFIG1_PAIRS", "1000000"))


def test_config1_full_batch_equals_the_oracle(kslam, oracle):
    import torch
    assert torch.cuda.is_available(), "torch sees no HIP device"
    W = importlib.import_module("kslam_amd.workload")
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev)
    gen.manual_seed(1)
    db, offs = W.make_database(dev, gen, 250, 5, 4_000_000)
    n_entries = len(offs) - 1
    gen.manual_seed(2)
    reads = W.make_reads(dev, gen, db, offs, PAIRS, read_len=150)
    ctx = kslam.Context()
    torch.cuda.synchronize()   # torch wrote the database on its own stream
    ctx.set_index_device(n_entries, db.data_ptr(), offs)
    flat = reads.reshape(-1)
    ctx.load_reads_device(reads.shape[0], flat.data_ptr(), np.arange(reads.shape[0] + 1, dtype=np.uint64) * np.uint64(150))
    n_out, n_cig = ctx.align_resident()
    got, gcig = ctx.fetch_results(n_out, n_cig)
    ctx.close()
    # ---- the checker: the whole batch against the whole database on the host ----
    host_db = db.cpu().numpy()
    gl = [host_db[int(offs[i]):int(offs[i + 1])].tobytes() for i in range(n_entries)]
    del host_db
    rn = reads.cpu().numpy()
    rl = [rn[i].tobytes() for i in range(rn.shape[0])]
    del db, reads
    torch.cuda.empty_cache()
    oracle.use_reference_ssw(True)          # the SSW core of the checker = the reference's own ssw.c when oracle/_ref has it
    oracle.set_num_threads(oracle.usable_cpus())
    try:
        exp, ecig, _ = oracle.align_to_database(rl, gl)
    finally:
        oracle.use_reference_ssw(False)
    assert len(exp) == len(got) and len(got) > 7 * PAIRS
    for f in ("read", "entry", "rel", "revcomp", "score", "ref_begin", "ref_end", "query_begin", "query_end", "cigar_len", "cigar_off"):
        assert (got[f] == exp[f]).all(), f
    assert np.array_equal(gcig, ecig)
