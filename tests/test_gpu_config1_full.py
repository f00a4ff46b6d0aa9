"""BASELINE configs[1] at its full size against the REFERENCE ITSELF, record for record: 1 M x 2 x 150 bp read pairs vs the
1 250-genome 5 Gb bacterial database -- ALL 8.15 M alignments of the batch (read, entry, rel, revComp, score, the four
coordinates, CIGAR length and offset, every CIGAR word), not a sub-database.  The checker is the reference's own
`alignToDatabase` (src/SLAM.h:59-79: KMer.h, Overlap.h, SmithWaterman.h, ssw.c compiled in place into
oracle/_ref/libslam_ref.so) on every CPU the job may use; rows may differ only where a multi-threaded run of the reference
is itself undecided (revComp ties), and each such row is checked to be one.  Without oracle/_ref (no
/root/reference where the tree was built) the checker is the restatement, which tests/test_oracle.py pins to the reference."""
import importlib
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
PAIRS = int(os.environ.get("KSLAM_TEST_CONFIG1_PAIRS", "1000000"))


# configs[1] itself; configs[4]'s read length and configs[2]'s database (bacterial + 10 k viral genomes of 5-200 kb: short
# entries, 11 250 ids) at half a batch each; a quarter batch against the repeat-rich database (rRNA-like segments and insertion
# elements shared by hundreds of entries: the join's long pile-ups, (read, entry) groups of tens of keys, 6.7 candidates per read
# instead of 4) -- every one against the reference's own alignToDatabase on the 5 Gb database
@pytest.mark.parametrize("read_len,pairs,n_viral,repeats", [(150, PAIRS, 0, False), (250, PAIRS // 2, 0, False), (150, PAIRS // 2, 10000, False),
                                                            (150, PAIRS // 4, 0, True)],
                         ids=["configs1", "250bp", "bacterial+viral", "repeat-rich"])
def test_config1_full_batch_equals_the_reference(kslam, oracle, read_len, pairs, n_viral, repeats):
    import torch
    assert torch.cuda.is_available(), "torch sees no HIP device"
    W = importlib.import_module("kslam_amd.workload")
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev)
    gen.manual_seed(1)
    db, offs = W.make_database(dev, gen, 250, 5, 4_000_000, n_viral=n_viral, repeats=repeats)
    n_entries = len(offs) - 1
    gen.manual_seed(2)
    reads = W.make_reads(dev, gen, db, offs, pairs, read_len=read_len, by_length=n_viral > 0)
    ctx = kslam.Context()
    torch.cuda.synchronize()   # torch wrote the database on its own stream
    ctx.set_index_device(n_entries, db.data_ptr(), offs)
    flat = reads.reshape(-1)
    ctx.load_reads_device(reads.shape[0], flat.data_ptr(), np.arange(reads.shape[0] + 1, dtype=np.uint64) * np.uint64(read_len))
    n_out, n_cig = ctx.align_resident()
    got, gcig = ctx.fetch_results(n_out, n_cig)
    ctx.close()
    # ---- the checker: the whole batch against the whole database on the host ----
    host_db = db.cpu().numpy()
    rn = np.ascontiguousarray(reads.cpu().numpy())
    del db, reads
    torch.cuda.empty_cache()
    rb = lambda i: rn[i].tobytes()
    eb = lambda j: host_db[int(offs[j]):int(offs[j + 1])].tobytes()
    if oracle.have_ref_slam():
        oracle.ref_slam_set_index_arrays(host_db, offs)
        exp, ecig, seconds, phases = oracle.ref_slam_align_to_database(
            rn.reshape(-1), np.arange(rn.shape[0] + 1, dtype=np.uint64) * np.uint64(read_len), threads=oracle.usable_cpus())
        oracle.ref_slam_set_index_arrays(host_db[:0], offs[:1])
        print("the reference's alignToDatabase: %.1f s on %d CPUs, phases %s" % (seconds, oracle.usable_cpus(), phases))
    else:
        gl = [eb(j) for j in range(n_entries)]
        rl = [rb(i) for i in range(rn.shape[0])]
        oracle.use_reference_ssw(True)
        oracle.set_num_threads(oracle.usable_cpus())
        try:
            exp, ecig, _ = oracle.align_to_database(rl, gl)
        finally:
            oracle.use_reference_ssw(False)
        del gl, rl
    assert len(exp) == len(got) and len(got) > 5 * pairs, (len(exp), len(got))
    v = oracle.compare_with_reference_rows(got, gcig, exp, ecig, rb, eb)
    print(v)
    assert v["identical"] and v["differing_rows_are_revcomp_ties"], v
    assert v["rows_differing"] <= 1000, v
