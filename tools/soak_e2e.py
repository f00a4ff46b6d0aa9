"""Soak of the whole GPU chain up to the SAM text against the oracle chain, random data sets until the time is up:
    python tools/soak_e2e.py [seconds] [first_seed]
reads + qualities -> pipelined entry (alignment, device pairing / screens / pseudo-assembly, per-row walk of the
referenced rows) -> host SAM formatter   ==   oracle alignToDatabase -> oracle host tail (restatement of the
reference's getPairedOverlaps ... writeSAMOutputPairs), byte for byte.  Shapes vary per seed: read length, divergence,
indel rate, N rate, reads over genome ends, plain or tandem-repeat genomes, tail flags."""
import ctypes as C
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
torch.cuda.init()
import oracle as O                      # noqa: E402  (test infrastructure: this tool is a checker)
from conftest import load_kslam        # noqa: E402
import test_gpu_parity as TP           # noqa: E402

K = load_kslam()
synth = importlib.import_module("kslam_amd.synth")
T = importlib.import_module("kslam_amd.tail")
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
FROM_FASTQ = os.environ.get("KSLAM_SOAK_FASTQ") == "1"     # through kslam_submit_batch_fastq_text instead of pointer arrays
t_end = time.time() + budget
seed, rounds, total_bytes = seed0, 0, 0
while time.time() < t_end:
    rng = np.random.default_rng(seed)
    read_len = int(rng.choice([60, 100, 150, 150, 250]))
    n_pairs = int(rng.choice([600, 1500, 3000]))
    if rng.random() < 0.3:
        reads_b, genomes_b = TP._low_complexity_dataset(synth, seed, int(rng.integers(2, 6)), n_pairs // 2 * 2, read_len)
        kind = "low-complexity"
    else:
        g = synth.make_genomes(seed, int(rng.integers(1, 4)), int(rng.integers(1, 5)), int(rng.integers(6000, 40000)),
                               strain_sub=float(rng.uniform(0.0, 0.05)), strain_indel=float(rng.uniform(0, 0.004)),
                               shared_segment=int(rng.choice([0, 1500])))
        rd, _ = synth.make_paired_reads(seed + 1, g, n_pairs, read_len=read_len, frag_mean=2 * read_len + 60, frag_sd=30,
                                        sub_rate=float(rng.uniform(0, 0.04)), indel_rate=float(rng.uniform(0, 0.01)),
                                        n_rate=float(rng.choice([0, 0.003])), edge_frac=float(rng.uniform(0, 0.2)))
        reads_b, genomes_b = synth.to_bytes(rd), synth.to_bytes(g)
        kind = "plain"
    n_reads = len(reads_b)
    quals = [bytes(rng.integers(33, 75, len(b), dtype=np.uint8)) for b in reads_b]
    ids = [b"f%d" % (i % (n_reads // 2)) for i in range(n_reads)]
    R = T.Reads(reads_b, quals, ids)
    genes = [[(50 + 700 * k, 700 * k + 650, b"g%d" % k, b"WP_%d" % k if k % 3 else b"", b"product %d" % k) for k in range(8)]
             for _ in genomes_b]
    I = T.Index(genomes_b, locus_tags=[b"NC_%06d" % i for i in range(len(genomes_b))],
                taxonomy_ids=[100 + i for i in range(len(genomes_b))], genes=genes)
    pseudo = bool(rng.random() < 0.6)
    P = T.TailParams.default(pseudo_assembly=pseudo, score_threshold=int(rng.choice([0, 0, 60])),
                             score_fraction=float(rng.choice([0.95, 0.8])), num_sam_alignments=int(rng.choice([10, 2])))
    ctx = K.Context(score_threshold=int(P.score_threshold))
    ctx.set_index(genomes_b)
    keep_b = [C.create_string_buffer(b, len(b) + 1) for b in reads_b]
    keep_q = [C.create_string_buffer(q, len(q) + 1) for q in quals]
    bp = (C.c_char_p * n_reads)(*[C.cast(x, C.c_char_p) for x in keep_b])
    qp = (C.c_char_p * n_reads)(*[C.cast(x, C.c_char_p) for x in keep_q])
    lens = np.array([len(b) for b in reads_b], dtype=np.uint32)
    ctx.set_pairing(paired=True, score_threshold=int(P.score_threshold), score_fraction=float(P.score_fraction), stages=7 if pseudo else 3)
    reads_view = R
    if FROM_FASTQ:     # the same batch as two FASTQ texts: records found, columns cut and identifiers taken on the device
        half = n_reads // 2
        eol = [b"\n", b"\r\n"][seed & 1]
        def text(lo, mate):
            return b"".join(b"@f%d/%d extra words%s%s%s+%s%s%s" % (i - lo, mate, eol, reads_b[i], eol, eol, quals[i], eol) for i in range(lo, lo + half))
        t1, t2 = text(0, 1), text(half, 2)
        h1, h2 = K.HostBuffer(len(t1) + 64), K.HostBuffer(len(t2) + 64)
        h1.a[:len(t1)] = np.frombuffer(t1, dtype=np.uint8)
        h2.a[:len(t2)] = np.frombuffer(t2, dtype=np.uint8)
        o, g_, d, m, release = ctx.collect_batch(ctx.submit_batch_fastq_text(h1.ptr, len(t1), h2.ptr, len(t2)))
        reads_view = ctx.last_reads
        assert reads_view.n_reads == n_reads and reads_view.consumed == (len(t1), len(t2))
    else:
        o, g_, d, m, release = ctx.collect_batch(ctx.submit_batch_full(n_reads, C.cast(bp, C.c_void_p), C.cast(qp, C.c_void_p), lens.ctypes.data))
    rp, pr, pst = ctx.last_pairs
    Pw = T.TailParams.default(pseudo_assembly=pseudo and not (pst["stages_done"] & 4), score_threshold=int(P.score_threshold),
                              score_fraction=float(P.score_fraction), num_sam_alignments=int(P.num_sam_alignments))
    out = []
    T.tail_finish_rows(Pw, reads_view, I, o, g_, d, m, rp.copy(), pr.copy(), out.append)
    got = b"".join(out)
    release()
    if FROM_FASTQ:
        h1.close()
        h2.close()
    ctx.close()
    eal, ecig, _ = O.align_to_database(reads_b, genomes_b, O.Params.default(score_threshold=int(P.score_threshold)))
    exp = O.tail_sam(P, R.view, I.view, eal, ecig)
    ok = got == exp
    rounds += 1
    total_bytes += len(exp)
    print("seed %d %s L=%d pairs %d pseudo %d thr %d frac %.2f nsam %d stages_done %d: %d SAM bytes %s" % (
        seed, kind, read_len, n_reads // 2, pseudo, P.score_threshold, P.score_fraction, P.num_sam_alignments, pst["stages_done"],
        len(exp), "ok" if ok else "DIFFERENT"), flush=True)
    if not ok:
        sys.exit(1)
    seed += 1
print("SOAK_E2E_OK rounds %d SAM bytes %d seeds %d..%d" % (rounds, total_bytes, seed0, seed - 1))
