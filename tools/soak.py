"""Randomised soak of the hot path against the oracle: python tools/soak.py [n_rounds] [first_seed]
Each round draws a data set shape (read length, divergence, indel rate, low-complexity or plain
genomes, scoring) from the seed, aligns it on the GPU and with the CPU oracle and compares every
field and CIGAR op.  Stops at the first difference and prints how to reproduce it."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle as O                      # noqa: E402  (test infrastructure: this tool is a checker)
from conftest import load_kslam        # noqa: E402
import test_gpu_parity as T            # noqa: E402
K = load_kslam()
synth = importlib.import_module("kslam_amd.synth")
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
SCORINGS = [None, None, (1, 4, 6, 1), (3, 2, 4, 1), (2, 5, 5, 2), (4, 4, 6, 2), (2, 2, 3, 1)]
total = 0
t0 = time.time()
for r in range(rounds):
    seed = seed0 + r
    rng = np.random.default_rng(seed)
    read_len = int(rng.choice([60, 100, 150, 150, 200, 250, 320]))
    scoring = SCORINGS[int(rng.integers(0, len(SCORINGS)))]
    if rng.random() < 0.5:
        reads, genomes = T._low_complexity_dataset(synth, seed, int(rng.integers(2, 7)), 2000, read_len)
        kind = "low-complexity"
    else:
        g = synth.make_genomes(seed, int(rng.integers(1, 4)), int(rng.integers(1, 5)), int(rng.integers(5000, 40000)),
                               strain_sub=float(rng.uniform(0.0, 0.06)), strain_indel=float(rng.uniform(0, 0.004)),
                               shared_segment=int(rng.choice([0, 800])))
        rd, _ = synth.make_paired_reads(seed + 1, g, 1200, read_len=read_len, frag_mean=2 * read_len + 40,
                                        sub_rate=float(rng.uniform(0, 0.05)), indel_rate=float(rng.uniform(0, 0.012)),
                                        n_rate=float(rng.choice([0, 0.003])), edge_frac=float(rng.uniform(0, 0.3)))
        if rng.random() < 0.2:     # round 3: a few reads beyond the packed kernels' 511 bases (runs of their own: k_sw_long)
            for k in rng.integers(0, len(rd), 25):
                gg = g[int(rng.integers(0, len(g)))]
                L = int(rng.integers(520, min(1800, len(gg) - 10)))
                at = int(rng.integers(0, len(gg) - L))
                frag = gg[at:at + L]
                if rng.random() < 0.5:
                    frag = synth.revcomp(frag)
                rd[int(k)] = synth.mutate(rng, frag, 0.02, 0.004)
        reads, genomes = synth.to_bytes(rd), synth.to_bytes(g)
        kind = "plain"
    kw = dict(zip(("match", "mismatch", "gap_open", "gap_extend"), scoring)) if scoring else {}
    got, gcig = K.align_to_database(reads, genomes, **kw)
    exp, ecig, _ = O.align_to_database(reads, genomes, O.Params.default(**kw))
    try:
        T._compare_alignments(got, gcig, exp, ecig)
    except AssertionError as e:
        print("MISMATCH seed %d (%s, read_len %d, scoring %s): %s" % (seed, kind, read_len, scoring, str(e)[:300]))
        sys.exit(1)
    total += len(exp)
    print("round %d seed %d %s L=%d scoring=%s: %d alignments, %d gapped cigars ok" % (
        r, seed, kind, read_len, scoring, len(exp), int((exp["cigar_len"] > 1).sum())), flush=True)
print("soak: %d rounds, %d alignments identical, %.0f s" % (rounds, total, time.time() - t0))
