"""Taxonomy stage (include/kslam_taxonomy.h; the per-read LCA of SURVEY.md section 8f row N1).

Tree queries are compared three ways -- product (dense parent/depth arrays), restatement
(oracle/taxonomy_oracle.cpp: hash map + root-ward paths) and the REAL reference
(oracle/_ref/libtaxonomy_ref.so = src/TaxonomyDatabase.h compiled in place), plus the answers the
reference gave for tests/golden/taxonomy_cases.json.  The per-read and summary steps
(src/MetagenomicResults.h, unbuildable here) are compared product vs restatement and against
hand-worked cases.  Host-only.
"""
import importlib
import json
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RANKS = [b"no rank", b"superkingdom", b"phylum", b"class", b"order", b"family", b"genus", b"species",
         b"subspecies", b"strain", b""]


@pytest.fixture(scope="module")
def X(kslam):
    return importlib.import_module("kslam_amd.taxonomy")


def make_tree(rng, n, dangling=3, duplicates=2):
    """-> (taxDB text, list of ids).  Node 1 is the root (its own parent); 131567 hangs below it."""
    ids = [1, 131567] + [int(x) for x in rng.choice(np.arange(2, 200000), n, replace=False) if x != 131567]
    parent = {1: 1, 131567: 1}
    for k, i in enumerate(ids[2:], start=2):
        parent[i] = ids[int(rng.integers(0, k))] if rng.random() < 0.9 else 1
    recs = []
    for i in ids:
        recs.append((i, parent[i], b"name of %d" % i if rng.random() < 0.95 else b"",
                     RANKS[int(rng.integers(0, len(RANKS)))]))
    for _ in range(dangling):       # a parent id that no record defines
        recs.append((int(rng.integers(300000, 400000)), int(rng.integers(400000, 500000)), b"orphan", b"genus"))
    order = rng.permutation(len(recs))
    recs = [recs[j] for j in order]
    for _ in range(duplicates):     # a repeated id: the first record wins
        i, p, nm, rk = recs[int(rng.integers(0, len(recs)))]
        recs.append((i, 1, b"DUPLICATE", b"species"))
    text = b"".join(b"%d\n%d\n%s\n%s\n" % r for r in recs)
    return text, [r[0] for r in recs] + [r[1] for r in recs]


def query_sets(rng, ids, n_sets):
    out = []
    for _ in range(n_sets):
        k = int(rng.integers(1, 6))
        s = [int(ids[int(rng.integers(0, len(ids)))]) for _ in range(k)]
        r = rng.random()
        if r < 0.08:
            s.append(0)
        elif r < 0.16:
            s.append(999999)            # an id the tree has never heard of
        elif r < 0.2:
            s = [999999, 999999]
        out.append(s)
    return out + [[], [1], [1, 1], [131567]]


@pytest.mark.parametrize("seed", range(6))
def test_tree_queries_three_ways(X, oracle, tmp_path, seed):
    rng = np.random.default_rng(seed)
    text, ids = make_tree(rng, 300)
    db, orc = X.TaxDB(text), oracle.taxonomy_tree(text)
    trees = [orc]
    if oracle.have_ref_taxonomy():
        path = str(tmp_path / "taxDB")
        open(path, "wb").write(text)
        trees.append(oracle.ref_taxonomy_tree(path))
    for t in trees:
        for s in query_sets(rng, ids, 400):
            assert db.lca(s) == t.lca(s), s
        for i in list(dict.fromkeys(ids))[:200] + [0, 1, 999999, 131567]:
            assert db.parent(i) == t.parent(i), i
            for which in (X.TaxDB.NAME, X.TaxDB.RANK, X.TaxDB.LINEAGE):
                assert db.text(i, which) == t.text(i, which), (i, which)
            assert db.is_subspecies(i) == t.is_subspecies(i), i
            for rank in (b"species", b"genus", b"nothing"):
                assert db.at_rank(i, rank) == t.at_rank(i, rank), (i, rank)
            j = int(ids[int(rng.integers(0, len(ids)))])
            assert db.is_below(j, i) == t.is_below(j, i) and db.is_below(i, i) == t.is_below(i, i)
    for t in trees:
        t.close()


def test_lca_conventions_by_hand(X):
    # 1 root; 2 and 3 top-level; 2 -> 20 -> 200, 201;  3 -> 30
    text = b"".join(b"%d\n%d\n%s\n%s\n" % r for r in [
        (1, 1, b"root", b"no rank"), (2, 1, b"Bacteria", b"superkingdom"), (3, 1, b"Viruses", b"superkingdom"),
        (20, 2, b"Escherichia", b"genus"), (200, 20, b"Escherichia coli", b"species"),
        (201, 20, b"Escherichia albertii", b"species"), (2000, 200, b"E. coli K-12", b"no rank"),
        (30, 3, b"Some virus", b"species")])
    db = X.TaxDB(text)
    assert len(db) == 8
    assert db.lca([200, 201]) == 20 and db.lca([2000, 201]) == 20 and db.lca([2000, 200]) == 200
    assert db.lca([200]) == 200 and db.lca([200, 200, 200]) == 200
    assert db.lca([200, 30]) == 0          # different top-level nodes: paths stop below the root
    assert db.lca([200, 0]) == 0 and db.lca([]) == 0 and db.lca([77]) == 77 and db.lca([77, 200]) == 0
    assert db.parent(2) == 0 and db.parent(20) == 2 and db.parent(77) == 0
    assert db.is_subspecies(2000) == 1 and db.is_subspecies(200) == 0 and db.is_subspecies(20) == 0
    assert db.at_rank(2000, b"species") == 200 and db.at_rank(2000, b"genus") == 20
    assert db.at_rank(2000, b"superkingdom") == 0   # the walk stops at a node whose parent is 1
    assert db.text(2000, X.TaxDB.LINEAGE) == b"Bacteria; Escherichia."   # everything from species down is cut
    assert db.text(20, X.TaxDB.LINEAGE) == b"Bacteria; Escherichia."
    assert db.text(77, X.TaxDB.NAME) == b""


def test_parse_errors(kslam, X):
    with pytest.raises(kslam.KslamError) as e:
        X.TaxDB(b"5\n1\nname\n")
    assert e.value.status == 1 and "multiple of four" in str(e.value)
    with pytest.raises(kslam.KslamError) as e:
        X.TaxDB(b"five\n1\nname\nrank\n")
    assert e.value.status == 1 and "not a number" in str(e.value)
    with pytest.raises(kslam.KslamError) as e:
        X.TaxDB(b"5\n6\na\nr\n6\n5\nb\nr\n")
    assert e.value.status == 1 and "cycle" in str(e.value)
    assert len(X.TaxDB(b"")) == 0


def test_summary_and_its_first_record_quirk(X, oracle):
    text = b"".join(b"%d\n%d\n%s\n%s\n" % r for r in [
        (1, 1, b"root", b""), (2, 1, b"Bacteria", b""), (20, 2, b"Escherichia", b"genus"),
        (200, 20, b"Escherichia coli", b"species")])
    db, orc = X.TaxDB(text), oracle.taxonomy_tree(text)
    # with unclassified pairs (id 0) present every group is complete
    ids = [200, 0, 20, 200, 200, 0, 20]
    assert db.summary(ids, 10) == b"Escherichia coli\t30\nEscherichia\t20\n"
    # without any id 0 the lowest id loses its first record (src/MetagenomicResults.h:159-175)
    ids = [200, 20, 200, 200, 20]
    assert db.summary(ids, 10) == b"Escherichia coli\t30\nEscherichia\t10\n"
    assert db.summary([200], 3) == b"Escherichia coli\t33.3333\n" and db.summary([], 3) == b""
    rng = np.random.default_rng(3)
    for trial in range(50):
        ids = rng.choice([0, 2, 20, 200, 999], int(rng.integers(0, 40)),
                         p=[0.1, 0.2, 0.3, 0.3, 0.1] if trial % 2 else [0.0, 0.3, 0.3, 0.3, 0.1])
        assert db.summary(ids, 1000) == oracle.taxonomy_summary(orc, ids, 1000)


def test_classify_read_pairs(kslam, X, oracle):
    T = importlib.import_module("kslam_amd.tail")
    from test_tail import _fuzz_overlaps
    rng = np.random.default_rng(11)
    text, ids = make_tree(rng, 60, dangling=0, duplicates=0)
    db, orc = X.TaxDB(text), oracle.taxonomy_tree(text)
    n_entries = 12
    entry_tax = [int(ids[int(rng.integers(0, 60))]) for _ in range(n_entries)]
    entry_tax[3] = 0                                     # an entry without a taxonomy id
    ov, n_reads = _fuzz_overlaps(kslam, rng, 3000, n_entries)
    reads = T.Reads([b"A" * 100] * n_reads, ids=[b"q%d" % (i % 3000) for i in range(n_reads)])
    index = T.Index([b"A" * 8000] * n_entries, taxonomy_ids=entry_tax)
    P = T.TailParams.default(report_cigar=False, threads=3)
    rp, pr, _ = T.tail_pairs(P, reads, ov)
    got, per_read = db.classify(P, reads, index, rp, pr)
    exp = [orc.lca([entry_tax[int(e)] for e in pr["entry"][int(g["first"]):int(g["first"]) + int(g["count"])]])
           for g in rp]
    assert got.tolist() == exp and len(set(exp)) > 5
    assert per_read == b"".join(b"q%d\t%d\n" % (int(g["r1_read"]), t) for g, t in zip(rp, exp))
    assert db.classify(P, reads, index, rp, pr, per_read=False)[1] is None


def test_golden_answers_from_the_real_reference(X, oracle):
    g = json.load(open(os.path.join(ROOT, "tests", "golden", "taxonomy_cases.json")))
    text = g["taxdb"].encode()
    db, orc = X.TaxDB(text), oracle.taxonomy_tree(text)
    assert len(g["lca"]) >= 200
    for ids, ans in g["lca"]:
        assert db.lca(ids) == ans and orc.lca(ids) == ans, ids
    for i, parent, name, rank, lineage, sub, species in g["nodes"]:
        assert (db.parent(i), db.is_subspecies(i), db.at_rank(i, b"species")) == (parent, sub, species)
        assert [db.text(i, w).decode() for w in (0, 1, 2)] == [name, rank, lineage]
        assert orc.text(i, 2).decode() == lineage


# ---- the XML report (writeResults, src/MetagenomicResults.h:213-224): a plain restatement of the reference's
# steps for small cases -- getGene (src/GenbankTools.h:170-185), geneSort / operator== (:84-91, 116-125),
# getResultFromPairedOverlaps (:88-112), combineTaxonomies + combineRangeOfIdentifiedTaxonomy (:118-177, records
# of equal id in input order), sortResults (:254-273), correctXML / getXML (:275-366).  Python's sort is stable
# where std::sort is not, so the exact comparison uses data without equal-but-different genes; the tie-heavy case
# compares what does not depend on the representative.
def _std_sort_front(keys):
    """index of the record libstdc++'s std::sort (by key alone) leaves at position 0: tests/std_sort_front.cpp, compiled once"""
    import subprocess
    import tempfile
    here = os.path.dirname(os.path.abspath(__file__))
    exe = os.path.join(tempfile.gettempdir(), "kslam_std_sort_front_%d" % os.getuid())
    if not os.path.exists(exe) or os.path.getmtime(exe) < os.path.getmtime(os.path.join(here, "std_sort_front.cpp")):
        subprocess.check_call(["g++", "-O2", "-std=c++11", "-o", exe, os.path.join(here, "std_sort_front.cpp")])
    with tempfile.NamedTemporaryFile(suffix=".u32") as f:
        f.write(np.asarray(keys, dtype="<u4").tobytes())
        f.flush()
        return int(subprocess.check_output([exe, f.name]).strip())


def _xml_restatement(db, entry_tax, genes, extras, batches, num_reads):
    import functools

    def less(a, b):
        if not a["protein"] and not b["protein"]:
            return a["name"] < b["name"]
        if a["protein"] == b["protein"]:
            return a["product"] < b["product"]
        return a["protein"] < b["protein"]

    def equal(a, b):
        if not a["protein"] and not b["protein"]:
            return a["name"] == b["name"]
        if a["protein"] == b["protein"]:
            return a["product"] == b["product"]
        return False
    key = functools.cmp_to_key(lambda a, b: -1 if less(a, b) else (1 if less(b, a) else 0))

    def i32(v):
        return v - (1 << 32) if v >= (1 << 31) else v
    recs = []
    for ids, rp, pr in batches:
        for g in rp:
            rec = {"tax": 0, "read": None, "genes": []}
            recs.append(rec)
            if int(g["count"]) == 0:
                continue
            taxs = []
            for p in pr[int(g["first"]):int(g["first"]) + int(g["count"])]:
                e = int(p["entry"])
                taxs.append(entry_tax[e])
                best, largest = None, 0
                for gene in genes[e]:
                    shared = min(int(p["ref_end"]), i32(gene["stop"])) - max(int(p["ref_start"]), i32(gene["start"]))
                    if shared > largest:
                        best, largest = gene, shared
                if best is not None:
                    rec["genes"].append(best)
            rec["genes"].sort(key=key)
            uniq = []
            for x in rec["genes"]:
                if not uniq or not equal(uniq[-1], x):
                    uniq.append(x)
            rec["genes"] = uniq
            rec["read"] = ids[int(g["r1_read"])]
            rec["tax"] = db.lca(taxs)
    order = sorted(range(len(recs)), key=lambda i: recs[i]["tax"])
    if order and recs[order[0]]["tax"] != 0:
        # no unclassified record: the reference's grouping loop leaves the FRONT record of its sorted vector out of its group
        # (src/MetagenomicResults.h:159-175), and its sort is std::sort when one thread runs it -- ask the real one
        front = _std_sort_front([r["tax"] for r in recs])
        at = order.index(front)
        assert recs[front]["tax"] == recs[order[0]]["tax"]
        order[0], order[at] = order[at], order[0]
    taxa = []

    def combine(a, b):
        allg, reads = [], []
        for i in order[a:b]:
            allg += recs[i]["genes"]
            if recs[i]["read"] is not None:
                reads.append(recs[i]["read"])
        allg.sort(key=key)
        merged = []
        for x in allg:
            if merged and equal(merged[-1][0], x):
                merged[-1][1] += 1
            else:
                merged.append([x, 1])
        taxa.append({"tax": recs[order[a]]["tax"], "reads": reads, "genes": merged})
    if recs:
        test, start = 0, 0
        for i in range(1, len(recs)):
            t = recs[order[i]]["tax"]
            if t != test:
                if test != 0:
                    combine(start, i)
                test, start = t, i
        if recs[order[start]]["tax"] != 0:
            combine(start, len(recs))
    taxa.sort(key=lambda t: (-len(t["reads"]), t["tax"]))

    def esc(b):
        return (b.replace(b"&", b"\0").replace(b"<", b"&lt;").replace(b">", b"&gt;").replace(b"'", b"&apos;")
                .replace(b'"', b"&quot;").replace(b"\0", b"&amp;"))
    out = []
    for t in taxa:
        t["reads"].sort()
        t["genes"].sort(key=lambda gc: (-gc[1], gc[0]["start"], gc[0]["locus"]))
        out.append(b'<taxon>\n  <abundance numReads="%d">%s</abundance>\n  <taxonomyID>%d</taxonomyID>\n  <lineage>%s</lineage>\n'
                   b'  <name>%s</name>\n  <genes>\n' % (len(t["reads"]), ("%f" % (len(t["reads"]) * 100.0 / num_reads)).encode(), t["tax"],
                                                       esc(db.text(t["tax"], 2)), esc(db.text(t["tax"], 0))))
        for gene, count in t["genes"]:
            out.append(b'    <gene protein="%s" locus="%s" product="%s" GeneID="%d" reference="%s" numReads="%d" cdsStart="%d" '
                       b'cdsEnd="%d">%s</gene>\n' % (esc(gene["protein"]), esc(gene["locus"]), esc(gene["product"]), gene["id"],
                                                     esc(gene["reference"]), count, gene["start"], gene["stop"], esc(gene["name"])))
        out.append(b"  </genes>\n  <reads>\n")
        for r in t["reads"]:
            out.append(b"    <read>%s</read>\n" % esc(r))
        out.append(b"  </reads>\n</taxon>\n")
    return b"".join(out), taxa


@pytest.mark.parametrize("seed,distinct", [(1, True), (2, True), (3, False), (4, False)])
def test_xml_report(kslam, X, seed, distinct):
    T = importlib.import_module("kslam_amd.tail")
    from test_tail import _fuzz_overlaps
    rng = np.random.default_rng(40 + seed)
    text, ids = make_tree(rng, 40, dangling=0, duplicates=0)
    db = X.TaxDB(text)
    n_entries = 10
    entry_tax = [int(ids[int(rng.integers(0, 40))]) for _ in range(n_entries)]
    if seed % 2:
        entry_tax[2] = 0                   # reads on this entry stay unclassified: no record is dropped by the quirk
    genes, flat = [], []
    names = [b"dnaA", b"rpoB", b"gyr<A>", b"", b"x&y", b"recA'", b'say"hi"']
    for e in range(n_entries):
        gl = []
        for k in range(int(rng.integers(0, 9))):
            start = int(rng.integers(0, 6000))
            stop = start + int(rng.integers(50, 1500))
            if distinct:                   # every gene its own class: the representative is not a question
                protein = b"WP_%d_%d" % (e, k) if rng.random() < 0.8 else b""
                name = b"g%d_%d" % (e, k)
            else:                          # classes shared between entries (strains): same protein and product, other fields differ
                protein = [b"WP_1", b"WP_2", b"WP_3", b""][int(rng.integers(0, 4))]
                name = names[int(rng.integers(0, len(names)))]
            product = [b"kinase", b"hypothetical protein", b"a < b"][int(rng.integers(0, 3))]
            g = {"start": start, "stop": stop, "name": name, "protein": protein, "product": product,
                 "locus": b"LT_%d_%d" % (e, k), "reference": b"NC_%06d" % e, "id": int(rng.integers(0, 1 << 31))}
            gl.append(g)
        if e == 5 and gl:
            gl[0]["start"], gl[0]["stop"] = 4000000000, 4000000500      # CDS fields are uint32; getGene compares them as int
        genes.append(gl)
        flat += gl
    index = T.Index([b"A" * 8000] * n_entries, taxonomy_ids=entry_tax,
                    genes=[[(g["start"], g["stop"], g["name"], g["protein"], g["product"]) for g in gl] for gl in genes])
    extras = X.GeneExtras.from_lists([g["locus"] for g in flat], [g["reference"] for g in flat], [g["id"] for g in flat]) if flat else None
    P = T.TailParams.default(report_cigar=False, threads=3)
    rep = X.Report()
    batches, n_pairs_total = [], 0
    for b in range(2):
        ov, n_reads = _fuzz_overlaps(kslam, rng, 400, n_entries, per_read=6.0 if not distinct else 3.0)
        rid = [b"r%d<&>%d" % (b, i % 400) for i in range(n_reads)]
        reads = T.Reads([b"A" * 100] * n_reads, ids=rid)
        rp, pr, _ = T.tail_pairs(P, reads, ov)
        tax, _ = db.classify(P, reads, index, rp, pr, per_read=False)
        rep.add_batch(reads, index, rp, pr, tax)
        batches.append((rid, rp, pr))
        n_pairs_total += 400
    got = db.report_xml(rep, index, extras, n_pairs_total)
    exp, taxa = _xml_restatement(db, entry_tax, genes, extras, batches, n_pairs_total)
    assert len(taxa) >= 3 and sum(len(t["genes"]) for t in taxa) > 5 and b"&lt;&amp;&gt;" in got
    if distinct:
        assert got == exp
    else:
        # equal-but-different genes: which one represents its class is libstdc++'s sort; everything else must agree
        import re
        strip = lambda x: re.sub(rb' locus="[^"]*"| GeneID="[^"]*"| reference="[^"]*"| cdsStart="[^"]*"| cdsEnd="[^"]*"|>[^<]*</gene>', b"", x)
        def canon(x):
            blocks = x.split(b"<taxon>\n")
            return [sorted(strip(l) for l in blk.split(b"\n")) for blk in blocks]
        assert canon(got) == canon(exp)
        assert max(c for t in taxa for _, c in t["genes"]) >= 3
    assert db.report_xml(X.Report(), index, extras, 10) == b""
    rep.close()


def test_xml_report_with_one_dominant_taxon(kslam, X):
    """A run dominated by one organism: its group of records is larger than one task's share, so its read names are
    sorted by all threads (chunks, then merges) before the taxon's text is written; the other taxa go through the
    parallel loop.  Names repeat and carry XML specials; the text must be the restatement's."""
    T = importlib.import_module("kslam_amd.tail")
    rng = np.random.default_rng(77)
    text, ids = make_tree(rng, 30, dangling=0, duplicates=0)
    db = X.TaxDB(text)
    n_entries = 6
    entry_tax = [int(ids[k]) for k in rng.choice(30, n_entries, replace=False)]
    genes = [[{"start": 500 * k, "stop": 500 * k + 450, "name": b"g%d_%d" % (e, k), "protein": b"WP_%d_%d" % (e, k),
               "product": b"p<%d>" % k, "locus": b"LT_%d_%d" % (e, k), "reference": b"NC_%06d" % e, "id": 10 * e + k}
              for k in range(4)] for e in range(n_entries)]
    flat = [g for gl in genes for g in gl]
    index = T.Index([b"A" * 4000] * n_entries, taxonomy_ids=entry_tax,
                    genes=[[(g["start"], g["stop"], g["name"], g["protein"], g["product"]) for g in gl] for gl in genes])
    extras = X.GeneExtras.from_lists([g["locus"] for g in flat], [g["reference"] for g in flat], [g["id"] for g in flat])
    rep, batches, n_total = X.Report(), [], 0
    for b in range(2):
        n = 160_000
        rid = [b"r<%d>&%06d" % (b, int(v)) for v in rng.integers(0, 100_000, n)]
        reads = T.Reads([b"A"] * (2 * n), ids=rid + rid)
        rp = np.zeros(n, dtype=T.READ_PAIR_DT)
        rp["r1_read"], rp["r2_read"], rp["first"], rp["count"] = np.arange(n), np.arange(n) + n, np.arange(n), 1
        pr = np.zeros(n, dtype=T.PAIRED_OVERLAP_DT)
        e = np.where(rng.random(n) < 0.92, 0, rng.integers(1, n_entries, n))
        pr["entry"], pr["ref_start"] = e, rng.integers(0, 1900, n)
        pr["ref_end"] = pr["ref_start"] + 120
        tax = np.asarray(entry_tax, dtype=np.uint32)[e]
        rep.add_batch(reads, index, rp, pr, tax)
        batches.append((rid + rid, rp, pr))
        n_total += n
    got = db.report_xml(rep, index, extras, n_total)
    exp, taxa = _xml_restatement(db, entry_tax, genes, extras, batches, n_total)
    assert len(taxa[0]["reads"]) > 262144 and len(taxa) == n_entries
    assert got == exp
    rep.close()
