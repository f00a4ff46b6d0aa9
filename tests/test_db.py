"""Database load (include/kslam_db.h): <db>/database, the
Boost.Serialization text archive of a GenbankIndex, to columns and back.

PARITY UNPINNED (no Boost in the image, no sample database in the reference): the product's
two-pass column parser is compared with oracle/db_oracle.py, an independently shaped plain-Python
restatement of the same published grammar, on seeded archives; plus round trips, the tolerated
variants, the error paths, and that the columns feed the host tail and kslam_set_index.  Host-only.
"""
import importlib
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def D(kslam):
    return importlib.import_module("kslam_amd.db")


@pytest.fixture(scope="module")
def ref():
    return importlib.import_module("oracle.db_oracle")


def make_entries(rng, n, max_len=400, genes=True):
    acgt = np.frombuffer(b"ACGTN", dtype=np.uint8)
    out = []
    for i in range(n):
        L = int(rng.integers(0, max_len))
        e = {"bases": acgt[rng.choice(5, L, p=[.248, .248, .248, .248, .008])].tobytes(),
             "taxonomyID": int(rng.integers(0, 2 ** 32)), "genbankID": int(rng.integers(0, 2 ** 32)),
             "isPlasmid": bool(rng.integers(0, 2)), "is16S": bool(rng.integers(0, 2)),
             "locusTag": b"NC_%06d.%d" % (i, rng.integers(1, 9)) if rng.random() < 0.9 else b"", "genes": []}
        if genes and rng.random() < 0.6:
            for _ in range(int(rng.integers(0, 5))):
                # free text with spaces, digits and punctuation: only the length prefix delimits it
                prod = b" ".join(rng.choice([b"30S ribosomal", b"protein", b"S1", b"2 4 0 0", b"", b"(EC 3.1.-.-)"],
                                            int(rng.integers(0, 4))))
                e["genes"].append({"geneName": rng.choice([b"rpsA", b"dnaA", b"", b"gene 7"]).item(),
                                   "locusTag": b"b%04d" % rng.integers(0, 9999), "proteinID": b"NP_%d.1" % rng.integers(1, 10 ** 6),
                                   "product": prod, "referenceSequence": rng.choice([b"", b"GeneID:945536"]).item(),
                                   "geneID": int(rng.integers(0, 2 ** 32)), "start": int(rng.integers(0, 2 ** 32)),
                                   "stop": int(rng.integers(0, 2 ** 32)), "complement": bool(rng.integers(0, 2))})
        out.append(e)
    return out


def as_entries(db):
    return [db.entry(i) for i in range(db.n_entries)]


@pytest.mark.parametrize("seed,n,genes", [(1, 0, True), (2, 1, False), (3, 1, True), (4, 40, True), (5, 200, False)])
def test_parse_equals_restatement(D, ref, seed, n, genes):
    rng = np.random.default_rng(seed)
    entries = make_entries(rng, n, genes=genes)
    text = ref.dump(entries, library_version=12 + seed)
    ver, exp = ref.parse(text)
    assert exp == entries and ver == 12 + seed          # the restatement round-trips itself
    db = D.Database.parse(text)
    assert db.library_version == 12 + seed and db.variant == 0
    assert as_entries(db) == entries
    assert db.entries() == [e["bases"] for e in entries]
    # CSR gene lists, int32 view of the CDS positions (getGene reads them as int, src/GenbankTools.h:170-185)
    assert list(db.gene_first) == list(np.cumsum([0] + [len(e["genes"]) for e in entries]))
    flat = [g for e in entries for g in e["genes"]]
    assert list(db.gene_start) == [int(np.uint32(g["start"]).view(np.int32)) for g in flat]
    db.close()


def test_write_round_trip_and_file_load(D, ref, tmp_path):
    rng = np.random.default_rng(11)
    entries = make_entries(rng, 60)
    path = tmp_path / "database"
    D.write(path, entries, library_version=17)
    raw = path.read_bytes()
    assert raw == ref.dump(entries, 17)                 # byte for byte what the restatement writes
    assert b"\n" not in raw                             # one line, as text_oarchive leaves it
    db = D.Database.load(path, threads=3)
    assert as_entries(db) == entries
    again = tmp_path / "database2"
    D.write(again, as_entries(db), library_version=17)
    assert again.read_bytes() == raw
    db.close()


def test_long_strings_are_copied_in_parallel(D, ref):
    rng = np.random.default_rng(5)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    entries = [{"bases": acgt[rng.integers(0, 4, L)].tobytes(), "taxonomyID": i, "genbankID": 0, "isPlasmid": False,
                "is16S": False, "locusTag": b"e%d" % i, "genes": []} for i, L in enumerate([30_000_000, 1, 45_000_000, 0])]
    db = D.Database.parse(ref.dump(entries), threads=4)
    assert db.entries() == [e["bases"] for e in entries]
    db.close()


def test_whitespace_and_grammar_variants(D, ref):
    rng = np.random.default_rng(21)
    entries = make_entries(rng, 12)
    text = ref.dump(entries)
    # line breaks / tabs between tokens (never inside a string: those are skipped by length)
    sig = b"22 serialization::archive"
    toks = text[len(sig) + 1:].split(b" ", 9)           # the nine numbers before the first string's length
    loose = sig + b"\n" + b"\n".join(toks[:4]) + b"\t " + b"  ".join(toks[4:9]) + b" \r\n" + toks[9] + b"\n"
    db = D.Database.parse(loose)
    assert as_entries(db) == entries and db.variant == 0
    db.close()
    # a writer without the class-info pair on the vector types / without item_version
    n = len(entries)

    def variant(no_vec_info, no_item_version):
        out, seen = [b"22 serialization::archive 9", b"0 0"], set()
        if not no_vec_info:
            out.append(b"0 0")
        out.append(b"%d" % n if no_item_version else b"%d 0" % n)
        for e in entries:
            if "e" not in seen:
                seen.add("e"); out.append(b"0 0")
            out += [b"%d %s" % (len(e["bases"]), e["bases"]), b"%d %d %d %d" % (e["taxonomyID"], e["genbankID"], e["isPlasmid"], e["is16S"]),
                    b"%d %s" % (len(e["locusTag"]), e["locusTag"])]
            if "vg" not in seen:
                seen.add("vg")
                if not no_vec_info:
                    out.append(b"0 0")
            out.append(b"%d" % len(e["genes"]) if no_item_version else b"%d 0" % len(e["genes"]))
            for g in e["genes"]:
                if "g" not in seen:
                    seen.add("g"); out.append(b"0 0")
                out += [b"%d %s" % (len(g[k]), g[k]) for k in ("geneName", "locusTag", "proteinID", "product", "referenceSequence")]
                out.append(b"%d" % g["geneID"])
                if "c" not in seen:
                    seen.add("c"); out.append(b"0 0")
                out.append(b"%d %d %d" % (g["start"], g["stop"], g["complement"]))
        return b" ".join(out)
    for nv, ni in ((True, False), (False, True), (True, True)):
        db = D.Database.parse(variant(nv, ni))
        assert as_entries(db) == entries
        assert db.variant == (1 if nv else 0) | (2 if ni else 0)
        db.close()


@pytest.mark.parametrize("case", ["empty", "signature", "truncated string", "truncated entry", "trailing", "bad flag",
                                  "count too large", "letters in a number", "missing file"])
def test_errors_are_reported_with_a_position(D, ref, kslam, case, tmp_path):
    rng = np.random.default_rng(31)
    entries = make_entries(rng, 5)
    text = ref.dump(entries)
    if case == "missing file":
        with pytest.raises(kslam.KslamError, match="Unable to open file"):
            D.Database.load(tmp_path / "nope")
        return
    bad = {"empty": b"", "signature": text.replace(b"serialization", b"serialisation", 1),
           "truncated string": text[:len(text) // 2], "truncated entry": text[:text.rindex(b" ")],
           "trailing": text + b" 7", "bad flag": None, "count too large": text.replace(b" %d 0 0 0 " % len(entries), b" 99999999999 0 0 0 ", 1),
           "letters in a number": text.replace(b" 0 0 0 0 5 ", b" 0 0 0 0 x5 ", 1)}[case]
    if case == "bad flag":
        e = [dict(x) for x in entries]
        t = ref.dump(e)
        # isPlasmid token of the first entry -> 2
        head = b"%d %s %d %d " % (len(e[0]["bases"]), e[0]["bases"], e[0]["taxonomyID"], e[0]["genbankID"])
        at = t.index(head) + len(head)
        bad = t[:at] + b"2" + t[at + 1:]
    with pytest.raises(kslam.KslamError, match=r"database archive: .*\(byte \d+\)"):
        D.Database.parse(bad)


def test_columns_feed_the_tail_and_the_index(D, ref, kslam):
    """The parsed columns ARE the tail's kslam_index_view and the arguments of kslam_set_index."""
    T = importlib.import_module("kslam_amd.tail")
    rng = np.random.default_rng(41)
    entries = make_entries(rng, 6, max_len=300, genes=False)
    db = D.Database.parse(ref.dump(entries))
    hdr = T.sam_header(db, b"SLAM --db d r1 r2")
    for e in entries:
        assert b"@SQ\tSN:%s\tLN:%d" % (e["locusTag"], len(e["bases"])) in hdr
    ptrs, lens = db.entry_pointers()
    import ctypes as C
    got = [C.string_at(C.cast(ptrs, C.POINTER(C.c_void_p))[i], C.cast(lens, C.POINTER(C.c_uint64))[i]) for i in range(len(entries))]
    assert got == [e["bases"] for e in entries]
    db.close()
