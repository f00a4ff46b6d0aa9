"""FASTQ ingest (include/kslam_fastq.h).

Three readers are compared record by record:
  product      k-slam_amd/host/fastq.cpp  (parallel line index -> read columns)
  restatement  oracle/fastq_oracle.cpp     (serial getline loop over a buffer)
  reference    oracle/_ref/libfastq_ref.so -- the reference's own src/FASTQsequence.h compiled in
               place (it needs no Boost), when that library is present
plus tests/golden/fastq_cases.json: inputs and the records the REAL reference returned for them
(tests/golden/make_golden.py wrote it), so the pin holds where the reference library is absent.
Host-only: nothing here needs a GPU.
"""
import base64
import importlib
import json
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def F(kslam):
    return importlib.import_module("kslam_amd.fastq")


HEADERS = [b"@r%d", b"@r%d desc text", b"@r%d/1", b"@r%d/2 x/y", b"@ r%d", b"@", b"", b"x", b"@r%d  two",
           b">r%d", b"@a/b/c%d", b"@/r%d", b"@r%d\tTAB/9"]
EOLS = [b"\n", b"\r\n", b"\r"]


def make_text(rng, n_records, eol_mix=True, truncate=False, blank_tail=0):
    out = []
    for k in range(n_records):
        h = HEADERS[int(rng.integers(0, len(HEADERS)))]
        h = h % k if b"%d" in h else h
        L = int(rng.choice([0, 1, 5, 36, 150]))
        seq = bytes(rng.choice(np.frombuffer(b"ACGTNacgt", dtype=np.uint8), L))
        qual = bytes(rng.integers(33, 75, L, dtype=np.uint8)) if rng.random() < 0.9 else b""
        plus = b"+" if rng.random() < 0.8 else b"+" + h[1:]
        for line in (h, seq, plus, qual):
            eol = EOLS[int(rng.integers(0, 3))] if eol_mix else b"\n"
            out.append(line + eol)
    text = b"".join(out) + b"\n" * blank_tail
    if truncate and len(text) > 4:
        text = text[:int(rng.integers(len(text) // 2, len(text)))]
    return text


def records(batch):
    return list(zip(batch.ids, batch.bases, batch.quality))


def oracle_records(oracle, text, **kw):
    b, q, i, pos = oracle.fastq_read(text, **kw)
    return list(zip(i, b, q)), pos


@pytest.mark.parametrize("seed", range(12))
def test_whole_stream_matches_restatement_and_reference(F, oracle, tmp_path, seed):
    rng = np.random.default_rng(seed)
    text = make_text(rng, int(rng.integers(0, 400)), truncate=seed % 3 == 1, blank_tail=seed % 4)
    if seed == 7:
        text = text.rstrip(b"\r\n")            # no terminator after the last line
    if seed == 9:
        text = text.rstrip(b"\r\n") + b"\r"     # a lone CR ends the stream
    batch, used = F.parse(text, threads=3)
    exp, pos = oracle_records(oracle, text)
    assert records(batch) == exp
    if exp:
        assert used <= len(text)
    if oracle.have_ref_fastq():
        path = str(tmp_path / "x.fq")
        open(path, "wb").write(text)
        b, q, i, _ = oracle.ref_fastq_read(path)
        assert list(zip(i, b, q)) == exp, "restatement differs from the real reference"
    one, used1 = F.parse(text, threads=1)
    assert records(one) == exp and used1 == used


def test_edge_streams(F, oracle):
    cases = [b"", b"\n", b"@a\nAC\n+\n", b"@a\nAC\n+", b"@a\nAC\n+\nII", b"@a\nAC\n+\nII\n", b"\r\r\r\r",
             b"@a\r\nAC\r\n+\r\nII\r\n@b", b"@a b/1\nA\n+\nI\n\n\n\n\n", b"@\nA\n+\nI\n", b"@a\n\n+\n\n"]
    for text in cases:
        batch, used = F.parse(text)
        exp, _ = oracle_records(oracle, text)
        assert records(batch) == exp, text
    # the missing quality line is completed by the empty line read at end of stream
    batch, _ = F.parse(b"@a\nAC\n+\n")
    assert records(batch) == [(b"a", b"AC", b"")]


@pytest.mark.parametrize("per_call", [1, 7, 64])
def test_streaming_in_calls_of_n_reads(F, oracle, tmp_path, per_call):
    """The low-memory loop (reference src/SLAM.h:193-207): N reads per call from one stream."""
    rng = np.random.default_rng(100 + per_call)
    text = make_text(rng, 150)
    pos, opos, got, exp, sizes = 0, 0, [], [], []
    while True:
        batch, used = F.parse(text[pos:], max_reads=per_call)
        e, opos2 = oracle_records(oracle, text, pos=opos, max_reads=per_call)
        assert records(batch) == e
        if not e:
            break
        assert pos + used == opos2       # the product leaves the stream where the reference would
        pos, opos = pos + used, opos2
        got += records(batch)
        sizes.append(len(e))
    whole, _ = F.parse(text)
    assert got == records(whole) and len(got) > 100   # (CR + LF of adjacent random lines may merge)
    if oracle.have_ref_fastq():
        path = str(tmp_path / "s.fq")
        open(path, "wb").write(text)
        b, q, i, calls = oracle.ref_fastq_read(path, per_call)
        assert list(zip(i, b, q)) == got and calls == sizes


def test_prefix_of_a_longer_stream(F, oracle):
    """at_eof = False: only records whose four lines are terminated are taken, and parsing the
    remainder later gives the same batch as parsing everything at once."""
    rng = np.random.default_rng(5)
    text = make_text(rng, 120)
    whole, _ = F.parse(text)
    for cut in [0, 1, 17, len(text) // 3, len(text) // 2, len(text) - 1, len(text)]:
        head, used = F.parse(text[:cut], at_eof=False)
        assert used <= cut
        rest, _ = F.parse(text[used:], at_eof=True)
        assert records(head) + records(rest) == records(whole), cut


def test_pairs_layout_and_mismatch(kslam, F, oracle):
    rng = np.random.default_rng(9)
    t1, t2 = make_text(rng, 80, eol_mix=False), make_text(rng, 80, eol_mix=False)
    batch, u1, u2 = F.parse_pair(t1, t2)
    e1, _ = oracle_records(oracle, t1)
    e2, _ = oracle_records(oracle, t2)
    assert batch.n_reads == 160 and records(batch) == e1 + e2     # [R1 block | R2 block]
    cat, off = batch.bases_array()
    assert bytes(cat[int(off[80]):int(off[81])]) == e2[0][1]
    few, u1, u2 = F.parse_pair(t1, t2, max_pairs=10)
    assert records(few) == e1[:10] + e2[:10]
    with pytest.raises(kslam.KslamError) as e:
        F.parse_pair(t1, make_text(rng, 79, eol_mix=False))
    assert e.value.status == 1 and "mismatch in R1 and R2 size" in str(e.value)


@pytest.mark.parametrize("seed,eol_mix,max_pairs", [(1, True, 0), (2, False, 0), (3, True, 37), (4, True, 0)])
def test_index_only_entry_describes_the_same_batch(kslam, F, seed, eol_mix, max_pairs):
    """kslam_fastq_index_pair (no bases / quality columns: their places in [r1 | r2] instead) against the
    full parser: same identifiers and offsets, and the text cut at the layout's positions IS the columns."""
    import ctypes as C
    rng = np.random.default_rng(500 + seed)

    def text(n, ragged):
        out = []
        for k in range(n):
            h = HEADERS[int(rng.integers(0, len(HEADERS)))]
            h = h % k if b"%d" in h else h
            L = int(rng.choice([0, 1, 5, 36, 150]))
            seq = bytes(rng.choice(np.frombuffer(b"ACGTNacgt", dtype=np.uint8), L))
            qual = bytes(rng.integers(33, 75, L + (1 if ragged and k == n // 2 else 0), dtype=np.uint8))
            for line in (h, seq, b"+", qual):
                # (no lone "\r" here: before an empty line it would fuse with that line's "\n" into one terminator)
                out.append(line + (EOLS[int(rng.integers(0, 2))] if eol_mix else b"\n"))
        return b"".join(out)
    t1, t2 = text(120, False), text(120, seed == 4)
    b1, b2 = C.create_string_buffer(t1, len(t1)), C.create_string_buffer(t2, len(t2))
    if seed == 4:
        with pytest.raises(kslam.KslamError, match="quality line"):
            F.index_pair(C.addressof(b1), len(t1), C.addressof(b2), len(t2))
        return
    full, f1, f2 = F.parse_pair(t1, t2, max_pairs=max_pairs)
    ix, u1, u2 = F.index_pair(C.addressof(b1), len(t1), C.addressof(b2), len(t2), max_pairs=max_pairs)
    assert (u1, u2) == (f1, f2) and ix.n_reads == full.n_reads
    n = ix.n_reads
    assert ix.ids == full.ids
    c = ix._cols
    off = np.frombuffer((C.c_char * (8 * (n + 1))).from_address(c.bases_off), dtype=np.uint64)
    qoff = np.frombuffer((C.c_char * (8 * (n + 1))).from_address(c.quality_off), dtype=np.uint64)
    _, foff = full.bases_array()
    assert (off == foff).all() and (qoff == foff).all() and not c.bases and not c.quality
    at = np.frombuffer((C.c_char * (8 * n)).from_address(ix.layout.bases_at), dtype=np.uint64)
    qat = np.frombuffer((C.c_char * (8 * n)).from_address(ix.layout.quality_at), dtype=np.uint64)
    both = t1 + t2
    for i in range(n):
        ln = int(off[i + 1] - off[i])
        assert both[int(at[i]):int(at[i]) + ln] == full.bases[i] and both[int(qat[i]):int(qat[i]) + ln] == full.quality[i]
    ix.close()
    full.close()


def test_golden_records_from_the_real_reference(F, oracle):
    cases = json.load(open(os.path.join(ROOT, "tests", "golden", "fastq_cases.json")))["cases"]
    assert len(cases) >= 8
    for c in cases:
        text = base64.b64decode(c["text_b64"])
        exp = [tuple(base64.b64decode(x) for x in r) for r in c["records"]]
        batch, _ = F.parse(text, threads=2)
        assert records(batch) == exp, c["name"]
        got, _ = oracle_records(oracle, text)
        assert got == exp, "restatement: " + c["name"]


def test_parsed_batch_feeds_the_tail(kslam, F):
    """The columns are a kslam_reads_view: the tail accepts them as they are."""
    T = importlib.import_module("kslam_amd.tail")
    t1 = b"".join(b"@p%d/1\n%s\n+\n%s\n" % (i, b"A" * 50, b"I" * 50) for i in range(4))
    t2 = b"".join(b"@p%d/2\n%s\n+\n%s\n" % (i, b"C" * 50, b"I" * 50) for i in range(4))
    batch, _, _ = F.parse_pair(t1, t2)
    ov = np.zeros(2, dtype=kslam.OVERLAP_DT)
    ov["read"], ov["entry"], ov["rel"], ov["revcomp"] = [1, 5], 0, [10, 200], [0, 1]
    ov["score"], ov["ref_begin"], ov["ref_end"], ov["query_end"] = 100, [10, 200], [59, 249], 49
    index = T.Index([b"A" * 400])
    sam, st = T.tail_sam(T.TailParams.default(report_cigar=False), batch, index, ov, np.zeros(0, np.uint32))
    lines = sam.split(b"\n")
    assert lines[0].startswith(b"p1\t") and lines[1].startswith(b"p1\t") and st.n_read_pairs == 1


@pytest.mark.parametrize("seed", range(10))
def test_batch_end_is_where_the_parser_stops(F, seed):
    """kslam_fastq_batch_end only counts line terminators; it must name the stream position the parser reports
    after max_records records (the reference's ifstream position between two calls of the batch loop,
    src/SLAM.h:193-207), for every line-ending flavour, at end of stream and on a prefix."""
    import ctypes as C
    rng = np.random.default_rng(500 + seed)
    text = make_text(rng, int(rng.integers(1, 300)), truncate=seed % 3 == 2, blank_tail=seed % 2)
    if seed == 4:
        text = text.rstrip(b"\r\n") + b"\r"
    if seed == 5:
        text = make_text(rng, 3000, eol_mix=False) * 30          # > 64 MiB would be a round; this spans many 1 MiB chunks
    buf = C.create_string_buffer(text, len(text) + 1)
    ptr = C.addressof(buf)
    total, _ = F.parse(text)
    for at_eof in (True, False):
        for n in (1, 2, 7, total.n_reads, total.n_reads + 1, 10 * total.n_reads + 3):
            if n == 0:
                continue
            batch, used = F.parse(text, max_reads=n, at_eof=at_eof, threads=2)
            end, complete = F.batch_end(ptr, len(text), n, at_eof, threads=3)
            if batch.n_reads == n and (complete or not at_eof):
                assert complete and end == used, (n, at_eof)
            else:           # fewer whole records than asked for
                assert end == len(text) and complete == at_eof, (n, at_eof)
                if at_eof:
                    assert used == len(text)
    # cutting a stream with batch_end and parsing the windows = parsing it in calls of n records
    pos, got = 0, []
    while pos < len(text):
        end, complete = F.batch_end(ptr + pos, len(text) - pos, 11, True)
        # (a window cut by batch_end ends right after a terminator, so "at end of stream" is safe for inner windows
        #  too -- and needed: a lone "\r" closing the window is a whole terminator, batch_end saw the byte after it)
        b, used = F.parse(text[pos:pos + end], at_eof=True)
        got += records(b)
        pos += end
    assert got == records(total)


def test_batch_windows_of_the_stream_driver(kslam, F):
    """k-slam_amd/stream.py: cut_batches (the windows the reference's batch loop would read, src/SLAM.h:193-207) --
    the windows tile both texts, each holds pairs_per_batch records per stream (the last one the rest), --num-reads caps the
    total, and parsing the windows one by one gives the records of the whole files in order."""
    import ctypes as C
    S = importlib.import_module("kslam_amd.stream")
    rng = np.random.default_rng(77)
    t1, t2 = make_text(rng, 103, eol_mix=False), make_text(rng, 103, eol_mix=False)   # (a lone "\r" before an empty line would merge two terminators)
    b1, b2 = C.create_string_buffer(t1, len(t1) + 1), C.create_string_buffer(t2, len(t2) + 1)
    p1, p2 = C.addressof(b1), C.addressof(b2)
    whole, _, _ = F.parse_pair(t1, t2)
    for per_batch, cap in ((10, 0), (103, 0), (50, 0), (1000, 0), (25, 60), (25, 100), (40, 40)):
        wins = list(S.cut_batches(p1, len(t1), p2, len(t2), per_batch, cap))
        want_pairs = min(103, cap) if cap else 103
        got1, got2, at1, at2 = [], [], 0, 0
        for a1, e1, a2, e2, last in wins:
            assert (a1, a2) == (at1, at2) and e1 > a1 and e2 > a2
            batch, _, _ = F.parse_pair(t1[a1:e1], t2[a2:e2], at_eof=True)
            half = batch.n_reads // 2
            got1 += records(batch)[:half]
            got2 += records(batch)[half:]
            at1, at2 = e1, e2
        assert len(got1) == want_pairs and len(wins) == -(-want_pairs // per_batch)
        assert got1 == records(whole)[:want_pairs] and got2 == records(whole)[103:103 + want_pairs]
        if not cap:
            assert (at1, at2) == (len(t1), len(t2))
