"""<db>/database restatement -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

A plain-Python reader and writer of the Boost.Serialization text archive the reference writes for
a GenbankIndex (src/GenbankTools.h:201-205 writer, :336-344 reader, serialize() members :58-62,
:101-109, :155-163, :198-200), shaped the way the reference holds the data (a list of entry
objects with a list of gene objects each) and tokenising by partition / slicing rather than by a
cursor, so that the product's column-wise two-pass parser is checked against an independently
shaped one.

PARITY UNPINNED: Boost is not in the build image and the reference ships no sample database; the
grammar is the one include/kslam_db.h states (published Boost.Serialization behaviour), restated
here independently -- both sides could share a misreading of it.
"""


class _Reader:
    def __init__(self, data):
        self.d, self.p = data, 0

    def token(self):
        d, p = self.d, self.p
        while p < len(d) and d[p] in b" \n\r\t":
            p += 1
        q = p
        while q < len(d) and d[q] not in b" \n\r\t":
            q += 1
        if q == p:
            raise ValueError("unexpected end of archive at byte %d" % p)
        self.p = q
        return d[p:q]

    def uint(self):
        t = self.token()
        if not t.isdigit():
            raise ValueError("not a number at byte %d: %r" % (self.p - len(t), t[:20]))
        return int(t)

    def string(self):
        n = self.uint()
        if self.d[self.p:self.p + 1] != b" " and not (n == 0 and self.p >= len(self.d)):
            raise ValueError("string bytes do not follow their length at byte %d" % self.p)
        s = self.d[self.p + 1:self.p + 1 + n]
        if len(s) != n:
            raise ValueError("string runs past the end of the archive")
        self.p += 1 + n
        return s

    def class_info(self):
        if self.uint() != 0 or self.uint() != 0:
            raise ValueError("tracked or versioned class at byte %d" % self.p)

    def done(self):
        return not self.d[self.p:].strip()


def parse(data):
    """bytes -> (library_version, [entry dict]) with the field names of the reference's classes."""
    r = _Reader(data)
    if r.string() != b"serialization::archive":
        raise ValueError("not a Boost.Serialization text archive")
    version = r.uint()
    r.class_info()                       # GenbankIndex
    r.class_info()                       # std::vector<GenbankEntry>
    n = r.uint()
    r.uint()                             # item_version
    entries, seen = [], set()

    def first(kind):
        if kind not in seen:
            seen.add(kind)
            r.class_info()
    for _ in range(n):
        first("GenbankEntry")
        e = {"bases": r.string(), "taxonomyID": r.uint(), "genbankID": r.uint(), "isPlasmid": bool(r.uint()),
             "is16S": bool(r.uint()), "locusTag": r.string(), "genes": []}
        first("vector<Gene>")
        ng = r.uint()
        r.uint()
        for _ in range(ng):
            first("Gene")
            g = {"geneName": r.string(), "locusTag": r.string(), "proteinID": r.string(), "product": r.string(),
                 "referenceSequence": r.string(), "geneID": r.uint()}
            first("CDS")
            g.update(start=r.uint(), stop=r.uint(), complement=bool(r.uint()))
            e["genes"].append(g)
        entries.append(e)
    if not r.done():
        raise ValueError("archive continues after the last entry")
    return version, entries


def dump(entries, library_version=17):
    """[entry dict] -> bytes, as text_oarchive << GenbankIndex would write them."""
    out = [b"22 serialization::archive %d" % library_version, b"0 0", b"0 0", b"%d 0" % len(entries)]
    seen = set()

    def first(kind):
        if kind not in seen:
            seen.add(kind)
            out.append(b"0 0")

    def s(x):
        return b"%d %s" % (len(x), x)
    for e in entries:
        first("GenbankEntry")
        out += [s(e["bases"]), b"%d %d %d %d" % (e["taxonomyID"], e["genbankID"], e["isPlasmid"], e["is16S"]),
                s(e["locusTag"])]
        first("vector<Gene>")
        out.append(b"%d 0" % len(e["genes"]))
        for g in e["genes"]:
            first("Gene")
            out += [s(g[k]) for k in ("geneName", "locusTag", "proteinID", "product", "referenceSequence")]
            out.append(b"%d" % g["geneID"])
            first("CDS")
            out.append(b"%d %d %d" % (g["start"], g["stop"], g["complement"]))
    return b" ".join(out)
