"""Plain-Python restatement of getCigarAndMD's sequence walk (reference src/SAM.h:101-237): per overlap
record NM, the log-probability and the MD text.  Checker for kslam_row_details (GPU) and for the host
tail's fast path that consumes those rows.  Test infrastructure only."""
import math

import numpy as np

_COMP = bytes.maketrans(b"ACGT", b"TGCA")          # src/sequenceTools.h:77-97: upper case only
MATCH = [math.log10(1.0 - math.pow(10.0, (i if i else 1.0) / -10.0)) for i in range(100)]   # src/SAM.h:33-40
MISMATCH = [(i if i else 1) / -10.0 for i in range(100)]                                     # src/SAM.h:41-48


def row_details(overlaps, cigar_pool, reads, quals, entries, dtype):
    """-> (details[dtype], md_pool uint8 array); reads / quals / entries: lists of bytes"""
    det = np.zeros(len(overlaps), dtype=dtype)
    pool = bytearray()
    for i, o in enumerate(overlaps):
        n = int(o["cigar_len"])
        if n == 0:
            continue
        read, qual, ref = reads[int(o["read"])], quals[int(o["read"])], entries[int(o["entry"])]
        if o["revcomp"]:
            query, q = read.translate(_COMP)[::-1], qual[::-1]
        else:
            query, q = read, qual
        rp, qp = int(o["ref_begin"]), max(int(o["query_begin"]), 0)
        comps, nm, logp, flags = [], 0, 0.0, 0
        for c in cigar_pool[int(o["cigar_off"]):int(o["cigar_off"]) + n]:
            ln, op = int(c) >> 4, int(c) & 15
            if op == 0:
                run = 0
                for _ in range(ln):
                    qv = q[qp] - 33
                    if not 0 <= qv < 100:
                        flags |= 1
                        qv = 0
                    if ref[rp] == query[qp]:
                        run += 1
                        logp += MATCH[qv]
                    else:
                        nm += 1
                        if run:
                            comps.append(run)
                        comps.append(bytes([ref[rp]]))
                        logp += MISMATCH[qv]
                        run = 0
                    rp += 1
                    qp += 1
                if run:
                    comps.append(run)
            elif op == 1:
                nm += ln
                qp += ln
            elif op == 2:
                comps.append(b"^" + ref[rp:rp + ln])
                nm += ln
                rp += ln
        md, k, after_del = bytearray(), 0, False       # the merge of SAM.h:204-235
        while k < len(comps):
            c = comps[k]
            if isinstance(c, int):
                tot = 0
                while k < len(comps) and isinstance(comps[k], int):
                    tot += comps[k]
                    k += 1
                md += b"%d" % tot
                after_del = False
                continue
            if c[:1] == b"^":
                md += c
                after_del = True
            else:
                if after_del:
                    md += b"0"
                    after_del = False
                md += c
            k += 1
        det[i] = (logp, len(pool), len(md), nm, flags, 0)
        pool += md
    return det, np.frombuffer(bytes(pool), dtype=np.uint8)
