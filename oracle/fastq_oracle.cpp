// fastq_oracle.cpp -- TEST INFRASTRUCTURE ONLY (see oracle/kslam_oracle.h header).
//
// Serial restatement of the reference's FASTQ reader:
// safeGetline (src/sequenceTools.h:45-73), the four-line loop of
// getSequencesFromFASTQFile (src/FASTQsequence.h:129-165) and the identifier rule of
// the FASTQSequence constructor (src/FASTQsequence.h:61-71), over a memory buffer that
// stands for the std::ifstream.
//
// PINNED: oracle/ref_fastq_driver.cpp includes the reference's own FASTQsequence.h where it
// lies (it needs no Boost) and is built into oracle/_ref/libfastq_ref.so; tests/test_fastq.py
// checks this file against it record by record, stream position included.
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace {

struct Stream {  // the part of std::istream the reader relies on
  const char *t;
  uint64_t len, pos = 0;
  bool eof = false, fail = false;
};

// safeGetline: false once the stream has failed (what `while (safeGetline(...))` tests)
bool get_line(Stream &s, std::string &line) {
  line.clear();
  if (s.eof || s.fail) {  // the sentry of a stream that is not good() sets failbit
    s.fail = true;
    return false;
  }
  for (;;) {
    if (s.pos >= s.len) {
      if (line.empty()) s.eof = true;
      return true;
    }
    char c = s.t[s.pos++];
    if (c == '\n') return true;
    if (c == '\r') {
      if (s.pos < s.len && s.t[s.pos] == '\n') s.pos++;
      return true;
    }
    line += c;
  }
}

std::string identifier(const std::string &header) {
  std::string id;
  if (header.size() > 1) {
    size_t space = header.find(' ');
    if (space > 0) space--;
    id = header.substr(1, space);
    id = id.substr(0, id.find('/'));
  }
  return id;
}

struct Rec {
  std::string id, bases, qual;
};

}  // namespace

extern "C" {

// Parses up to max_reads records (the reference's `numReads`; UINT32_MAX = its default) starting at
// *pos; returns the records flattened: three buffers + offsets, malloc'ed.  *pos is advanced to
// where the stream stands afterwards.
int orc_fastq_read(const char *text, uint64_t len, uint64_t *pos, uint32_t max_reads, uint64_t *n_out,
                   char **bases, uint64_t **bases_off, char **qual, uint64_t **qual_off, char **ids,
                   uint64_t **ids_off) {
  Stream s{text, len, *pos};
  std::vector<Rec> recs;
  std::string line, header, b;
  unsigned type = 0, added = 0;
  while (get_line(s, line)) {
    switch (type) {
      case 0: header = line; type++; break;
      case 1: b = line; type++; break;
      case 2: type++; break;
      case 3:
        recs.push_back(Rec{identifier(header), b, line});
        added++;
        type = 0;
        break;
    }
    if (added >= max_reads) break;
  }
  *pos = s.pos;
  *n_out = recs.size();
  std::string cb, cq, ci;
  *bases_off = (uint64_t *)malloc(8 * (recs.size() + 1));
  *qual_off = (uint64_t *)malloc(8 * (recs.size() + 1));
  *ids_off = (uint64_t *)malloc(8 * (recs.size() + 1));
  for (size_t i = 0; i < recs.size(); i++) {
    (*bases_off)[i] = cb.size();
    (*qual_off)[i] = cq.size();
    (*ids_off)[i] = ci.size();
    cb += recs[i].bases;
    cq += recs[i].qual;
    ci += recs[i].id;
  }
  (*bases_off)[recs.size()] = cb.size();
  (*qual_off)[recs.size()] = cq.size();
  (*ids_off)[recs.size()] = ci.size();
  auto dup = [](const std::string &x) {
    char *p = (char *)malloc(x.size() + 1);
    memcpy(p, x.data(), x.size());
    p[x.size()] = 0;
    return p;
  };
  *bases = dup(cb);
  *qual = dup(cq);
  *ids = dup(ci);
  return 0;
}

void orc_fastq_free(void *p) { free(p); }
}
