// db.cpp -- <db>/database (Boost.Serialization text archive of a GenbankIndex) to columns
// and back: include/kslam_db.h, SURVEY.md section 8f row N2.
//
// Reference: getIndexFromBoostSerial (src/GenbankTools.h:336-344), writeIndexToBoostSerial
// (:201-205) and the serialize() members (:58-62, :101-109, :155-163, :198-200).  The grammar
// and its PARITY UNPINNED status are stated in the header.
//
// Reader = two passes.  Pass 1 walks the tokens and records, per string, where its bytes lie
// and how long they are; a string's bytes are never looked at (the length prefix says where the
// next token starts), so the walk over a 5 GB line reads a few megabytes.  Pass 2 copies the
// strings into their columns on all usable CPUs (the bases are > 99.9 % of the bytes).
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <memory>
#include <vector>

#include "../../include/kslam_db.h"
#include "workers.hpp"

using namespace kslam_host;

// a string column's bytes: plain malloc, so that 5 GB of bases are not zero-filled before they are copied in
struct CharBuf {
  char *p = nullptr;
  size_t n = 0;
  CharBuf() = default;
  CharBuf(const CharBuf &) = delete;
  CharBuf &operator=(const CharBuf &) = delete;
  ~CharBuf() { free(p); }
  void resize(size_t m) {
    free(p);
    p = (char *)malloc(m ? m : 1);
    if (!p) throw std::bad_alloc();
    advise_huge(p, m);   // the tail reads ~150 bases at a random place of this column per alignment
    n = m;
  }
  char *data() { return p; }
  const char *data() const { return p; }
};

struct kslam_db {
  uint32_t library_version = 0, variant = 0;
  // columns (see kslam_db_columns)
  CharBuf bases, locus, gname, gprot, gprod, glocus, gref;
  std::vector<uint64_t> bases_off, locus_off, gene_first, gname_off, gprot_off, gprod_off, glocus_off, gref_off;
  std::vector<uint32_t> tax_id, genbank_id, gene_id;
  std::vector<uint8_t> is_plasmid, is_16s, gcomp;
  std::vector<int32_t> gstart, gstop;
  std::vector<const char *> entry_ptr;
  std::vector<uint64_t> entry_len;
  kslam_db_columns view;
};

namespace {

struct Span {
  uint64_t at, len;   // bytes text[at .. at + len)
};

// what pass 1 collects
struct Walk {
  std::vector<Span> bases, locus, gname, glocus, gprot, gprod, gref;
  std::vector<uint32_t> tax_id, genbank_id, gene_id;
  std::vector<uint8_t> is_plasmid, is_16s, gcomp;
  std::vector<int32_t> gstart, gstop;
  std::vector<uint64_t> gene_first;
};

struct ParseFail {
  uint64_t at;
  std::string why;
};

class Cursor {
 public:
  Cursor(const char *t, uint64_t n) : t_(t), n_(n) {}
  uint64_t pos() const { return p_; }
  bool at_end() {
    skip_ws();
    return p_ >= n_;
  }
  // an unsigned decimal token
  uint64_t number(const char *what, uint64_t max) {
    skip_ws();
    const uint64_t start = p_;
    if (p_ >= n_) throw ParseFail{p_, std::string("text ends where ") + what + " was expected"};
    uint64_t v = 0;
    int digits = 0;
    while (p_ < n_ && t_[p_] >= '0' && t_[p_] <= '9') {
      if (v > (UINT64_MAX - 9) / 10) throw ParseFail{start, std::string(what) + " does not fit 64 bits"};
      v = v * 10 + (uint64_t)(t_[p_] - '0');
      p_++;
      digits++;
    }
    if (!digits || (p_ < n_ && !is_ws(t_[p_])))
      throw ParseFail{start, std::string(what) + " is not an unsigned decimal token"};
    if (v > max) throw ParseFail{start, std::string(what) + " = " + std::to_string(v) + " is out of range"};
    return v;
  }
  bool flag(const char *what) { return number(what, 1) != 0; }
  // <length> ' ' <length raw bytes>
  Span string(const char *what) {
    const uint64_t len = number(what, n_);
    if (p_ >= n_ || t_[p_] != ' ') {
      if (len == 0 && p_ >= n_) return Span{p_, 0};   // an empty string as the very last token
      throw ParseFail{p_, std::string("no single space between the length and the bytes of ") + what};
    }
    p_++;
    if (len > n_ - p_) throw ParseFail{p_, std::string(what) + ": length " + std::to_string(len) + " runs past the end of the text"};
    const Span s{p_, len};
    p_ += len;
    if (p_ < n_ && !is_ws(t_[p_])) throw ParseFail{p_, std::string(what) + ": no delimiter after its " + std::to_string(len) + " bytes"};
    return s;
  }

 private:
  static bool is_ws(char c) { return c == ' ' || c == '\n' || c == '\r' || c == '\t'; }
  void skip_ws() {
    while (p_ < n_ && is_ws(t_[p_])) p_++;
  }
  const char *t_;
  uint64_t n_, p_ = 0;
};

// tracking + version of a class type met for the first time: both must be 0 here (nothing is
// saved through a pointer, no class declares a version)
void class_info(Cursor &c, const char *type) {
  c.number((std::string("tracking flag of ") + type).c_str(), 0);
  c.number((std::string("class version of ") + type).c_str(), 0);
}

constexpr uint64_t MAX_ITEMS = 1ull << 32;

void walk(const char *text, uint64_t len, uint32_t variant, uint32_t *libver, Walk &w) {
  const bool vec_info = !(variant & 1u), item_version = !(variant & 2u);
  Cursor c(text, len);
  // "22 serialization::archive": the signature is itself saved as a std::string
  const Span sig = c.string("archive signature");
  if (sig.len != 22 || memcmp(text + sig.at, "serialization::archive", 22) != 0)
    throw ParseFail{sig.at, "not a Boost.Serialization text archive (signature)"};
  *libver = (uint32_t)c.number("library version", 1000);
  class_info(c, "GenbankIndex");
  if (vec_info) class_info(c, "std::vector<GenbankEntry>");
  const uint64_t n_entries = c.number("entry count", std::min<uint64_t>(MAX_ITEMS, len));
  if (item_version) c.number("item version of std::vector<GenbankEntry>", 0);
  w.gene_first.reserve(n_entries + 1);
  w.bases.reserve(n_entries);
  bool seen_entry = false, seen_genes = false, seen_gene = false, seen_cds = false;
  for (uint64_t e = 0; e < n_entries; e++) {
    if (!seen_entry) {
      class_info(c, "GenbankEntry");
      seen_entry = true;
    }
    w.bases.push_back(c.string("GenbankEntry::bases"));
    w.tax_id.push_back((uint32_t)c.number("GenbankEntry::taxonomyID", UINT32_MAX));
    w.genbank_id.push_back((uint32_t)c.number("GenbankEntry::genbankID", UINT32_MAX));
    w.is_plasmid.push_back(c.flag("GenbankEntry::isPlasmid"));
    w.is_16s.push_back(c.flag("GenbankEntry::is16S"));
    w.locus.push_back(c.string("GenbankEntry::locusTag"));
    if (!seen_genes) {
      if (vec_info) class_info(c, "std::vector<Gene>");
      seen_genes = true;
    }
    const uint64_t n_genes = c.number("gene count", std::min<uint64_t>(MAX_ITEMS, len));
    if (item_version) c.number("item version of std::vector<Gene>", 0);
    w.gene_first.push_back(w.gname.size());
    for (uint64_t g = 0; g < n_genes; g++) {
      if (!seen_gene) {
        class_info(c, "Gene");
        seen_gene = true;
      }
      w.gname.push_back(c.string("Gene::geneName"));
      w.glocus.push_back(c.string("Gene::locusTag"));
      w.gprot.push_back(c.string("Gene::proteinID"));
      w.gprod.push_back(c.string("Gene::product"));
      w.gref.push_back(c.string("Gene::referenceSequence"));
      w.gene_id.push_back((uint32_t)c.number("Gene::geneID", UINT32_MAX));
      if (!seen_cds) {
        class_info(c, "CDS");
        seen_cds = true;
      }
      w.gstart.push_back((int32_t)(uint32_t)c.number("CDS::start", UINT32_MAX));   // getGene reads them as int
      w.gstop.push_back((int32_t)(uint32_t)c.number("CDS::stop", UINT32_MAX));
      w.gcomp.push_back(c.flag("CDS::complement"));
    }
  }
  w.gene_first.push_back(w.gname.size());
  if (!c.at_end()) throw ParseFail{c.pos(), "text continues after the last entry"};
}

// spans -> one contiguous column + offsets; big columns are copied by all workers
void gather(const char *text, const std::vector<Span> &spans, CharBuf &col, std::vector<uint64_t> &off,
            int threads) {
  off.resize(spans.size() + 1);
  uint64_t total = 0;
  for (size_t i = 0; i < spans.size(); i++) {
    off[i] = total;
    total += spans[i].len;
  }
  off[spans.size()] = total;
  col.resize(total);
  if (!total) return;
  // tasks of about 8 MB: split long strings, batch short ones
  struct Piece {
    uint64_t src, dst, len;
  };
  std::vector<Piece> pieces;
  constexpr uint64_t CHUNK = 8ull << 20;
  for (size_t i = 0; i < spans.size(); i++)
    for (uint64_t o = 0; o < spans[i].len; o += CHUNK)
      pieces.push_back(Piece{spans[i].at + o, off[i] + o, std::min(CHUNK, spans[i].len - o)});
  if (total < (64ull << 20) || threads <= 1) {
    for (const Piece &p : pieces) memcpy(col.data() + p.dst, text + p.src, p.len);
    return;
  }
  const size_t per = std::max<size_t>(1, pieces.size() / ((size_t)threads * 8));
  Pool::get().tasks(threads, (pieces.size() + per - 1) / per, [&](size_t t) {
    for (size_t k = t * per; k < std::min(pieces.size(), (t + 1) * per); k++)
      memcpy(col.data() + pieces[k].dst, text + pieces[k].src, pieces[k].len);
  });
}

void finish_view(kslam_db &d) {
  kslam_db_columns &v = d.view;
  memset(&v, 0, sizeof v);
  v.index.n_entries = d.tax_id.size();
  v.index.bases = d.bases.data();
  v.index.bases_off = d.bases_off.data();
  v.index.locus_tag = d.locus.data();
  v.index.locus_tag_off = d.locus_off.data();
  v.index.taxonomy_id = d.tax_id.data();
  v.index.n_genes = d.gene_id.size();
  v.index.gene_first = d.gene_first.data();
  v.index.gene_start = d.gstart.data();
  v.index.gene_stop = d.gstop.data();
  v.index.gene_name = d.gname.data();
  v.index.gene_name_off = d.gname_off.data();
  v.index.protein_id = d.gprot.data();
  v.index.protein_id_off = d.gprot_off.data();
  v.index.product = d.gprod.data();
  v.index.product_off = d.gprod_off.data();
  v.genbank_id = d.genbank_id.data();
  v.is_plasmid = d.is_plasmid.data();
  v.is_16s = d.is_16s.data();
  v.gene_locus_tag = d.glocus.data();
  v.gene_locus_tag_off = d.glocus_off.data();
  v.gene_reference = d.gref.data();
  v.gene_reference_off = d.gref_off.data();
  v.gene_id = d.gene_id.data();
  v.gene_complement = d.gcomp.data();
  const size_t n = d.tax_id.size();
  d.entry_ptr.resize(n);
  d.entry_len.resize(n);
  for (size_t i = 0; i < n; i++) {
    d.entry_ptr[i] = d.bases.data() + d.bases_off[i];
    d.entry_len[i] = d.bases_off[i + 1] - d.bases_off[i];
  }
}

void parse_into(const char *text, uint64_t len, int threads, kslam_db &d) {
  if (threads <= 0) threads = usable_cpus();
  std::string first_error;
  for (uint32_t variant = 0; variant < 4; variant++) {
    Walk w;
    try {
      walk(text, len, variant, &d.library_version, w);
    } catch (const ParseFail &f) {
      if (variant == 0) first_error = "database archive: " + f.why + " (byte " + std::to_string(f.at) + ")";
      continue;
    }
    d.variant = variant;
    gather(text, w.bases, d.bases, d.bases_off, threads);
    gather(text, w.locus, d.locus, d.locus_off, threads);
    gather(text, w.gname, d.gname, d.gname_off, threads);
    gather(text, w.gprot, d.gprot, d.gprot_off, threads);
    gather(text, w.gprod, d.gprod, d.gprod_off, threads);
    gather(text, w.glocus, d.glocus, d.glocus_off, threads);
    gather(text, w.gref, d.gref, d.gref_off, threads);
    d.tax_id.swap(w.tax_id);
    d.genbank_id.swap(w.genbank_id);
    d.gene_id.swap(w.gene_id);
    d.is_plasmid.swap(w.is_plasmid);
    d.is_16s.swap(w.is_16s);
    d.gcomp.swap(w.gcomp);
    d.gstart.swap(w.gstart);
    d.gstop.swap(w.gstop);
    d.gene_first.swap(w.gene_first);
    finish_view(d);
    return;
  }
  fail(KSLAM_ERR_ARG, first_error);
}

struct Mapping {
  const char *p = nullptr;
  uint64_t n = 0;
  int fd = -1;
  ~Mapping() {
    if (p && n) munmap(const_cast<char *>(p), n);
    if (fd >= 0) close(fd);
  }
};

// buffered text output
struct Out {
  FILE *f;
  explicit Out(FILE *f_) : f(f_) {}
  void num(uint64_t v) {
    if (fprintf(f, " %llu", (unsigned long long)v) < 0) fail(KSLAM_ERR_INTERNAL, "database archive: write failed");
  }
  void str(const char *col, const uint64_t *off, uint64_t i) {
    const uint64_t a = col && off ? off[i] : 0, b = col && off ? off[i + 1] : 0;
    num(b - a);
    if (fputc(' ', f) == EOF) fail(KSLAM_ERR_INTERNAL, "database archive: write failed");
    if (b > a && fwrite(col + a, 1, b - a, f) != b - a) fail(KSLAM_ERR_INTERNAL, "database archive: write failed");
  }
};

}  // namespace

extern "C" {

kslam_status kslam_db_parse(const char *text, uint64_t len, int threads, kslam_db **out) {
  return guarded([&] {
    if (!out || (len && !text)) fail(KSLAM_ERR_ARG, "null argument");
    *out = nullptr;
    std::unique_ptr<kslam_db> d(new kslam_db());
    parse_into(text, len, threads, *d);
    *out = d.release();
  });
}

kslam_status kslam_db_load(const char *path, int threads, kslam_db **out) {
  return guarded([&] {
    if (!out || !path) fail(KSLAM_ERR_ARG, "null argument");
    *out = nullptr;
    Mapping m;
    m.fd = open(path, O_RDONLY);
    if (m.fd < 0) fail(KSLAM_ERR_ARG, std::string("Unable to open file ") + path);   // message of src/GenbankTools.h:339
    struct stat st;
    if (fstat(m.fd, &st) != 0) fail(KSLAM_ERR_ARG, std::string("cannot stat ") + path);
    m.n = (uint64_t)st.st_size;
    if (m.n) {
      void *p = mmap(nullptr, m.n, PROT_READ, MAP_PRIVATE, m.fd, 0);
      if (p == MAP_FAILED) {
        m.n = 0;
        fail(KSLAM_ERR_OOM, std::string("cannot map ") + path);
      }
      m.p = (const char *)p;
    }
    std::unique_ptr<kslam_db> d(new kslam_db());
    parse_into(m.p, m.n, threads, *d);
    *out = d.release();
  });
}

void kslam_db_free(kslam_db *db) { delete db; }
const kslam_db_columns *kslam_db_view(const kslam_db *db) { return db ? &db->view : nullptr; }
uint32_t kslam_db_library_version(const kslam_db *db) { return db ? db->library_version : 0; }
uint32_t kslam_db_variant(const kslam_db *db) { return db ? db->variant : 0; }
const char *const *kslam_db_entry_bases(const kslam_db *db) { return db ? db->entry_ptr.data() : nullptr; }
const uint64_t *kslam_db_entry_lengths(const kslam_db *db) { return db ? db->entry_len.data() : nullptr; }

kslam_status kslam_db_write(const char *path, const kslam_db_columns *c, uint32_t library_version) {
  return guarded([&] {
    if (!path || !c) fail(KSLAM_ERR_ARG, "null argument");
    const kslam_index_view &iv = c->index;
    if (iv.n_entries && (!iv.bases || !iv.bases_off)) fail(KSLAM_ERR_ARG, "database archive: no bases column");
    if (iv.n_genes && !iv.gene_first) fail(KSLAM_ERR_ARG, "database archive: genes without gene_first");
    FILE *f = fopen(path, "wb");
    if (!f) fail(KSLAM_ERR_ARG, std::string("Unable to open file ") + path);
    std::unique_ptr<FILE, int (*)(FILE *)> closer(f, fclose);
    static thread_local std::vector<char> buf;
    buf.resize(8u << 20);
    setvbuf(f, buf.data(), _IOFBF, buf.size());
    Out o(f);
    if (fputs("22 serialization::archive", f) == EOF) fail(KSLAM_ERR_INTERNAL, "database archive: write failed");
    o.num(library_version);
    o.num(0), o.num(0);   // GenbankIndex
    o.num(0), o.num(0);   // std::vector<GenbankEntry>
    o.num(iv.n_entries), o.num(0);
    bool seen_gene = false;
    for (uint64_t e = 0; e < iv.n_entries; e++) {
      if (e == 0) o.num(0), o.num(0);   // GenbankEntry
      o.str(iv.bases, iv.bases_off, e);
      o.num(iv.taxonomy_id ? iv.taxonomy_id[e] : 0);
      o.num(c->genbank_id ? c->genbank_id[e] : 0);
      o.num(c->is_plasmid ? (c->is_plasmid[e] != 0) : 0);
      o.num(c->is_16s ? (c->is_16s[e] != 0) : 0);
      o.str(iv.locus_tag, iv.locus_tag_off, e);
      if (e == 0) o.num(0), o.num(0);   // std::vector<Gene>
      const uint64_t g0 = iv.n_genes ? iv.gene_first[e] : 0, g1 = iv.n_genes ? iv.gene_first[e + 1] : 0;
      o.num(g1 - g0), o.num(0);
      for (uint64_t g = g0; g < g1; g++) {
        if (!seen_gene) o.num(0), o.num(0);   // Gene
        o.str(iv.gene_name, iv.gene_name_off, g);
        o.str(c->gene_locus_tag, c->gene_locus_tag_off, g);
        o.str(iv.protein_id, iv.protein_id_off, g);
        o.str(iv.product, iv.product_off, g);
        o.str(c->gene_reference, c->gene_reference_off, g);
        o.num(c->gene_id ? c->gene_id[g] : 0);
        if (!seen_gene) o.num(0), o.num(0);   // CDS
        seen_gene = true;
        o.num(iv.gene_start ? (uint32_t)iv.gene_start[g] : 0);
        o.num(iv.gene_stop ? (uint32_t)iv.gene_stop[g] : 0);
        o.num(c->gene_complement ? (c->gene_complement[g] != 0) : 0);
      }
    }
    if (fflush(f) != 0) fail(KSLAM_ERR_INTERNAL, "database archive: write failed");
  });
}

}  // extern "C"
