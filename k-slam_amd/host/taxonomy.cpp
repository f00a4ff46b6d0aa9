// taxonomy.cpp -- the taxonomy stage behind include/kslam_taxonomy.h (per-read LCA of SURVEY.md
// section 8f row N1).
//
// The reference keeps the NCBI tree in an unordered_map<taxid, entry> and answers every query by
// chains of hash lookups (src/TaxonomyDatabase.h); its LCA builds one root-ward path vector per
// input id and compares them level by level.  Here the tree is dense: nodes numbered in file
// order, a parent-index and a depth array, and the LCA of a set is a fold of pairwise walks that
// first level the depths.  The conventions that make the answers equal are in the header.
#include <algorithm>
#include <cmath>
#include <memory>
#include <string>
#include <string_view>
#include <unordered_map>
#include <vector>

#include "../../include/kslam_taxonomy.h"
#include "../csrc/gnu_sort.h"
#include "workers.hpp"

struct kslam_taxdb {
  static constexpr uint32_t NONE = 0xFFFFFFFFu;
  std::unordered_map<uint32_t, uint32_t> node_of;  // taxonomy id -> node
  std::vector<uint32_t> tax_id, parent_id;         // as in the file
  std::vector<uint32_t> up;                        // node of getParentTaxID(), NONE when that is 0
  std::vector<uint32_t> depth;                     // nodes on the path up to the top-level node
  std::vector<std::string> name, rank;
  uint64_t n_real = 0;  // nodes read from the file; nodes past this stand for parent ids the file
                        // mentions but never defines (the reference still puts them on paths)
  uint32_t node(uint32_t id) const {
    auto it = node_of.find(id);
    return it == node_of.end() ? NONE : it->second;
  }
  bool known(uint32_t id) const {
    uint32_t n = node(id);
    return n != NONE && n < n_real;
  }
  // getParentTaxID, src/TaxonomyDatabase.h:225-231
  uint32_t parent_tax(uint32_t id) const {
    uint32_t n = node(id);
    if (n != NONE && n < n_real && parent_id[n] != 1) return parent_id[n];
    return 0;
  }
};

// One IdentifiedTaxonomy (src/MetagenomicResults.h:32-43) as collected per read pair: the genes are numbers
// into the index view's gene columns
// (pooled: record i's read name is names[name_off[i] .. name_off[i + 1]), its genes genes[gene_off[i] .. gene_off[i + 1]);
// a batch appends a million records, so there is no object -- and no allocation -- per record)
struct kslam_taxreport {
  std::vector<uint32_t> tax;
  std::vector<uint8_t> has_read;
  std::vector<uint64_t> name_off{0}, gene_off{0};
  std::vector<char> names;
  std::vector<uint64_t> genes;
  size_t size() const { return tax.size(); }
};

namespace {
using namespace kslam_host;

std::string_view col(const char *text, const uint64_t *off, uint64_t i) {
  if (!text || !off) return std::string_view();
  return std::string_view(text + off[i], off[i + 1] - off[i]);
}
// geneSort and Gene::operator==, src/GenbankTools.h:84-91, 116-125 (std::string compares bytes as unsigned)
struct GeneOrder {
  const kslam_index_view *ix;
  std::string_view name(uint64_t g) const { return col(ix->gene_name, ix->gene_name_off, g); }
  std::string_view protein(uint64_t g) const { return col(ix->protein_id, ix->protein_id_off, g); }
  std::string_view product(uint64_t g) const { return col(ix->product, ix->product_off, g); }
  bool less(uint64_t i, uint64_t j) const {
    if (protein(i).empty() && protein(j).empty()) return name(i) < name(j);
    if (protein(i) == protein(j)) return product(i) < product(j);
    return protein(i) < protein(j);
  }
  bool equal(uint64_t i, uint64_t j) const {
    if (protein(i).empty() && protein(j).empty()) return name(i) == name(j);
    if (protein(i) == protein(j)) return product(i) == product(j);
    return false;
  }
};
// GenbankEntry::getGene, src/GenbankTools.h:170-185: largest overlap, the first gene on ties, none at <= 0
int64_t best_gene_of(const kslam_index_view *ix, uint32_t e, int32_t start, int32_t stop) {
  if (!ix->n_genes || !ix->gene_first) return -1;
  int64_t best = -1;
  int32_t largest = 0;
  for (uint64_t g = ix->gene_first[e]; g < ix->gene_first[e + 1]; g++) {
    const int32_t shared = std::min<int>(stop, ix->gene_stop[g]) - std::max<int>(start, ix->gene_start[g]);
    if (shared > largest) {
      best = (int64_t)g;
      largest = shared;
    }
  }
  return best;
}
void xml_escaped(std::string &out, std::string_view in) {   // correctXML, src/MetagenomicResults.h:275-301
  for (char c : in) {
    switch (c) {
      case '<': out += "&lt;"; break;
      case '>': out += "&gt;"; break;
      case '&': out += "&amp;"; break;
      case '\'': out += "&apos;"; break;
      case '"': out += "&quot;"; break;
      default: out += c;
    }
  }
}

bool to_number(const char *s, size_t n, uint32_t *out) {
  // std::stoi: optional whitespace, optional sign, digits; trailing text ignored
  size_t i = 0;
  while (i < n && (s[i] == ' ' || s[i] == '\t' || s[i] == '\r' || s[i] == '\v' || s[i] == '\f')) i++;
  bool neg = false;
  if (i < n && (s[i] == '+' || s[i] == '-')) neg = s[i++] == '-';
  if (i >= n || s[i] < '0' || s[i] > '9') return false;
  int64_t v = 0;
  for (; i < n && s[i] >= '0' && s[i] <= '9'; i++) {
    v = v * 10 + (s[i] - '0');
    if (v > 2147483648ll) return false;  // std::out_of_range in the reference
  }
  if (neg) v = -v;
  if (v > 2147483647ll) return false;
  *out = (uint32_t)(int32_t)v;
  return true;
}

uint32_t lca_nodes(const kslam_taxdb &db, uint32_t a, uint32_t b) {
  while (db.depth[a] > db.depth[b]) a = db.up[a];
  while (db.depth[b] > db.depth[a]) b = db.up[b];
  while (a != b) {
    a = db.up[a];
    b = db.up[b];
    if (a == kslam_taxdb::NONE || b == kslam_taxdb::NONE) return kslam_taxdb::NONE;
  }
  return a;
}

// getLowestCommonAncestor, src/TaxonomyDatabase.h:185-223
uint32_t lca_ids(const kslam_taxdb &db, const uint32_t *ids, uint64_t n) {
  if (n == 0) return 0;
  uint32_t acc = kslam_taxdb::NONE;
  uint32_t lone = 0;  // an id the tree does not know: a path of just itself
  bool have_lone = false;
  for (uint64_t i = 0; i < n; i++) {
    const uint32_t id = ids[i];
    if (id == 0) return 0;  // empty path: nothing in common
    const uint32_t nd = db.node(id);
    if (nd == kslam_taxdb::NONE) {
      if (acc != kslam_taxdb::NONE || (have_lone && lone != id)) return 0;
      lone = id;
      have_lone = true;
      continue;
    }
    if (have_lone) return 0;
    acc = acc == kslam_taxdb::NONE ? nd : lca_nodes(db, acc, nd);
    if (acc == kslam_taxdb::NONE) return 0;
  }
  return have_lone ? lone : db.tax_id[acc];
}

std::string lineage_of(const kslam_taxdb &db, uint32_t id) {  // src/TaxonomyDatabase.h:249-265
  std::string lineage;
  for (;;) {
    if (id != 131567) {
      if (!lineage.empty()) lineage.insert(0, "; ");
      if (db.known(id)) lineage.insert(0, db.name[db.node(id)]);
      if (db.known(id) && db.rank[db.node(id)] == "species") lineage.clear();
    }
    id = db.parent_tax(id);
    if (id == 0) {
      if (!lineage.empty()) lineage.append(".");
      break;
    }
  }
  return lineage;
}

char *dup_text(const std::string &s, uint64_t *len) {
  char *p = (char *)malloc(s.size() + 1);
  if (!p) fail(KSLAM_ERR_OOM, "out of host memory");
  memcpy(p, s.data(), s.size());
  p[s.size()] = 0;
  *len = s.size();
  return p;
}

}  // namespace

extern "C" {

kslam_status kslam_taxdb_parse(const char *text, uint64_t len, kslam_taxdb **out) {
  return guarded([&] {
    if (!out || (len && !text)) fail(KSLAM_ERR_ARG, "null argument");
    *out = nullptr;
    std::unique_ptr<kslam_taxdb> db(new kslam_taxdb());
    // std::getline lines: '\n' only; a final line without '\n' still counts
    std::vector<std::pair<uint64_t, uint64_t>> lines;
    for (uint64_t p = 0; p < len;) {
      const void *nl = memchr(text + p, '\n', len - p);
      uint64_t e = nl ? (uint64_t)((const char *)nl - text) : len;
      lines.push_back({p, e - p});
      p = e + 1;
    }
    if (lines.size() % 4) fail(KSLAM_ERR_ARG, "taxonomy index: line count is not a multiple of four");
    for (size_t i = 0; i < lines.size(); i += 4) {
      uint32_t id, parent;
      if (!to_number(text + lines[i].first, lines[i].second, &id) ||
          !to_number(text + lines[i + 1].first, lines[i + 1].second, &parent))
        fail(KSLAM_ERR_ARG, "taxonomy index: id line " + std::to_string(i + 1) + " is not a number");
      if (db->node_of.count(id)) continue;  // map::insert keeps the first
      db->node_of[id] = (uint32_t)db->tax_id.size();
      db->tax_id.push_back(id);
      db->parent_id.push_back(parent);
      db->name.emplace_back(text + lines[i + 2].first, lines[i + 2].second);
      db->rank.emplace_back(text + lines[i + 3].first, lines[i + 3].second);
    }
    db->n_real = db->tax_id.size();
    // parents the file never defines still appear on the reference's paths
    for (uint64_t n = 0; n < db->n_real; n++) {
      const uint32_t p = db->parent_id[n];
      if (p != 1 && p != 0 && !db->node_of.count(p)) {
        db->node_of[p] = (uint32_t)db->tax_id.size();
        db->tax_id.push_back(p);
        db->parent_id.push_back(1);
        db->name.emplace_back();
        db->rank.emplace_back();
      }
    }
    const size_t N = db->tax_id.size();
    db->up.assign(N, kslam_taxdb::NONE);
    for (size_t n = 0; n < N; n++) {
      const uint32_t p = n < db->n_real ? db->parent_id[n] : 1;
      if (p != 1 && p != 0) db->up[n] = db->node_of[p];
    }
    // depths, and a cycle check, by walking up with path marking
    db->depth.assign(N, 0);
    std::vector<uint32_t> stack;
    std::vector<uint8_t> on_path(N, 0);
    for (size_t n = 0; n < N; n++) {
      if (db->depth[n]) continue;
      stack.clear();
      uint32_t v = (uint32_t)n;
      while (v != kslam_taxdb::NONE && !db->depth[v]) {
        if (on_path[v]) fail(KSLAM_ERR_ARG, "taxonomy index: the parent links contain a cycle");
        on_path[v] = 1;
        stack.push_back(v);
        v = db->up[v];
      }
      uint32_t d = v == kslam_taxdb::NONE ? 0 : db->depth[v];
      for (size_t k = stack.size(); k-- > 0;) {
        db->depth[stack[k]] = ++d;
        on_path[stack[k]] = 0;
      }
    }
    *out = db.release();
  });
}

void kslam_taxdb_free(kslam_taxdb *db) { delete db; }
uint64_t kslam_taxdb_size(const kslam_taxdb *db) { return db ? db->n_real : 0; }

kslam_status kslam_taxdb_dense(const kslam_taxdb *db, uint64_t *n_nodes, const uint32_t **up, const uint32_t **depth,
                               const uint32_t **node_tax) {
  return guarded([&] {
    if (!db || !n_nodes || !up || !depth || !node_tax) fail(KSLAM_ERR_ARG, "null argument");
    *n_nodes = db->up.size();
    *up = db->up.data();
    *depth = db->depth.data();
    *node_tax = db->tax_id.data();
  });
}
uint32_t kslam_taxdb_node(const kslam_taxdb *db, uint32_t tax_id) { return db ? db->node(tax_id) : kslam_taxdb::NONE; }

uint32_t kslam_taxdb_lca(const kslam_taxdb *db, const uint32_t *tax_ids, uint64_t n) {
  if (!db || (!tax_ids && n)) return 0;
  return lca_ids(*db, tax_ids, n);
}

uint32_t kslam_taxdb_parent(const kslam_taxdb *db, uint32_t tax_id) { return db ? db->parent_tax(tax_id) : 0; }

// getTaxIDAtRank, src/TaxonomyDatabase.h:306-317
uint32_t kslam_taxdb_at_rank(const kslam_taxdb *db, uint32_t tax_id, const char *rank) {
  if (!db || !rank) return 0;
  uint32_t n = db->node(tax_id);
  while (n != kslam_taxdb::NONE && n < db->n_real && db->parent_id[n] != 1) {
    if (db->rank[n] == rank) return db->tax_id[n];
    n = db->node(db->parent_id[n]);
  }
  return 0;
}

// isBelowInTree, src/TaxonomyDatabase.h:318-331
int32_t kslam_taxdb_is_below(const kslam_taxdb *db, uint32_t upper, uint32_t lower) {
  if (!db) return -1;
  uint32_t n = db->node(lower);
  unsigned level = 0;
  while (n != kslam_taxdb::NONE && n < db->n_real && db->parent_id[n] != 1) {
    if (db->tax_id[n] == upper) return (int32_t)level;
    n = db->node(db->parent_id[n]);
    level++;
  }
  return -1;
}

// isSubSpecies, src/TaxonomyDatabase.h:332-349
int32_t kslam_taxdb_is_subspecies(const kslam_taxdb *db, uint32_t tax_id) {
  if (!db) return 0;
  uint32_t n = db->node(tax_id);
  int levels = 0;
  while (n != kslam_taxdb::NONE && n < db->n_real && db->parent_id[n] != 1) {
    if (db->rank[n] == "species") return levels > 0;
    n = db->node(db->parent_id[n]);
    levels++;
  }
  return 0;
}

kslam_status kslam_taxdb_text(const kslam_taxdb *db, uint32_t tax_id, int which, char **text,
                              uint64_t *text_len) {
  return guarded([&] {
    if (!db || !text || !text_len) fail(KSLAM_ERR_ARG, "null argument");
    std::string s;
    if (which == 0 || which == 1) {
      if (db->known(tax_id)) s = which == 0 ? db->name[db->node(tax_id)] : db->rank[db->node(tax_id)];
    } else if (which == 2)
      s = lineage_of(*db, tax_id);
    else
      fail(KSLAM_ERR_ARG, "which must be 0 (name), 1 (rank) or 2 (lineage)");
    *text = dup_text(s, text_len);
  });
}

kslam_status kslam_tail_classify(const kslam_tail_params *params, const kslam_reads_view *reads,
                                 const kslam_index_view *index, const kslam_taxdb *db,
                                 const kslam_read_pair *read_pairs, uint64_t n_read_pairs,
                                 const kslam_paired_overlap *pairs, uint64_t n_pairs, uint32_t *tax_ids,
                                 char **per_read_text, uint64_t *per_read_len) {
  return guarded([&] {
    if (!params || !reads || !index || !db || !tax_ids || (n_read_pairs && (!read_pairs || !pairs)))
      fail(KSLAM_ERR_ARG, "null argument");
    if (!index->taxonomy_id) fail(KSLAM_ERR_ARG, "index view needs taxonomy_id");
    if (per_read_text && (!per_read_len || !reads->ids || !reads->ids_off))
      fail(KSLAM_ERR_ARG, "per-read text needs the read identifiers and a length output");
    const int threads = std::max(1, std::min(params->threads > 0 ? params->threads : usable_cpus(), 512));
    const uint64_t grain = 4096, n_tasks = (n_read_pairs + grain - 1) / grain;
    Pool::get().tasks(threads, n_tasks, [&](size_t t) {
      std::vector<uint32_t> ids;
      for (uint64_t g = t * grain; g < std::min(n_read_pairs, (t + 1) * grain); g++) {
        const kslam_read_pair &rp = read_pairs[g];
        if (rp.first + rp.count > n_pairs) fail(KSLAM_ERR_ARG, "read pair slice outside the pairs array");
        ids.clear();
        for (uint64_t k = 0; k < rp.count; k++) {
          const uint32_t e = pairs[rp.first + k].entry;
          if (e >= index->n_entries) fail(KSLAM_ERR_ARG, "alignment pair refers outside the index");
          ids.push_back(index->taxonomy_id[e]);
        }
        tax_ids[g] = lca_ids(*db, ids.data(), ids.size());  // src/MetagenomicResults.h:88-112
      }
    });
    if (per_read_text) {  // writePerReadResults, src/MetagenomicResults.h:455-463
      std::vector<std::string> part(n_tasks);
      Pool::get().tasks(threads, n_tasks, [&](size_t t) {
        std::string &out = part[t];
        out.reserve(grain * 24);
        char num[16];
        for (uint64_t g = t * grain; g < std::min(n_read_pairs, (t + 1) * grain); g++) {
          if (!read_pairs[g].count) continue;  // (a result without alignments has no read name)
          const uint32_t r = read_pairs[g].r1_read;
          if (r >= reads->n_reads) fail(KSLAM_ERR_ARG, "read pair refers to a read outside the batch");
          out.append(reads->ids + reads->ids_off[r], reads->ids_off[r + 1] - reads->ids_off[r]);
          out += '\t';
          int k = 0;
          uint32_t v = tax_ids[g];
          do { num[k++] = (char)('0' + v % 10); v /= 10; } while (v);
          while (k) out += num[--k];
          out += '\n';
        }
      });
      uint64_t total = 0;
      std::vector<uint64_t> at(n_tasks + 1, 0);
      for (size_t t = 0; t < n_tasks; t++) at[t + 1] = at[t] + part[t].size();
      total = at[n_tasks];
      char *buf = (char *)malloc(total + 1);
      if (!buf) fail(KSLAM_ERR_OOM, "out of host memory");
      Pool::get().tasks(threads, n_tasks, [&](size_t t) {
        if (!part[t].empty()) memcpy(buf + at[t], part[t].data(), part[t].size());
      });
      buf[total] = 0;
      *per_read_text = buf;
      *per_read_len = total;
    }
  });
}

kslam_status kslam_taxonomy_summary(const kslam_taxdb *db, const uint32_t *tax_ids, uint64_t n,
                                    uint64_t num_reads, char **text, uint64_t *text_len) {
  return guarded([&] {
    if (!db || !text || !text_len || (n && !tax_ids)) fail(KSLAM_ERR_ARG, "null argument");
    // combineTaxonomies, src/MetagenomicResults.h:149-177, on (id) records in stable id order: only the group sizes
    // matter downstream, so the ids are counted (chunks sorted in parallel, their runs merged) instead of sorted as a
    // whole.  The reference's loop starts with `test = 0`: the group of id 0 is never emitted and, when no record has
    // id 0, the first record of the sorted array is not counted with its group (:158-176)
    std::vector<std::pair<uint32_t, uint64_t>> groups;  // (taxonomy id, reads)
    if (n == 1) {
      if (tax_ids[0] != 0) groups.push_back({tax_ids[0], 1});
    } else if (n) {
      const int threads = std::max(1, std::min(usable_cpus(), 512));
      const size_t chunks = std::max<size_t>(1, std::min<size_t>((size_t)threads * 4, (n + 65535) / 65536));
      const size_t per = (n + chunks - 1) / chunks;
      std::vector<std::vector<std::pair<uint32_t, uint64_t>>> runs(chunks);
      Pool::get().tasks(threads, chunks, [&](size_t c) {
        const size_t lo = c * per, hi = std::min<size_t>(n, (c + 1) * per);
        if (lo >= hi) return;
        std::vector<uint32_t> part(tax_ids + lo, tax_ids + hi);
        std::sort(part.begin(), part.end());
        for (size_t i = 0, j; i < part.size(); i = j) {
          for (j = i + 1; j < part.size() && part[j] == part[i]; j++) {}
          runs[c].push_back({part[i], j - i});
        }
      });
      std::vector<std::pair<uint32_t, uint64_t>> all;
      for (auto &r : runs) all.insert(all.end(), r.begin(), r.end());
      std::sort(all.begin(), all.end());
      for (size_t i = 0, j; i < all.size(); i = j) {
        uint64_t count = 0;
        for (j = i; j < all.size() && all[j].first == all[i].first; j++) count += all[j].second;
        if (i == 0 && all[i].first != 0) count--;       // the first record of the sorted array
        if (all[i].first != 0 && count) groups.push_back({all[i].first, count});
      }
    }
    // sortResults, src/MetagenomicResults.h:254-262
    std::sort(groups.begin(), groups.end(), [](const std::pair<uint32_t, uint64_t> &a,
                                               const std::pair<uint32_t, uint64_t> &b) {
      return a.second == b.second ? a.first < b.first : a.second > b.second;
    });
    // writeAbbreviatedResultsFile, src/MetagenomicResults.h:237-248 (ostream << double = "%g")
    std::string out;
    char num[64];
    const unsigned reads32 = (unsigned)num_reads;  // the reference's `const unsigned numReads`
    for (auto &g : groups) {
      if (db->known(g.first)) out += db->name[db->node(g.first)];
      snprintf(num, sizeof num, "\t%g\n", g.second * 100.0 / reads32);
      out += num;
    }
    *text = dup_text(out, text_len);
  });
}
kslam_status kslam_taxreport_create(kslam_taxreport **out) {
  return guarded([&] {
    if (!out) fail(KSLAM_ERR_ARG, "null argument");
    *out = new kslam_taxreport();
  });
}

void kslam_taxreport_free(kslam_taxreport *report) { delete report; }

// getResultFromPairedOverlaps for every read pair of the batch, src/MetagenomicResults.h:88-112
kslam_status kslam_taxreport_add_batch(kslam_taxreport *report, const kslam_reads_view *reads,
                                       const kslam_index_view *index, const kslam_read_pair *read_pairs,
                                       uint64_t n_read_pairs, const kslam_paired_overlap *pairs, uint64_t n_pairs,
                                       const uint32_t *tax_ids) {
  return guarded([&] {
    if (!report || !reads || !index || (n_read_pairs && (!read_pairs || !pairs || !tax_ids)))
      fail(KSLAM_ERR_ARG, "null argument");
    if (!reads->ids || !reads->ids_off) fail(KSLAM_ERR_ARG, "the report needs the read identifiers");
    const GeneOrder ord{index};
    const size_t base = report->size();
    if (n_read_pairs == 0) return;
    // pass 1, parallel: every record's genes (best-overlapping gene of each alignment pair, sorted, unique) into the
    // task's own list; pass 2: the lists and the read names are appended to the report's pools in record order
    const int threads = std::max(1, std::min(usable_cpus(), 512));
    const uint64_t grain = 8192, n_tasks = (n_read_pairs + grain - 1) / grain;
    std::vector<std::vector<uint64_t>> task_genes(n_tasks);
    std::vector<uint32_t> n_genes(n_read_pairs, 0), name_len(n_read_pairs, 0);
    report->tax.resize(base + n_read_pairs, 0);
    report->has_read.resize(base + n_read_pairs, 0);
    Pool::get().tasks(threads, n_tasks, [&](size_t t) {
      std::vector<uint64_t> &out = task_genes[t];
      for (uint64_t g = t * grain; g < std::min(n_read_pairs, (t + 1) * grain); g++) {
        const kslam_read_pair &rp = read_pairs[g];
        if (rp.first + rp.count > n_pairs) fail(KSLAM_ERR_ARG, "read pair slice outside the pairs array");
        if (rp.count == 0) continue;                      // an empty result: id 0, no read, no genes (:93)
        const size_t first = out.size();
        for (uint64_t k = 0; k < rp.count; k++) {
          const kslam_paired_overlap &p = pairs[rp.first + k];
          if (p.entry >= index->n_entries) fail(KSLAM_ERR_ARG, "alignment pair refers outside the index");
          const int64_t gene = best_gene_of(index, p.entry, p.ref_start, p.ref_end);
          if (gene >= 0) out.push_back((uint64_t)gene);
        }
        std::sort(out.begin() + first, out.end(), [&](uint64_t a, uint64_t b) { return ord.less(a, b); });
        out.erase(std::unique(out.begin() + first, out.end(), [&](uint64_t a, uint64_t b) { return ord.equal(a, b); }), out.end());
        n_genes[g] = (uint32_t)(out.size() - first);
        if (rp.r1_read >= reads->n_reads) fail(KSLAM_ERR_ARG, "read pair refers to a read outside the batch");
        name_len[g] = (uint32_t)(reads->ids_off[rp.r1_read + 1] - reads->ids_off[rp.r1_read]);
        report->has_read[base + g] = 1;
        report->tax[base + g] = tax_ids[g];
      }
    });
    report->name_off.resize(base + n_read_pairs + 1);
    report->gene_off.resize(base + n_read_pairs + 1);
    uint64_t no = report->name_off[base], go = report->gene_off[base];
    for (uint64_t g = 0; g < n_read_pairs; g++) {
      no += name_len[g];
      go += n_genes[g];
      report->name_off[base + g + 1] = no;
      report->gene_off[base + g + 1] = go;
    }
    report->names.resize(no);
    report->genes.resize(go);
    Pool::get().tasks(threads, n_tasks, [&](size_t t) {
      const uint64_t g0 = t * grain, g1 = std::min(n_read_pairs, (t + 1) * grain);
      if (!task_genes[t].empty())
        memcpy(report->genes.data() + report->gene_off[base + g0], task_genes[t].data(), task_genes[t].size() * sizeof(uint64_t));
      for (uint64_t g = g0; g < g1; g++)
        if (name_len[g])
          memcpy(report->names.data() + report->name_off[base + g], reads->ids + reads->ids_off[read_pairs[g].r1_read], name_len[g]);
    });
  });
}

kslam_status kslam_taxreport_xml(const kslam_taxreport *report, const kslam_index_view *index,
                                 const kslam_gene_extras *extras, const kslam_taxdb *db, uint64_t num_reads,
                                 char **text, uint64_t *text_len) {
  return guarded([&] {
    if (!report || !index || !db || !text || !text_len) fail(KSLAM_ERR_ARG, "null argument");
    const GeneOrder ord{index};
    struct Counted { uint64_t gene; int count; };
    struct Taxon { uint32_t tax = 0; size_t from = 0, to = 0; uint64_t n_reads = 0; std::string xml; };
    const size_t n = report->size();
    const int threads = std::max(1, std::min(usable_cpus(), 512));
    // combineTaxonomies, src/MetagenomicResults.h:149-177: records in stable taxonomy-id order (records of equal id in
    // input order: see the header).  A stable LSD radix sort of the record numbers, 16 bits of the id per pass, the
    // chunks of a pass counted and scattered in parallel -- the same permutation as std::stable_sort by id
    std::vector<uint32_t> order(n), other;
    {
      uint32_t max_tax = 0;
      for (size_t i = 0; i < n; i++) max_tax = std::max(max_tax, report->tax[i]);
      const size_t chunks = std::max<size_t>(1, std::min<size_t>((size_t)threads * 2, (n + 65535) / 65536));
      const size_t per = (n + chunks - 1) / std::max<size_t>(chunks, 1);
      Pool::get().tasks(threads, chunks, [&](size_t c) {
        for (size_t i = c * per; i < std::min(n, (c + 1) * per); i++) order[i] = (uint32_t)i;
      });
      if (max_tax) other.resize(n);
      std::vector<uint32_t> hist;
      for (int shift = 0; shift < 32 && (max_tax >> shift); shift += 16) {
        hist.assign(chunks * 65536, 0);
        Pool::get().tasks(threads, chunks, [&](size_t c) {
          uint32_t *h = hist.data() + c * 65536;
          for (size_t i = c * per; i < std::min(n, (c + 1) * per); i++) h[(report->tax[order[i]] >> shift) & 0xFFFF]++;
        });
        uint64_t run = 0;
        for (size_t d = 0; d < 65536; d++)
          for (size_t c = 0; c < chunks; c++) {
            const uint32_t t = hist[c * 65536 + d];
            hist[c * 65536 + d] = (uint32_t)run;
            run += t;
          }
        Pool::get().tasks(threads, chunks, [&](size_t c) {
          uint32_t *h = hist.data() + c * 65536;
          for (size_t i = c * per; i < std::min(n, (c + 1) * per); i++) {
            const uint32_t r = order[i];
            other[h[(report->tax[r] >> shift) & 0xFFFF]++] = r;
          }
        });
        order.swap(other);
      }
    }
    // The reference sorts with __gnu_parallel::sort, which is NOT stable, and one thing in its output depends on that: its
    // grouping loop (below) leaves the record at the FRONT of the sorted vector out of its group when no record has id 0.
    // Which record that is depends on the sort: with one thread (and below 1 000 records with any number) the parallel
    // sort IS std::sort, whose front element kslam_gnu::front_after_sort names exactly; with more threads the reference's
    // multiway mergesort picks by its thread count, and its XML differs from run to run of the reference itself (measured:
    // the C1 golden at 1 and 8 threads differ in exactly this read).  The product writes what the one-thread reference
    // writes (tests/golden/c1_golden.json): the stable order above decides the groups, this decides who stands in front.
    if (n && report->tax[order[0]] != 0) {
      struct KeyRec { uint32_t key, rec; };
      std::vector<KeyRec> kr(n);
      Pool::get().tasks(threads, (n + 65535) / 65536, [&](size_t c) {
        for (size_t i = c * 65536; i < std::min(n, (c + 1) * 65536); i++) kr[i] = KeyRec{report->tax[i], (uint32_t)i};
      });
      const KeyRec *f = kslam_gnu::front_after_sort(kr.data(), kr.data() + n, [](const KeyRec &a, const KeyRec &b) { return a.key < b.key; });
      const uint32_t front = f->rec;
      for (size_t i = 0; i < n && report->tax[order[i]] == report->tax[order[0]]; i++)
        if (order[i] == front) {
          std::swap(order[0], order[i]);
          break;
        }
    }
    // the groups, cut by the reference's loop (:158-176: `test` starts at 0, so the group of id 0 is never combined and,
    // when no record has id 0, the very first record stays out of its group)
    std::vector<Taxon> taxa;
    auto group = [&](size_t from, size_t to) {
      Taxon t;
      t.tax = report->tax[order[from]];
      t.from = from;
      t.to = to;
      taxa.push_back(std::move(t));
    };
    if (n) {
      uint32_t test = 0;
      size_t start = 0;
      for (size_t i = 1; i < n; i++) {
        const uint32_t id = report->tax[order[i]];
        if (id != test) {
          if (test != 0) group(start, i);
          test = id;
          start = i;
        }
      }
      if (report->tax[order[start]] != 0) group(start, n);
    }
    auto locus = [&](uint64_t g) { return extras ? col(extras->gene_locus_tag, extras->gene_locus_tag_off, g) : std::string_view(); };
    const unsigned reads32 = (unsigned)num_reads;   // the reference's `const unsigned numReads`
    // one taxon: combineRangeOfIdentifiedTaxonomy (:118-142), its share of sortResults (:263-273) and of getXML
    // (:302-366).  `sorted_reads` arrives sorted when the caller did that in parallel (large groups)
    auto finish = [&](Taxon &t, std::vector<std::string_view> &reads, bool reads_sorted) {
      std::vector<uint64_t> all;
      for (size_t i = t.from; i < t.to; i++) {
        const uint32_t r = order[i];
        all.insert(all.end(), report->genes.begin() + report->gene_off[r], report->genes.begin() + report->gene_off[r + 1]);
      }
      std::sort(all.begin(), all.end(), [&](uint64_t a, uint64_t b) { return ord.less(a, b); });
      std::vector<Counted> genes;
      for (size_t i = 0; i < all.size(); i++) {
        if (!genes.empty() && ord.equal(genes.back().gene, all[i])) genes.back().count++;
        else genes.push_back(Counted{all[i], 1});
      }
      if (!reads_sorted) std::sort(reads.begin(), reads.end());
      std::sort(genes.begin(), genes.end(), [&](const Counted &a, const Counted &b) {
        if (a.count == b.count) {
          const uint32_t sa = (uint32_t)index->gene_start[a.gene], sb = (uint32_t)index->gene_start[b.gene];
          if (sa == sb) return locus(a.gene) < locus(b.gene);
          return sa < sb;
        }
        return a.count > b.count;
      });
      t.n_reads = reads.size();
      std::string &out = t.xml;
      size_t name_bytes = 0;
      for (const std::string_view &r : reads) name_bytes += r.size();
      out.reserve(512 + genes.size() * 256 + reads.size() * 20 + name_bytes + name_bytes / 8);
      out += "<taxon>\n  <abundance numReads=\"";
      out += std::to_string(reads.size());
      out += "\">";
      out += std::to_string(reads.size() * 100.0 / reads32);
      out += "</abundance>\n  <taxonomyID>";
      out += std::to_string(t.tax);
      out += "</taxonomyID>\n  <lineage>";
      xml_escaped(out, lineage_of(*db, t.tax));
      out += "</lineage>\n  <name>";
      if (db->known(t.tax)) xml_escaped(out, db->name[db->node(t.tax)]);
      out += "</name>\n  <genes>\n";
      for (const Counted &c : genes) {
        const uint64_t g = c.gene;
        out += "    <gene protein=\"";
        xml_escaped(out, ord.protein(g));
        out += "\" locus=\"";
        xml_escaped(out, locus(g));
        out += "\" product=\"";
        xml_escaped(out, ord.product(g));
        out += "\" GeneID=\"";
        out += std::to_string(extras && extras->gene_id ? extras->gene_id[g] : 0u);
        out += "\" reference=\"";
        xml_escaped(out, extras ? col(extras->gene_reference, extras->gene_reference_off, g) : std::string_view());
        out += "\" numReads=\"";
        out += std::to_string(c.count);
        out += "\" cdsStart=\"";
        out += std::to_string((uint32_t)index->gene_start[g]);
        out += "\" cdsEnd=\"";
        out += std::to_string((uint32_t)index->gene_stop[g]);
        out += "\">";
        xml_escaped(out, ord.name(g));
        out += "</gene>\n";
      }
      out += "  </genes>\n  <reads>\n";
      for (const std::string_view &r : reads) {
        out += "    <read>";
        xml_escaped(out, r);
        out += "</read>\n";
      }
      out += "  </reads>\n</taxon>\n";
    };
    auto reads_of = [&](const Taxon &t, std::vector<std::string_view> &reads) {
      reads.clear();
      reads.reserve(t.to - t.from);
      for (size_t i = t.from; i < t.to; i++) {
        const uint32_t r = order[i];
        if (report->has_read[r])
          reads.emplace_back(report->names.data() + report->name_off[r], report->name_off[r + 1] - report->name_off[r]);
      }
    };
    // taxa in parallel, the largest first; a group too large to be one task's (a run dominated by one organism) has its
    // read names sorted by all threads first (equal names are indistinguishable, so any correct sort gives the text)
    std::vector<uint32_t> by_size(taxa.size());
    for (size_t i = 0; i < taxa.size(); i++) by_size[i] = (uint32_t)i;
    std::sort(by_size.begin(), by_size.end(), [&](uint32_t a, uint32_t b) {
      const size_t sa = taxa[a].to - taxa[a].from, sb = taxa[b].to - taxa[b].from;
      return sa == sb ? a < b : sa > sb;
    });
    const size_t large = std::max<size_t>(1 << 18, n / std::max(1, threads));
    size_t n_large = 0;
    while (threads > 1 && n_large < by_size.size() && taxa[by_size[n_large]].to - taxa[by_size[n_large]].from > large) n_large++;
    for (size_t k = 0; k < n_large; k++) {
      Taxon &t = taxa[by_size[k]];
      std::vector<std::string_view> reads;
      reads_of(t, reads);
      const size_t parts = (size_t)threads, m = reads.size(), step = (m + parts - 1) / parts;
      Pool::get().tasks(threads, parts, [&](size_t c) {
        if (c * step < m) std::sort(reads.begin() + c * step, reads.begin() + std::min(m, (c + 1) * step));
      });
      for (size_t width = step; width < m; width *= 2) {
        const size_t merges = (m + 2 * width - 1) / (2 * width);
        Pool::get().tasks(threads, merges, [&](size_t c) {
          const size_t lo = c * 2 * width, mid = std::min(m, lo + width), hi = std::min(m, lo + 2 * width);
          if (mid < hi) std::inplace_merge(reads.begin() + lo, reads.begin() + mid, reads.begin() + hi);
        });
      }
      finish(t, reads, true);
    }
    Pool::get().tasks(threads, taxa.size() - n_large, [&](size_t k) {
      Taxon &t = taxa[by_size[n_large + k]];
      std::vector<std::string_view> reads;
      reads_of(t, reads);
      finish(t, reads, false);
    });
    // sortResults, src/MetagenomicResults.h:254-262 (a total order: taxonomy ids are distinct)
    std::vector<uint32_t> rank(taxa.size());
    for (size_t i = 0; i < taxa.size(); i++) rank[i] = (uint32_t)i;
    std::sort(rank.begin(), rank.end(), [&](uint32_t a, uint32_t b) {
      return taxa[a].n_reads == taxa[b].n_reads ? taxa[a].tax < taxa[b].tax : taxa[a].n_reads > taxa[b].n_reads;
    });
    std::vector<uint64_t> at(taxa.size() + 1, 0);
    for (size_t i = 0; i < rank.size(); i++) at[i + 1] = at[i] + taxa[rank[i]].xml.size();
    char *buf = (char *)malloc(at.back() + 1);
    if (!buf) fail(KSLAM_ERR_OOM, "out of host memory");
    Pool::get().tasks(threads, rank.size(), [&](size_t i) {
      const std::string &x = taxa[rank[i]].xml;
      memcpy(buf + at[i], x.data(), x.size());
    });
    buf[at.back()] = 0;
    *text = buf;
    *text_len = at.back();
  });
}
}
