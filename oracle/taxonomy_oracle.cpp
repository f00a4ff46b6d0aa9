// taxonomy_oracle.cpp -- TEST INFRASTRUCTURE ONLY (see oracle/kslam_oracle.h header).
//
// Restatement of the reference's taxonomy stage the way the reference does it: a hash map of
// entries, one root-ward path vector per id, level-by-level comparison
// (src/TaxonomyDatabase.h:166-349), and the per-read / combine / abbreviated-report steps of
// src/MetagenomicResults.h:88-112, 149-177, 237-262, 455-463 reduced to what they do to
// taxonomy ids and read names.
//
// PINNED (tree queries): oracle/ref_taxonomy_driver.cpp includes the reference's own
// src/TaxonomyDatabase.h where it lies (std headers only) -> oracle/_ref/libtaxonomy_ref.so;
// tests/test_taxonomy.py checks LCA, parent, rank, lineage, at-rank, is-below, is-subspecies
// against it on random trees.  PARITY UNPINNED (per-read + summary steps):
// src/MetagenomicResults.h includes Boost-dependent headers; those steps are a few lines each.
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>

namespace {

struct Entry {
  uint32_t id = 0, parent = 0;
  std::string name, rank;
};

struct Tree {
  std::unordered_map<uint32_t, Entry> m;
  uint32_t parent_of(uint32_t id) const {  // getParentTaxID
    auto e = m.find(id);
    if (e != m.end() && e->second.parent != 1) return e->second.parent;
    return 0;
  }
  uint32_t lca(const std::vector<uint32_t> &ids) const {  // getLowestCommonAncestor
    if (ids.empty()) return 0;
    std::vector<std::vector<uint32_t>> paths;
    for (uint32_t id : ids) {
      std::vector<uint32_t> path;
      for (uint32_t t = id; t != 0; t = parent_of(t)) path.push_back(t);
      std::reverse(path.begin(), path.end());
      paths.push_back(path);
    }
    size_t shortest = paths[0].size();
    for (auto &p : paths) shortest = std::min(shortest, p.size());
    uint32_t consensus = 0;
    for (size_t i = 0; i < shortest; i++) {
      uint32_t here = 0;
      for (auto &p : paths) {
        if (here == 0)
          here = p[i];
        else if (here != p[i])
          return consensus;
      }
      consensus = here;
    }
    return consensus;
  }
  std::string name(uint32_t id) const {
    auto e = m.find(id);
    return e == m.end() ? std::string() : e->second.name;
  }
  std::string rank(uint32_t id) const {
    auto e = m.find(id);
    return e == m.end() ? std::string() : e->second.rank;
  }
  std::string lineage(uint32_t id) const {  // getLineage
    std::string l;
    while (true) {
      if (id != 131567) {
        if (l.size()) l.insert(0, "; ");
        l.insert(0, name(id));
        if (rank(id) == "species") l.clear();
      }
      id = parent_of(id);
      if (id == 0) {
        if (l.size()) l.append(".");
        break;
      }
    }
    return l;
  }
};

}  // namespace

extern "C" {

void *orc_taxdb_parse(const char *text, uint64_t len) {  // readTaxonomyIndex
  Tree *t = new Tree();
  std::vector<std::string> lines;
  for (uint64_t p = 0; p < len;) {
    const void *nl = memchr(text + p, '\n', len - p);
    uint64_t e = nl ? (uint64_t)((const char *)nl - text) : len;
    lines.emplace_back(text + p, e - p);
    p = e + 1;
  }
  for (size_t i = 0; i + 3 < lines.size(); i += 4) {
    Entry e;
    e.id = std::stoi(lines[i]);
    e.parent = std::stoi(lines[i + 1]);
    e.name = lines[i + 2];
    e.rank = lines[i + 3];
    t->m.insert({e.id, e});
  }
  return t;
}
void orc_taxdb_free(void *t) { delete (Tree *)t; }
uint32_t orc_taxdb_lca(const void *t, const uint32_t *ids, uint64_t n) {
  return ((const Tree *)t)->lca(std::vector<uint32_t>(ids, ids + n));
}
uint32_t orc_taxdb_parent(const void *t, uint32_t id) { return ((const Tree *)t)->parent_of(id); }

uint32_t orc_taxdb_at_rank(const void *tv, uint32_t id, const char *rank) {  // getTaxIDAtRank
  const Tree *t = (const Tree *)tv;
  auto e = t->m.find(id);
  while (e != t->m.end() && e->second.parent != 1) {
    if (e->second.rank == rank) return e->second.id;
    e = t->m.find(e->second.parent);
  }
  return 0;
}
int32_t orc_taxdb_is_below(const void *tv, uint32_t upper, uint32_t lower) {  // isBelowInTree
  const Tree *t = (const Tree *)tv;
  auto e = t->m.find(lower);
  unsigned level = 0;
  while (e != t->m.end() && e->second.parent != 1) {
    if (e->first == upper) return level;
    e = t->m.find(e->second.parent);
    level++;
  }
  return -1;
}
int32_t orc_taxdb_is_subspecies(const void *tv, uint32_t id) {  // isSubSpecies
  const Tree *t = (const Tree *)tv;
  bool sub = false;
  auto e = t->m.find(id);
  int levels = 0;
  while (e != t->m.end() && e->second.parent != 1) {
    if (e->second.rank == "species") {
      if (levels > 0) sub = true;
      break;
    }
    e = t->m.find(e->second.parent);
    levels++;
  }
  return sub;
}
// which: 0 name, 1 rank, 2 lineage; returns malloc'ed text
char *orc_taxdb_text(const void *tv, uint32_t id, int which) {
  const Tree *t = (const Tree *)tv;
  std::string s = which == 0 ? t->name(id) : which == 1 ? t->rank(id) : t->lineage(id);
  char *p = (char *)malloc(s.size() + 1);
  memcpy(p, s.c_str(), s.size() + 1);
  return p;
}

// combineTaxonomies + sortResults + writeAbbreviatedResultsFile on (id, one read) records
char *orc_taxonomy_summary(const void *tv, const uint32_t *ids, uint64_t n, uint32_t num_reads) {
  const Tree *t = (const Tree *)tv;
  struct Rec {
    uint32_t id;
    uint64_t reads;
  };
  std::vector<Rec> all;
  for (uint64_t i = 0; i < n; i++) all.push_back(Rec{ids[i], 1});
  std::stable_sort(all.begin(), all.end(), [](const Rec &a, const Rec &b) { return a.id < b.id; });
  std::vector<Rec> combined;
  auto combine = [&](size_t b, size_t e) {
    Rec r = all[b];
    for (size_t k = b + 1; k < e; k++) r.reads += all[k].reads;
    return r;
  };
  if (!all.empty()) {
    uint32_t test = 0;
    size_t start = 0;
    for (size_t i = 0; i < all.size(); i++) {
      if (i == 0) continue;
      if (all[i].id != test) {
        if (test != 0) combined.push_back(combine(start, i));
        test = all[i].id;
        start = i;
      }
    }
    if (start != all.size() && all[start].id != 0) combined.push_back(combine(start, all.size()));
  }
  std::sort(combined.begin(), combined.end(), [](const Rec &a, const Rec &b) {
    if (a.reads == b.reads) return a.id < b.id;
    return a.reads > b.reads;
  });
  std::string out;
  char num[64];
  for (auto &r : combined) {
    snprintf(num, sizeof num, "%g", r.reads * 100.0 / num_reads);
    out += t->name(r.id) + "\t" + num + "\n";
  }
  char *p = (char *)malloc(out.size() + 1);
  memcpy(p, out.c_str(), out.size() + 1);
  return p;
}

void orc_tax_free(void *p) { free(p); }
}
