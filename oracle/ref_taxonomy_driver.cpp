// ref_taxonomy_driver.cpp -- TEST INFRASTRUCTURE ONLY.
//
// Thin C entry points around the REAL reference taxonomy tree: includes the reference's
// src/TaxonomyDatabase.h (and through it src/sequenceTools.h) where they lie under
// /root/reference -- nothing is copied -- and is compiled by oracle/Makefile into
// oracle/_ref/libtaxonomy_ref.so.
#include <inttypes.h>
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "TaxonomyDatabase.h"

extern "C" {

void *ref_taxdb_open(const char *path) {
  try {
    return new SLAM::TaxonomyDB(std::string(path));
  } catch (...) {
    return nullptr;
  }
}
void ref_taxdb_close(void *t) { delete (SLAM::TaxonomyDB *)t; }
uint64_t ref_taxdb_size(const void *t) { return ((const SLAM::TaxonomyDB *)t)->taxIDsAndEntries.size(); }
uint32_t ref_taxdb_lca(const void *t, const uint32_t *ids, uint64_t n) {
  return ((const SLAM::TaxonomyDB *)t)->getLowestCommonAncestor(std::vector<uint32_t>(ids, ids + n));
}
uint32_t ref_taxdb_parent(const void *t, uint32_t id) { return ((const SLAM::TaxonomyDB *)t)->getParentTaxID(id); }
uint32_t ref_taxdb_at_rank(const void *t, uint32_t id, const char *rank) {
  return ((const SLAM::TaxonomyDB *)t)->getTaxIDAtRank(id, std::string(rank));
}
int32_t ref_taxdb_is_below(const void *t, uint32_t upper, uint32_t lower) {
  return ((const SLAM::TaxonomyDB *)t)->isBelowInTree(upper, lower);
}
int32_t ref_taxdb_is_subspecies(const void *t, uint32_t id) { return ((const SLAM::TaxonomyDB *)t)->isSubSpecies(id); }
char *ref_taxdb_text(const void *tv, uint32_t id, int which) {
  const SLAM::TaxonomyDB *t = (const SLAM::TaxonomyDB *)tv;
  std::string s = which == 0 ? t->getScientificName(id) : which == 1 ? t->getRank(id) : t->getLineage(id);
  char *p = (char *)malloc(s.size() + 1);
  memcpy(p, s.c_str(), s.size() + 1);
  return p;
}
void ref_tax_free(void *p) { free(p); }
}
