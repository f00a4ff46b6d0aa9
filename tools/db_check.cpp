// db_check.cpp -- small driver over include/kslam_db.h for the sanitizer pass (tools/sanitize_host.sh):
// writes a synthetic archive, loads it with several thread counts, writes it back, compares, and
// feeds the parser truncated / corrupted copies.   db_check [n_entries] [dir]
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "../include/kslam_db.h"

static std::string slurp(const std::string &p) {
  FILE *f = fopen(p.c_str(), "rb");
  std::string s;
  char buf[1 << 16];
  size_t k;
  while (f && (k = fread(buf, 1, sizeof buf, f)) > 0) s.append(buf, k);
  if (f) fclose(f);
  return s;
}

int main(int argc, char **argv) {
  const size_t n = argc > 1 ? strtoull(argv[1], 0, 10) : 200;
  const std::string dir = argc > 2 ? argv[2] : "/tmp";
  std::mt19937_64 rng(3);
  std::string bases, locus, gname, gprot, gprod;
  std::vector<uint64_t> boff{0}, loff{0}, gfirst{0}, gnoff{0}, gpoff{0}, groff{0};
  std::vector<uint32_t> tax, gid;
  std::vector<int32_t> gs, ge;
  for (size_t i = 0; i < n; i++) {
    const size_t L = i == 7 ? 70u << 20 : rng() % 5000;   // one long string: the parallel copy path
    for (size_t k = 0; k < L; k++) bases.push_back("ACGT"[rng() & 3]);
    boff.push_back(bases.size());
    locus += "NC_" + std::to_string(i);
    loff.push_back(locus.size());
    tax.push_back((uint32_t)rng());
    for (size_t g = 0; g < rng() % 4; g++) {
      gname += "gene " + std::to_string(g);
      gnoff.push_back(gname.size());
      gprot += "NP_" + std::to_string(rng() % 1000);
      gpoff.push_back(gprot.size());
      gprod += "30S ribosomal protein 1 0 0";
      groff.push_back(gprod.size());
      gs.push_back((int32_t)rng());
      ge.push_back((int32_t)rng());
      gid.push_back((uint32_t)rng());
    }
    gfirst.push_back(gid.size());
  }
  kslam_db_columns c;
  memset(&c, 0, sizeof c);
  c.index.n_entries = n;
  c.index.bases = bases.data(); c.index.bases_off = boff.data();
  c.index.locus_tag = locus.data(); c.index.locus_tag_off = loff.data();
  c.index.taxonomy_id = tax.data();
  c.index.n_genes = gid.size(); c.index.gene_first = gfirst.data();
  c.index.gene_start = gs.data(); c.index.gene_stop = ge.data();
  c.index.gene_name = gname.data(); c.index.gene_name_off = gnoff.data();
  c.index.protein_id = gprot.data(); c.index.protein_id_off = gpoff.data();
  c.index.product = gprod.data(); c.index.product_off = groff.data();
  c.gene_id = gid.data();
  const std::string p1 = dir + "/kslam_db_check.1", p2 = dir + "/kslam_db_check.2";
  if (kslam_db_write(p1.c_str(), &c, 17) != KSLAM_OK) return fprintf(stderr, "write: %s\n", kslam_tail_last_error()), 1;
  const std::string text = slurp(p1);
  for (int threads : {1, 3, 0}) {
    kslam_db *db = nullptr;
    if (kslam_db_load(p1.c_str(), threads, &db) != KSLAM_OK) return fprintf(stderr, "load: %s\n", kslam_tail_last_error()), 1;
    const kslam_db_columns *v = kslam_db_view(db);
    if (v->index.n_entries != n || memcmp(v->index.bases, bases.data(), bases.size()) != 0) return fprintf(stderr, "bases differ\n"), 1;
    if (kslam_db_write(p2.c_str(), v, 17) != KSLAM_OK || slurp(p2) != text) return fprintf(stderr, "round trip differs\n"), 1;
    kslam_db_free(db);
  }
  // damaged copies must fail cleanly
  size_t rejected = 0, tried = 0;
  for (size_t cut : {size_t(0), size_t(10), size_t(40), text.size() / 3, text.size() - 1}) {
    kslam_db *db = nullptr;
    tried++;
    if (kslam_db_parse(text.data(), cut, 2, &db) != KSLAM_OK) rejected++; else kslam_db_free(db);
  }
  for (int k = 0; k < 40; k++) {
    std::string t = text.substr(0, 200000);
    t[rng() % 300] = "x 9\n"[rng() & 3];
    kslam_db *db = nullptr;
    tried++;
    if (kslam_db_parse(t.data(), t.size(), 2, &db) != KSLAM_OK) rejected++; else kslam_db_free(db);
  }
  remove(p1.c_str());
  remove(p2.c_str());
  printf("db_check: %zu entries, %zu bytes, round trips ok, %zu of %zu damaged copies rejected\n", n, text.size(), rejected, tried);
  return 0;
}
