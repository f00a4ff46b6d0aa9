// tail_bench.cpp -- host-only timing of the tail (include/kslam_tail.h) on synthetic overlaps.
//   g++ -O3 -std=c++17 -pthread tools/tail_bench.cpp k-slam_amd/host/tail.cpp -o /tmp/tail_bench
//   /tmp/tail_bench [n_pairs] [threads] [iters] [mode: 0 = one malloc'ed text, 1 = writer callback,
//                    2 = the batch loop's host stage (host/stream.cpp): kslam_tail_finish_prepare, THEN the SAM text and the
//                        classification side by side on two threads, read pairs of up to 24 alignment pairs -- the shape
//                        tools/sanitize_host.sh runs under TSan (link host/taxonomy.cpp as well)]
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include <thread>

#include "../include/kslam_taxonomy.h"

int main(int argc, char **argv) {
  const uint64_t n_pairs = argc > 1 ? strtoull(argv[1], 0, 10) : 500000;
  const int threads = argc > 2 ? atoi(argv[2]) : 0, iters = argc > 3 ? atoi(argv[3]) : 5, mode = argc > 4 ? atoi(argv[4]) : 0;
  const uint32_t n_entries = 1250, L = 150;
  std::mt19937_64 rng(7);
  std::vector<kslam_overlap> ov;
  std::vector<uint32_t> pool;
  // per pair: a proper pair on one entry, sometimes extra hits on other entries
  auto add = [&](uint32_t read, uint32_t entry, int32_t rel, bool rc) {
    kslam_overlap o;
    memset(&o, 0, sizeof o);
    o.read = read; o.entry = entry; o.rel = rel; o.revcomp = rc;
    o.score = mode == 2 ? 290 : 250 + rng() % 50;   // mode 2: ties, so that the screens leave large groups
    o.ref_begin = std::max(rel, 0); o.ref_end = o.ref_begin + L - 1;
    o.query_begin = 0; o.query_end = L - 1;
    o.cigar_off = pool.size(); o.cigar_len = 1; pool.push_back(L << 4);
    ov.push_back(o);
  };
  for (int mate = 0; mate < 2; mate++)
    for (uint64_t p = 0; p < n_pairs; p++) {
      std::mt19937_64 pr(p * 977 + 1);
      uint32_t e = pr() % n_entries; int32_t pos = pr() % 7000; bool flip = pr() & 1;
      int extra = (pr() % 4 == 0) ? 1 + pr() % 2 : 0;
      if (mode == 2 && pr() % 8 == 0) extra = 17 + pr() % 7;   // read pairs of more than 16 alignment pairs: introsort really moves records
      std::vector<std::pair<uint32_t, int32_t>> hits{{e, pos}};
      for (int k = 0; k < extra; k++) hits.push_back({(uint32_t)((e + 1 + k) % n_entries), pos});
      std::sort(hits.begin(), hits.end());
      for (auto &h : hits) add(mate * n_pairs + p, h.first, h.second + (mate ? 200 : 0), mate ? !flip : flip);
    }
  const uint64_t n_reads = 2 * n_pairs;
  // every entry carries the same random 8000-base sequence, so each overlap is a true
  // alignment: a read is the entry window at its position (reverse-complemented for
  // revcomp overlaps) with ~1% substitutions
  std::string genome(8000, 'A');
  for (auto &c : genome) c = "ACGT"[rng() & 3];
  std::string ent, bases(n_reads * L, 'A'), qual(n_reads * L, 'I'), ids, loc;
  for (uint32_t e = 0; e < n_entries; e++) ent += genome;
  for (auto &o : ov) {
    if (o.rel < 0 || o.rel + (int)L > 8000) { o.ref_begin = o.rel = 100; o.ref_end = 100 + L - 1; }
    char *r = &bases[(uint64_t)o.read * L];
    for (uint32_t i = 0; i < L; i++) {
      char c = genome[o.rel + i];
      if (rng() % 100 == 0) c = "ACGT"[rng() & 3];
      if (o.revcomp) { c = c == 'A' ? 'T' : c == 'T' ? 'A' : c == 'C' ? 'G' : 'C'; r[L - 1 - i] = c; }
      else r[i] = c;
    }
  }
  for (auto &c : qual) c = (char)(33 + 20 + rng() % 21);
  std::vector<uint64_t> boff(n_reads + 1), ioff(n_reads + 1), eoff(n_entries + 1), loff(n_entries + 1);
  for (uint64_t i = 0; i <= n_reads; i++) boff[i] = i * L;
  for (uint64_t i = 0; i < n_reads; i++) { ioff[i] = ids.size(); ids += "frag" + std::to_string(i % n_pairs); }
  ioff[n_reads] = ids.size();
  std::vector<uint32_t> tax(n_entries, 9);
  for (uint32_t e = 0; e < n_entries; e++) { eoff[e] = e * 8000ull; loff[e] = loc.size(); loc += "NC_" + std::to_string(e); }
  eoff[n_entries] = n_entries * 8000ull; loff[n_entries] = loc.size();
  kslam_reads_view rv{n_reads, bases.data(), boff.data(), qual.data(), boff.data(), ids.data(), ioff.data()};
  kslam_index_view iv;
  memset(&iv, 0, sizeof iv);
  iv.n_entries = n_entries; iv.bases = ent.data(); iv.bases_off = eoff.data();
  iv.locus_tag = loc.data(); iv.locus_tag_off = loff.data(); iv.taxonomy_id = tax.data();
  kslam_tail_params P{0, 10, 0.95, 1, 0, 1, 1, 0, threads};
  printf("%zu overlaps, %llu pairs\n", ov.size(), (unsigned long long)n_pairs);
  if (mode == 2) {
    const char *taxtext = "1\n1\nroot\nno rank\n9\n1\nx\nspecies\n";
    kslam_taxdb *db = nullptr;
    if (kslam_taxdb_parse(taxtext, strlen(taxtext), &db)) { printf("taxdb: %s\n", kslam_tail_last_error()); return 1; }
    for (int it = 0; it < iters; it++) {
      kslam_tail_params front = P;
      front.pseudo_assembly = 0;
      front.stages = 3;
      kslam_read_pair *rp; kslam_paired_overlap *pr; uint64_t nrp, npr;
      if (kslam_tail_pairs(&front, &rv, ov.data(), ov.size(), &rp, &nrp, &pr, &npr, nullptr)) { printf("pairs: %s\n", kslam_tail_last_error()); return 1; }
      kslam_tail_params all = P;              // pseudo-assembly on the host + second screen + per-pair sort: everything that mutates
      if (kslam_tail_finish_prepare(&all, &rv, ov.data(), ov.size(), rp, nrp, pr, npr, 1, nullptr)) { printf("prepare: %s\n", kslam_tail_last_error()); return 1; }
      kslam_tail_params ro = P;
      ro.pseudo_assembly = 0;
      ro.stages = 7u | KSLAM_TAIL_GROUPS_SORTED;
      uint64_t bytes = 0, biggest = 0;
      for (uint64_t g = 0; g < nrp; g++) biggest = std::max<uint64_t>(biggest, rp[g].count);
      std::vector<uint32_t> ids(nrp + 1);
      kslam_status s1 = KSLAM_OK, s2 = KSLAM_OK;
      std::thread tax([&] { char *t = nullptr; uint64_t tl = 0; s2 = kslam_tail_classify(&ro, &rv, &iv, db, rp, nrp, pr, npr, ids.data(), &t, &tl); free(t); });
      s1 = kslam_tail_finish_write_rows(&ro, &rv, &iv, ov.data(), ov.size(), pool.data(), pool.size(), nullptr, nullptr, 0, rp, nrp, pr, npr,
                                        [](void *u, const char *, uint64_t n) -> int { *(uint64_t *)u += n; return 0; }, &bytes, nullptr);
      tax.join();
      if (s1 || s2) { printf("error %d %d %s\n", s1, s2, kslam_tail_last_error()); return 1; }
      printf("iter %d: %llu read pairs (largest %llu alignment pairs), %.1f MB text, text and classification side by side\n", it,
             (unsigned long long)nrp, (unsigned long long)biggest, bytes / 1e6);
      free(rp);   // (kslam_free is free(): the standalone build has no kslam_api.hip)
      free(pr);
    }
    kslam_taxdb_free(db);
    return 0;
  }
  for (int it = 0; it < iters; it++) {
    char *txt; uint64_t len; kslam_tail_stats st;
    auto t0 = std::chrono::steady_clock::now();
    kslam_status rc;
    if (mode == 0)
      rc = kslam_tail_sam(&P, &rv, &iv, ov.data(), ov.size(), pool.data(), pool.size(), &txt, &len, &st);
    else {
      txt = nullptr;
      len = 0;
      rc = kslam_tail_sam_write(&P, &rv, &iv, ov.data(), ov.size(), pool.data(), pool.size(),
                                [](void *u, const char *d, uint64_t n) -> int {
                                  *(uint64_t *)u += n + (unsigned char)d[n - 1];  // touch the chunk's end
                                  return 0;
                                },
                                &len, &st);
      len = st.sam_bytes;
    }
    double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (rc) { printf("error %d %s\n", rc, kslam_tail_last_error()); return 1; }
    printf("iter %d: %.1f ms total | pairing %.1f insert %.1f screens %.1f pseudo %.1f sam %.1f | %.1f MB text, %u threads, %.2f M reads/s\n",
           it, ms, st.ms_pairing, st.ms_insert, st.ms_screens, st.ms_pseudo, st.ms_sam, len / 1e6, st.threads,
           n_reads / ms / 1e3);
    free(txt);
  }
  return 0;
}
