// slam_main.cpp -> k-slam_amd/SLAM: the reference's command line over the C ABI (the OUTER drop-in boundary,
// SURVEY.md section 8b last row).  Plain C++ + getopt_long, no Boost; links libkslam_hip.so and nothing else.
//
// Replaces, in the reference (citations into /root/reference/):
//   main                              src/main.cpp:24-157   flags, defaults, dispatch
//   metagenomicAnalysis_Low_Mem       src/SLAM.h:159-268    open files, batch loop, end-of-run outputs
//   metagenomicAnalysis               src/SLAM.h:82-157     (--num-reads-at-once 4294967295: one batch of --num-reads)
//   createIndexFromFASTA              src/GenbankTools.h:224-260   (--parse-fasta)
//   log                               src/sequenceTools.h:171-179  (./log.txt, "[t = 0.00s]\t<message>")
// Same flag names and defaults (src/main.cpp:36-71), same files: <sam-file>, <output-file>, <output-file>_abbreviated,
// <output-file>_PerRead, ./log.txt.  Not built: --parse-genbank, --parse-taxonomy, --server (database builders and a
// dead code path, out of the hot path's scope).
//
//   SLAM [option] --db=DATABASE R1FILE [R2FILE]
#include <algorithm>
#include <cerrno>
#include <chrono>
#include <climits>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <getopt.h>
#include <string>
#include <sys/stat.h>
#include <thread>
#include <unistd.h>
#include <vector>

#include "../include/kslam_db.h"
#include "../include/kslam_stream.h"

namespace {

// src/sequenceTools.h:140-179: a Log that opens ./log.txt on first use, fixed notation, two decimals
struct Log {
  FILE *f = nullptr;
  std::chrono::steady_clock::time_point t0;
  void line(const std::string &s) {
    if (!f) {
      f = fopen("log.txt", "w");
      if (!f) {
        fprintf(stderr, "unable to open log file\n");
        exit(1);
      }
      t0 = std::chrono::steady_clock::now();
    }
    const double t = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    fprintf(f, "[t = %.2fs]\t%s\n", t, s.c_str());
    fflush(f);
  }
};
Log g_log;
void logl(const std::string &s) { g_log.line(s); }

[[noreturn]] void die(const std::string &m) {
  fprintf(stderr, "SLAM: %s\n", m.c_str());
  logl("error: " + m);
  exit(1);
}

struct Options {
  std::string db, out, sam;
  uint32_t score_threshold = 0, match = 2, mismatch = 3, gap_open = 5, gap_extend = 2;
  uint32_t num_reads = UINT32_MAX, num_reads_at_once = 10000000, num_alignments = 10;
  double score_fraction = 0.95;
  bool sam_xa = false, just_align = false, no_pseudo = false, help = false, version = false, parse_fasta = false;
  int device = 0;
  std::vector<std::string> inputs;
};

uint32_t to_u32(const char *s, const char *name) {
  char *end = nullptr;
  errno = 0;
  const unsigned long long v = strtoull(s, &end, 10);
  if (errno || !end || *end || end == s || v > UINT32_MAX || s[0] == '-') die(std::string("the argument ('") + s + "') for option '--" + name + "' is invalid");
  return (uint32_t)v;
}

void usage(FILE *o) {
  // src/main.cpp:102-113 + the option table of :36-71
  fputs("Usage\tSLAM [option] --db=DATABASE R1FILE R2FILE\n"
        "\tAlign paired reads from R1FILE and R2FILE against DATABASE and perform metagenomic analysis\n"
        "or\tSLAM [option] --db=DATABASE R1FILE\n"
        "\tAlign reads from R1FILE against DATABASE and perform metagenomic analysis\n"
        "Allowed options:\n"
        "  --help                                produce help message\n"
        "  --db arg                              SLAM database directory which reads will be aligned against\n"
        "  --min-alignment-score arg (=0)        alignment score cutoff\n"
        "  --score-fraction-threshold arg (=0.95) screen alignments with scores < this*top score\n"
        "  --match-score arg (=2)                match score\n"
        "  --mismatch-penalty arg (=3)           mismatch penalty (positive)\n"
        "  --gap-open arg (=5)                   gap opening penalty (positive)\n"
        "  --gap-extend arg (=2)                 gap extend penalty (positive)\n"
        "  --num-reads arg (=4294967295)         Number of reads from R1/R2 File to align\n"
        "  --num-reads-at-once arg (=10000000)   Reduce RAM usage by only analysing \"arg\" reads at once, this will increase execution time\n"
        "  --output-file arg                     write to this file instead of stdout\n"
        "  --sam-file arg                        write SAM output to this file\n"
        "  --num-alignments arg (=10)            Number of alignments to report in SAM file\n"
        "  --sam-xa                              only output primary alignment lines, use XA field for secondary alignments\n"
        "  --version                             print version number\n"
        "  --just-align                          only perform alignments, not metagenomics\n"
        "  --no-pseudo-assembly                  do not link alignments together\n"
        "\n", o);
}

Options parse(int argc, char **argv) {
  Options o;
  enum { DB = 256, MINSCORE, FRACTION, MATCH, MISMATCH, GAPO, GAPE, NREADS, ATONCE, OUT, SAM, NALIGN, XA, VERSION, JUST, NOPSEUDO, HELP, INPUT,
         PARSE_FASTA, UNSUPPORTED, DEVICE, IGNORED };
  static const option longopts[] = {
      {"db", required_argument, nullptr, DB}, {"min-alignment-score", required_argument, nullptr, MINSCORE},
      {"score-fraction-threshold", required_argument, nullptr, FRACTION}, {"match-score", required_argument, nullptr, MATCH},
      {"mismatch-penalty", required_argument, nullptr, MISMATCH}, {"gap-open", required_argument, nullptr, GAPO},
      {"gap-extend", required_argument, nullptr, GAPE}, {"num-reads", required_argument, nullptr, NREADS},
      {"num-reads-at-once", required_argument, nullptr, ATONCE}, {"output-file", required_argument, nullptr, OUT},
      {"sam-file", required_argument, nullptr, SAM}, {"num-alignments", required_argument, nullptr, NALIGN},
      {"sam-xa", no_argument, nullptr, XA}, {"version", no_argument, nullptr, VERSION}, {"just-align", no_argument, nullptr, JUST},
      {"no-pseudo-assembly", no_argument, nullptr, NOPSEUDO}, {"help", no_argument, nullptr, HELP},
      {"input-file", required_argument, nullptr, INPUT}, {"parse-fasta", no_argument, nullptr, PARSE_FASTA},
      {"parse-genbank", no_argument, nullptr, UNSUPPORTED}, {"parse-taxonomy", no_argument, nullptr, UNSUPPORTED},
      {"server", no_argument, nullptr, UNSUPPORTED}, {"alignment-only", no_argument, nullptr, IGNORED},   // declared, never read (src/main.cpp:80-82)
      {"device", required_argument, nullptr, DEVICE},   // not in the reference: the HIP device ordinal (default 0)
      {nullptr, 0, nullptr, 0}};
  opterr = 0;
  int c;
  // a leading '-' in the option string: positional arguments arrive in order as option 1 (boost's positional "input-file")
  while ((c = getopt_long(argc, argv, "-", longopts, nullptr)) != -1) {
    switch (c) {
      case 1: o.inputs.push_back(optarg); break;
      case INPUT: o.inputs.push_back(optarg); break;
      case DB: o.db = optarg; break;
      case MINSCORE: o.score_threshold = to_u32(optarg, "min-alignment-score"); break;
      case FRACTION: {
        char *end = nullptr;
        o.score_fraction = strtod(optarg, &end);
        if (!end || *end || end == optarg) die(std::string("the argument ('") + optarg + "') for option '--score-fraction-threshold' is invalid");
        break;
      }
      case MATCH: o.match = to_u32(optarg, "match-score"); break;
      case MISMATCH: o.mismatch = to_u32(optarg, "mismatch-penalty"); break;
      case GAPO: o.gap_open = to_u32(optarg, "gap-open"); break;
      case GAPE: o.gap_extend = to_u32(optarg, "gap-extend"); break;
      case NREADS: o.num_reads = to_u32(optarg, "num-reads"); break;
      case ATONCE: o.num_reads_at_once = to_u32(optarg, "num-reads-at-once"); break;
      case OUT: o.out = optarg; break;
      case SAM: o.sam = optarg; break;
      case NALIGN: o.num_alignments = to_u32(optarg, "num-alignments"); break;
      case XA: o.sam_xa = true; break;
      case VERSION: o.version = true; break;
      case JUST: o.just_align = true; break;
      case NOPSEUDO: o.no_pseudo = true; break;
      case HELP: o.help = true; break;
      case PARSE_FASTA: o.parse_fasta = true; break;
      case DEVICE: o.device = (int)to_u32(optarg, "device"); break;
      case IGNORED: break;
      case UNSUPPORTED: die("the database builders --parse-genbank / --parse-taxonomy and --server are not part of this build");
      default: die(std::string("unrecognised option '") + (optind > 0 && optind <= argc ? argv[optind - 1] : "?") + "'");
    }
  }
  return o;
}

struct FileText {   // a whole file in memory -- page-locked (DMA straight from it) unless `pageable`; empty when missing
  char *p = nullptr;
  uint64_t len = 0, cap = 0;
  bool good = false, pageable = false;
  explicit FileText(bool pageable_ = false) : pageable(pageable_) {}
  char *get(uint64_t bytes) { return (char *)(pageable ? malloc(bytes) : kslam_host_alloc(bytes)); }
  void load(const std::string &path) {
    const int fd = open(path.c_str(), O_RDONLY);
    if (fd < 0) return;
    struct stat sb;
    if (fstat(fd, &sb) != 0 || !S_ISREG(sb.st_mode)) {   // a pipe or device: read until EOF into a growing buffer
      std::vector<char> v;
      char buf[1 << 16];
      ssize_t r;
      while ((r = read(fd, buf, sizeof buf)) > 0) v.insert(v.end(), buf, buf + r);
      close(fd);
      cap = v.size() + 64;
      p = get(cap);
      if (!p) die("out of page-locked memory for " + path);
      memcpy(p, v.data(), v.size());
      len = v.size();
      good = true;
      return;
    }
    len = (uint64_t)sb.st_size;
    cap = len + 64;
    p = get(cap);
    if (!p) die("out of page-locked memory for " + path);
    const unsigned n_thr = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(8, len >> 26));
    std::vector<std::thread> th;
    bool ok = true;
    for (unsigned t = 0; t < n_thr; t++)
      th.emplace_back([&, t] {
        uint64_t at = len * t / n_thr;
        const uint64_t end = len * (t + 1) / n_thr;
        while (at < end) {
          const ssize_t r = pread(fd, p + at, (size_t)std::min<uint64_t>(end - at, 1u << 30), (off_t)at);
          if (r <= 0) {
            if (r < 0 && errno == EINTR) continue;
            ok = false;
            return;
          }
          at += (uint64_t)r;
        }
      });
    for (auto &x : th) x.join();
    close(fd);
    if (!ok) die("reading " + path + " failed");
    good = true;
  }
  ~FileText() {
    if (p && pageable) free(p);
    else if (p) kslam_host_free(p, cap);
  }
};

bool write_file(const std::string &path, const char *p, uint64_t n) {
  const int fd = open(path.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644);
  if (fd < 0) return false;
  int user = fd;
  const bool ok = n == 0 || kslam_write_fd(&user, p, n) == 0;
  close(fd);
  return ok;
}

// createIndexFromFASTA, src/GenbankTools.h:224-260: locusTag = the text between '>' and the first space (empty when the
// header has no space, or the space comes first), bases upper-cased, entries without bases dropped, line ends \n, \r\n, \r
int parse_fasta(const Options &o) {
  logl("Parsing FASTA");
  std::vector<std::string> bases, tags;
  for (const std::string &name : o.inputs) {
    logl("Parsing\t" + name);
    FileText t(true);   // no GPU needed for the database builder
    t.load(name);
    if (!t.good) die("unable to open FASTA file");
    std::string cur_b, cur_t;
    auto flush = [&] {
      if (!cur_b.empty()) {
        bases.push_back(std::move(cur_b));
        tags.push_back(cur_t);
      }
      cur_b.clear();
      cur_t.clear();
    };
    uint64_t i = 0;
    while (i < t.len) {   // safeGetline, src/sequenceTools.h:45-73
      uint64_t j = i;
      while (j < t.len && t.p[j] != '\n' && t.p[j] != '\r') j++;
      const char *line = t.p + i;
      const uint64_t n = j - i;
      if (j < t.len && t.p[j] == '\r' && j + 1 < t.len && t.p[j + 1] == '\n') j++;
      i = j + 1;
      if (n == 0) continue;
      if (line[0] == '>') {
        flush();
        const char *sp = (const char *)memchr(line, ' ', n);
        if (sp && sp != line) cur_t.assign(line + 1, sp - line - 1);
      } else {
        cur_b.append(line, n);
      }
    }
    flush();
  }
  for (auto &b : bases)
    for (auto &ch : b) ch = (char)toupper((unsigned char)ch);   // inPlaceConvertToUpperCase, src/sequenceTools.h:117-122
  const uint64_t n = bases.size();
  std::string all_b, all_t;
  std::vector<uint64_t> b_off(n + 1, 0), t_off(n + 1, 0), gene_first(n + 1, 0);
  std::vector<uint32_t> tax(n, 0);
  for (uint64_t e = 0; e < n; e++) {
    all_b += bases[e];
    all_t += tags[e];
    b_off[e + 1] = all_b.size();
    t_off[e + 1] = all_t.size();
  }
  kslam_db_columns c;
  memset(&c, 0, sizeof c);
  c.index.n_entries = n;
  c.index.bases = all_b.data();
  c.index.bases_off = b_off.data();
  c.index.locus_tag = all_t.data();
  c.index.locus_tag_off = t_off.data();
  c.index.taxonomy_id = tax.data();
  c.index.gene_first = gene_first.data();
  if (kslam_db_write(o.out.c_str(), &c, 17) != KSLAM_OK) die(kslam_tail_last_error());
  return 0;
}

std::string cat(const std::string &a, uint64_t v, const std::string &b) { return a + std::to_string(v) + b; }

int run(const Options &o, const std::string &command_line) {
  const std::string r1 = o.inputs[0], r2 = o.inputs.size() > 1 ? o.inputs[1] : std::string();
  const bool paired = !r2.empty(), want_sam = !o.sam.empty();
  logl("Performing metagenomic analysis");
  if (o.db.empty()) die("the option '--db' is required but missing");
  // ---- taxDB + database (src/SLAM.h:172-176) ----
  kslam_taxdb *taxdb = nullptr;
  if (!o.just_align) {
    logl("Building taxonomy index");
    FileText t(true);
    t.load(o.db + "/taxDB");
    if (!t.good) die("unable to open taxonomy index file");
    if (kslam_taxdb_parse(t.p, t.len, &taxdb) != KSLAM_OK) die(kslam_tail_last_error());
    logl(cat("Built a taxonomy tree with ", kslam_taxdb_size(taxdb), " nodes"));
  }
  kslam_db *db = nullptr;
  if (kslam_db_load((o.db + "/database").c_str(), 0, &db) != KSLAM_OK) die(std::string("database: ") + kslam_tail_last_error());
  const kslam_db_columns *cols = kslam_db_view(db);
  const kslam_index_view *index = &cols->index;
  kslam_params kp;
  memset(&kp, 0, sizeof kp);
  kp.match = o.match;
  kp.mismatch = o.mismatch;
  kp.gap_open = o.gap_open;
  kp.gap_extend = o.gap_extend;
  kp.score_threshold = o.score_threshold;
  kp.report_cigar = want_sam ? 1 : 0;   // reportCigar = a SAM file was asked for (src/SLAM.h:169)
  kp.device = o.device;
  kslam_ctx *ctx = nullptr;
  if (kslam_create(&kp, &ctx) != KSLAM_OK) die(std::string("GPU context: ") + (ctx ? kslam_last_error(ctx) : "kslam_create failed"));
  logl("Getting k-mers from index");
  if (kslam_set_index(ctx, index->n_entries, kslam_db_entry_bases(db), kslam_db_entry_lengths(db)) != KSLAM_OK)
    die(std::string("index: ") + kslam_last_error(ctx));
  // ---- the input streams (src/SLAM.h:178-188): a file that cannot be read is logged and behaves as an empty stream ----
  FileText t1, t2;
  t1.load(r1);
  if (!t1.good) logl("FASTQ file " + r1 + " bad");
  if (paired) {
    t2.load(r2);
    if (!t2.good) logl("FASTQ file " + r2 + " bad");
  }
  // ---- outputs ----
  int sam_fd = -1, per_read_fd = -1;
  char *header = nullptr;
  uint64_t header_len = 0;
  if (want_sam) {
    sam_fd = open(o.sam.c_str(), O_RDWR | O_CREAT | O_TRUNC, 0644);   // O_RDWR: the writer may map the file (KSLAM_WRITER=mmap)
    if (sam_fd < 0) die("unable to open SAM file " + o.sam);
    if (kslam_sam_header(index, command_line.c_str(), &header, &header_len) != KSLAM_OK) die(kslam_tail_last_error());
  }
  kslam_taxreport *report = nullptr;
  if (!o.just_align) {
    per_read_fd = open((o.out + "_PerRead").c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644);
    if (per_read_fd < 0) die("unable to open " + o.out + "_PerRead");
    if (kslam_taxreport_create(&report) != KSLAM_OK) die(kslam_tail_last_error());
  }
  kslam_stream_params sp;
  memset(&sp, 0, sizeof sp);
  sp.pairs_per_batch = o.num_reads_at_once;
  sp.max_pairs_total = o.num_reads == UINT32_MAX ? 0 : o.num_reads;
  if (o.num_reads_at_once == UINT32_MAX && sp.max_pairs_total) sp.pairs_per_batch = sp.max_pairs_total;   // metagenomicAnalysis: one batch
  sp.tail.score_threshold = o.score_threshold;
  sp.tail.num_sam_alignments = o.num_alignments;
  sp.tail.score_fraction = o.score_fraction;
  sp.tail.pseudo_assembly = o.no_pseudo ? 0 : 1;
  sp.tail.sam_xa = o.sam_xa ? 1 : 0;
  sp.tail.report_cigar = want_sam ? 1 : 0;
  sp.tail.paired = paired ? 1 : 0;
  sp.sam_fd = sam_fd;
  sp.per_read_fd = per_read_fd;
  sp.sam_header = header;
  sp.sam_header_len = header_len;
  if (paired)
    logl("Getting reads from FASTQ files " + r1 + " and " + r2);
  else
    logl("Getting reads from FASTQ file " + r1);
  logl("Aligning reads to database using k = 32");
  uint32_t *tax_ids = nullptr;
  uint64_t n_tax = 0;
  kslam_stream_stats st;
  memset(&st, 0, sizeof st);
  uint64_t n_pairs = 0;
  if (o.num_reads != 0 && sp.pairs_per_batch != 0) {
    const kslam_status rc = kslam_stream_classify(ctx, index, taxdb, report, t1.p ? t1.p : "", t1.len, paired ? (t2.p ? t2.p : "") : nullptr,
                                                  paired ? t2.len : 0, &sp, &tax_ids, &n_tax, &st);
    if (rc != KSLAM_OK) {
      const char *m = kslam_tail_last_error();
      die(std::string("batch loop: ") + (m && *m ? m : kslam_last_error(ctx)));
    }
    n_pairs = st.n_pairs;
  }
  if (sam_fd >= 0) close(sam_fd);
  if (per_read_fd >= 0) close(per_read_fd);
  logl(cat("Found ", st.n_overlaps, " k-mer overlaps"));
  logl(cat("", st.n_read_pairs_aligned, " entries have k-mer overlaps"));
  if (paired && st.n_batches) logl(cat("Screening all alignment pairs with insert size >= ", st.first_max_insert_size, ""));
  if (want_sam) logl(cat("Writing SAM output (", st.sam_bytes + header_len, " bytes)"));
  logl("Processed\t" + std::to_string(n_pairs) + "\t reads");
  if (o.just_align) {
    logl("Done");
  } else {
    // ---- end of run (src/SLAM.h:256-266): <out>_PerRead is complete; <out> and <out>_abbreviated, or the XML on stdout ----
    logl("Combining taxonomies");
    kslam_gene_extras ex = {cols->gene_locus_tag, cols->gene_locus_tag_off, cols->gene_reference, cols->gene_reference_off, cols->gene_id};
    char *xml = nullptr, *abbr = nullptr;
    uint64_t xml_len = 0, abbr_len = 0;
    if (kslam_taxreport_xml(report, index, &ex, taxdb, n_pairs, &xml, &xml_len) != KSLAM_OK) die(kslam_tail_last_error());
    if (!o.out.empty()) {
      if (!write_file(o.out, xml, xml_len)) die("unable to write " + o.out);
      if (kslam_taxonomy_summary(taxdb, tax_ids, n_tax, n_pairs, &abbr, &abbr_len) != KSLAM_OK) die(kslam_tail_last_error());
      if (!write_file(o.out + "_abbreviated", abbr, abbr_len)) die("unable to write " + o.out + "_abbreviated");
    } else {
      fwrite(xml, 1, xml_len, stdout);
    }
    kslam_free(xml);
    kslam_free(abbr);
    logl("Done");
  }
  kslam_free(tax_ids);
  kslam_free(header);
  if (report) kslam_taxreport_free(report);
  kslam_destroy(ctx);
  kslam_db_free(db);
  if (taxdb) kslam_taxdb_free(taxdb);
  return 0;
}

}  // namespace

int main(int argc, char **argv) {
  std::string command_line;   // src/main.cpp:25-30: the @PG line's CL field
  for (int i = 0; i < argc; i++) {
    if (i) command_line += ' ';
    command_line += argv[i];
  }
  const Options o = parse(argc, argv);
  if (o.version) {   // src/main.cpp:95-98
    puts("1.0");
    return 1;
  }
  if (o.help || argc == 1) {
    usage(stdout);
    return 1;
  }
  if (o.parse_fasta) return parse_fasta(o);
  if (o.inputs.size() == 1 || o.inputs.size() == 2) return run(o, command_line);
  return 0;   // src/main.cpp:136: nothing to do without an input file
}
