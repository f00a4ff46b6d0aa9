// slam_hot_path.hpp -- host-side mirror of the reference's operator for this path, in the
// reference's own language (C++11), layered on the C ABI of include/kslam.h.
//
// It offers what src/SLAM.h:59-79 offers:
//
//   template <class FASTQType>
//   std::vector<Overlap> alignToDatabase(const std::vector<FASTQType>& reads,
//                                        const GenbankIndex& genbankIndex);
//
// with the same argument meaning (reads[i].bases / genbankIndex.entries[j].bases used verbatim,
// scoring taken from the globals match / misMatch / gapOpen / gapExtend / scoreThreshold /
// reportCigar of src/Globals.h:27-36) and the same result: Overlap records sorted by
// (readPosInArray, entryPosInArray, relativePosition) after the reference's dedupe, each with its
// StripedSmithWaterman::Alignment (malloc-owned BAM cigar, src/ssw_cpp.h:10-87).
// Errors surface as std::runtime_error, like the reference's I/O errors do.
//
// The header is generic over the caller's types so it compiles both inside the reference (with
// its own Overlap / Alignment / GenbankIndex) and stand-alone (tests/host_mirror_check.cpp).
#pragma once
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>
#include "../../include/kslam.h"

namespace kslam_host {

class HotPath {
 public:
  // one context per process and GPU; the index upload + genome k-mer sort happens once
  HotPath(uint32_t match, uint32_t misMatch, uint32_t gapOpen, uint32_t gapExtend,
          uint32_t scoreThreshold, bool reportCigar, int device = 0) {
    kslam_params p;
    std::memset(&p, 0, sizeof p);
    p.match = match; p.mismatch = misMatch; p.gap_open = gapOpen; p.gap_extend = gapExtend;
    p.score_threshold = scoreThreshold; p.report_cigar = reportCigar ? 1 : 0; p.device = device;
    kslam_status st = kslam_create(&p, &ctx_);
    if (st != KSLAM_OK) {
      std::string msg = ctx_ ? kslam_last_error(ctx_) : "kslam_create failed";
      if (ctx_) kslam_destroy(ctx_);
      ctx_ = nullptr;
      throw std::runtime_error("kslam: " + msg);
    }
  }
  ~HotPath() { if (ctx_) kslam_destroy(ctx_); }
  HotPath(const HotPath&) = delete;
  HotPath& operator=(const HotPath&) = delete;

  // const GenbankIndex& : anything with .entries[j].bases (std::string)
  template <class Index> void setIndex(const Index& index) {
    std::vector<const char*> ptr(index.entries.size());
    std::vector<uint64_t> len(index.entries.size());
    for (size_t j = 0; j < index.entries.size(); j++) {
      ptr[j] = index.entries[j].bases.data();
      len[j] = index.entries[j].bases.size();
    }
    check(kslam_set_index(ctx_, ptr.size(), ptr.data(), len.data()));
  }

  // alignToDatabase: OverlapT needs the members of src/Overlap.h:53-74 (readPosInArray,
  // entryPosInArray, relativePosition, revComp, alignment{ref_begin, ref_end, query_begin,
  // query_end, sw_score, cigarLen, cigar})
  template <class OverlapT, class FASTQType>
  std::vector<OverlapT> alignToDatabase(const std::vector<FASTQType>& reads) {
    std::vector<const char*> ptr(reads.size());
    std::vector<uint32_t> len(reads.size());
    for (size_t i = 0; i < reads.size(); i++) {
      ptr[i] = reads[i].bases.data();
      len[i] = (uint32_t)reads[i].bases.size();
    }
    kslam_overlap* ov = nullptr; uint32_t* pool = nullptr; uint64_t n = 0, nc = 0;
    check(kslam_align_batch(ctx_, ptr.size(), ptr.data(), len.data(), &ov, &n, &pool, &nc));
    std::vector<OverlapT> out;
    convert(ov, n, pool, out);
    kslam_free_batch(ctx_, ov, pool);
    return out;
  }

  // kslam_overlap records + pool -> the caller's Overlap objects (each Alignment owns a malloc'ed
  // cigar array, src/ssw_cpp.h:23-77)
  template <class OverlapT>
  static void convert(const kslam_overlap* ov, uint64_t n, const uint32_t* pool, std::vector<OverlapT>& out) {
    out.resize(n);
    for (uint64_t i = 0; i < n; i++) {
      OverlapT& o = out[i];
      o.readPosInArray = ov[i].read;
      o.entryPosInArray = ov[i].entry;
      o.relativePosition = ov[i].rel;
      o.revComp = ov[i].revcomp != 0;
      o.alignment.ref_begin = ov[i].ref_begin;
      o.alignment.ref_end = ov[i].ref_end;
      o.alignment.query_begin = ov[i].query_begin;
      o.alignment.query_end = ov[i].query_end;
      o.alignment.sw_score = ov[i].score;
      o.alignment.cigarLen = (int32_t)ov[i].cigar_len;
      o.alignment.cigar = nullptr;
      if (ov[i].cigar_len) {
        o.alignment.cigar = (uint32_t*)std::malloc(sizeof(uint32_t) * ov[i].cigar_len);
        std::memcpy(o.alignment.cigar, pool + ov[i].cigar_off, sizeof(uint32_t) * ov[i].cigar_len);
      }
    }
  }

 private:
  void check(kslam_status st) {
    if (st != KSLAM_OK) throw std::runtime_error(std::string("kslam: ") + kslam_last_error(ctx_));
  }
  kslam_ctx* ctx_ = nullptr;
};

// The same operator over several GPUs of one node (kslam_multi_*, include/kslam.h): the batch's read
// pairs are sharded, the index is replicated, the result is byte for byte the single-GPU one.
// `pairedData` as the reference's global (src/Globals.h): reads = [R1 block | R2 block].
class HotPathMulti {
 public:
  HotPathMulti(const std::vector<int>& devices, uint32_t match, uint32_t misMatch, uint32_t gapOpen,
               uint32_t gapExtend, uint32_t scoreThreshold, bool reportCigar) {
    kslam_params p;
    std::memset(&p, 0, sizeof p);
    p.match = match; p.mismatch = misMatch; p.gap_open = gapOpen; p.gap_extend = gapExtend;
    p.score_threshold = scoreThreshold; p.report_cigar = reportCigar ? 1 : 0;
    std::vector<int32_t> dv(devices.begin(), devices.end());
    kslam_status st = kslam_multi_create(&p, dv.data(), (uint32_t)dv.size(), &m_);
    if (st != KSLAM_OK) {
      std::string msg = m_ ? kslam_multi_last_error(m_) : "kslam_multi_create failed";
      if (m_) kslam_multi_destroy(m_);
      m_ = nullptr;
      throw std::runtime_error("kslam: " + msg);
    }
  }
  ~HotPathMulti() { if (m_) kslam_multi_destroy(m_); }
  HotPathMulti(const HotPathMulti&) = delete;
  HotPathMulti& operator=(const HotPathMulti&) = delete;

  template <class Index> void setIndex(const Index& index) {
    std::vector<const char*> ptr(index.entries.size());
    std::vector<uint64_t> len(index.entries.size());
    for (size_t j = 0; j < index.entries.size(); j++) {
      ptr[j] = index.entries[j].bases.data();
      len[j] = index.entries[j].bases.size();
    }
    check(kslam_multi_set_index(m_, ptr.size(), ptr.data(), len.data()));
  }

  template <class OverlapT, class FASTQType>
  std::vector<OverlapT> alignToDatabase(const std::vector<FASTQType>& reads, bool pairedData) {
    std::vector<const char*> ptr(reads.size());
    std::vector<uint32_t> len(reads.size());
    for (size_t i = 0; i < reads.size(); i++) {
      ptr[i] = reads[i].bases.data();
      len[i] = (uint32_t)reads[i].bases.size();
    }
    kslam_overlap* ov = nullptr; uint32_t* pool = nullptr; uint64_t n = 0, nc = 0;
    check(kslam_multi_align_batch(m_, ptr.size(), ptr.data(), len.data(), pairedData ? 1 : 0, &ov, &n, &pool, &nc));
    std::vector<OverlapT> out;
    HotPath::convert(ov, n, pool, out);
    kslam_multi_free_batch(m_, ov, pool);
    return out;
  }

 private:
  void check(kslam_status st) {
    if (st != KSLAM_OK) throw std::runtime_error(std::string("kslam: ") + kslam_multi_last_error(m_));
  }
  kslam_multi* m_ = nullptr;
};

}  // namespace kslam_host
