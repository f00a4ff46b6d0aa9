// Compiles k-slam_amd/host/slam_hot_path.hpp stand-alone with look-alike types (the members the
// reference's Overlap / Alignment / GenbankEntry / FASTQ types have) and, on a GPU box, runs one
// tiny batch through it.  Build (one line): g++ -std=c++11 tests/host_mirror_check.cpp -o /tmp/hmc
//   -Lk-slam_amd -lkslam_hip -Wl,-rpath,$PWD/k-slam_amd
#include <cstdio>
#include "../k-slam_amd/host/slam_hot_path.hpp"

struct Alignment { int32_t ref_begin = 0, ref_end = 0, query_begin = 0, query_end = 0; uint16_t sw_score = 0;
                   int32_t cigarLen = 0; uint32_t* cigar = nullptr; };
struct Overlap { uint32_t readPosInArray = 0, entryPosInArray = 0; int32_t relativePosition = 0; bool revComp = false;
                 Alignment alignment; };
struct Entry { std::string bases; };
struct Index { std::vector<Entry> entries; };
struct Read { std::string bases; };

int main() {
  try {
    kslam_host::HotPath hp(2, 3, 5, 2, 0, true, 0);
    Index idx;
    std::string g;
    uint64_t x = 88172645463325252ull;
    for (int i = 0; i < 5000; i++) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; g.push_back("ACGT"[x & 3]); }
    idx.entries.push_back(Entry{g});
    hp.setIndex(idx);
    std::vector<Read> reads{Read{g.substr(1600, 150)}, Read{g.substr(3216, 150)}};
    std::vector<Overlap> ov = hp.alignToDatabase<Overlap>(reads);
    for (auto& o : ov) {
      std::printf("read %u entry %u rel %d rc %d score %u ref %d..%d cigar", o.readPosInArray, o.entryPosInArray,
                  o.relativePosition, (int)o.revComp, o.alignment.sw_score, o.alignment.ref_begin, o.alignment.ref_end);
      for (int k = 0; k < o.alignment.cigarLen; k++) std::printf(" %u%c", o.alignment.cigar[k] >> 4, "MID"[o.alignment.cigar[k] & 15]);
      std::printf("\n");
      std::free(o.alignment.cigar);
    }
    // the same two reads as one PAIR through the multi-device mirror (two shards on device 0)
    kslam_host::HotPathMulti mp({0, 0}, 2, 3, 5, 2, 0, true);
    mp.setIndex(idx);
    std::vector<Overlap> mv = mp.alignToDatabase<Overlap>(reads, /*pairedData=*/true);
    bool same = mv.size() == ov.size();
    for (size_t i = 0; same && i < mv.size(); i++) {
      same = mv[i].readPosInArray == ov[i].readPosInArray && mv[i].relativePosition == ov[i].relativePosition &&
             mv[i].alignment.sw_score == ov[i].alignment.sw_score && mv[i].alignment.cigarLen == ov[i].alignment.cigarLen;
      std::free(mv[i].alignment.cigar);
    }
    std::printf("multi %s\n", same ? "identical" : "DIFFERENT");
    return (same && ov.size() == 2 && ov[0].alignment.sw_score == 300 && ov[1].relativePosition == 3216) ? 0 : 1;
  } catch (const std::exception& e) {
    std::printf("error: %s\n", e.what());
    return 2;
  }
}
