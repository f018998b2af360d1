// Host-side work planning for one batch: which reads share a wavefront, and how
// the haplotypes are chained into streams.  Pure C++ (no HIP), so it is unit-tested
// on CPU through the C ABI's plan introspection and reused by every launch path.
//
// There is no counterpart in the reference: its batch loop is one OpenMP
// `schedule(dynamic,1)` over independent pairs (IntelPairHmm.cc:151-154).
#pragma once
#include <cstdint>
#include <vector>

namespace gklhip {

struct PlanLane {
  int32_t read;   // -1 = idle
  int32_t block;  // RPL-row block of the read held by this lane
};

struct PlanGroup {
  int32_t hap_begin, hap_end, stream_begin, pad_;
};

struct Plan {
  int rows_per_lane = 0;
  int n_chunks = 0;
  // Read packing, compact form (what travels to the device; prep_kernel expands it into the lane map):
  // read r sits in chunk place_chunk[r] from lane place_lane[r] on (-1: a long read, striped kernel);
  // chunk c has chunk_used[c] lanes taken, the rest idle.
  std::vector<int32_t> place_chunk;  // [n_reads]
  std::vector<uint8_t> place_lane;   // [n_reads]
  std::vector<uint8_t> chunk_used;   // [n_chunks]
  std::vector<PlanLane> lanes;      // [n_chunks*64], only with want_lanes (introspection / tests)
  std::vector<PlanGroup> groups;
  std::vector<int32_t> hap_len;     // stream order
  std::vector<int32_t> hap_pos;     // stream index of column 1
  std::vector<int32_t> hap_pos_flat;// same in the FLAT stream: all haplotypes back to back, one drain gap at the end
  std::vector<int32_t> hap_orig;    // stream order -> caller index
  std::vector<int32_t> hap_sidx;    // caller index -> stream order
  std::vector<int32_t> hap_group;   // stream order -> index into groups
  std::vector<int32_t> hap_src;     // stream order: offset of the haplotype's first base in hap_bases
  // Stream layout (built on the device by prep_kernel): per group, for each of its haplotypes hap_len column
  // entries + 1 separator, then 64 idle entries of drain room.
  int32_t n_stream = 0;             // total stream entries
  int32_t n_stream_flat = 0;        // entries of the flat stream (columns + separators + 64 idle)
  std::vector<int32_t> long_reads;  // reads with more than 64*rows_per_lane-1 bases: striped kernel
  int64_t useful_rows = 0;
  int max_read_len = 0;
  int max_hap_len = 0;
};

// Lanes a read of length R occupies at RPL rows per lane: its R rows plus >= 1 pad row.
inline int blocks_for(int R, int rpl) { return (R + rpl) / rpl; }

// Build the haplotype streams (always) and, when rows_per_lane > 0, the read packing.
// Haplotypes are streamed in order of increasing length (ties by index); target_cols:
// desired columns per haplotype group (job length).
// `out` is reused from call to call (its vectors keep their capacity: this runs on the caller's thread).
void build_plan(int n_reads, int n_haps, const int64_t* read_off, const int64_t* hap_off,
                int rows_per_lane, int target_cols, Plan* out, bool want_lanes = false);

// The lane map of a compact packing (what prep_kernel builds on the device).
void expand_lanes(const Plan& p, int n_reads, const int64_t* read_off, std::vector<PlanLane>* lanes);

// Pack the given reads (in the given order) into 64-lane chunks: best-fit-decreasing inside
// consecutive windows of `window` reads, never across windows, so reads that are adjacent in
// `order` (e.g. sorted by how many haplotypes they must be recomputed against) share chunks.
// Appends to lanes; returns the number of chunks created.
int pack_reads_windowed(const int32_t* order, int n, const int64_t* read_off, int rows_per_lane,
                        int window, std::vector<PlanLane>* lanes, int64_t* useful_rows);

// Same packing, compact output: place_chunk / place_lane indexed by read, chunk_used appended per chunk
// (chunk numbers start at chunk_used->size() on entry).  Scratch vectors live in `scratch` across calls.
struct PackScratch {
  std::vector<uint8_t> need_of, sorted_need;
  std::vector<int32_t> sorted, next;
};
int pack_reads_place(const int32_t* order, int n, const int64_t* read_off, int rows_per_lane, int window,
                     int32_t* place_chunk, uint8_t* place_lane, std::vector<uint8_t>* chunk_used,
                     int64_t* useful_rows, PackScratch* scratch);

}  // namespace gklhip
