#include "pairhmm_plan.h"

#include <algorithm>
#include <cstdlib>
#include <numeric>

namespace gklhip {

namespace {
constexpr int kLanes = 64;
constexpr int kWantedJobs = 4096;
}

void build_plan(int n_reads, int n_haps, const int64_t* read_off, const int64_t* hap_off,
                int rows_per_lane, int target_cols, Plan* out, bool want_lanes) {
  Plan& p = *out;
  // reset, keeping every vector's capacity (a fresh Plan per call cost more than the planning itself:
  // 1.5 MB of lane map allocated, filled and freed per 10k-read batch)
  p.rows_per_lane = rows_per_lane;
  p.n_chunks = 0;
  p.n_stream = p.n_stream_flat = 0;
  p.useful_rows = 0;
  p.max_read_len = p.max_hap_len = 0;
  p.groups.clear(); p.long_reads.clear(); p.lanes.clear(); p.chunk_used.clear();

  // ---- haplotype streams: caller order, cut into groups of ~target_cols columns ----
  p.hap_len.resize(n_haps);
  p.hap_pos.resize(n_haps);
  p.hap_pos_flat.resize(n_haps);
  p.hap_orig.resize(n_haps);
  p.hap_sidx.resize(n_haps);
  p.hap_group.resize(n_haps);
  p.hap_src.resize(n_haps);
  int64_t total_cols = 0;
  std::iota(p.hap_orig.begin(), p.hap_orig.end(), 0);
  std::stable_sort(p.hap_orig.begin(), p.hap_orig.end(), [&](int32_t a, int32_t b) {
    return hap_off[a + 1] - hap_off[a] < hap_off[b + 1] - hap_off[b];
  });
  for (int k = 0; k < n_haps; k++) {
    const int h = p.hap_orig[k];
    const int len = (int)(hap_off[h + 1] - hap_off[h]);
    p.hap_len[k] = len;
    p.hap_sidx[h] = k;
    p.max_hap_len = std::max(p.max_hap_len, len);
    total_cols += len + 1;
  }
  if (target_cols < 256) target_cols = 256;
  int n_groups = (int)((total_cols + target_cols - 1) / target_cols);
  bool equal_split = true, graded = false;
  if (rows_per_lane > 0) {
    // Small batches (one active region of GATK is a few hundred reads x a few dozen haplotypes): a job is one
    // (chunk, group) and the chip has 1024 SIMDs x 4 wavefront slots, so cut the stream finer -- down to one
    // haplotype per group -- until there are about that many jobs; the price is 63 fill/drain steps per job.
    // (lanes needed by all reads that fit a chunk: total rows / rows per lane, + 1 pad row and < 1 lane of rounding each)
    int64_t blocks = 0;
    {
      int shift = -1;
      for (int b = 0; b < 8; b++) if ((1 << b) == rows_per_lane) shift = b;
      const int long_from = kLanes * rows_per_lane;
      for (int r = 0; r < n_reads; r++) {
        const int R = (int)(read_off[r + 1] - read_off[r]);
        p.max_read_len = std::max(p.max_read_len, R);
        if (R < long_from) blocks += shift >= 0 ? (R + rows_per_lane) >> shift : blocks_for(R, rows_per_lane);
      }
    }
    const int64_t chunks_est = std::max<int64_t>(1, (blocks + kLanes - 1) / kLanes);
    static const int wanted_env = [] { const char* v = getenv("GKLHIP_WANTED_JOBS"); return v ? atoi(v) : 0; }();
    // Mid-size calls (a few hundred reads x tens of haplotypes: up to ~650 chunks) and tall ones (thousands of reads x a
    // few haplotypes: up to four full-size groups of columns; 5000 x 16: 938 -> 686 us) get three times as many, shorter
    // jobs of GRADED length (below): 250 x 128 532 -> 320 us, 450 x 100 563 -> 435, 1000 x 50 583 -> 467 (fp32
    // kernel; docs/NOTES.md 23).  A count just above one set of wavefront slots (4096 = 1024 SIMDs x 4) is cut back to
    // one set: the few jobs of a second set would double the time.
    const int64_t slots = kWantedJobs, chunks_hi = chunks_est + chunks_est / 100 + 1;  // (the packing may need a chunk more than the estimate)
    const int64_t wanted = wanted_env > 0 ? wanted_env : (chunks_est <= 650 || total_cols <= 4 * (int64_t)target_cols ? 3 * slots : slots);
    int64_t by_jobs = std::min<int64_t>((wanted + chunks_est - 1) / chunks_est, n_haps);
    if (chunks_hi * by_jobs > slots && chunks_hi * by_jobs * 10 <= slots * 13) by_jobs = std::max<int64_t>(1, slots / chunks_hi);
    graded = chunks_hi * by_jobs > slots;
    if (by_jobs >= n_groups) n_groups = (int)by_jobs;
    else equal_split = false;
  }
  n_groups = std::max(1, std::min(n_groups, n_haps));
  // Group sizes.  Jobs are dispatched group by group (all chunks of group 0, then of group 1, ...), and the
  // kernel ends when the slowest wavefront does, so big batches get full-size groups first and a short tail of
  // halving groups (T/2, T/4, T/8, T/8): the last jobs in flight are 8x shorter than the first ones.
  std::vector<int64_t> want;
  if (equal_split || total_cols < 3 * (int64_t)target_cols) {
    const int64_t per_group = (total_cols + n_groups - 1) / n_groups;
    want.assign((size_t)n_groups, per_group);
    // Graded sizes (1.35 x ... 0.65 x the mean, long ones first) when there is more than one set of jobs: jobs of ONE
    // length fill the wavefront slots in lockstep -- every set ends at the same moment and the next one is dispatched
    // and set up into an empty chip (two or three exact sets of 4096 equal jobs ran 1.3-1.65 x slower than 2.7 or 3.7
    // sets); within a single set equal jobs are the shortest way through.
    if (graded && equal_split && n_groups >= 4)
      for (int i = 0; i < n_groups; i++)
        want[(size_t)i] = std::max<int64_t>(1, (int64_t)((double)per_group * (1.35 - 0.7 * (double)i / (double)(n_groups - 1))));
  } else {
    const int64_t tail[4] = {target_cols / 2, target_cols / 4, target_cols / 8, target_cols / 8};
    const int64_t head = total_cols - target_cols;  // the tail sums to target_cols
    const int n_big = (int)std::max<int64_t>(1, (head + target_cols / 2) / target_cols);
    want.assign((size_t)n_big, (head + n_big - 1) / n_big);
    want.insert(want.end(), tail, tail + 4);
  }
  {
    int h = 0;
    size_t gi = 0;
    while (h < n_haps) {
      PlanGroup g;
      g.hap_begin = h;
      g.stream_begin = p.n_stream;
      g.pad_ = 0;
      int64_t cols = 0;
      const bool last = gi + 1 >= want.size();
      const int64_t share = want[std::min(gi, want.size() - 1)];
      // at least one haplotype per group; stop once the group reached its share (the last group takes the rest)
      do {  // h is a stream-order index here
        p.hap_pos[h] = p.n_stream;
        p.hap_pos_flat[h] = p.n_stream_flat;
        p.n_stream_flat += p.hap_len[h] + 1;
        p.hap_group[h] = (int32_t)p.groups.size();
        p.hap_src[h] = (int32_t)hap_off[p.hap_orig[h]];
        p.n_stream += p.hap_len[h] + 1;  // columns + separator
        cols += p.hap_len[h] + 1;
        h++;
      } while (h < n_haps && (last || cols + (p.hap_len[h] + 1) / 2 < share));
      g.hap_end = h;
      p.n_stream += kLanes;  // drain room (idle entries)
      p.groups.push_back(g);
      gi++;
    }
  }

  p.n_stream_flat += kLanes;  // drain room behind the last haplotype

  // ---- read packing: best-fit decreasing into 64-lane chunks (one window = all reads) ----
  if (rows_per_lane <= 0)
    for (int r = 0; r < n_reads; r++)
      p.max_read_len = std::max(p.max_read_len, (int)(read_off[r + 1] - read_off[r]));
  p.place_chunk.resize((size_t)n_reads);
  p.place_lane.resize((size_t)n_reads);
  if (rows_per_lane <= 0 || n_reads == 0) return;
  static thread_local std::vector<int32_t> order;
  static thread_local PackScratch scratch;
  order.clear();
  const int long_from = kLanes * rows_per_lane;  // reads with this many bases or more need more than 64 lanes
  if (p.max_read_len < long_from) {
    order.resize((size_t)n_reads);
    std::iota(order.begin(), order.end(), 0);
  } else {
    for (int r = 0; r < n_reads; r++) {
      if ((int)(read_off[r + 1] - read_off[r]) < long_from) order.push_back(r);
      else { p.long_reads.push_back(r); p.place_chunk[r] = -1; p.place_lane[r] = 0; }
    }
  }
  p.n_chunks = pack_reads_place(order.data(), (int)order.size(), read_off, rows_per_lane,
                                (int)std::max<size_t>(order.size(), 1), p.place_chunk.data(), p.place_lane.data(),
                                &p.chunk_used, &p.useful_rows, &scratch);
  if (want_lanes) expand_lanes(p, n_reads, read_off, &p.lanes);
}

void expand_lanes(const Plan& p, int n_reads, const int64_t* read_off, std::vector<PlanLane>* lanes) {
  lanes->assign((size_t)p.n_chunks * kLanes, PlanLane{-1, 0});
  for (int r = 0; r < n_reads; r++) {
    if (p.place_chunk[r] < 0) continue;
    const int nb = blocks_for((int)(read_off[r + 1] - read_off[r]), p.rows_per_lane);
    PlanLane* dst = lanes->data() + (size_t)p.place_chunk[r] * kLanes + p.place_lane[r];
    for (int b = 0; b < nb; b++) dst[b] = PlanLane{r, b};
  }
}

int pack_reads_place(const int32_t* order_in, int n, const int64_t* read_off, int rows_per_lane, int window,
                     int32_t* place_chunk, uint8_t* place_lane, std::vector<uint8_t>* chunk_used,
                     int64_t* useful_rows, PackScratch* sc) {
  // counting sort by lanes needed (descending), then best fit through one stack of open chunks per free size
  const int rpl = rows_per_lane;
  if (window < 1) window = 1;
  int shift = -1;
  for (int b = 0; b < 8; b++) if ((1 << b) == rpl) shift = b;
  sc->need_of.resize((size_t)n);
  int64_t rows = 0;
  for (int i = 0; i < n; i++) {
    const int32_t r = order_in[i];
    const int R = (int)(read_off[r + 1] - read_off[r]);
    sc->need_of[i] = (uint8_t)(shift >= 0 ? (R + rpl) >> shift : blocks_for(R, rpl));
    rows += R;
  }
  if (useful_rows) *useful_rows += rows;
  const int wcap = std::min(n, window);
  sc->sorted.resize((size_t)wcap);
  sc->sorted_need.resize((size_t)wcap);
  const uint8_t* need_of = sc->need_of.data();
  int chunks_total = 0;
  for (int w0 = 0; w0 < n; w0 += window) {
    const int cnt = std::min(window, n - w0);
    int32_t bucket[kLanes + 2] = {0};
    for (int i = 0; i < cnt; i++) bucket[need_of[w0 + i]]++;
    int32_t start[kLanes + 2];
    int acc = 0;
    for (int c = kLanes; c >= 1; c--) { start[c] = acc; acc += bucket[c]; }
    for (int i = 0; i < cnt; i++) {
      const int at = start[need_of[w0 + i]]++;
      sc->sorted[at] = order_in[w0 + i];
      sc->sorted_need[at] = need_of[w0 + i];
    }
    int32_t head[kLanes + 1];
    for (int c = 0; c <= kLanes; c++) head[c] = -1;
    uint64_t open_sizes = 0;  // bit c-1: some open chunk has exactly c free lanes (c = 1..63; a fresh chunk has 64)
    const size_t base = chunk_used->size();  // chunk numbers of this window start here
    sc->next.clear();
    for (int i = 0; i < cnt; i++) {
      const int32_t r = sc->sorted[i];
      const int nb = sc->sorted_need[i];
      const uint64_t fits = nb <= 63 ? open_sizes >> (nb - 1) : 0;  // free sizes >= nb
      int chunk, c;
      if (!fits) {
        chunk = (int)sc->next.size();
        sc->next.push_back(-1);
        chunk_used->push_back(0);
        c = kLanes;
      } else {
        c = nb + __builtin_ctzll(fits);  // the smallest free size that fits
        chunk = head[c];
        head[c] = sc->next[chunk];
        if (head[c] < 0) open_sizes &= ~(1ull << (c - 1));
      }
      uint8_t& used = (*chunk_used)[base + (size_t)chunk];
      place_chunk[r] = (int32_t)(base + (size_t)chunk);
      place_lane[r] = used;
      used = (uint8_t)(used + nb);
      const int left = c - nb;
      if (left > 0) { sc->next[chunk] = head[left]; head[left] = chunk; open_sizes |= 1ull << (left - 1); }
    }
    chunks_total += (int)sc->next.size();
  }
  return chunks_total;
}

int pack_reads_windowed(const int32_t* order_in, int n, const int64_t* read_off, int rows_per_lane,
                        int window, std::vector<PlanLane>* lanes, int64_t* useful_rows) {
  // Allocation-free inner loops (this runs on the caller's thread for every batch): counting
  // sort by lanes needed, then best fit through per-free-size stacks threaded through `next`.
  const int rpl = rows_per_lane;
  if (window < 1) window = 1;
  int chunks_total = 0;
  // lanes needed per read, computed once (a shift when rpl is a power of two, as it always is)
  int shift = -1;
  for (int b = 0; b < 8; b++) if ((1 << b) == rpl) shift = b;
  std::vector<uint8_t> need_of((size_t)n);
  int64_t need_sum = 0;
  for (int i = 0; i < n; i++) {
    const int32_t r = order_in[i];
    const int R = (int)(read_off[r + 1] - read_off[r]);
    need_of[i] = (uint8_t)(shift >= 0 ? (R + rpl) >> shift : blocks_for(R, rpl));
    need_sum += need_of[i];
    if (useful_rows) *useful_rows += R;
  }
  {  // grow geometrically: callers (the PDHMM planner) may append thousands of small packings
    const size_t want = lanes->size() + (size_t)(need_sum + need_sum / 8 + 2 * kLanes);
    if (lanes->capacity() < want) lanes->reserve(std::max(want, 2 * lanes->capacity()));
  }
  std::vector<int32_t> sorted((size_t)std::min(n, window));
  std::vector<uint8_t> sorted_need((size_t)std::min(n, window));
  std::vector<int32_t> next;  // per chunk of the current window: next chunk with the same free size
  std::vector<int32_t> used;  // per chunk: lanes taken
  for (int w0 = 0; w0 < n; w0 += window) {
    const int cnt = std::min(window, n - w0);
    int32_t bucket[kLanes + 2] = {0};
    for (int i = 0; i < cnt; i++) bucket[need_of[w0 + i]]++;
    int32_t start[kLanes + 2];
    int acc = 0;
    for (int c = kLanes; c >= 1; c--) { start[c] = acc; acc += bucket[c]; }  // descending, stable
    for (int i = 0; i < cnt; i++) {
      const int at = start[need_of[w0 + i]]++;
      sorted[at] = order_in[w0 + i];
      sorted_need[at] = need_of[w0 + i];
    }
    int32_t head[kLanes + 1];
    for (int c = 0; c <= kLanes; c++) head[c] = -1;
    next.clear();
    used.clear();
    const size_t base = lanes->size();
    for (int i = 0; i < cnt; i++) {
      const int32_t r = sorted[i];
      const int nb = sorted_need[i];  // 1..64 (caller guarantees the read fits a chunk)
      int c = nb;
      while (c <= kLanes && head[c] < 0) c++;  // smallest free size that fits
      int chunk;
      if (c > kLanes) {
        chunk = (int)used.size();
        used.push_back(0);
        next.push_back(-1);
        lanes->resize(base + (size_t)(chunk + 1) * kLanes, PlanLane{-1, 0});
        c = kLanes;
      } else {
        chunk = head[c];
        head[c] = next[chunk];
      }
      PlanLane* dst = lanes->data() + base + (size_t)chunk * kLanes + used[chunk];
      for (int b = 0; b < nb; b++) dst[b] = PlanLane{r, b};
      used[chunk] += nb;
      const int left = c - nb;
      if (left > 0) { next[chunk] = head[left]; head[left] = chunk; }
    }
    chunks_total += (int)used.size();
  }
  return chunks_total;
}

}  // namespace gklhip
