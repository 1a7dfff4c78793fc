// scan.hip -- prefix sums for gdf_prefixsum_{i8,i32,i64,generic} and for the
// library's own histogram / offset scans.
//
// Replaces the reference's cub::DeviceScan calls (src/scan.cu:11-76).  Semantics
// kept: sum in the column's own dtype with wrap-around, inclusive or exclusive,
// equal size & dtype required, valid masks rejected.
//
// Shape (default from 2^22 elements on, when the scan is not in place -- round 6): ONE pass in lockstep ROUNDS, 2*w bytes per element
// (scan_lookback<.., ROUNDS = true>, launched as "scan_rounds"): G resident workgroups, workgroup b takes tiles b, b + G, ...; a
// tile's aggregate is published one pipeline step before the others' are needed, and every workgroup reads ALL G aggregates of a round
// in one batch of loads and adds them up itself -- those of the workgroups before it are its offset inside the round, their total
// advances its own carry.  No chain and no walk (what the decoupled look-back below dies of on this part: every hop a cross-XCD round
// trip), one round trip per round, hidden behind the next tile's loads.  A tile is only SUMMED when its aggregate goes out and scanned
// when it is written: 142 VGPRs for int64, three workgroups per CU.  1e9 int64: 2.97 ms = 5.4 TB/s = 0.67 of the peak (the three
// launches: 4.1 ms; profiles/r6_q_scan_rounds.jsonl).  The workgroups wait for one another, so all must be resident: the grid is what
// hipOccupancyMaxActiveBlocksPerMultiprocessor says fits, and a poll that lasts a quarter of a second (another process holds CUs) sets a
// flag, everybody leaves and the host starts over with the three launches below -- which is why an in-place scan never takes this path.
//
// Shape (smaller inputs, in-place scans, columns that are not 16-byte aligned): reduce-then-scan, three launches on the default stream, 3*w bytes per element --
//   1. scan_reduce_v : every block sums one contiguous chunk                  (read N)
//   2. scan_spine    : one block scans the <= 2048 chunk sums
//   3. scan_apply_v  : every block re-reads its chunk and writes the scan seeded with its chunk offset (read N, write N)
// with coalesced 16-byte non-temporal accesses: wave w of a 256-thread tile owns 1024 consecutive elements and reads them
// as vectors k * 64 + lane, so the element order inside a wave is (round, lane, element) and the local scan is one wave
// scan per round.  (Round 1's scan_apply gave every thread 16 consecutive elements -- 64 separate 64-byte requests per
// wave instruction -- and capped at ~4.9 TB/s.)  Columns that are not 16-byte aligned keep those element-wise kernels.
//
// The single-pass alternative (scan_lookback, GDF_SCAN_LOOKBACK=1: decoupled look-back, 2*w bytes) is in the file and
// correct, but LOSES on this part: a tile has to learn the sum of everything before it from other workgroups, through
// words that cross the XCDs' non-coherent L2s (agent-scope 8-byte {flag, data} stores / loads, ~2 us per round trip under
// load), and it can only finish after the slowest load among the few hundred tiles in flight before it.  Measured on 1e8
// int64 (profiles/r2_c_scan_ablation.md; GDF_SCAN_LOOKBACK=2 replaces the look-back by a SPINE workgroup that publishes every
// tile's exclusive prefix, one polled word per tile: 0.50 ms): the same kernel without the look-back 0.27 ms (5.9 TB/s), with ticket order
// 0.32, with a one-wave / 256-wide / software-pipelined look-back 0.64 / 0.64 / 0.55 ms -- against 0.49 ms for round 1's
// three launches.  The streaming half of that kernel is what the coalesced kernels above reuse.
#include "internal.h"

#include <cstdlib>

namespace gdf_amd {

constexpr int SCAN_THREADS = 256;
constexpr int SCAN_MAX_CHUNKS = 2048;

// ACC: accumulator type (uint32 for 1- and 4-byte columns, uint64 for 8-byte);
// ELEM: storage type.  Unsigned arithmetic gives the same bits as signed wrap.
template <class ACC, class ELEM, int ITEMS>
__global__ __launch_bounds__(SCAN_THREADS) void scan_reduce(const ELEM *__restrict__ in, ACC *__restrict__ chunk_sum,
                                                            size_t n, size_t chunk) {
  __shared__ ACC wsum[SCAN_THREADS / WAVE];
  const size_t begin = (size_t)blockIdx.x * chunk;
  const size_t end = begin + chunk < n ? begin + chunk : n;
  ACC acc = 0;
  size_t i = begin + threadIdx.x;
  for (; i + 7 * SCAN_THREADS < end; i += 8 * SCAN_THREADS) {      // 8 independent loads in flight per thread
    ELEM v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = in[i + (size_t)k * SCAN_THREADS];
#pragma unroll
    for (int k = 0; k < 8; ++k) acc += (ACC)v[k];
  }
  for (; i < end; i += SCAN_THREADS) acc += (ACC)in[i];
  acc = wave_reduce_add(acc);
  if (lane_id() == 0) wsum[threadIdx.x / WAVE] = acc;
  block_sync();
  if (threadIdx.x == 0) {
    ACC s = 0;
    for (int w = 0; w < SCAN_THREADS / WAVE; ++w) s += wsum[w];
    chunk_sum[blockIdx.x] = s;
  }
}

template <class ACC>
__global__ __launch_bounds__(SCAN_THREADS) void scan_spine(ACC *chunk_sum, int nchunks, ACC *running) {
  // exclusive scan of <= SCAN_MAX_CHUNKS values by one block, seeded with *running (the total of the
  // segments before this one), which is advanced by this segment's total
  __shared__ ACC wsum[SCAN_THREADS / WAVE];
  __shared__ ACC carry;
  if (threadIdx.x == 0) carry = running ? *running : (ACC)0;       // (null: a scan in one segment, nothing before it)
  block_sync();
  for (int base = 0; base < nchunks; base += SCAN_THREADS) {
    const int i = base + threadIdx.x;
    ACC v = i < nchunks ? chunk_sum[i] : 0;
    ACC incl = wave_scan_incl(v);
    if (lane_id() == WAVE - 1) wsum[threadIdx.x / WAVE] = incl;
    block_sync();
    ACC woff = 0;
    for (int w = 0; w < (int)(threadIdx.x / WAVE); ++w) woff += wsum[w];
    const ACC c = carry;
    if (i < nchunks) chunk_sum[i] = c + woff + incl - v;
    block_sync();
    if (threadIdx.x == SCAN_THREADS - 1) carry = c + woff + incl;
    block_sync();
  }
  if (running && threadIdx.x == 0) *running = carry;
}

template <class ACC, class ELEM, int ITEMS>
__global__ __launch_bounds__(SCAN_THREADS) void scan_apply(const ELEM *in, ELEM *out,   // in == out allowed
                                                           const ACC *__restrict__ chunk_off, size_t n, size_t chunk,
                                                           int inclusive) {
  __shared__ ACC wsum[SCAN_THREADS / WAVE];
  const size_t begin = (size_t)blockIdx.x * chunk;
  const size_t end = begin + chunk < n ? begin + chunk : n;
  ACC carry = chunk_off[blockIdx.x];
  constexpr size_t TILE = (size_t)SCAN_THREADS * ITEMS;
  for (size_t tile = begin; tile < end; tile += TILE) {
    const size_t t0 = tile + (size_t)threadIdx.x * ITEMS;   // this thread owns ITEMS consecutive elements
    ACC v[ITEMS];
    ACC run = 0;
    const bool full = tile + TILE <= end;                  // block-uniform: whole tiles load and store without guards
    if (full) {
#pragma unroll
      for (int k = 0; k < ITEMS; ++k) v[k] = (ACC)in[t0 + k];
    } else {
#pragma unroll
      for (int k = 0; k < ITEMS; ++k) v[k] = (t0 + k < end) ? (ACC)in[t0 + k] : 0;
    }
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) run += v[k];
    const ACC incl = wave_scan_incl(run);
    if (lane_id() == WAVE - 1) wsum[threadIdx.x / WAVE] = incl;
    block_sync();
    ACC woff = 0, total = 0;
#pragma unroll
    for (int w = 0; w < SCAN_THREADS / WAVE; ++w) {
      if (w < (int)(threadIdx.x / WAVE)) woff += wsum[w];
      total += wsum[w];
    }
    ACC pre = carry + woff + incl - run;   // exclusive prefix of this thread's first element
    ELEM o[ITEMS];
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
      o[k] = (ELEM)(inclusive ? pre + v[k] : pre);
      pre += v[k];
    }
    if (full) {
#pragma unroll
      for (int k = 0; k < ITEMS; ++k) out[t0 + k] = o[k];
    } else {
#pragma unroll
      for (int k = 0; k < ITEMS; ++k)
        if (t0 + k < end) out[t0 + k] = o[k];
    }
    carry += total;
    block_sync();
  }
}


// ---------------------------------------------------------------------------
// single pass, decoupled look-back
// ---------------------------------------------------------------------------
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));     // native vector: non-temporal builtins take it
constexpr int LB_THREADS = 256;
constexpr int LB_ITEMS = 16;
constexpr int LB_TILE = LB_THREADS * LB_ITEMS;
constexpr unsigned long long LB_FLAG = 1ull << 32;
constexpr size_t SCAN_ROUNDS_MIN = (size_t)1 << 22;    // elements from which the single-pass rounds are the default (device_scan)

// tile state: NW = sizeof(ACC) / 4 words of aggregate, then NW words of inclusive prefix; word = LB_FLAG | 32 data bits
template <class ACC>
__device__ __forceinline__ void lb_publish(unsigned long long *slot, ACC v) {
#pragma unroll
  for (int i = 0; i < (int)sizeof(ACC) / 4; ++i)
    __hip_atomic_store(slot + i, LB_FLAG | (unsigned long long)(uint32_t)((uint64_t)v >> (32 * i)), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
}
template <class ACC>
__device__ __forceinline__ bool lb_read(const unsigned long long *slot, ACC &v) {
  unsigned long long w[sizeof(ACC) / 4];
#pragma unroll
  for (int i = 0; i < (int)sizeof(ACC) / 4; ++i) w[i] = __hip_atomic_load(slot + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  bool ok = true;
  uint64_t r = 0;
#pragma unroll
  for (int i = 0; i < (int)sizeof(ACC) / 4; ++i) {
    ok = ok && (w[i] & LB_FLAG);
    r |= (uint64_t)(uint32_t)w[i] << (32 * i);
  }
  v = (ACC)r;
  return ok;
}

template <class ACC, class ELEM, bool ROUNDS = false>
__global__ __launch_bounds__(LB_THREADS) void scan_lookback(const ELEM *in, ELEM *out, size_t n, int inclusive,
                                                            unsigned long long *state, uint32_t *ticket, uint32_t ntiles, int dbg) {
  // dbg & 8: ROUNDS mode (round 6).  No chain and no walk: workgroup b takes tiles b, b + G, b + 2G, ... (G = gridDim.x, all resident),
  // round r is the G tiles r G ... r G + G - 1.  A workgroup publishes its tile's aggregate in slot [r & 3][b] (words tagged r + 1) one
  // pipeline step before it needs the others', then reads ALL G slots of the round in one batch of loads: the aggregates of the
  // workgroups before it are its tile's offset inside the round, their total advances its own running carry -- every workgroup adds
  // up every round itself, nobody waits for a prefix somebody else computed.  state[] holds 4 x G slots; ticket[1] is the bail-out flag
  // (a poll that never completes -- the workgroups are not all resident -- sets it, everybody leaves, the host takes the three launches).
  constexpr bool rounds = ROUNDS;    // (its own instantiation: the look-back's state does not cost the rounds their third workgroup per CU)
  uint32_t my_step = 0;              // rounds: the round of the next tile this workgroup takes
  constexpr int NW = sizeof(ACC) / 4;
  constexpr int VEC = 16 / sizeof(ELEM);           // elements per 16-byte vector
  constexpr int VPT = LB_ITEMS / VEC;              // vectors per thread (i8: 1, i32: 4, i64: 8)
  constexpr int NWAVES = LB_THREADS / WAVE;
  constexpr int SEG = WAVE * LB_ITEMS;             // elements per wave
  union Vec { u32x4 q; ELEM e[VEC]; };
  struct Round { uint32_t code; ACC partial; };    // code 0: this wave's 64 tiles hold aggregates only, 1: an inclusive prefix
                                                   // among them (partial = sum up to it), 2: a tile before that has nothing yet
  __shared__ ACC wsum[NWAVES];
  __shared__ Round s_round[NWAVES];
  __shared__ uint32_t s_tile[2];
  __shared__ ACC s_excl;
  __shared__ uint32_t s_first[NWAVES];
  const int wave = threadIdx.x / WAVE, lane = lane_id();

  // dbg & 4: SPINE mode.  Workgroup 0 does nothing but turn tile aggregates into exclusive prefixes, in tile order: every
  // round it reads the next 1024 aggregates (four per thread, all loads in flight together), takes the leading run that
  // is already published, scans it and publishes the prefixes.  A worker then polls ONE word -- its own prefix -- instead
  // of walking back over hundreds of predecessors, each round of which is a cross-XCD round trip.  The spine consumes up
  // to 1024 tiles per round trip (~2.5 us): several times the ~80 tiles per us the data path needs.
  if (!ROUNDS && (dbg & 4) && blockIdx.x == 0) {
    constexpr int PT = 4;
    ACC carry = 0;
    uint32_t base = 0;
    while (base < ntiles) {
      ACC agg[PT];
      bool have[PT];
#pragma unroll
      for (int k = 0; k < PT; ++k) {
        const uint32_t t = base + threadIdx.x * PT + k;
        agg[k] = 0;
        have[k] = t < ntiles && lb_read<ACC>(state + (size_t)t * 2 * NW, agg[k]);
      }
      int lead = 0;
      ACC mine = 0, part[PT];
#pragma unroll
      for (int k = 0; k < PT; ++k) {
        part[k] = mine;                       // sum of this thread's published tiles before tile k
        if (lead == k && have[k]) { mine += agg[k]; ++lead; }
      }
      const unsigned long long short_m = __ballot(lead < PT);
      block_sync();                           // s_first / wsum readers of the previous round are done
      if (lane == 0) s_first[wave] = short_m ? (uint32_t)(wave * WAVE + __ffsll((long long)short_m) - 1) : 0xffffffffu;
      block_sync();
      uint32_t tstar = 0xffffffffu;
#pragma unroll
      for (int w = NWAVES - 1; w >= 0; --w) if (s_first[w] != 0xffffffffu) tstar = s_first[w];
      if (threadIdx.x > tstar) { mine = 0; lead = 0; }
      const ACC inc = wave_scan_incl(mine);
      if (lane == WAVE - 1) wsum[wave] = inc;
      block_sync();
      ACC woff = 0, total = 0;
#pragma unroll
      for (int w = 0; w < NWAVES; ++w) {
        if (w < wave) woff += wsum[w];
        total += wsum[w];
      }
      const ACC mybase = carry + woff + inc - mine;
#pragma unroll
      for (int k = 0; k < PT; ++k)
        if (k < lead) lb_publish<ACC>(state + ((size_t)(base + threadIdx.x * PT + k) * 2 + 1) * NW, (ACC)(mybase + part[k]));
      const uint32_t done = tstar == 0xffffffffu ? (uint32_t)(LB_THREADS * PT) : tstar * PT;
      // (+ the leading tiles of thread tstar itself)
      uint32_t extra = 0;
      if (tstar != 0xffffffffu) {
        block_sync();
        if (threadIdx.x == tstar) s_first[0] = (uint32_t)lead;
        block_sync();
        extra = s_first[0];
      }
      carry += total;
      base += done + extra;
      if (done + extra == 0) __builtin_amdgcn_s_sleep(4);
    }
    return;
  }

  // No LDS transposition (its 37 KB per workgroup halved the occupancy of a kernel that lives on bytes in flight): wave w
  // owns SEG consecutive elements and reads them as 16-byte vectors, vector k * 64 + lane in round k -- 1 KB contiguous
  // per load instruction.  The order of the elements is then (round, lane, element within the vector): one wave scan of
  // the per-vector sums per ROUND.  The partial last tile takes guarded loads of 16 consecutive elements per thread.
  auto load_tile = [&](Vec (&v)[VPT], uint32_t t) {
    const size_t base = (size_t)t * LB_TILE;
    if (base + LB_TILE <= n) {
      const u32x4 *src = reinterpret_cast<const u32x4 *>(in + base + (size_t)wave * SEG);     // 16-byte aligned: the host checked
#pragma unroll
      for (int k = 0; k < VPT; ++k) v[k].q = __builtin_nontemporal_load(src + k * WAVE + lane);
    } else {
      const size_t t0 = base + (size_t)threadIdx.x * LB_ITEMS;
#pragma unroll
      for (int k = 0; k < VPT; ++k)
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          const size_t i = t0 + k * VEC + e;
          v[k].e[e] = i < n ? in[i] : (ELEM)0;
        }
    }
  };

  // Three tiles per workgroup are in flight, a software pipeline over the ticket sequence:
  //   Z  loads issued (its ticket was taken one step ago)
  //   Y  data arrived: scanned locally and its AGGREGATE published at once -- successors must not wait for it
  //   X  aggregate published one step ago: look back, publish the inclusive prefix, write the outputs.
  // X looks back a whole step (a tile's load latency) after its aggregate went out, so its predecessors -- older tickets
  // -- have had that long to publish theirs: the poll rarely meets a tile with nothing published, and the round trip of the
  // poll overlaps Z's loads.  (Looking back right after the local scan, every tile stalled on the slowest load among the
  // ~200 tiles started just before it: 0.64 ms per 1e8 int64 against 0.32 ms for the same kernel without the look-back.)
  struct Scanned {
    uint32_t tile;
    // prefix of vector (k, lane) inside the wave's segment.  ROUNDS keeps none: a tile is only SUMMED when its aggregate goes out and
    // scanned when it is written -- sixteen registers less per tile in flight, the third workgroup per CU
    ACC excl_in_wave[ROUNDS ? 1 : VPT];
    ACC woff, aggregate;
  };
  auto ticket_to = [&](int slot) {
    if (rounds) {
      if (threadIdx.x == 0) {
        const unsigned long long t = (unsigned long long)my_step * gridDim.x + blockIdx.x;
        s_tile[slot] = t < ntiles ? (uint32_t)t : 0xffffffffu;
      }
      ++my_step;                     // (every thread counts: workgroup-uniform)
      return;
    }
    if (threadIdx.x == 0) s_tile[slot] = atomicAdd(ticket, 1u);
  };
  // local scan of a loaded tile + publication of its aggregate; contains one block_sync (which also makes the ticket
  // written just before it visible)
  auto scan_and_publish = [&](const Vec (&v)[VPT], uint32_t t, Scanned &sc) {
    const size_t base = (size_t)t * LB_TILE;
    ACC wave_total = 0;
    if constexpr (ROUNDS) {
      ACC sum = 0;
#pragma unroll
      for (int k = 0; k < VPT; ++k)
#pragma unroll
        for (int e = 0; e < VEC; ++e) sum += (ACC)v[k].e[e];
      wave_total = wave_reduce_add(sum);
    } else if (base + LB_TILE <= n) {
#pragma unroll
      for (int k = 0; k < VPT; ++k) {
        ACC sum = 0;
#pragma unroll
        for (int e = 0; e < VEC; ++e) sum += (ACC)v[k].e[e];
        const ACC inc = wave_scan_incl(sum);
        sc.excl_in_wave[k] = wave_total + inc - sum;
        wave_total += __shfl(inc, WAVE - 1, WAVE);
      }
    } else {
      ACC sum = 0;
#pragma unroll
      for (int k = 0; k < VPT; ++k)
#pragma unroll
        for (int e = 0; e < VEC; ++e) sum += (ACC)v[k].e[e];
      const ACC inc = wave_scan_incl(sum);
      sc.excl_in_wave[0] = inc - sum;
      wave_total = __shfl(inc, WAVE - 1, WAVE);
    }
    block_sync();                               // readers of wsum from the previous tile are done
    if (lane == 0) wsum[wave] = wave_total;
    block_sync();
    sc.woff = 0;
    sc.aggregate = 0;
#pragma unroll
    for (int w = 0; w < NWAVES; ++w) {
      if (w < wave) sc.woff += wsum[w];
      sc.aggregate += wsum[w];
    }
    sc.tile = t;
    if (threadIdx.x == 0) {
      if (rounds) {
        const uint32_t r = t / gridDim.x;
        unsigned long long *slot = state + ((size_t)(r & 3u) * gridDim.x + blockIdx.x) * NW;
#pragma unroll
        for (int i = 0; i < NW; ++i)
          __hip_atomic_store(slot + i, ((unsigned long long)(r + 1u) << 32) | (unsigned long long)(uint32_t)((uint64_t)sc.aggregate >> (32 * i)),
                             __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        lb_publish<ACC>(state + (size_t)t * 2 * NW, sc.aggregate);
      }
    }
  };
  // ROUNDS: this tile's offset = carry (everything before its round) + the aggregates of the workgroups before this one in the round
  ACC carry = 0;
  __shared__ ACC s_before[NWAVES], s_total[NWAVES];
  __shared__ int s_ok;
  auto resolve_rounds = [&](const Scanned &sc) -> ACC {
    const uint32_t G = gridDim.x, r = sc.tile / G;
    const uint32_t left = ntiles - r * G, pubs = left < G ? left : G;          // publishers of this round (the last one may be short)
    const unsigned long long *slots = state + (size_t)(r & 3u) * G * NW;
    constexpr int PT = 2;                                                     // slots per thread and page; G <= 4 * LB_THREADS (the host's grid)
    ACC before = 0, total = 0;
    unsigned long long waiting_since = 0;          // wall_clock64(): 100 MHz
    for (uint32_t spins = 0;; ++spins) {
      bool ok = true;
      before = 0;
      total = 0;
      for (uint32_t page = 0; page < pubs; page += PT * LB_THREADS) {           // (one page up to two workgroups per CU)
        unsigned long long w[PT][NW];
#pragma unroll
        for (int k = 0; k < PT; ++k) {
          const uint32_t b = page + threadIdx.x + (uint32_t)k * LB_THREADS;
#pragma unroll
          for (int i = 0; i < NW; ++i)
            w[k][i] = b < pubs ? __hip_atomic_load(slots + (size_t)b * NW + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
        }
#pragma unroll
        for (int k = 0; k < PT; ++k) {
          const uint32_t b = page + threadIdx.x + (uint32_t)k * LB_THREADS;
          if (b < pubs) {
            uint64_t v = 0;
#pragma unroll
            for (int i = 0; i < NW; ++i) { ok = ok && (uint32_t)(w[k][i] >> 32) == r + 1u; v |= (uint64_t)(uint32_t)w[k][i] << (32 * i); }
            total += (ACC)v;
            if (b < blockIdx.x) before += (ACC)v;
          }
        }
      }
      block_sync();                 // s_ok's previous readers are done
      if (threadIdx.x == 0) s_ok = 1;
      block_sync();
      if (!ok) s_ok = 0;
      block_sync();
      if (s_ok) break;
      // not everybody has published: look again; a poll that goes on for a quarter of a second (a round takes microseconds) means
      // the grid is not resident -- another process holds CUs, two such scans wait for one another -- bail out
      if (threadIdx.x == 0) {
        const unsigned long long now = wall_clock64();
        if (spins == 0) waiting_since = now;
        if (now - waiting_since > 25000000ull || __hip_atomic_load(ticket + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
          __hip_atomic_store(ticket + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          s_ok = 2;
        }
      }
      block_sync();
      if (s_ok == 2) return (ACC)0;                  // (the caller sees ticket[1] and leaves)
      __builtin_amdgcn_s_sleep(1);
    }
    before = wave_reduce_add(before);
    total = wave_reduce_add(total);
    if (lane == 0) { s_before[wave] = before; s_total[wave] = total; }
    block_sync();
    ACC bsum = 0, tsum = 0;
#pragma unroll
    for (int w2 = 0; w2 < NWAVES; ++w2) { bsum += s_before[w2]; tsum += s_total[w2]; }
    const ACC exclusive = carry + bsum;
    carry += tsum;
    return exclusive;
  };
  // look-back for a scanned tile, the whole workgroup: thread t examines tile - 1 - t (a tile before the first one has
  // prefix 0); both words of a predecessor are requested together -- one round trip, not two
  auto resolve = [&](const Scanned &sc) -> ACC {
    if constexpr (ROUNDS) return resolve_rounds(sc);
    if (dbg & 4) {                  // spine mode: the tile's exclusive prefix arrives in its own slot
      block_sync();                 // s_excl's previous readers are done
      if (threadIdx.x == 0) {
        ACC e = 0;
        while (!lb_read<ACC>(state + ((size_t)sc.tile * 2 + 1) * NW, e)) __builtin_amdgcn_s_sleep(1);
        s_excl = e;
      }
      block_sync();
      return s_excl;
    }
    ACC exclusive = 0;
    long long nearest = (long long)sc.tile - 1;
    for (; !(LAB_BITS(dbg) & 2);) {           // dbg & 2 (experiment): no look-back, wrong prefixes
      const long long p = nearest - (long long)threadIdx.x;
      int st = 2;
      ACC val = 0;
      if (p >= 0) {
        const unsigned long long *theirs = state + (size_t)p * 2 * NW;
        ACC agg, inc;
        const bool has_agg = lb_read<ACC>(theirs, agg);
        const bool has_inc = lb_read<ACC>(theirs + NW, inc);
        st = has_inc ? 2 : (has_agg ? 1 : 0);
        val = has_inc ? inc : agg;
      }
      const unsigned long long m2 = __ballot(st == 2), m0 = __ballot(st == 0);
      const int c = m2 ? __ffsll((long long)m2) - 1 : WAVE;
      const bool blocked = (m0 & (c == WAVE ? ~0ull : ((1ull << c) - 1ull))) != 0;
      const ACC part = wave_reduce_add(lane <= c ? val : (ACC)0);
      block_sync();                 // s_round's previous readers are done
      if (lane == 0) s_round[wave] = Round{blocked ? 2u : (m2 ? 1u : 0u), part};
      block_sync();
      bool wait = false, found = false;
      ACC add = 0;
#pragma unroll
      for (int w = 0; w < NWAVES; ++w) {
        if (!wait && !found) {
          const Round r = s_round[w];
          if (r.code == 2) wait = true;
          else { add += r.partial; found = r.code == 1; }
        }
      }
      if (!wait) exclusive += add;
      if (found) break;
      if (wait) __builtin_amdgcn_s_sleep(2);
      else nearest -= LB_THREADS;
    }
    if (threadIdx.x == 0) lb_publish<ACC>(state + ((size_t)sc.tile * 2 + 1) * NW, (ACC)(exclusive + sc.aggregate));
    return exclusive;
  };
  auto write_tile = [&](const Vec (&v)[VPT], const Scanned &sc, ACC exclusive) {
    const size_t base = (size_t)sc.tile * LB_TILE;
    const ACC wave_base = exclusive + sc.woff;
    if (base + LB_TILE <= n) {
      u32x4 *dst = reinterpret_cast<u32x4 *>(out + base + (size_t)wave * SEG);
      ACC running = 0;              // ROUNDS: the wave's elements before round k
#pragma unroll
      for (int k = 0; k < VPT; ++k) {
        ACC pre;
        if constexpr (ROUNDS) {
          ACC sum = 0;
#pragma unroll
          for (int e = 0; e < VEC; ++e) sum += (ACC)v[k].e[e];
          const ACC inc = wave_scan_incl(sum);
          pre = wave_base + running + inc - sum;
          running += __shfl(inc, WAVE - 1, WAVE);
        } else {
          pre = wave_base + sc.excl_in_wave[k];
        }
        Vec o;
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          const ACC x = (ACC)v[k].e[e];
          o.e[e] = (ELEM)(inclusive ? pre + x : pre);
          pre += x;
        }
        __builtin_nontemporal_store(o.q, dst + k * WAVE + lane);
      }
    } else {
      const size_t t0 = base + (size_t)threadIdx.x * LB_ITEMS;
      ACC pre;
      if constexpr (ROUNDS) {       // (the partial last tile: sixteen consecutive elements per thread, one wave scan of the threads' sums)
        ACC sum = 0;
#pragma unroll
        for (int k = 0; k < VPT; ++k)
#pragma unroll
          for (int e = 0; e < VEC; ++e) sum += (ACC)v[k].e[e];
        pre = wave_base + wave_scan_incl(sum) - sum;
      } else {
        pre = wave_base + sc.excl_in_wave[0];
      }
#pragma unroll
      for (int k = 0; k < VPT; ++k)
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          const size_t i = t0 + k * VEC + e;
          const ACC x = (ACC)v[k].e[e];
          if (i < n) out[i] = (ELEM)(inclusive ? pre + x : pre);
          pre += x;
        }
    }
  };

  // prologue: X loaded and scanned, Y's loads in flight
  if constexpr (ROUNDS) {           // (the bail-out flag set before the launch: the test switch GDF_SCAN_FORCE_BAIL)
    if (threadIdx.x == 0) s_ok = __hip_atomic_load(ticket + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u ? 2 : 1;
    block_sync();
    if (s_ok == 2) return;
  }
  ticket_to(0);
  block_sync();
  uint32_t tx = s_tile[0];
  if (tx >= ntiles) return;
  Vec vx[VPT], vy[VPT];
  Scanned sx, sy;
  load_tile(vx, tx);
  ticket_to(1);
  scan_and_publish(vx, tx, sx);
  uint32_t ty = s_tile[1];
  if (ty < ntiles) load_tile(vy, ty);
  int slot = 0;
  for (;;) {
    // ticket for Z; Y scanned and published (this waits for Y's data); Z's loads issued; then X resolved and written
    uint32_t tz = ntiles;
    Vec vz[VPT];
    if (ty < ntiles) {
      ticket_to(slot);
      scan_and_publish(vy, ty, sy);
      tz = s_tile[slot];
      slot ^= 1;
      if (tz < ntiles) load_tile(vz, tz);
    }
    const ACC exclusive = resolve(sx);
    if (rounds && s_ok == 2) return;                 // bail-out (workgroup-uniform: s_ok was read behind a barrier)
    write_tile(vx, sx, exclusive);
    if (ty >= ntiles) break;
#pragma unroll
    for (int k = 0; k < VPT; ++k) { vx[k] = vy[k]; vy[k] = vz[k]; }
    sx = sy;
    ty = tz;
  }
}

template <class ACC, class ELEM, bool ROUNDS>
static gdf_error device_scan_lookback_impl(const ELEM *in, ELEM *out, size_t n, bool inclusive, long long mode) {
  constexpr int NW = sizeof(ACC) / 4;
  const size_t ntiles = (n + LB_TILE - 1) / LB_TILE;
  const int dbg = (int)lab::knob_int("GDF_SCAN_DBG", 0) | (mode == 2 ? 4 : 0) | (ROUNDS ? 8 : 0);      // 2: spine mode, 3: rounds
  DevBuf st;
  const size_t state_bytes = sizeof(unsigned long long) * (ROUNDS ? (size_t)4 * 4 * LB_THREADS * NW : ntiles * 2 * NW);
  RMM_TRY(st.alloc(state_bytes + 2 * sizeof(unsigned long long)));
  HIP_TRY(hipMemsetAsync(st.p, 0, state_bytes + 2 * sizeof(unsigned long long), stream0()));
  uint32_t *ticket = reinterpret_cast<uint32_t *>(st.as<unsigned char>() + state_bytes);
  // persistent workgroups, each takes tiles from the ticket counter until they run out: a few per CU (every one keeps two
  // tiles in flight).  With dbg & 1 every workgroup handles exactly the tile of its blockIdx.
  const int per_cu_env = (int)lab::knob_int("GDF_SCAN_WGS_PER_CU", 0);
  int fit = 1;
  HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&fit, (const void *)scan_lookback<ACC, ELEM, ROUNDS>, LB_THREADS, 0));
  if (fit < 1) fit = 1;
  int per_cu = per_cu_env > 0 ? per_cu_env : fit;
  size_t grid = (dbg & 1) ? ntiles : (size_t)NUM_CU * (size_t)per_cu;
  // rounds: every workgroup must be resident (they wait for one another) and the poll reads <= 4 * LB_THREADS slots
  if (ROUNDS) grid = (size_t)device_cu_count() * (size_t)std::max(1, std::min(std::min(per_cu_env > 0 ? per_cu_env : 4, fit), 4));
  if (ROUNDS && grid > (size_t)4 * LB_THREADS) grid = (size_t)4 * LB_THREADS;        // (the slots one poll covers)
  if (grid > ntiles + ((dbg & 4) ? 1 : 0)) grid = ntiles + ((dbg & 4) ? 1 : 0);      // spine mode: workgroup 0 takes no tiles
  if ((dbg & 4) && grid < 2) grid = 2;
  if (ROUNDS && lab::path_on("GDF_SCAN_FORCE_BAIL")) {      // test switch: the kernel leaves at once, the caller takes the three launches
    const uint32_t one = 1;
    HIP_TRY(hipMemcpyAsync(ticket + 1, &one, sizeof(one), hipMemcpyHostToDevice, stream0()));
    HIP_TRY(hipStreamSynchronize(stream0()));
  }
  GDF_LAUNCH(ROUNDS ? "scan_rounds" : "scan_lookback", (scan_lookback<ACC, ELEM, ROUNDS>), dim3((unsigned)grid), dim3(LB_THREADS), 0, stream0(), in, out, n,
             inclusive ? 1 : 0, st.as<unsigned long long>(), ticket, (uint32_t)ntiles, dbg);
  HIP_CHECK_LAST();
  if (ROUNDS) {                               // a grid that was not resident bailed out -- the caller takes the three launches
    uint32_t bailed = 0;
    HIP_TRY(read_back(&bailed, ticket + 1, sizeof(bailed)));
    if (bailed) return GDF_UNSUPPORTED_METHOD;
    return GDF_SUCCESS;
  }
  HIP_TRY(hipStreamSynchronize(stream0()));   // scratch is released on return
  return GDF_SUCCESS;
}
template <class ACC, class ELEM>
static gdf_error device_scan_lookback(const ELEM *in, ELEM *out, size_t n, bool inclusive, long long mode) {
  return mode == 3 ? device_scan_lookback_impl<ACC, ELEM, true>(in, out, n, inclusive, mode)
                   : device_scan_lookback_impl<ACC, ELEM, false>(in, out, n, inclusive, mode);
}


// ---------------------------------------------------------------------------
// reduce-then-scan with COALESCED accesses (the default): the same three launches as scan_reduce / scan_spine / scan_apply
// above, but every load and store is a 16-byte vector, lanes on consecutive vectors (1 KB contiguous per wave instruction),
// non-temporal.  scan_apply's 16 consecutive elements per thread are 64 separate 64-byte requests per wave instruction and
// cap that kernel at ~4.9 TB/s; the striped order needs one wave scan per round of 64 vectors instead of one per tile,
// which is VALU time the kernel has to spare.
// ---------------------------------------------------------------------------
template <class ACC, class ELEM>
__global__ __launch_bounds__(LB_THREADS) void scan_reduce_v(const ELEM *__restrict__ in, ACC *__restrict__ chunk_sum, size_t n, size_t chunk) {
  constexpr int VEC = 16 / sizeof(ELEM);
  union Vec { u32x4 q; ELEM e[VEC]; };
  __shared__ ACC wsum[LB_THREADS / WAVE];
  const size_t begin = (size_t)blockIdx.x * chunk;                 // a multiple of LB_TILE, so of VEC
  const size_t end = begin + chunk < n ? begin + chunk : n;
  const size_t nvec = (end - begin) / VEC;
  const u32x4 *src = reinterpret_cast<const u32x4 *>(in + begin);
  ACC acc = 0;
  size_t i = threadIdx.x;
  for (; i + 7 * LB_THREADS < nvec; i += 8 * LB_THREADS) {         // 8 independent 16-byte loads in flight per thread
    Vec v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k].q = __builtin_nontemporal_load(src + i + (size_t)k * LB_THREADS);
#pragma unroll
    for (int k = 0; k < 8; ++k)
#pragma unroll
      for (int e = 0; e < VEC; ++e) acc += (ACC)v[k].e[e];
  }
  for (; i < nvec; i += LB_THREADS) {
    Vec v;
    v.q = __builtin_nontemporal_load(src + i);
#pragma unroll
    for (int e = 0; e < VEC; ++e) acc += (ACC)v.e[e];
  }
  if (threadIdx.x == 0)
    for (size_t j = begin + nvec * VEC; j < end; ++j) acc += (ACC)in[j];      // fewer than VEC elements behind the last whole vector
  acc = wave_reduce_add(acc);
  if (lane_id() == 0) wsum[threadIdx.x / WAVE] = acc;
  block_sync();
  if (threadIdx.x == 0) {
    ACC t = 0;
    for (int w = 0; w < LB_THREADS / WAVE; ++w) t += wsum[w];
    chunk_sum[blockIdx.x] = t;
  }
}

template <class ACC, class ELEM>
__global__ __launch_bounds__(LB_THREADS) void scan_apply_v(const ELEM *in, ELEM *out,   // in == out allowed
                                                          const ACC *__restrict__ chunk_off, size_t n, size_t chunk, int inclusive) {
  constexpr int VEC = 16 / sizeof(ELEM);
  constexpr int VPT = LB_ITEMS / VEC;
  constexpr int NWAVES = LB_THREADS / WAVE;
  constexpr int SEG = WAVE * LB_ITEMS;
  union Vec { u32x4 q; ELEM e[VEC]; };
  __shared__ ACC wsum[NWAVES];
  const size_t begin = (size_t)blockIdx.x * chunk;
  const size_t end = begin + chunk < n ? begin + chunk : n;
  const int wave = threadIdx.x / WAVE, lane = lane_id();
  ACC carry = chunk_off[blockIdx.x];
  Vec cur[VPT], nxt[VPT];
  auto load_full = [&](Vec (&v)[VPT], size_t tile) {
    const u32x4 *src = reinterpret_cast<const u32x4 *>(in + tile + (size_t)wave * SEG);
#pragma unroll
    for (int k = 0; k < VPT; ++k) v[k].q = __builtin_nontemporal_load(src + k * WAVE + lane);
  };
  if (begin + LB_TILE <= end) load_full(cur, begin);
  for (size_t tile = begin; tile < end; tile += LB_TILE) {
    const bool full = tile + LB_TILE <= end;
    const bool next_full = tile + 2 * (size_t)LB_TILE <= end;
    if (next_full) load_full(nxt, tile + LB_TILE);                // the next tile's loads fly over this tile's scan and stores
    ACC excl_in_wave[VPT];
    ACC wave_total = 0;
    if (full) {
#pragma unroll
      for (int k = 0; k < VPT; ++k) {
        ACC sum = 0;
#pragma unroll
        for (int e = 0; e < VEC; ++e) sum += (ACC)cur[k].e[e];
        const ACC inc = wave_scan_incl(sum);
        excl_in_wave[k] = wave_total + inc - sum;
        wave_total += __shfl(inc, WAVE - 1, WAVE);
      }
    } else {                                                       // the chunk's partial last tile: 16 consecutive elements per thread
      const size_t t0 = tile + (size_t)threadIdx.x * LB_ITEMS;
      ACC sum = 0;
#pragma unroll
      for (int k = 0; k < VPT; ++k)
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          const size_t i = t0 + k * VEC + e;
          cur[k].e[e] = i < end ? in[i] : (ELEM)0;
          sum += (ACC)cur[k].e[e];
        }
      const ACC inc = wave_scan_incl(sum);
      excl_in_wave[0] = inc - sum;
      wave_total = __shfl(inc, WAVE - 1, WAVE);
    }
    block_sync();                                                  // the previous tile's readers of wsum are done
    if (lane == 0) wsum[wave] = wave_total;
    block_sync();
    ACC woff = 0, total = 0;
#pragma unroll
    for (int w = 0; w < NWAVES; ++w) {
      if (w < wave) woff += wsum[w];
      total += wsum[w];
    }
    const ACC wave_base = carry + woff;
    if (full) {
      u32x4 *dst = reinterpret_cast<u32x4 *>(out + tile + (size_t)wave * SEG);
#pragma unroll
      for (int k = 0; k < VPT; ++k) {
        ACC pre = wave_base + excl_in_wave[k];
        Vec o;
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          const ACC x = (ACC)cur[k].e[e];
          o.e[e] = (ELEM)(inclusive ? pre + x : pre);
          pre += x;
        }
        __builtin_nontemporal_store(o.q, dst + k * WAVE + lane);
      }
    } else {
      const size_t t0 = tile + (size_t)threadIdx.x * LB_ITEMS;
      ACC pre = wave_base + excl_in_wave[0];
#pragma unroll
      for (int k = 0; k < VPT; ++k)
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          const size_t i = t0 + k * VEC + e;
          const ACC x = (ACC)cur[k].e[e];
          if (i < end) out[i] = (ELEM)(inclusive ? pre + x : pre);
          pre += x;
        }
    }
    carry += total;
#pragma unroll
    for (int k = 0; k < VPT; ++k) cur[k] = nxt[k];
  }
}

template <class ACC, class ELEM>
static gdf_error device_scan_coalesced(const ELEM *in, ELEM *out, size_t n, bool inclusive, DevBuf *keep = nullptr) {
  DevBuf local;
  DevBuf &sums = keep ? *keep : local;         // keep: the caller holds the scratch, nothing waits here
  RMM_TRY(sums.alloc(sizeof(ACC) * SCAN_MAX_CHUNKS));
  const size_t tiles = (n + LB_TILE - 1) / LB_TILE;
  const size_t tiles_per_chunk = (tiles + SCAN_MAX_CHUNKS - 1) / SCAN_MAX_CHUNKS;
  const size_t chunk = tiles_per_chunk * LB_TILE;
  const int nchunks = (int)((n + chunk - 1) / chunk);
  GDF_LAUNCH("scan_reduce", (scan_reduce_v<ACC, ELEM>), dim3(nchunks), dim3(LB_THREADS), 0, stream0(), in, sums.as<ACC>(), n, chunk);
  hipLaunchKernelGGL((scan_spine<ACC>), dim3(1), dim3(SCAN_THREADS), 0, stream0(), sums.as<ACC>(), nchunks, (ACC *)nullptr);
  GDF_LAUNCH("scan_apply", (scan_apply_v<ACC, ELEM>), dim3(nchunks), dim3(LB_THREADS), 0, stream0(), in, out, sums.as<ACC>(), n, chunk,
             inclusive ? 1 : 0);
  HIP_CHECK_LAST();
  if (!keep) HIP_TRY(hipStreamSynchronize(stream0()));   // scratch is released on return
  return GDF_SUCCESS;
}

// GDF_SCAN_SEG_MB=<n> scans the input in segments of n MiB (reduce -> spine -> apply per segment) so that the apply
// pass re-reads what the reduce pass has just pulled through the 256 MiB Infinity Cache.  Measured on 1e8 int64:
// scan_apply drops from 0.33 to 0.27 ms at 128 MiB segments, but the smaller grids slow scan_reduce (0.17 -> 0.24 ms)
// and the extra launches add gaps -- 0.64 ms against 0.55 ms for the whole array in one go.  So the default is one
// segment; the switch stays for larger inputs / other parts.
template <class ACC, class ELEM>
gdf_error device_scan(const ELEM *in, ELEM *out, size_t n, bool inclusive) {
  if (n == 0) return GDF_SUCCESS;
  // 16-byte accesses need 16-byte-aligned columns (a column may be a slice of a larger buffer): those take the coalesced
  // kernels, or -- GDF_SCAN_LOOKBACK=1, an experiment that lost, see the header -- the single-pass kernel; the rest, and
  // GDF_SCAN_BLOCKED=1, the element-wise kernels of round 1
  // GDF_SCAN_LOOKBACK (test switch, read per call): 1 look-back, 2 spine, 3 rounds whatever the size, 0 never; default (-1): the rounds
  // from 2^22 elements on when the scan is not in place (they can bail out half-way -- a grid that is not resident -- and leave `out`
  // partly written; the three launches then start over from `in`)
  long long mode = lab::path_int("GDF_SCAN_LOOKBACK", -1);
  const bool blocked = lab::path_on("GDF_SCAN_BLOCKED");
  // (1-byte elements keep the three launches: a 4096-element tile is 4 KB, and a round's cadence -- not the memory -- sets the pace:
  // 0.84 against 0.64 ms per 1e9 int8; int32 1.57 against 2.08, int64 2.98 against 4.07 -- profiles/r6_q_scan_rounds.jsonl)
  if (mode < 0) mode = (n >= SCAN_ROUNDS_MIN && sizeof(ELEM) >= 4) ? 3 : 0;
  if (mode == 3 && (const void *)in == (const void *)out) mode = 0;
  if (!blocked && ((uintptr_t)in % 16 == 0) && ((uintptr_t)out % 16 == 0)) {
    if (mode > 0 && n / LB_TILE < 0x7fffffffULL) {
      const gdf_error e = device_scan_lookback<ACC, ELEM>(in, out, n, inclusive, mode);
      if (e != GDF_UNSUPPORTED_METHOD) return e;
    }
    return device_scan_coalesced<ACC, ELEM>(in, out, n, inclusive);
  }
  constexpr int ITEMS = 16 / sizeof(ELEM) >= 4 ? 8 : 4;
  constexpr size_t TILE = (size_t)SCAN_THREADS * ITEMS;
  const size_t seg_bytes = (size_t)lab::knob_int("GDF_SCAN_SEG_MB", 0) << 20;
  size_t seg = seg_bytes ? seg_bytes / sizeof(ELEM) / TILE * TILE : (n + TILE - 1) / TILE * TILE;
  if (seg < TILE) seg = TILE;
  DevBuf sums, running;
  RMM_TRY(sums.alloc(sizeof(ACC) * SCAN_MAX_CHUNKS));
  RMM_TRY(running.alloc(sizeof(ACC)));
  HIP_TRY(hipMemsetAsync(running.p, 0, sizeof(ACC), stream0()));
  for (size_t s0 = 0; s0 < n; s0 += seg) {
    const size_t m = n - s0 < seg ? n - s0 : seg;
    // chunks are whole tiles so that thread-contiguous loads stay aligned
    const size_t tiles = (m + TILE - 1) / TILE;
    const size_t tiles_per_chunk = (tiles + SCAN_MAX_CHUNKS - 1) / SCAN_MAX_CHUNKS;
    const size_t chunk = tiles_per_chunk * TILE;
    const int nchunks = (int)((m + chunk - 1) / chunk);
    GDF_LAUNCH("scan_reduce", (scan_reduce<ACC, ELEM, ITEMS>), dim3(nchunks), dim3(SCAN_THREADS), 0, stream0(), in + s0, sums.as<ACC>(), m, chunk);
    hipLaunchKernelGGL((scan_spine<ACC>), dim3(1), dim3(SCAN_THREADS), 0, stream0(), sums.as<ACC>(), nchunks, running.as<ACC>());
    GDF_LAUNCH("scan_apply", (scan_apply<ACC, ELEM, ITEMS>), dim3(nchunks), dim3(SCAN_THREADS), 0, stream0(), in + s0, out + s0, sums.as<ACC>(), m, chunk,
               inclusive ? 1 : 0);
  }
  HIP_CHECK_LAST();
  HIP_TRY(hipStreamSynchronize(stream0()));   // scratch is released on return
  return GDF_SUCCESS;
}

// internal entry points used by partition / join / group-by host code
gdf_error scan_u32(const uint32_t *in, uint32_t *out, size_t n, bool inclusive) {
  return device_scan<uint32_t, uint32_t>(in, out, n, inclusive);
}
gdf_error scan_u64(const uint64_t *in, uint64_t *out, size_t n, bool inclusive) {
  return device_scan<uint64_t, uint64_t>(in, out, n, inclusive);
}
gdf_error scan_u32_async(const uint32_t *in, uint32_t *out, size_t n, bool inclusive, DevBuf *scratch) {
  if (n == 0) return GDF_SUCCESS;
  if (((uintptr_t)in % 16 == 0) && ((uintptr_t)out % 16 == 0) && !lab::path_on("GDF_SCAN_BLOCKED") && lab::path_int("GDF_SCAN_LOOKBACK", -1) <= 0)
    return device_scan_coalesced<uint32_t, uint32_t>(in, out, n, inclusive, scratch);
  return device_scan<uint32_t, uint32_t>(in, out, n, inclusive);      // (the other kernels synchronise)
}

}  // namespace gdf_amd

using namespace gdf_amd;

static gdf_error prefixsum_checked(gdf_column *inp, gdf_column *out, int inclusive, gdf_dtype expect) {
  (void)expect;
  GDF_REQUIRE(inp && out, GDF_DATASET_EMPTY);
  GDF_REQUIRE(inp->size == out->size, GDF_COLUMN_SIZE_MISMATCH);   // scan.cu:55
  GDF_REQUIRE(inp->dtype == out->dtype, GDF_UNSUPPORTED_DTYPE);     // scan.cu:56
  GDF_REQUIRE(!inp->valid, GDF_VALIDITY_UNSUPPORTED);               // scan.cu:57
  GDF_REQUIRE(!out->valid, GDF_VALIDITY_UNSUPPORTED);               // scan.cu:58
  return GDF_SUCCESS;
}

extern "C" {

gdf_error gdf_prefixsum_i8(gdf_column *inp, gdf_column *out, int inclusive) {
  return gdf_amd::guarded([&]() -> gdf_error {
  GDF_TRY(prefixsum_checked(inp, out, inclusive, GDF_INT8));
  return device_scan<uint32_t, uint8_t>((const uint8_t *)inp->data, (uint8_t *)out->data, inp->size, inclusive != 0);
  });
}
gdf_error gdf_prefixsum_i32(gdf_column *inp, gdf_column *out, int inclusive) {
  return gdf_amd::guarded([&]() -> gdf_error {
  GDF_TRY(prefixsum_checked(inp, out, inclusive, GDF_INT32));
  return device_scan<uint32_t, uint32_t>((const uint32_t *)inp->data, (uint32_t *)out->data, inp->size,
                                         inclusive != 0);
  });
}
gdf_error gdf_prefixsum_i64(gdf_column *inp, gdf_column *out, int inclusive) {
  return gdf_amd::guarded([&]() -> gdf_error {
  GDF_TRY(prefixsum_checked(inp, out, inclusive, GDF_INT64));
  return device_scan<uint64_t, uint64_t>((const uint64_t *)inp->data, (uint64_t *)out->data, inp->size,
                                         inclusive != 0);
  });
}
gdf_error gdf_prefixsum_generic(gdf_column *inp, gdf_column *out, int inclusive) {
  return gdf_amd::guarded([&]() -> gdf_error {
  GDF_REQUIRE(inp, GDF_DATASET_EMPTY);
  switch (inp->dtype) {   // scan.cu:66-76: other dtypes silently succeed
    case GDF_INT8: return gdf_prefixsum_i8(inp, out, inclusive);
    case GDF_INT32: return gdf_prefixsum_i32(inp, out, inclusive);
    case GDF_INT64: return gdf_prefixsum_i64(inp, out, inclusive);
    default: return GDF_SUCCESS;
  }
  });
}

}  // extern "C"
