// nms_step_device.h — the device side of the one-launch detector-step NMS (tvmi_nms_step): constants, workspace, hand-over
// helpers and the workgroup body.  A header because two kernels run the body: nms_step_fused (nms.hip) and the RoIAlign launch
// that carries the step's NMS workgroups in front of its own grid (roi_align.hip, round 6).  Moved verbatim from nms.hip; the
// body takes its workgroup coordinates and its LDS block as arguments instead of blockIdx / gridDim / __shared__ declarations.
#pragma once

#include <algorithm>

#include "nms_device.h"

namespace tvmi {
// the hand-over words of a stream's step launches: a block the library owns, one per (device, stream) (nms.hip); nullptr while the
// stream is being captured into a graph
int* nms_step_sync_block(hipStream_t stream);

namespace {

constexpr int kStepThreads = 256;
constexpr int kStepMaxSegments = 64;
constexpr int kStepMaxGroups = 512;
constexpr int kStepMaxImages = 16;
// hand-over words of a launch (ints): [0] sweepers finished (ticket), [1] input error, [2] a poll gave up (never cleared),
// then ONE FLAG PER PRODUCER — publishers [S][16], tile workgroups [S * G], counting workgroups — instead of shared counters:
// agent-scope atomics on one address retire one after the other (~150 ns each on the MI355X: 250 counting workgroups reporting
// to one counter took 35 us), flag stores go out in parallel and a poll is one load per lane.
constexpr int kStepFlagsPub = 4;
constexpr int kStepFlagsTile = kStepFlagsPub + kStepMaxSegments * kSmallSegBlocks;
constexpr int kStepFlagsRank = kStepFlagsTile + kStepMaxGroups;
constexpr int kStepSyncWords = kStepFlagsRank + kStepMaxGroups;

struct StepWorkspace {
  int* sync;     // kStepSyncWords ints, all zero when the launch starts; the last sweeper re-zeroes them (see step_sync_block)
  int* sidx;     // [S][1024] box index of the p-th best box of segment s
  int* part;     // [4][n] partial global ranks
  int* slot;     // [n] by global rank: 0, or (kept box index + 1) | image << 13
  u64* tiles;    // [S][136][64]
  u64* sbox;     // [S][1024][2] the boxes of a segment in score order
};
inline size_t step_workspace_layout(int64_t n, int64_t S, char* base, StepWorkspace* w) {
  size_t off = 0;
  auto take = [&](size_t bytes) {
    char* ptr = base ? base + off : nullptr;
    off += (bytes + 255) & ~(size_t)255;
    return ptr;
  };
  char* sy = take((size_t)kStepSyncWords * sizeof(int));
  char* si = take((size_t)S * kSmallSegBoxes * sizeof(int));
  char* pa = take((size_t)4 * n * sizeof(int));
  char* sl = take((size_t)n * sizeof(int));
  char* tl = take((size_t)S * kSmallSegTiles * 64 * sizeof(u64));
  char* sb = take((size_t)S * kSmallSegBoxes * 2 * sizeof(u64));
  if (w) {
    w->sync = reinterpret_cast<int*>(sy);
    w->sidx = reinterpret_cast<int*>(si);
    w->part = reinterpret_cast<int*>(pa);
    w->slot = reinterpret_cast<int*>(sl);
    w->tiles = reinterpret_cast<u64*>(tl);
    w->sbox = reinterpret_cast<u64*>(sb);
  }
  return off;
}
inline int step_groups_per_segment(int64_t n, int64_t S) {
  const int nbmax = (int)std::min<int64_t>(kSmallSegBlocks, ceil_div(n, 64));
  const int want = (int)ceil_div(nbmax * (nbmax + 1) / 2, 4);
  return (int)std::max<int64_t>(1, std::min<int64_t>(want, kStepMaxGroups / S));
}

__device__ __forceinline__ int ld_agent(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ u64 ld_agent64(const u64* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent64(u64* p, u64 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// every flag of flags[0 .. count) is non-zero: the whole workgroup polls (one load per lane and round), bounded like
// poll_until_equal; returns with the give-up word raised when the producers do not show up
__device__ __forceinline__ void poll_flags(const int* flags, int count, int* gave_up, int (&votes)[4]) {
  // (`votes`: four words of the workgroup's LDS block.  __syncthreads_and keeps a word of static LDS of its own, which the launch
  // that carries this body next to a 40 KB RoIAlign image cannot afford: 256 bytes more and a CU holds three workgroups, not four)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int polls = 0;; ++polls) {
    bool ok = true;
    for (int i = threadIdx.x; i < count; i += blockDim.x) ok = ok && ld_agent(&flags[i]) != 0;
    const bool wave_ok = __ballot(!ok) == 0ull;
    if (lane == 0) votes[wave] = wave_ok ? 1 : 0;
    __syncthreads();
    const bool all = (votes[0] & votes[1] & votes[2] & votes[3]) != 0;
    __syncthreads();   // (the next round's votes overwrite these)
    if (all) return;
    __builtin_amdgcn_s_sleep(4);
    if ((polls & 255) == 255 && ld_agent(gave_up) != 0) return;
    if (polls > (1 << 18)) {
      if (threadIdx.x == 0) st_agent(gave_up, 1);
      return;
    }
  }
}

struct StepPack {   // optional payload of the last phase (payload == nullptr: keep list only)
  const int64_t* image_idx;
  const int64_t* labels;
  float* payload;
  int32_t* counts;
  int64_t row_stride;
  int num_images, max_dets, count_in_row;
};

// tiles above the column blocks of one sweep wave: block w + 4k has w + 4k earlier blocks, at most 3 + 4k
template <int K>
struct StepAbove {
  u64 v[4 * K + 3];
};

// number of the 64 keys held one per lane in `kk` that are below `ki` (per lane): the keys are read lane by lane into SGPRs
// (v_readlane) — no LDS round trip, no barrier
// Number of the 64 score words held one per lane in `dk` (a group of 64 consecutive boxes) that rank before the lane's own box
// (score word `di`).  Keys are (score word, box index) pairs and the groups are in index order, so against a whole group only the
// score words need comparing: `rel` < 0 — the group's boxes all have a smaller index, a tie counts (d <= di, i.e. d < di + 1: no
// live score word is 0xffffffff); `rel` > 0 — a larger index, a tie does not.  One full-rate v_cmp_lt_u32 against an SGPR per key
// (a 64-bit integer compare is several times slower); the loops stay rolled (x8): this code runs once per launch, straight-line
// unrolling of everything made the kernel 130 KB of instructions that were all fetched cold.
// max(a - b, 0) for unsigned a (VGPR), b (wave-uniform): v_sub_u32 with the integer clamp.  Opaque on purpose — written as
// __builtin_elementwise_sub_sat the compiler folds min(sat(thr - d), 1) back into v_cmp_lt + v_cndmask / v_addc.
__device__ __forceinline__ unsigned sub_sat_u32(unsigned a, unsigned b_uniform) {
  unsigned r;
  asm("v_sub_u32_e64 %0, %1, %2 clamp" : "=v"(r) : "v"(a), "s"(b_uniform));
  return r;
}
__device__ __forceinline__ int count_below_group(unsigned dk, unsigned di, int rel) {
  // [d < thr] as min(sat(thr - d), 1): plain VALU results, no VCC / SGPR-pair round trip between the compare and the add (a
  // v_cmp -> v_addc chain stalls on the condition code for every key: ~45 cycles per key measured, one wave per SIMD)
  const unsigned thr = di + (rel < 0 ? 1u : 0u);
  unsigned c0 = 0, c1 = 0;
#pragma unroll 4
  for (int t = 0; t < 64; t += 2) {
    const unsigned d0 = (unsigned)__builtin_amdgcn_readlane((int)dk, t), d1 = (unsigned)__builtin_amdgcn_readlane((int)dk, t + 1);
    c0 += min(sub_sat_u32(thr, d0), 1u);
    c1 += min(sub_sat_u32(thr, d1), 1u);
  }
  return (int)(c0 + c1);
}
// ... and against the lane's OWN group: key t of the group against lane `lane` — a tie counts when t < lane, i.e. the
// threshold of key t is di + 1 for the lanes above t
__device__ __forceinline__ int count_below_own(unsigned dk, unsigned di) {
  const int lane = threadIdx.x & 63;
  unsigned c0 = 0, c1 = 0;
#pragma unroll 4
  for (int t = 0; t < 64; t += 2) {
    const unsigned d0 = (unsigned)__builtin_amdgcn_readlane((int)dk, t), d1 = (unsigned)__builtin_amdgcn_readlane((int)dk, t + 1);
    c0 += min(sub_sat_u32(di + (t < lane ? 1u : 0u), d0), 1u);
    c1 += min(sub_sat_u32(di + (t + 1 < lane ? 1u : 0u), d1), 1u);
  }
  return (int)(c0 + c1);
}

#ifdef TVMI_STEP_TIMING
__device__ unsigned long long g_step_stamp[64];
#define STEP_STAMP(slot, cond) do { if ((cond) && threadIdx.x == 0) g_step_stamp[slot] = wall_clock64(); } while (0)
#else
#define STEP_STAMP(slot, cond) do { } while (0)
#endif
constexpr int kStepPer = kSortMax / kStepThreads;   // boxes per thread of a pass over all n (16)
// the LDS of one workgroup of the step (26 KB): a struct, so that a kernel which also runs other code can overlay it
struct StepShared {
  u64 s_keys[kSmallSegBoxes];                                 // members of my segment (ascending box index)
  __attribute__((aligned(16))) float s_row[3][5][64];         // row blocks of a workgroup's four tiles
  u64 s_keepbits[kSmallSegBlocks];
  int s_pre[kStepMaxImages + 1][kStepPer * 4];                    // counts -> exclusive prefixes per (round, wave)
  int s_tot[kStepMaxImages + 1];
  int s_lr[4][64];                                            // local-rank partials of a member group
  unsigned s_rk[4][8][64];                                    // global rank counting: eight groups of score words per wave
  int s_flag;
  int s_votes[4];                                                        // poll_flags
};

// workgroup (bx, by) of a launch of S x gdim_y workgroups of kStepThreads threads
__device__ __forceinline__ void nms_step_body(StepShared& sh, const int bx, const int by, const int gdim_y, const float* __restrict__ dets, const float* __restrict__ scores,
                                                               const int64_t* __restrict__ seg, int n, int S, int G, double thr,
                                                               ThrBand band, StepWorkspace ws, int64_t* __restrict__ keep_out,
                                                               int64_t* __restrict__ num_keep, StepPack pk) {

  constexpr int kPer = kStepPer;
  u64 (&s_keys)[kSmallSegBoxes] = sh.s_keys;
  float (&s_row)[3][5][64] = sh.s_row;
  u64 (&s_keepbits)[kSmallSegBlocks] = sh.s_keepbits;
  int (&s_pre)[kStepMaxImages + 1][kStepPer * 4] = sh.s_pre;
  int (&s_tot)[kStepMaxImages + 1] = sh.s_tot;
  int (&s_lr)[4][64] = sh.s_lr;
  unsigned (&s_rk)[4][8][64] = sh.s_rk;
  int& s_flag = sh.s_flag;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(tid >> 6));
  const int s = bx, j = by;
  int* const err = ws.sync + 1;
  int* const gave_up = ws.sync + 2;
  int* const pub_flags = ws.sync + kStepFlagsPub + s * kSmallSegBlocks;   // [16] of my segment
  int* const tile_flags = ws.sync + kStepFlagsTile + s * G;               // [G] of my segment
  int* const rank_flags = ws.sync + kStepFlagsRank;                       // one per counting workgroup of the launch
  const u64 lt_mask = (1ull << lane) - 1ull;
  STEP_STAMP(0, s == 0 && j == 0);

  // The counting is spread over the counting-only workgroups of the launch (blockIdx.y >= G: a launch has up to kStepMaxGroups
  // workgroups, the tile workgroups use S * G of them — 16M key pairs at 4000 boxes want every SIMD of the chip) or, without
  // them, over all workgroups before their tiles.  A counting workgroup takes 64 boxes (lane = box) and a Q-th of the keys, its
  // four waves a quarter of that each; the four counts meet in LDS, so a box has Q <= 4 partials to add up later (16 scattered
  // words per box made the sweepers' gather the longest phase of the kernel).
  const int n_groups = (n + 63) >> 6;
  const int rank_j0 = gdim_y > G ? G : 0;
  const int n_rank_wgs = (gdim_y - rank_j0) * S;
  const int Q = min(4, max(1, n_rank_wgs / n_groups));
  const int per = ((n_groups + 4 * Q - 1) / (4 * Q)) << 6;   // keys per wave, a multiple of 64
  auto rank_share = [&]() {
    for (int rw = (j - rank_j0) * S + s; rw < n_groups * Q; rw += n_rank_wgs) {   // workgroup-uniform
      const int eg = rw % n_groups, q = rw / n_groups;
      const int i = eg * 64 + lane;
      const unsigned di = i < n ? (unsigned)(score_key(scores[i], 0) >> 32) : 0xffffffffu;
      const int j_begin = (q * 4 + wave) * per, j_end = min(n, j_begin + per);
      int c = 0;
      for (int jb = j_begin; jb < j_end; jb += 8 * 64) {   // eight 64-key chunks per batch: their loads are in flight together
        float sv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int qq = jb + u * 64 + lane;
          sv[u] = scores[min(qq, n - 1)];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {                        // parked in LDS so that the counting loop below can stay rolled
          const int qq = jb + u * 64 + lane;
          s_rk[wave][u][lane] = qq < j_end ? (unsigned)(score_key(sv[u], 0) >> 32) : 0xffffffffu;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        for (int u = 0; u < 8 && jb + u * 64 < j_end; ++u) {
          const int chunk = (jb >> 6) + u;
          const unsigned dk = s_rk[wave][u][lane];
          c += chunk == eg ? count_below_own(dk, di) : count_below_group(dk, di, chunk < eg ? -1 : 1);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      }
      s_lr[wave][lane] = c;
      __syncthreads();
      if (wave == 0 && i < n) st_agent(&ws.part[(size_t)q * n + i], s_lr[0][lane] + s_lr[1][lane] + s_lr[2][lane] + s_lr[3][lane]);
      __syncthreads();
    }
  };
  if (j >= G) {   // counting-only workgroup
    STEP_STAMP(21, s == 0 && j == G);
    STEP_STAMP(23, s == 0 && j == gdim_y - 1);
    rank_share();
    STEP_STAMP(22, s == 0 && j == G);
    STEP_STAMP(24, s == 0 && j == gdim_y - 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) st_agent(&rank_flags[(j - G) * S + s], 1);
    return;
  }
  // ---- A1: the members of my segment, as keys (score bits << 32 | box index) in ascending box order.  EVERY workgroup of the
  // segment builds the same list (redundant work on an idle chip instead of a publish / poll round trip).
  int cnt;
  {
    int64_t sg[kPer];
    float sc[kPer];
#pragma unroll
    for (int r = 0; r < kPer; ++r) {           // all loads of the pass in flight together
      const int g = min(r * kStepThreads + tid, n - 1);
      sg[r] = seg ? seg[g] : 0;
      sc[r] = scores[g];
    }
    u64 bal[kPer];
    bool bad = false;
#pragma unroll
    for (int r = 0; r < kPer; ++r) {
      const int g = r * kStepThreads + tid;
      const bool in = g < n;
      bad |= in && (sg[r] < 0 || sg[r] >= (int64_t)S);
      bal[r] = __ballot(in && sg[r] == (int64_t)s);
      if (lane == 0) s_pre[0][r * 4 + wave] = __popcll(bal[r]);
    }
    __syncthreads();
    if (wave == 0) {
      const int c = s_pre[0][lane];
      int incl = c;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(incl, d);
        if (lane >= d) incl += o;
      }
      s_pre[1][lane] = incl - c;
      if (lane == 63) s_tot[0] = incl;
    }
    // (a vote through the workgroup's own LDS block instead of __syncthreads_or, which keeps static LDS of its own: poll_flags)
    {
      const bool wave_bad = __ballot(bad) != 0ull;
      if (lane == 0) sh.s_votes[wave] = wave_bad ? 1 : 0;
    }
    __syncthreads();
    const bool any_bad = (sh.s_votes[0] | sh.s_votes[1] | sh.s_votes[2] | sh.s_votes[3]) != 0;
    cnt = s_tot[0];
    if ((any_bad || cnt > kSmallSegBoxes) && tid == 0 && j == 0) st_agent(err, 1);
#pragma unroll
    for (int r = 0; r < kPer; ++r) {
      const int g = r * kStepThreads + tid;
      if ((bal[r] >> lane) & 1ull) {
        const int pos = s_pre[1][r * 4 + wave] + __popcll(bal[r] & lt_mask);
        if (pos < kSmallSegBoxes) s_keys[pos] = score_key(sc[r], g);
      }
    }
    cnt = min(cnt, kSmallSegBoxes);
    __syncthreads();
  }
  STEP_STAMP(1, s == 0 && j == 0);
  STEP_STAMP(16, s == 0 && j == 16);
  const int nb = (cnt + 63) >> 6, ntiles = nb * (nb + 1) / 2;
  const int npub = min(G, nb);               // workgroups that publish member groups
  // ---- A3: score order of the segment by rank counting: workgroup j takes the member groups j, j + G, ... (64 members each),
  // its four waves count over a quarter of the segment's keys each; the lane that holds member p then knows its rank r and
  // writes the box (and its index) to position r of the segment's sorted lists.
  for (int mg = j; mg < nb; mg += G) {       // workgroup-uniform
    const int p = mg * 64 + lane;
    const bool have = p < cnt;
    const u64 ki = have ? s_keys[p] : ~0ull;
    float4 bx = make_float4(0.f, 0.f, 0.f, 0.f);
    if (have && wave == 0) bx = *reinterpret_cast<const float4*>(dets + (int64_t)(unsigned)ki * 4);   // in flight under the counting
    const int cpw = (nb + 3) >> 2;            // 64-key chunks per wave
    const unsigned di = (unsigned)(ki >> 32);
    int c = 0;
    for (int ch = wave * cpw; ch < min(nb, (wave + 1) * cpw); ++ch) {
      const int q = ch * 64 + lane;
      const unsigned dk = q < cnt ? (unsigned)(s_keys[q] >> 32) : 0xffffffffu;   // padding: above every live score word
      c += ch == mg ? count_below_own(dk, di) : count_below_group(dk, di, ch < mg ? -1 : 1);
    }
    s_lr[wave][lane] = c;
    __syncthreads();
    if (wave == 0 && have) {
      const int r = s_lr[0][lane] + s_lr[1][lane] + s_lr[2][lane] + s_lr[3][lane];
      u64* dst = ws.sbox + ((size_t)s * kSmallSegBoxes + r) * 2;
      st_agent64(dst, ((u64)__builtin_bit_cast(unsigned, bx.y) << 32) | __builtin_bit_cast(unsigned, bx.x));
      st_agent64(dst + 1, ((u64)__builtin_bit_cast(unsigned, bx.w) << 32) | __builtin_bit_cast(unsigned, bx.z));
      st_agent(&ws.sidx[(size_t)s * kSmallSegBoxes + r], (int)(unsigned)ki);
    }
    __syncthreads();
  }
  if (j < npub) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // my part of the sorted lists is acknowledged
    __syncthreads();
    if (tid == 0) st_agent(&pub_flags[j], 1);
  }
  STEP_STAMP(2, s == 0 && j == 0);
  if (gdim_y == G) rank_share();   // no counting-only workgroups in this launch: everybody takes a share first
  STEP_STAMP(3, s == 0 && j == 0);
  STEP_STAMP(17, s == 0 && j == 16);
  // ---- B: suppression tiles of my segment, from the sorted boxes
  poll_flags(pub_flags, npub, gave_up, sh.s_votes);
  STEP_STAMP(4, s == 0 && j == 0);
  STEP_STAMP(18, s == 0 && j == 16);
  {
    const u64* my_sbox = ws.sbox + (size_t)s * kSmallSegBoxes * 2;
    auto box_at = [&](int p, float& x1, float& y1, float& x2, float& y2) {
      const u64 lo = ld_agent64(&my_sbox[2 * p]), hi = ld_agent64(&my_sbox[2 * p + 1]);
      x1 = __builtin_bit_cast(float, (unsigned)lo);
      y1 = __builtin_bit_cast(float, (unsigned)(lo >> 32));
      x2 = __builtin_bit_cast(float, (unsigned)hi);
      y2 = __builtin_bit_cast(float, (unsigned)(hi >> 32));
    };
    auto row_of = [&](int t, int& rb, int& cb) {   // tile index -> (row block, column block) of the upper triangle, row-major
      rb = 0;
      int rem = t;
      while (rem >= nb - rb) {
        rem -= nb - rb;
        ++rb;
      }
      cb = rb + rem;
    };
    u64* tiles = ws.tiles + (size_t)s * kSmallSegTiles * 64;
    for (int t0 = j * 4; t0 < ntiles; t0 += G * 4) {   // workgroup-uniform
      int rb0, cb0;
      row_of(t0, rb0, cb0);
      const int t = t0 + wave;
      int rb = 0, cb = 0;
      if (t < ntiles) row_of(t, rb, cb);
      // column box of this lane and (waves 0..2) one of the at most three row blocks of the four tiles: loads issued together
      const int c = cb * 64 + lane;
      const bool jvalid = t < ntiles && c < cnt;
      float jx1 = 0, jy1 = 0, jx2 = 0, jy2 = 0, x1 = 0, y1 = 0, x2 = 0, y2 = 0;
      const bool stage = wave < 3 && rb0 + wave < nb;
      const int p = (rb0 + wave) * 64 + lane;
      if (jvalid) box_at(c, jx1, jy1, jx2, jy2);
      if (stage && p < cnt) box_at(p, x1, y1, x2, y2);
      if (stage) {
        s_row[wave][0][lane] = x1;
        s_row[wave][1][lane] = y1;
        s_row[wave][2][lane] = x2;
        s_row[wave][3][lane] = y2;
        s_row[wave][4][lane] = (x2 - x1) * (y2 - y1);
      }
      __syncthreads();
      if (t < ntiles) {
        const float jarea = (jx2 - jx1) * (jy2 - jy1);
        const u64 mine = suppression_tile<float, 64>(&s_row[rb - rb0][0][0], nullptr, min(64, cnt - rb * 64), jx1, jy1, jx2, jy2,
                                                     jarea, 0, jvalid, cb == rb, thr, band);
        st_agent64(&tiles[(size_t)t * 64 + lane], mine);
      }
      __syncthreads();
    }
  }
  STEP_STAMP(19, s == 0 && j == 16);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // my tiles and rank partials are acknowledged
  __syncthreads();
  STEP_STAMP(20, s == 0 && j == 16);
  if (tid == 0) {
    st_agent(&tile_flags[j], 1);
    if (gdim_y == G) st_agent(&rank_flags[j * S + s], 1);   // this launch has no counting-only workgroups: my share is in
  }
  if (j != 0) return;
  // workgroup (s, 0) sweeps the segment once every tile workgroup has reported
  poll_flags(tile_flags, G, gave_up, sh.s_votes);
  STEP_STAMP(5, s == 0);

  // ---- C: sweep of my segment (4 waves; wave w owns column blocks w, w + 4, w + 8, w + 12)
  int midx[4];                                    // box index of my members p = tid + 256 r (score order), loaded under the sweep
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int p = tid + kStepThreads * r;
    midx[r] = p < cnt ? ld_agent(&ws.sidx[(size_t)s * kSmallSegBoxes + p]) : 0;
  }
  {
    const u64* tiles = ws.tiles + (size_t)s * kSmallSegTiles * 64;
    auto tile_index = [nb](int rb, int cb) { return rb * nb - (rb * (rb - 1)) / 2 + (cb - rb); };
    u64 diag[4];
    StepAbove<0> a0;
    StepAbove<1> a1;
    StepAbove<2> a2;
    StepAbove<3> a3;
    auto load_col = [&](int k, u64* above, int cap) {
      const int c = wave + 4 * k;
      const bool have = c < nb;
      diag[k] = have ? ld_agent64(&tiles[(size_t)tile_index(c, c) * 64 + lane]) : 0ull;
#pragma unroll
      for (int q = 0; q < 15; ++q)
        if (q < cap) above[q] = (have && q < c) ? ld_agent64(&tiles[(size_t)tile_index(q, c) * 64 + lane]) : 0ull;
    };
    load_col(0, a0.v, 3);
    load_col(1, a1.v, 7);
    load_col(2, a2.v, 11);
    load_col(3, a3.v, 15);
    STEP_STAMP(6, s == 0);
    u64 pend[4] = {0ull, 0ull, 0ull, 0ull};   // per-lane pending removals of my blocks (OR-reduced when the block's turn comes)
#pragma unroll
    for (int step = 0; step < kSmallSegBlocks; ++step) {
      if (step < nb) {   // workgroup-uniform
        const int ow = step & 3, ok = step >> 2;
        if (wave == ow) {
          const int rows_here = min(64, cnt - step * 64);
          const u64 valid = rows_here >= 64 ? ~0ull : ((1ull << rows_here) - 1ull);
          u64 r = wave_or64(pend[ok]);
          const u64 dg = diag[ok];
          u64 active = uniform64(__ballot(dg != 0ull)) & ~r & valid;
          while (active) {
            const int k = __builtin_ctzll(active);
            r |= readlane64(dg, k);
            active &= ~(r | (1ull << k));
          }
          if (lane == 0) s_keepbits[step] = ~r & valid;
        }
        __syncthreads();
        const bool kept_lane = (s_keepbits[step] >> lane) & 1ull;
        // rows of block `step` that were kept push their suppression words to my later column blocks
        if (step < 3) pend[0] |= kept_lane ? a0.v[step < 3 ? step : 0] : 0ull;
        if (step < 7) pend[1] |= kept_lane ? a1.v[step < 7 ? step : 0] : 0ull;
        if (step < 11) pend[2] |= kept_lane ? a2.v[step < 11 ? step : 0] : 0ull;
        if (step < 15) pend[3] |= kept_lane ? a3.v[step < 15 ? step : 0] : 0ull;
      }
    }
  }
  __syncthreads();
  STEP_STAMP(7, s == 0);
  // ---- slots of my boxes, by global rank (needs the rank partials of every workgroup of the launch)
  poll_flags(rank_flags, n_rank_wgs, gave_up, sh.s_votes);
  STEP_STAMP(14, s == 0);
  {
    const bool pack = pk.payload != nullptr;
    int g[4] = {0, 0, 0, 0}, img[4] = {0, 0, 0, 0};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const bool have = tid + kStepThreads * r < cnt;
#pragma unroll
      for (int q = 0; q < 4; ++q)                         // all loads in flight together
        if (q < Q && have) g[r] += ld_agent(&ws.part[(size_t)q * n + midx[r]]);
      if (pack && have) img[r] = (int)pk.image_idx[midx[r]];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int p = tid + kStepThreads * r;
      if (p < cnt) {
        const bool kept = (s_keepbits[p >> 6] >> (p & 63)) & 1ull;
        const int im = (img[r] >= 0 && img[r] < kStepMaxImages) ? img[r] : kStepMaxImages;   // outside the payload: no row
        st_agent(&ws.slot[min(max(g[r], 0), n - 1)], kept ? ((midx[r] + 1) | (im << 13)) : 0);
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  STEP_STAMP(8, s == 0);
  if (tid == 0) s_flag = __hip_atomic_fetch_add(&ws.sync[0], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == S - 1;
  __syncthreads();
  if (!s_flag) return;
  STEP_STAMP(9, true);

  // ---- D: last sweeper out — keep list in global score order (+ payload).  Round r covers the ranks [256 r, 256 r + 256); all
  // slot words are loaded together, the ballots of the 16 rounds go to LDS as counts per (round, wave), one wave-scan per list
  // turns them into exclusive prefixes, and the stores follow with one more round of loads (the payload fields).
  const bool pack = pk.payload != nullptr;
  const bool failed = ld_agent(err) != 0 || ld_agent(gave_up) != 0;
  const int B = pack ? pk.num_images : 0;
  int run = 0;
  if (!failed) {
    int v[kPer];
#pragma unroll
    for (int r = 0; r < kPer; ++r) {
      const int g = r * kStepThreads + tid;
      v[r] = g < n ? ld_agent(&ws.slot[g]) : 0;
    }
    STEP_STAMP(11, v[0] != 0x7fffffff);
    u64 bal[kPer], ibal[kPer];
#pragma unroll
    for (int r = 0; r < kPer; ++r) {
      const int img = v[r] >> 13;
      bal[r] = __ballot(v[r] != 0);
      ibal[r] = 0ull;
      if (lane == 0) s_pre[kStepMaxImages][r * 4 + wave] = __popcll(bal[r]);
      for (int b = 0; b < B; ++b) {
        const u64 bb = __ballot(v[r] != 0 && img == b);
        if (lane == 0) s_pre[b][r * 4 + wave] = __popcll(bb);
        if (img == b) ibal[r] = bb;
      }
    }
    __syncthreads();
    // exclusive scan of the 64 (round, wave) counts of every list: lane = entry, wave w takes lists w, w + 4, ...
    for (int l = wave; l <= kStepMaxImages; l += 4) {
      if (l < B || l == kStepMaxImages) {
        const int c = s_pre[l][lane];
        int incl = c;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
          const int o = __shfl_up(incl, d);
          if (lane >= d) incl += o;
        }
        s_pre[l][lane] = incl - c;
        if (lane == 63) s_tot[l] = incl;
      }
    }
    __syncthreads();
    STEP_STAMP(12, true);
    run = s_tot[kStepMaxImages];
#pragma unroll
    for (int r = 0; r < kPer; ++r) {
      if (v[r] != 0) {
        const int src = (v[r] & 0x1fff) - 1, img = v[r] >> 13;
        keep_out[s_pre[kStepMaxImages][r * 4 + wave] + __popcll(bal[r] & lt_mask)] = (int64_t)src;
        if (img < B) {
          const int rr = s_pre[img][r * 4 + wave] + __popcll(ibal[r] & lt_mask);
          if (rr < pk.max_dets) {
            float* d = pk.payload + (int64_t)img * pk.row_stride + (int64_t)rr * 6;
            const float4 bx = *reinterpret_cast<const float4*>(dets + (int64_t)src * 4);
            d[0] = bx.x;
            d[1] = bx.y;
            d[2] = bx.z;
            d[3] = bx.w;
            d[4] = scores[src];
            d[5] = pk.labels ? (float)pk.labels[src] : 0.f;
          }
        }
      }
    }
  }
  STEP_STAMP(13, true);
  if (tid == 0) *num_keep = failed ? -1 : run;
  for (int i = run + tid; i < n; i += kStepThreads) keep_out[i] = 0;   // the tail behind num: a deterministic output
  for (int b = 0; b < B; ++b) {
    const int c = failed ? 0 : min(s_tot[b], pk.max_dets);
    float* my = pk.payload + (int64_t)b * pk.row_stride;
    for (int i = c * 6 + tid; i < pk.max_dets * 6; i += kStepThreads) my[i] = 0.f;
    if (tid == 0) {
      const int cn = failed ? -1 : c;
      if (pk.counts) pk.counts[b] = cn;
      if (pk.count_in_row) my[(int64_t)pk.max_dets * 6] = (float)cn;
    }
  }
  // nobody reads the hand-over words any more (every workgroup has passed its last poll: the tickets say so): clean for the
  // next launch on this stream.  The word of a poll that gave up stays: such a stream keeps answering num = -1.
  for (int i = tid; i < kStepSyncWords; i += kStepThreads)
    if (i != 2) st_agent(&ws.sync[i], 0);
  STEP_STAMP(10, true);
}

// zero fill of the caller's hand-over words (the capture path of step_prepare).  A kernel node on purpose: with a hipMemsetAsync
// node in its place every hipGraph after the first one captured in a process replayed correctly ONCE and then left the outputs
// untouched (tools/_dbg_step.py of round 6; the test that was meant to cover replays compared against stale memory and passed)
__global__ __launch_bounds__(256) void step_zero_words(int* __restrict__ words) {
  for (int i = threadIdx.x; i < kStepSyncWords; i += 256) words[i] = 0;
}

// everything a launch hands to nms_step_body
struct StepArgs {
  const float* dets;
  const float* scores;
  const int64_t* seg;
  int n, S, G, gdim_y;   // gdim_y = tile groups + counting-only groups per segment: the launch has S x gdim_y workgroups
  double thr;
  ThrBand band;
  StepWorkspace ws;
  int64_t* keep_out;
  int64_t* num_keep;
  StepPack pk;
};

// argument checks, workspace carving, the stream's hand-over block (or the caller's words + a memset while capturing) and the
// shape of the launch — shared by tvmi_nms_step and the RoIAlign launch that carries the step
inline int step_prepare(StepArgs* a, const float* dets, const float* scores, const int64_t* seg, int64_t n, int64_t num_segments,
                        double iou_threshold, void* workspace, size_t workspace_bytes, int64_t* keep_out, int64_t* num_keep_out,
                        const int64_t* image_idx, const int64_t* labels, int64_t num_images, int64_t max_dets, float* payload,
                        int64_t row_stride, int32_t* counts, int count_in_row, hipStream_t s) {
  TVMI_CHECK_ARG(n >= 1 && n <= kSortMax, "nms_step: 1 <= n <= 4096");
  TVMI_CHECK_ARG(num_segments >= 1 && num_segments <= kStepMaxSegments, "nms_step: 1 <= num_segments <= 64");
  TVMI_CHECK_ARG(dets && scores && keep_out && num_keep_out && workspace, "nms_step: null pointer");
  TVMI_CHECK_ARG(seg != nullptr || num_segments == 1, "nms_step: segment ids are needed for more than one segment");
  TVMI_CHECK_ARG(workspace_bytes >= step_workspace_layout(n, num_segments, nullptr, nullptr), "nms_step: workspace too small");
  StepPack pk{};
  if (payload) {
    TVMI_CHECK_ARG(image_idx != nullptr, "nms_step: the payload needs the image index of every box");
    TVMI_CHECK_ARG(num_images >= 1 && num_images <= kStepMaxImages, "nms_step: 1 <= num_images <= 16 for the fused payload");
    TVMI_CHECK_ARG(max_dets >= 1 && row_stride >= max_dets * 6 + (count_in_row ? 1 : 0), "nms_step: bad payload shape");
    pk.image_idx = image_idx;
    pk.labels = labels;
    pk.payload = payload;
    pk.counts = counts;
    pk.row_stride = row_stride;
    pk.num_images = (int)num_images;
    pk.max_dets = (int)max_dets;
    pk.count_in_row = count_in_row;
  }
  StepWorkspace w;
  step_workspace_layout(n, num_segments, static_cast<char*>(workspace), &w);
  if (int* sync = nms_step_sync_block(s)) {
    w.sync = sync;                       // the library's block of this stream: clean, re-armed by the kernel
  } else {                               // graph capture (or no memory): the caller's words, zeroed by a launch in front
    step_zero_words<<<dim3(1), dim3(256), 0, s>>>(w.sync);
  }
  const int G = step_groups_per_segment(n, num_segments);
  // counting-only workgroups: enough for 16 key splits per 64 boxes, inside the cap on the launch's workgroups
  const int64_t want_rank = ceil_div(ceil_div(n, 64) * 4, num_segments);
  const int Rg = (int)std::max<int64_t>(0, std::min<int64_t>(want_rank, (kStepMaxGroups - num_segments * G) / num_segments));
  *a = StepArgs{dets, scores, seg, (int)n, (int)num_segments, G, G + Rg, iou_threshold, thr_band(iou_threshold), w, keep_out, num_keep_out, pk};
  return 0;
}

}  // namespace
}  // namespace tvmi
