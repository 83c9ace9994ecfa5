// nms.hip — greedy IoU non-maximum suppression for gfx950 (MI355X), wave64 bitmask form.
//
// Semantics and arithmetic: torchvision/csrc/ops/cpu/nms_kernel.cpp:17-95.  The result is
// the same index list, bit for bit, as the reference CPU kernel:
//   * areas[k] = (x2-x1)*(y2-y1); inter = max(0,xx2-xx1)*max(0,yy2-yy1);
//     ovr = inter / ((iarea + areas[j]) - inter), every operation individually rounded
//     (this TU is compiled with -ffp-contract=off; `/` is the IEEE-correct division);
//   * the test is `(double)ovr > iou_threshold` exactly as cpu/nms_kernel.cpp:88 promotes
//     it (the reference CUDA kernel compares against a float-rounded threshold instead,
//     cuda/nms_kernel.cu:45 — that is NOT what we reproduce);
//   * 0/0 = NaN compares false, so degenerate boxes neither suppress nor get suppressed.
//
// Structure (differs from cuda/nms_kernel.cu:56-148, whose second kernel is one block doing
// N serial steps with a barrier each):
//   K1 nms_mask_tiles  : one wave per upper-triangular 64x64 tile.  Lane = column box (in
//      registers), the 64 row boxes sit in LDS and are broadcast; for every row the 64-lane
//      predicate becomes the 64-bit suppression word through the compare itself (__ballot =
//      an SGPR pair) and is parked in lane `row`.  The mask is stored TILE-major: tile
//      (rb, cb) is 64 consecutive words, so both this store and every later read of a tile
//      is one coalesced 512-byte access.  Boxes are gathered through `order` in-kernel.
//      Two rows per step with packed fp32 and a division-free band test (see thr_band below); above 4096 boxes the
//      kernel runs in chunks of 64 row blocks and drops the rows the sweep already knows to be suppressed: the row block
//      is compacted as it is staged, the pair loop covers the survivors only.
//   The greedy sweep is blocked and runs UNDER the mask kernels on side streams (see launch()):
//   K2 nms_colreduce   : removed[cb] |= OR over the rows kept in a range of row blocks of tile(rb, cb)[row] —
//      thousands of independent waves, each a masked 512-byte load + a DPP OR-reduction + one atomic.  Used as the
//      PUSH of a resolved chunk's suppression to the later column blocks (near part on the sweep stream, far part on
//      its own stream).
//   K3 nms_resolve_wide: one 16-wave workgroup per chunk of 64 blocks, walked in mini super-blocks of 16.  Wave c owns
//      column block c: it pulls the tiles of the earlier mini super-blocks (keep bits final, in LDS), then the 16
//      diagonal blocks are resolved by parallel fixed-point rounds (exact when two rounds agree) with the serial walk
//      over the not-yet-removed bits (s_ff1 + v_readlane) as fallback.  Kept original indices are written in score
//      order and the running count stays on the device — no masked_select pass.
//   Large problems are RE-PLANNED on their survivors (tvmi_nms_blocking, see launch()): after the first chunks have been swept
//   and pushed to every later column, the score order is compacted to the boxes still alive and the pipeline starts over
//   on that list.  The resolver and the push kernels hand over through memory words (agent-scope atomics, no fences:
//   SweepSync) instead of stream events wherever the three streams are known to run concurrently.
//   Up to 4096 boxes one mask launch + nms_sweep_small; batched NMS: segment-major kernels further down.
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <utility>

#include <memory>
#include <mutex>
#include <vector>

#include "tvmi_common.h"
#include "nms_step_device.h"

namespace tvmi {
namespace {

constexpr int kMaskWaves = 4;    // waves (= tiles of one row block) per workgroup in K1
constexpr int kSuper = 16;       // 64-box blocks per super-block
constexpr int kReduceRows = 8;   // row blocks folded per wave in K2
constexpr int kWide = 64;        // 64-box blocks per chunk of the large path (one resolve launch)
// (the helpers shared with roi_align.hip — boxes, wave reductions, the suppression tile, score keys — live in nms_device.h)
// mask layout: tile (rb, cb) = 64 words at mask + (rb*CB + cb)*64; word r = row rb*64+r.
template <typename T>
__global__ __launch_bounds__(kMaskWaves * kWave) void nms_mask_tiles(
    const T* __restrict__ dets, const int64_t* __restrict__ order, const int64_t* __restrict__ seg, int n, int CB,
    double thr, ThrBand band, u64* __restrict__ mask, int rb0, const u64* removed) {
  __shared__ __attribute__((aligned(16))) T s_row[5][64];  // x1,y1,x2,y2,area of the row block, component-major
  __shared__ long long s_seg[64];
  __shared__ u64 s_skip;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int rb = rb0 + blockIdx.y;  // large problems are launched in chunks of row blocks (see launch())
  const int cb = blockIdx.x * kMaskWaves + wave;
  if ((int)(blockIdx.x * kMaskWaves + kMaskWaves - 1) < rb) return;  // whole workgroup left of the diagonal
  const int row0 = rb * 64;
  if (threadIdx.x < 64) {
    // Rows already known to be suppressed (by kept boxes of chunks the sweep has finished) are never read by anybody: a
    // kept row is by definition not removed, and both the column reduction and the resolve step only use kept rows.  The
    // word may be stale (the sweep runs concurrently on other streams and only ever ADDS bits): stale = fewer rows
    // dropped.  The surviving rows are COMPACTED to the front of the LDS block (the dropped ones and the rows past n
    // become zero boxes behind them), so that the tile loop below runs over ceil(survivors / 2) row pairs instead of
    // testing pair by pair whether both rows happen to be dropped; every wave of the workgroup uses the one word read here.
    u64 skip = 0ull;
    if (removed) skip = uniform64(__hip_atomic_load(&removed[rb], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    const int r = row0 + lane;
    const u64 below = (1ull << lane) - 1ull;
    const u64 surv = ~skip & __ballot(r < n);
    const bool live = (surv >> lane) & 1ull;
    const int slot = live ? __popcll(surv & below) : __popcll(surv) + __popcll(~surv & below);
    T x1 = 0, y1 = 0, x2 = 0, y2 = 0;
    long long sg = 0;
    if (live) {
      const int64_t oi = order[r];
      const Box<T> b = load_box<T>(dets, oi);
      x1 = b.x1;
      y1 = b.y1;
      x2 = b.x2;
      y2 = b.y2;
      if (seg) sg = seg[oi];
    }
    s_row[0][slot] = x1;
    s_row[1][slot] = y1;
    s_row[2][slot] = x2;
    s_row[3][slot] = y2;
    s_row[4][slot] = (x2 - x1) * (y2 - y1);
    s_seg[slot] = sg;
    if (lane == 0) s_skip = ~surv;  // dropped or non-existent
  }
  __syncthreads();
  if (cb < rb || cb >= CB) return;
  const int j = cb * 64 + lane;
  T jx1 = 0, jy1 = 0, jx2 = 0, jy2 = 0;
  long long jseg = 0;
  const bool jvalid = j < n;
  if (jvalid) {
    const int64_t oj = order[j];
    const Box<T> b = load_box<T>(dets, oj);
    jx1 = b.x1;
    jy1 = b.y1;
    jx2 = b.x2;
    jy2 = b.y2;
    if (seg) jseg = seg[oj];
  }
  const T jarea = (jx2 - jx1) * (jy2 - jy1);
  const u64 dropped = uniform64(s_skip);
  const int nsurv = 64 - __popcll(dropped);
  // compact row k sits in lane k of `packed`; the row pairs at and beyond nsurv are skipped as a whole
  const u64 packed = suppression_tile<T, 64>(&s_row[0][0], seg ? s_seg : nullptr, nsurv, jx1, jy1, jx2, jy2, jarea, jseg, jvalid,
                                             false, thr, band, nsurv >= 64 ? 0ull : ~0ull << nsurv);
  u64 mine = packed;
  if (dropped) {  // wave-uniform: back to one word per ORIGINAL row (lane r <- compact slot of row r; dropped rows: 0)
    const int src = __popcll(~dropped & ((1ull << lane) - 1ull)) << 2;
    const unsigned lo = (unsigned)__builtin_amdgcn_ds_bpermute(src, (int)(unsigned)packed);
    const unsigned hi = (unsigned)__builtin_amdgcn_ds_bpermute(src, (int)(unsigned)(packed >> 32));
    mine = ((dropped >> lane) & 1ull) ? 0ull : (((u64)hi << 32) | (u64)lo);
  }
  if (cb == rb) mine &= lane < 63 ? ~0ull << (lane + 1) : 0ull;  // diagonal tile: only the columns after the row
  mask[((size_t)rb * CB + cb) * 64 + lane] = mine;
}

// K2: fold the rows kept in a range of row blocks into removed[cb] for a range of column blocks (see launch()).
__global__ __launch_bounds__(256) void nms_colreduce(const u64* __restrict__ mask, const u64* __restrict__ keepbits,
                                                     u64* __restrict__ removed, int CB, int rlo, int rhi, int c0, int c1) {
  // removed[cb] |= OR over the rows kept in row blocks [rlo, rhi) of tile(rb, cb)[row], for cb in [c0, c1)
  __builtin_amdgcn_s_setprio(2);  // on the sweep's critical path, under the mask kernels of later chunks
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int cb = c0 + blockIdx.x;
  if (cb >= c1) return;
  const int rb_begin = rlo + (blockIdx.y * 4 + wave) * kReduceRows;
  u64 acc = 0ull;
#pragma unroll
  for (int q = 0; q < kReduceRows; ++q) {
    const int rb = rb_begin + q;
    if (rb < rhi) {
      const u64 w = mask[((size_t)rb * CB + cb) * 64 + lane];
      const u64 kb = keepbits[rb];
      if ((kb >> lane) & 1ull) acc |= w;
    }
  }
  const u64 red = wave_or64(acc);
  if (lane == 0 && red) atomicOr(&removed[cb], red);
}

// ---- device-side hand-offs of the sweep ("nms.device_handoff", see launch()).  The resolve workgroup of chunk c and the push
// kernel of chunk c run in DIFFERENT launches on different streams and meet through two words per chunk: `flag` (set by the
// resolver once the chunk's keep bits are in memory) and the arrival counter of the near-push kernel, which the resolver of
// the next chunk polls.  No fences: on the 8-XCD part an agent-scope release is a write-back
// of the whole L2 slice — which the mask kernel keeps full of dirty tiles — and cost more than the event hops it replaced
// (measured).  Instead everything that crosses between the two launches is an agent-scope (write-through /
// L2-bypassing) atomic — keepbits[], removed[], the three words — and a signal is preceded by `s_waitcnt vmcnt(0)`: the
// writer's stores and atomics are acknowledged before the word that announces them is written (the idiom nms_small_seg_sweep's
// ticket already uses).  The mask tiles a push kernel reads were written by a mask kernel that completed before the resolver
// started (stream event), and nobody reads a tile before it is written, so no stale copy of one can sit in any L2.
// A poll that outlives its bound (about a second: the launches it waits for are not running beside it — a tool that
// serialises kernels, a debugger, a co-tenant that keeps the resolver off the CUs) GIVES UP: it raises the call's `failed`
// word (one more SweepSync behind the per-chunk ones) and goes on with whatever is in memory; every later poll of the call
// sees the word and returns at once, so the launch chain drains in bounded time.  The host reads the word when the call has
// finished (tvmi_nms_blocking synchronises anyway) and re-runs the call in the stream-event form: a slower correct answer
// instead of a GPU fault (VERDICT r03 item 4 — this used to be __builtin_trap()).
struct SweepSync {  // one per chunk, zeroed with removed[] at the start of a level
  int flag, near_done, pad[2];
};
__device__ __forceinline__ void poll_until_equal(const int* p, int want, int* failed) {
  int polls = 0;
  while (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != want) {
    __builtin_amdgcn_s_sleep(4);
    ++polls;
    if ((polls & 1023) == 0 && __hip_atomic_load(failed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return;
    if (polls > (1 << 20)) {
      __hip_atomic_store(failed, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return;
    }
  }
}
__device__ __forceinline__ void signal_after_my_stores(int* p) {  // one thread, after the workgroup's barrier
  __hip_atomic_fetch_add(p, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// The NEAR push of chunk [b0, b1) with device-side hand-offs: a fixed grid of 4-wave workgroups that is ALREADY RESIDENT
// (polling the chunk's flag) when the resolver finishes; folds the chunk's kept rows into removed[] of the next chunk's column
// blocks [b1, b2) — all the next resolver still lacks — and reports.  The FAR push (every later column block) is the wide
// nms_colreduce launch right behind it on the same stream, whose workgroups report to the chunk's far counter.
constexpr int kPushGroups = 128;
__global__ __launch_bounds__(256) void nms_push(const u64* __restrict__ mask, const u64* keepbits, u64* __restrict__ removed, int CB,
                                                int b0, int b1, int b2, SweepSync* sync, int* failed) {
  __shared__ u64 s_kb[kWide];
  __builtin_amdgcn_s_setprio(2);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  if (threadIdx.x == 0) poll_until_equal(&sync->flag, 1, failed);
  __syncthreads();
  // this kernel started before the resolver wrote them: agent-scope loads (the far push, a later launch, reads them plainly)
  if ((int)threadIdx.x < b1 - b0) s_kb[threadIdx.x] = __hip_atomic_load(&keepbits[b0 + threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  const int groups = (b1 - b0 + kReduceRows - 1) / kReduceRows;
  const int gw = blockIdx.x * 4 + wave, nw = gridDim.x * 4;
  for (int item = gw; item < (b2 - b1) * groups; item += nw) {
    const int cb = b1 + item / groups, g = item % groups;
    u64 acc = 0ull;
#pragma unroll
    for (int q = 0; q < kReduceRows; ++q) {
      const int r = g * kReduceRows + q;   // row block within the chunk
      if (r < b1 - b0) {
        const u64 w = mask[((size_t)(b0 + r) * CB + cb) * 64 + lane];
        if ((s_kb[r] >> lane) & 1ull) acc |= w;
      }
    }
    const u64 red = wave_or64(acc);
    if (lane == 0 && red) atomicOr(&removed[cb], red);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // my atomics are acknowledged
  __syncthreads();
  if (threadIdx.x == 0) signal_after_my_stores(&sync->near_done);
}

// K3 for large problems: ONE workgroup resolves a WIDE super-block (chunk) of up to kWide 64-box blocks [b0, b1): it
// walks it in mini super-blocks of kSuper blocks; removed[] (filled by the pushes of the earlier chunks, nms_colreduce)
// carries what every earlier chunk suppresses, the rows of earlier mini super-blocks of THIS chunk are pulled here by
// the wave that owns the column (keep bits in LDS), then the diagonal blocks are resolved in registers.
constexpr int kJacobiRounds = 24;  // parallel fixed-point rounds tried per mini super-block before the serial walk
__global__ __launch_bounds__(kSuper * kWave) void nms_resolve_wide(const u64* __restrict__ mask,
                                                                   const int64_t* __restrict__ order,
                                                                   const u64* __restrict__ removed,
                                                                   u64* __restrict__ keepbits, int n, int CB, int b0, int b1,
                                                                   int64_t* __restrict__ keep_out,
                                                                   int64_t* __restrict__ num_keep, SweepSync* sync, int chunk,
                                                                   int* failed, int lose_flag) {
  __shared__ u64 s_keep[kWide];
  __shared__ u64 s_jac[2][kSuper];
  __shared__ int s_changed[3];
  __shared__ int s_base[kWide + 1];
  // this workgroup is the serial link of the sweep and shares the chip with the mask kernels of later chunks:
  // its waves take issue priority over whatever else is resident on the CU
  __builtin_amdgcn_s_setprio(3);
  const int lane = threadIdx.x & 63;
  const int c_loc = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int nb = b1 - b0;
  if (sync) {  // device hand-offs: removed[b0..b1) is complete once the push kernels of the two previous chunks have reported
    if (threadIdx.x == 0) {
      // (the far push of chunk - 2 precedes the near push of chunk - 1 on their in-order stream: one poll covers both)
      if (chunk >= 1) poll_until_equal(&sync[chunk - 1].near_done, kPushGroups, failed);
    }
    __syncthreads();
  }
  for (int m0 = 0; m0 < nb; m0 += kSuper) {
    const int lb = m0 + c_loc;  // local block of this wave in the wide super-block
    const int cb = b0 + lb;
    const bool have = lb < nb;
    u64 diag = 0ull, above[kSuper - 1];
    u64 rem = 0ull;
    if (have) {
      diag = mask[((size_t)cb * CB + cb) * 64 + lane];
      // written by atomics of push kernels that may still be running for LATER chunks: read at agent scope
      rem = __hip_atomic_load(&removed[cb], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
#pragma unroll
    for (int q = 0; q < kSuper - 1; ++q) {
      above[q] = 0ull;
      if (have && q < c_loc) above[q] = mask[((size_t)(b0 + m0 + q) * CB + cb) * 64 + lane];
    }
    u64 acc = 0ull;
    if (have) {
      // earlier mini super-blocks of this wide block: keep bits are final, in LDS.  16 independent tile loads per
      // round (one coalesced 512-byte access each) so that the memory latency is paid once per round, not per tile.
      // (Requesting all of them up front was measured: 128 VGPRs + spills at 1024 threads, slower.)
      for (int r0 = 0; r0 < m0; r0 += kSuper) {
        u64 w[kSuper];
#pragma unroll
        for (int q = 0; q < kSuper; ++q) w[q] = mask[((size_t)(b0 + r0 + q) * CB + cb) * 64 + lane];
#pragma unroll
        for (int q = 0; q < kSuper; ++q) acc |= ((s_keep[r0 + q] >> lane) & 1ull) ? w[q] : 0ull;
      }
    }
    rem = uniform64(rem) | (m0 > 0 ? wave_or64(acc) : 0ull);
    const int rows_here = have ? min(64, n - cb * 64) : 0;
    const u64 valid = rows_here >= 64 ? ~0ull : ((1ull << rows_here) - 1ull);
    // ---- the kSuper x kSuper tiles of this mini super-block form a strictly upper-triangular system
    //          keep_c = ~(rem_c | OR over kept rows r of blocks q <= c of tile(q, c)[r]) & valid,
    // whose unique solution is the greedy sweep's answer.  Instead of walking its 16 blocks one barrier-separated step
    // at a time, iterate it in parallel (all 16 waves at once, Jacobi): after t rounds every box whose chain of
    // "suppressed by a box that is itself suppressed by ..." is at most t long is final, and a round without change
    // is the fixed point.  Real detections need a handful of rounds; a chain longer than kJacobiRounds falls back to
    // the serial walk below (same answer, old speed).
    u64 my_keep = ~rem & valid;
    bool converged = false;
    if (lane == 0) s_jac[0][c_loc] = have ? my_keep : 0ull;
    if (threadIdx.x == 0) s_changed[0] = 0;
    for (int it = 0; it < kJacobiRounds; ++it) {
      if (threadIdx.x == 0) s_changed[(it + 1) % 3] = 0;
      __syncthreads();
      const u64* cur = s_jac[it & 1];
      u64 contrib = 0ull;
#pragma unroll
      for (int q = 0; q < kSuper - 1; ++q) contrib |= ((cur[q] >> lane) & 1ull) ? above[q] : 0ull;  // above[q] = 0 for q >= c_loc
      contrib |= ((cur[c_loc] >> lane) & 1ull) ? diag : 0ull;
      const u64 nk = have ? (~(rem | wave_or64(contrib)) & valid) : 0ull;
      if (lane == 0) {
        s_jac[(it + 1) & 1][c_loc] = nk;
        if (nk != my_keep) s_changed[it % 3] = 1;
      }
      my_keep = nk;
      __syncthreads();
      if (s_changed[it % 3] == 0) {
        converged = true;
        break;
      }
    }
    if (converged) {
      if (lane == 0 && have) {
        s_keep[lb] = my_keep;
        __hip_atomic_store(&keepbits[cb], my_keep, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      __syncthreads();
      continue;
    }
#pragma unroll
    for (int step = 0; step < kSuper; ++step) {
      if (step == c_loc && have) {
        u64 r = uniform64(rem);
        u64 active = uniform64(__ballot(diag != 0ull)) & ~r & valid;
        while (active) {
          const int k = __builtin_ctzll(active);
          r |= readlane64(diag, k);
          active &= ~(r | (1ull << k));
        }
        if (lane == 0) {
          s_keep[lb] = ~r & valid;
          __hip_atomic_store(&keepbits[cb], ~r & valid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      __syncthreads();
      if (step < kSuper - 1 && have && step < c_loc) {
        const u64 kb = s_keep[m0 + step];
        const u64 contrib = ((kb >> lane) & 1ull) ? above[step] : 0ull;
        rem |= wave_or64(contrib);
      }
    }
  }
  if (sync) {  // the keep bits of the chunk are acknowledged stores: release the push kernel before the index list is emitted
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // lose_flag: test hook ("nms.handoff_lose_flag"): chunk 0 never announces itself, as if this workgroup had not run
    // beside the push kernel — the recovery path (poll gives up, host re-runs with stream events) is then exercised for real
    if (threadIdx.x == 0 && !(lose_flag && chunk == 0)) signal_after_my_stores(&sync[chunk].flag);
  }
  // append the kept original indices in score order
  if (threadIdx.x == 0) {
    int run = 0;
    for (int b = 0; b < nb; ++b) {
      s_base[b] = run;
      run += __popcll(s_keep[b]);
    }
    s_base[nb] = run;
  }
  __syncthreads();
  const int64_t base = *num_keep;
  for (int b = c_loc; b < nb; b += kSuper) {
    const u64 kb = s_keep[b];
    if ((kb >> lane) & 1ull) {
      const u64 below = kb & ((1ull << lane) - 1ull);
      keep_out[base + s_base[b] + __popcll(below)] = order[(int64_t)(b0 + b) * 64 + lane];
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) *num_keep = base + s_base[nb];
}

// K3': the whole sweep of a small problem (CB <= kSmallCB column blocks, i.e. N <= 4096) in ONE
// workgroup and ONE launch: the super-blocks are walked in order, the column reduction over
// earlier super-blocks is done by the wave that owns the column (its lanes accumulate the masked
// tile words, one DPP reduction at the end) with the keep bits held in LDS, then the same
// register-resident resolve chain as nms_resolve.  Removes 2 memsets + 2 launches per
// super-block from the latency-bound case (RPN / box-head sizes).
constexpr int kSmallCB = 64;
__global__ __launch_bounds__(kSuper * kWave) void nms_sweep_small(const u64* __restrict__ mask,
                                                                  const int64_t* __restrict__ order, int n, int CB,
                                                                  int64_t* __restrict__ keep_out,
                                                                  int64_t* __restrict__ num_keep, bool append) {
  __shared__ u64 s_keepbits[kSmallCB];
  __shared__ int s_base[kSmallCB + 1];
  const int lane = threadIdx.x & 63;
  const int c_loc = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  for (int b0 = 0; b0 < CB; b0 += kSuper) {
    const int b1 = min(CB, b0 + kSuper);
    const int cb = b0 + c_loc;
    const bool have = cb < b1;
    u64 diag = 0ull, above[kSuper - 1];
    if (have) diag = mask[((size_t)cb * CB + cb) * 64 + lane];
#pragma unroll
    for (int q = 0; q < kSuper - 1; ++q) {
      above[q] = 0ull;
      if (have && q < c_loc) above[q] = mask[((size_t)(b0 + q) * CB + cb) * 64 + lane];
    }
    // pull from every earlier super-block (their keep bits are final and in LDS)
    u64 acc = 0ull;
    if (have) {
      for (int rb = 0; rb < b0; ++rb) {
        const u64 w = mask[((size_t)rb * CB + cb) * 64 + lane];
        if ((s_keepbits[rb] >> lane) & 1ull) acc |= w;
      }
    }
    u64 rem = b0 > 0 ? wave_or64(acc) : 0ull;
    const int rows_here = have ? min(64, n - cb * 64) : 0;
    const u64 valid = rows_here >= 64 ? ~0ull : ((1ull << rows_here) - 1ull);
#pragma unroll
    for (int step = 0; step < kSuper; ++step) {
      if (step == c_loc && have) {
        u64 r = uniform64(rem);
        u64 active = uniform64(__ballot(diag != 0ull)) & ~r & valid;
        while (active) {
          const int k = __builtin_ctzll(active);
          r |= readlane64(diag, k);
          active &= ~(r | (1ull << k));
        }
        if (lane == 0) s_keepbits[cb] = ~r & valid;
      }
      __syncthreads();
      if (step < kSuper - 1 && have && step < c_loc) {
        const u64 kb = s_keepbits[b0 + step];
        const u64 contrib = ((kb >> lane) & 1ull) ? above[step] : 0ull;
        rem |= wave_or64(contrib);
      }
    }
  }
  // exclusive prefix of the per-block keep counts, then every wave appends its blocks' indices
  if (threadIdx.x == 0) {
    int run = 0;
    for (int b = 0; b < CB; ++b) {
      s_base[b] = run;
      run += __popcll(s_keepbits[b]);
    }
    s_base[CB] = run;
  }
  const int64_t base = append ? *num_keep : 0;  // append: the survivors of a re-planned large problem (see launch())
  __syncthreads();
  for (int b = c_loc; b < CB; b += kSuper) {
    const u64 kb = s_keepbits[b];
    if ((kb >> lane) & 1ull) {
      const u64 below = kb & ((1ull << lane) - 1ull);
      keep_out[base + s_base[b] + __popcll(below)] = order[(int64_t)b * 64 + lane];
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) *num_keep = base + s_base[CB];
}

// ---- re-planning on the survivors (see launch()): the boxes of column blocks >= pb that no kept box has removed,
// in score order.  Two small kernels: per-block survivor counts -> exclusive offsets (+ the total, also written to
// pinned host memory), then one wave per block copies its surviving entries of `order`.
__global__ __launch_bounds__(1024) void nms_survivor_offsets(const u64* __restrict__ removed, int n, int CB, int pb,
                                                             int* __restrict__ offsets, int* __restrict__ total_host) {
  __shared__ int s_part[1024];
  const int words = CB - pb;
  const int per = (words + 1023) / 1024;
  const int w0 = threadIdx.x * per, w1 = min(words, w0 + per);
  int sum = 0;
  for (int w = w0; w < w1; ++w) {
    const int rows = min(64, n - (pb + w) * 64);
    const u64 valid = rows >= 64 ? ~0ull : ((1ull << rows) - 1ull);
    sum += __popcll(~removed[pb + w] & valid);
  }
  s_part[threadIdx.x] = sum;
  __syncthreads();
  for (int d = 1; d < 1024; d <<= 1) {  // inclusive Hillis-Steele scan of the 1024 partial sums
    const int v = threadIdx.x >= d ? s_part[threadIdx.x - d] : 0;
    __syncthreads();
    s_part[threadIdx.x] += v;
    __syncthreads();
  }
  int run = s_part[threadIdx.x] - sum;
  for (int w = w0; w < w1; ++w) {
    const int rows = min(64, n - (pb + w) * 64);
    const u64 valid = rows >= 64 ? ~0ull : ((1ull << rows) - 1ull);
    offsets[w] = run;
    run += __popcll(~removed[pb + w] & valid);
  }
  if (threadIdx.x == 1023) *total_host = s_part[1023];
}
__global__ __launch_bounds__(256) void nms_compact_order(const u64* __restrict__ removed, const int64_t* __restrict__ order,
                                                         int n, int CB, int pb, const int* __restrict__ offsets,
                                                         int64_t* __restrict__ order_out) {
  const int lane = threadIdx.x & 63;
  const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (w >= CB - pb) return;
  const int rows = min(64, n - (pb + w) * 64);
  const u64 valid = rows >= 64 ? ~0ull : ((1ull << rows) - 1ull);
  const u64 surv = ~removed[pb + w] & valid;
  if ((surv >> lane) & 1ull) order_out[offsets[w] + __popcll(surv & ((1ull << lane) - 1ull))] = order[(int64_t)(pb + w) * 64 + lane];
}

__global__ void nms_report_failed(const int* __restrict__ failed, int* __restrict__ host_word) { *host_word = *failed; }

// Fork / join helpers of the large-problem path.  The mask kernel (throughput-bound, fills the chip) is launched in
// chunks of kWide row blocks on one internal stream; the sweep of chunk c (latency-bound: one resolve workgroup plus
// the near push) runs on a second, higher-priority stream as soon as chunk c of the mask is complete — i.e. UNDER the
// mask kernels of the later chunks instead of after them; the far pushes run on a third stream (see launch()).  All
// three are forked from and joined back into the caller's stream with events, so the caller sees ordinary
// stream-ordered behaviour (and the pattern is capturable).  Streams / events are cached per host thread and device.
struct SweepStreams {
  hipStream_t mask_stream = nullptr, sweep_stream = nullptr, far_stream = nullptr;
  hipEvent_t fork = nullptr, join = nullptr, join_far = nullptr;
  std::vector<hipEvent_t> chunk_done, resolved, far_done;
  int* host_count = nullptr;  // pinned host word the survivor count of a re-planned problem is read back through
  bool ready = false;
  bool ensure(int nchunks) {
    if (!ready) {  // first use on this thread for this device (the caller has made it current)
      int lo = 0, hi = 0;
      (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
      if (hipStreamCreateWithPriority(&mask_stream, hipStreamNonBlocking, lo) != hipSuccess) return false;
      if (hipStreamCreateWithPriority(&sweep_stream, hipStreamNonBlocking, hi) != hipSuccess) return false;
      if (hipStreamCreateWithPriority(&far_stream, hipStreamNonBlocking, hi) != hipSuccess) return false;
      if (hipEventCreateWithFlags(&fork, hipEventDisableTiming) != hipSuccess) return false;
      if (hipEventCreateWithFlags(&join, hipEventDisableTiming) != hipSuccess) return false;
      if (hipEventCreateWithFlags(&join_far, hipEventDisableTiming) != hipSuccess) return false;
      if (hipHostMalloc(reinterpret_cast<void**>(&host_count), 64, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        host_count = nullptr;  // no re-planning, everything else works
      }
      ready = true;
    }
    for (std::vector<hipEvent_t>* v : {&chunk_done, &resolved, &far_done})
      while ((int)v->size() < nchunks) {
        hipEvent_t e;
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return false;
        v->push_back(e);
      }
    return true;
  }
  // Handles are destroyed when the owning host thread exits.  Errors are ignored on purpose: at process teardown the
  // HIP runtime may already be gone (hipErrorDeinitialized), and a handle whose work is still queued is released by
  // the runtime once that work has drained.
  ~SweepStreams() {
    for (std::vector<hipEvent_t>* v : {&chunk_done, &resolved, &far_done})
      for (hipEvent_t e : *v) (void)hipEventDestroy(e);
    for (hipEvent_t e : {fork, join, join_far})
      if (e) (void)hipEventDestroy(e);
    for (hipStream_t st : {mask_stream, sweep_stream, far_stream})
      if (st) (void)hipStreamDestroy(st);
    if (host_count) (void)hipHostFree(host_count);
  }
};
// one set per (host thread, device): a thread that alternates between GPUs keeps both sets instead of re-creating
// (and leaking) streams on every switch
struct SweepStreamsByDevice {
  std::vector<std::unique_ptr<SweepStreams>> per_device;
  SweepStreams* current() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0) return nullptr;
    if ((int)per_device.size() <= dev) per_device.resize(dev + 1);
    if (!per_device[dev]) per_device[dev] = std::make_unique<SweepStreams>();
    return per_device[dev].get();
  }
};
thread_local SweepStreamsByDevice g_sweep_streams;

// ONE call per device at a time may use the device-side hand-offs.  Every poll in them targets work that the SAME host thread
// enqueued earlier, so one call can never block itself however its streams are mapped to hardware queues — but two host
// threads whose polling kernels sit at the heads of each other's shared in-order queues could.  A second call that arrives
// while the first one's work is still on the device therefore takes the stream-event form (whose kernels never wait).
struct HandoffClaim {
  std::mutex mu;
  bool enqueuing = false;     // a call that took the hand-off form is still enqueueing its launches
  hipEvent_t done = nullptr;  // recorded behind the last such call
};
HandoffClaim g_handoff_claim[64];
// A profiler that collects hardware counters per dispatch runs ONE kernel at a time (rocprofv3 --pmc: measured — the polling
// push kernel then waits for a resolver that is never started).  rocprofv3 announces that mode to the process it launches.
// HIP_LAUNCH_BLOCKING / AMD_SERIALIZE_KERNEL serialise every launch of the process, and the older rocprof generations announce
// counter collection through ROCP_* variables.  What none of these catch (a debugger, a co-tenant that starves the resolver) is
// caught on the device: a poll gives up after about a second, the call is re-run in the stream-event form and the process
// stops using the hand-offs (g_handoff_broken).
inline bool env_nonzero(const char* name) {
  const char* v = std::getenv(name);
  return v != nullptr && v[0] != '\0' && !(v[0] == '0' && v[1] == '\0');
}
inline bool kernels_are_serialised() {
  static const bool yes = std::getenv("ROCPROF_COUNTER_COLLECTION") != nullptr || std::getenv("ROCPROF_COUNTERS") != nullptr ||
                          std::getenv("ROCP_METRICS") != nullptr || std::getenv("ROCP_INPUT") != nullptr ||
                          env_nonzero("HIP_LAUNCH_BLOCKING") || env_nonzero("AMD_SERIALIZE_KERNEL") ||
                          env_nonzero("CUDA_LAUNCH_BLOCKING") || env_nonzero("TVMI_TOOL_SERIALIZED_PROFILER");
  return yes;
}
std::atomic<bool> g_handoff_broken{false};   // a poll of this process has timed out once: stream events from now on
inline bool claim_handoff(int dev) {
  if (dev < 0 || dev >= 64 || kernels_are_serialised() || g_handoff_broken.load(std::memory_order_relaxed)) return false;
  HandoffClaim& h = g_handoff_claim[dev];
  std::lock_guard<std::mutex> lock(h.mu);
  if (h.enqueuing) return false;
  if (h.done && hipEventQuery(h.done) != hipSuccess) {
    (void)hipGetLastError();   // hipErrorNotReady is not an error of this call
    return false;
  }
  h.enqueuing = true;
  return true;
}
inline void release_handoff(int dev, hipStream_t stream) {
  HandoffClaim& h = g_handoff_claim[dev];
  std::lock_guard<std::mutex> lock(h.mu);
  if (!h.done && hipEventCreateWithFlags(&h.done, hipEventDisableTiming) != hipSuccess) h.done = nullptr;
  if (h.done) (void)hipEventRecord(h.done, stream);
  h.enqueuing = false;
}

// Options of the large path (tvmi_set_option): "nms.replan_min_boxes" — problems at least this large are re-planned on
// their survivors (0 = never); "nms.replan_divisor" — the first 1/divisor of the row chunks is swept before the re-plan;
// "nms.replan_max" — how many times one call may re-plan; "nms.mask_lds_bytes" — dynamic LDS per mask workgroup.
std::atomic<int64_t> g_replan_min_boxes{24576};
std::atomic<int> g_replan_divisor{16}, g_replan_max{3}, g_mask_lds_bytes{36000}, g_device_handoff{1}, g_handoff_lose_flag{0};
std::atomic<bool> g_step_fused{true};    // "nms.step_fused": detector-step sizes through the one-launch kernel (read by the glue)

// Workspace of the large path: mask tiles | removed[CB] | keepbits[CB] | survivor offsets[CB] (int) | two score-order
// buffers of n indices (re-planning ping-pongs between them).
struct LargeWorkspace {
  u64 *mask, *removed, *keepbits;
  SweepSync* sync;   // one per chunk, directly behind keepbits[] (cleared with them)
  int* offsets;
  int64_t* order_buf[2];
  int* failed;       // raised by a hand-off poll that gave up (sticky for the whole call: outside the per-level clears)
};
inline size_t large_state_bytes(size_t n) {
  const size_t CB = ceil_div(n, (size_t)64);
  return 2 * CB * sizeof(u64) + ceil_div(CB, (size_t)kWide) * sizeof(SweepSync) + ceil_div(CB, (size_t)2) * 2 * sizeof(int) +
         2 * n * sizeof(int64_t) + 16;
}
inline size_t sweep_state_bytes(size_t CB) { return 2 * CB * sizeof(u64) + ceil_div(CB, (size_t)kWide) * sizeof(SweepSync); }
inline LargeWorkspace carve(void* workspace, size_t n) {
  const size_t CB = ceil_div(n, (size_t)64);
  LargeWorkspace w;
  w.mask = static_cast<u64*>(workspace);
  w.removed = w.mask + CB * CB * 64;
  w.keepbits = w.removed + CB;
  w.sync = reinterpret_cast<SweepSync*>(w.keepbits + CB);
  w.offsets = reinterpret_cast<int*>(w.sync + ceil_div(CB, (size_t)kWide));
  w.order_buf[0] = reinterpret_cast<int64_t*>(w.offsets + ceil_div(CB, (size_t)2) * 2);
  w.order_buf[1] = w.order_buf[0] + n;
  w.failed = reinterpret_cast<int*>(w.order_buf[1] + n);
  return w;
}

template <typename T>
int launch(const void* dets, const int64_t* order, const int64_t* seg, int64_t n, double thr, void* workspace,
           int64_t* keep_out, int64_t* num_keep, hipStream_t stream, bool may_sync, bool allow_handoff = true) {
  const T* d = static_cast<const T*>(dets);
  const int64_t n_call = n;   // a hand-off that gave up re-runs the call from here
  const ThrBand band = thr_band(thr);
  if (ceil_div(n, 64) <= kSmallCB) {  // latency-bound sizes: one mask launch, the whole sweep is one more
    const int CB = (int)ceil_div(n, 64);
    u64* mask = static_cast<u64*>(workspace);
    const dim3 grid((unsigned)ceil_div(CB, kMaskWaves), (unsigned)CB);
    nms_mask_tiles<T><<<grid, dim3(kMaskWaves * kWave), 0, stream>>>(d, order, seg, (int)n, CB, thr, band, mask, 0, nullptr);
    nms_sweep_small<<<dim3(1), dim3(kSuper * kWave), 0, stream>>>(mask, order, (int)n, CB, keep_out, num_keep, false);
    TVMI_RETURN_LAUNCH_STATUS("tvmi_nms");
  }
  const LargeWorkspace ws = carve(workspace, (size_t)n);
  u64 *mask = ws.mask, *removed = ws.removed, *keepbits = ws.keepbits;
  const size_t state_bytes = sweep_state_bytes((size_t)ceil_div(n, 64));   // removed[] + keepbits[] + the hand-off words
  hipError_t e = hipMemsetAsync(num_keep, 0, sizeof(int64_t), stream);
  if (e == hipSuccess) e = hipMemsetAsync(ws.failed, 0, sizeof(int), stream);
  if (e != hipSuccess) return set_error((int)e, "tvmi_nms: memset");
  static SweepStreams no_streams;  // never ensure()d: only names the members below when the device query failed
  SweepStreams* ssp = g_sweep_streams.current();
  SweepStreams& ss = ssp ? *ssp : no_streams;
  // RE-PLANNING ON THE SURVIVORS.  The pair tests of a sorted list are wasted on boxes that an early, high-scoring
  // box has already removed: after the first eighth of the rows has been swept (and its removals pushed to EVERY later
  // column) typically half (sparse scenes) to nine tenths (dense scenes) of the remaining boxes are gone.  Dropping
  // the removed ROWS of later tiles (nms_mask_tiles) only saves that fraction once; restarting on the compacted list
  // of survivors saves it in both dimensions, and shortens the serial sweep by the same factor.  The price is one
  // host synchronisation per re-plan (the survivor count sizes the next grids), so it is only done for large
  // problems, never under stream capture, and `tvmi_set_option("nms.replan_min_boxes", 0)` turns it off.
  hipStreamCaptureStatus capture = hipStreamCaptureStatusNone;
  const bool capturing = !(hipStreamIsCapturing(stream, &capture) == hipSuccess && capture == hipStreamCaptureStatusNone);
  const bool may_replan = may_sync && ssp && !capturing;
  const size_t mask_lds = (size_t)g_mask_lds_bytes.load(std::memory_order_relaxed);
  const int64_t replan_min = g_replan_min_boxes.load(std::memory_order_relaxed);
  const int divisor = std::max(2, g_replan_divisor.load(std::memory_order_relaxed));
  int replans_left = g_replan_max.load(std::memory_order_relaxed);
  const int64_t* cur = order;
  int flip = 0;
  bool ok = true;
  bool claim_tried = false, claimed = false;   // the device-side hand-offs of this call (see HandoffClaim)
  int claim_dev = -1;
  struct ClaimRelease {   // on every way out: "done" is recorded behind whatever this call enqueued on the caller's stream
    bool& claimed;
    int& dev;
    hipStream_t stream;
    ~ClaimRelease() {
      if (claimed) release_handoff(dev, stream);
    }
  } claim_release{claimed, claim_dev, stream};
  bool side_streams_open = false;  // a re-plan leaves the three side streams forked, idle and ahead of `stream`
  // Everything the side streams did is ordered before what the caller enqueues next: the sweep stream has waited for every
  // mask chunk, so joining it and the far stream joins all three.  BOTH on every way out: with device hand-offs
  // the far stream is released by the resolver's flag, which is raised BEFORE the resolver appends to keep_out / num_keep.
  auto join_side_streams = [&]() {
    bool j = hipEventRecord(ss.join, ss.sweep_stream) == hipSuccess && hipStreamWaitEvent(stream, ss.join, 0) == hipSuccess;
    j = (hipEventRecord(ss.join_far, ss.far_stream) == hipSuccess && hipStreamWaitEvent(stream, ss.join_far, 0) == hipSuccess) && j;
    side_streams_open = false;
    return j;
  };
  while (true) {
    const int CB = (int)ceil_div(n, 64);
    if (CB <= kSmallCB) {  // what survived a re-plan fits the one-workgroup sweep: append to the list (streams are idle)
      const dim3 grid((unsigned)ceil_div(CB, kMaskWaves), (unsigned)CB);
      nms_mask_tiles<T><<<grid, dim3(kMaskWaves * kWave), 0, stream>>>(d, cur, seg, (int)n, CB, thr, band, mask, 0, nullptr);
      nms_sweep_small<<<dim3(1), dim3(kSuper * kWave), 0, stream>>>(mask, cur, (int)n, CB, keep_out, num_keep, true);
      break;
    }
    const int nchunks = (int)ceil_div(CB, kWide);
    bool forked = side_streams_open;
    if (!forked) {
      e = hipMemsetAsync(removed, 0, state_bytes, stream);
      if (e != hipSuccess) return set_error((int)e, "tvmi_nms: memset");   // (the side streams are not forked here)
      forked = ssp && ss.ensure(nchunks) && hipEventRecord(ss.fork, stream) == hipSuccess &&
               hipStreamWaitEvent(ss.mask_stream, ss.fork, 0) == hipSuccess &&
               hipStreamWaitEvent(ss.sweep_stream, ss.fork, 0) == hipSuccess &&
               hipStreamWaitEvent(ss.far_stream, ss.fork, 0) == hipSuccess;
    }
    hipStream_t ms = forked ? ss.mask_stream : stream, sw = forked ? ss.sweep_stream : stream, fs = forked ? ss.far_stream : stream;
    // device-side hand-offs need the three streams to really run concurrently: not under capture (a graph may serialise
    // its branches — may_sync is false there anyway only for tvmi_nms, so ask), and not without the side streams
    // ... and only where the call may synchronise at its end to learn whether a poll gave up (tvmi_nms_blocking)
    if (!claim_tried && forked && !capturing && may_sync && allow_handoff && ss.host_count &&
        g_device_handoff.load(std::memory_order_relaxed) != 0) {
      claim_tried = true;
      (void)hipGetDevice(&claim_dev);
      claimed = claim_handoff(claim_dev);
    }
    const bool handoff = forked && claimed;
    const bool replan = may_replan && forked && ss.host_count && replans_left > 0 && replan_min > 0 && n >= replan_min;
    const int limit = replan ? std::max(1, (nchunks + divisor / 2) / divisor) : nchunks;  // row chunks swept at this level
    auto mask_chunk = [&](int c, bool skip_known) {
      const int r0 = c * kWide, r1 = std::min(CB, r0 + kWide);
      // row block r only has tiles for column blocks >= r: workgroups left of the chunk's first row block exit at once
      const dim3 grid((unsigned)ceil_div(CB, kMaskWaves), (unsigned)(r1 - r0));
      // From the second chunk on a resolve step runs under the mask kernel.  Its ONE 16-wave workgroup can only start on
      // a CU with four free wave slots on every SIMD, which a chip saturated by 4-wave workgroups offers rarely
      // (kernel-trace: resolve 33 us alone, 110-180 us under a full mask kernel).  Dynamic LDS the kernel never touches
      // caps these launches at four workgroups (16 of 32 wave slots) per CU: the resolver starts at once (37-47 us),
      // and the rows its pushes remove are dropped by the mask workgroups that start after them.  The cap costs the
      // mask kernel ~13 % of its throughput, so the first chunk — which runs alone — is launched without it.
      nms_mask_tiles<T><<<grid, dim3(kMaskWaves * kWave), c >= 1 ? mask_lds : 0, ms>>>(d, cur, seg, (int)n, CB, thr, band, mask, r0,
                                                                                      skip_known ? removed : nullptr);
    };
    auto push = [&](hipStream_t st, int rlo, int rhi, int c0, int c1) {  // rows kept in [rlo, rhi) -> removed[c0..c1)
      if (c1 <= c0 || rhi <= rlo) return;
      const dim3 rgrid((unsigned)(c1 - c0), (unsigned)ceil_div(rhi - rlo, 4 * kReduceRows));
      nms_colreduce<<<rgrid, dim3(256), 0, st>>>(mask, keepbits, removed, CB, rlo, rhi, c0, c1);
    };
    auto resolve = [&](int c) {
      const int b0 = c * kWide, b1 = std::min(CB, b0 + kWide);
      nms_resolve_wide<<<dim3(1), dim3(kSuper * kWave), 0, sw>>>(mask, cur, removed, keepbits, (int)n, CB, b0, b1, keep_out,
                                                                 num_keep, handoff ? ws.sync : nullptr, c, ws.failed,
                                                                 handoff ? g_handoff_lose_flag.load(std::memory_order_relaxed) : 0);
    };
    if (forked) {
      // PUSH pipeline on three streams.  Chunk c: its mask tiles (mask stream), then — sweep stream, the serial link —
      // resolve(c) and the NEAR push (rows kept in chunk c -> removed[] of chunk c + 1, all resolve(c + 1) still lacks),
      // then — far stream, off the critical path — the FAR push to every later column block, which has the whole of
      // resolve(c + 1) to finish before resolve(c + 2) needs it.  Because removed[] of a chunk is now filled as the
      // sweep advances (not just before its own resolve), the mask kernel of chunk c can DROP the rows that are already
      // known to be suppressed; it is held back until the far push of chunk c - 3 is done, so that it sees (at least)
      // every removal caused by chunks <= c - 3 — in a sorted list that is nearly all of them.
      for (int c = 0; c < limit; ++c) {
        const int b0 = c * kWide, b1 = std::min(CB, b0 + kWide), b2 = std::min(CB, b1 + kWide);
        const bool last_before_replan = replan && c == limit - 1 && limit < nchunks;
        if (c >= 3) ok = ok && hipStreamWaitEvent(ms, ss.far_done[c - 3], 0) == hipSuccess;
        mask_chunk(c, true);
        ok = ok && hipEventRecord(ss.chunk_done[c], ms) == hipSuccess && hipStreamWaitEvent(sw, ss.chunk_done[c], 0) == hipSuccess;
        if (handoff) {
          // Device-side hand-offs: the sweep stream carries ONE event wait per chunk (the chunk's tiles) and back-to-back
          // resolve launches; the push kernel of the chunk is already resident on the far stream, polling the resolver's
          // flag, and the resolvers of the next two chunks poll its two arrival counters.  An event hop between streams
          // costs ~10 us of command-processor latency here and the event form has two of them on the serial chain of
          // every chunk (58 us per chunk for 33 us of resolve).  Every wait targets work that was enqueued earlier, so
          // any interleaving of the three queues — even a single shared one — makes progress.
          resolve(c);
          if (b1 < CB) nms_push<<<dim3(kPushGroups), dim3(256), 0, fs>>>(mask, keepbits, removed, CB, b0, b1, b2, ws.sync + c, ws.failed);
          // the FAR push needs no counter: the far stream is in order, so the near push of chunk c + 1 — whose counter the
          // resolver of chunk c + 2 polls — cannot even start before this launch has finished
          push(fs, b0, b1, b2, CB);
          ok = ok && hipEventRecord(ss.far_done[c], fs) == hipSuccess;
          (void)last_before_replan;
          continue;
        }
        if (c >= 2) ok = ok && hipStreamWaitEvent(sw, ss.far_done[c - 2], 0) == hipSuccess;
        resolve(c);
        if (!last_before_replan) push(sw, b0, b1, b1, b2);
        ok = ok && hipEventRecord(ss.resolved[c], sw) == hipSuccess && hipStreamWaitEvent(fs, ss.resolved[c], 0) == hipSuccess;
        push(fs, b0, b1, last_before_replan ? b1 : b2, CB);  // nobody waits for a near part before a re-plan: one launch
        ok = ok && hipEventRecord(ss.far_done[c], fs) == hipSuccess;
      }
    } else {  // no side streams (creation failed): same kernels, serially on the caller's stream
      for (int c = 0; c < nchunks; ++c) mask_chunk(c, false);
      for (int c = 0; c < nchunks; ++c) {
        const int b0 = c * kWide, b1 = std::min(CB, b0 + kWide);
        resolve(c);
        push(stream, b0, b1, b1, CB);
      }
    }
    if (!ok || limit >= nchunks) {
      if (forked) ok = join_side_streams() && ok;
      break;
    }
    // Every kept row of blocks < pb has been pushed to every column once the far stream is through: compact the rest
    // there (no event hop), clear the sweep state for the next level, and wait for exactly that stream — everything
    // the other two did at this level precedes it.  The side streams stay forked: the next level's launches go
    // straight to them, an idle machine needs no fork.
    const int pb = limit * kWide;
    int64_t* next = ws.order_buf[flip];
    flip ^= 1;
    nms_survivor_offsets<<<dim3(1), dim3(1024), 0, fs>>>(removed, (int)n, CB, pb, ws.offsets, ss.host_count);
    nms_compact_order<<<dim3((unsigned)ceil_div(CB - pb, 4)), dim3(256), 0, fs>>>(removed, cur, (int)n, CB, pb, ws.offsets, next);
    e = hipMemsetAsync(removed, 0, state_bytes, fs);
    if (e == hipSuccess) e = hipStreamSynchronize(fs);
    if (e != hipSuccess) {
      (void)join_side_streams();   // work may still be queued on the side streams while the caller frees the workspace
      return set_error((int)e, "tvmi_nms: synchronising for the survivor count");
    }
    side_streams_open = true;
    const int survivors = *static_cast<volatile int*>(ss.host_count);
    cur = next;
    n = survivors;
    --replans_left;
    if (survivors <= 0) {  // nothing left: close the fork
      ok = join_side_streams() && ok;
      break;
    }
    if (ceil_div(n, 64) <= kSmallCB) {  // the one-workgroup sweep runs on the caller's stream: it follows BOTH side streams
      ok = join_side_streams() && ok;   // (it appends to keep_out / num_keep, which the last resolver writes after its flag)
    }
  }
  if (!ok) return set_error((int)hipErrorUnknown, "tvmi_nms: stream fork / join failed");
  if (claimed) {
    // Did a hand-off poll give up?  The call may synchronise (its caller reads num_keep next): read the word through the
    // pinned host slot; if it is raised the result is unspecified — re-run the whole call in the stream-event form.
    nms_report_failed<<<dim3(1), dim3(1), 0, stream>>>(ws.failed, ss.host_count + 1);
    e = hipStreamSynchronize(stream);
    if (e != hipSuccess) return set_error((int)e, "tvmi_nms: synchronising for the hand-off status");
    if (static_cast<volatile int*>(ss.host_count)[1] != 0) {
      g_handoff_broken.store(true, std::memory_order_relaxed);
      release_handoff(claim_dev, stream);
      claimed = false;
      return launch<T>(dets, order, seg, n_call, thr, workspace, keep_out, num_keep, stream, may_sync, /*allow_handoff=*/false);
    }
  }
  TVMI_RETURN_LAUNCH_STATUS("tvmi_nms");
}


// ---------------------------------------------------------------------------------------
// Segment-major batched NMS.  With segment ids (classes / levels / images) the suppression
// matrix is block diagonal once the boxes are laid out segment by segment (score order kept
// inside a segment): only tiles whose two 64-box blocks share a segment are evaluated, and
// every segment is swept by its own workgroup, concurrently — instead of N x N pair tests
// and ONE serial sweep over all N boxes in global score order.  The caller provides the
// stable partition of the score order by segment (`perm`, `keys`: a second stable sort);
// the result is emitted in global score order, identical to the unsegmented formulation.
//   position p (segment-major)  --perm-->  g (rank in global score order)  --order-->  box
// Blocks are NOT padded per segment: a 64-box block may hold the tail of one segment and the
// heads of others; cross-segment mask bits are zero, so a workgroup sweeping the block range
// of "its" segments may resolve foreign rows of its first block incorrectly — it never writes
// them (each keep bit is published by exactly one workgroup, with an atomicOr).
constexpr int kSegMaxBlocks = 128;  // blocks one sweep workgroup handles (8,192 boxes per segment); its pull over
                                   // earlier super-blocks is O(blocks^2) tile reads by ONE workgroup, so larger
                                   // segments are sent back to the global-order pipeline (colreduce + resolve)

// The same bound found by a whole wave (all 64 lanes call it with the same arguments): 64 probes per round trip instead of
// one — 3 dependent loads for 100k keys instead of 17.  (Round 4's lane-0 binary searches, two per row block, were most of
// nms_mask_tiles_seg's 70 us at 100k boxes x 80 classes: every workgroup began with 34 serial global round trips.)
__device__ __forceinline__ int upper_bound_key_wave(const int64_t* __restrict__ keys, int n, int64_t v) {
  const int lane = threadIdx.x & 63;
  int lo = 0, hi = n;   // invariant: every p < lo has keys[p] <= v, every p >= hi has keys[p] > v
  while (hi - lo > 64) {
    const int step = (hi - lo + 63) >> 6;
    const int p = lo + lane * step;
    const bool le = p < hi && keys[p] <= v;
    const int cnt = __builtin_popcountll(__ballot(le));   // keys ascending: the lanes with le form a prefix
    const int nlo = cnt > 0 ? lo + (cnt - 1) * step + 1 : lo;   // probe cnt - 1 holds <= v
    const int nhi = min(hi, lo + cnt * step);                   // probe cnt (if any) holds > v
    lo = __builtin_amdgcn_readfirstlane(nlo);
    hi = __builtin_amdgcn_readfirstlane(nhi);
  }
  const int p = lo + lane;
  const bool le = p < hi && keys[p] <= v;
  return lo + __builtin_popcountll(__ballot(le));
}

__device__ __forceinline__ int upper_bound_key(const int64_t* __restrict__ keys, int n, int64_t v) {
  int lo = 0, hi = n;  // first p with keys[p] > v
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (keys[mid] <= v) lo = mid + 1; else hi = mid;
  }
  return lo;
}

// `n_dev` (all kernels of the segment-major and the small-segment path): the boxes that take part, read on the device.
// The caller's lists are laid out for the CAPACITY n, and the boxes that take part are the first *n_dev of BOTH the
// score order and the segment-major order (masked-out candidates carry score -inf and the largest key, so both sorts
// put them last): every kernel clamps its n and the launch grids, sized for the capacity, retire their surplus blocks.
__device__ __forceinline__ int live_boxes(int n, const int64_t* __restrict__ n_dev) {
  if (n_dev == nullptr) return n;
  const int64_t v = *n_dev;
  return (int)min(max(v, (int64_t)0), (int64_t)n);
}

__global__ __launch_bounds__(256) void nms_seg_layout(const int64_t* __restrict__ order, const int64_t* __restrict__ perm,
                                                      int n, const int64_t* __restrict__ n_dev, int64_t* __restrict__ oidx,
                                                      int* __restrict__ invperm) {
  n = live_boxes(n, n_dev);
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= n) return;
  const int64_t g = perm[p];
  oidx[p] = order[g];
  invperm[g] = p;
}

// one workgroup per row block; its waves walk the column blocks that share a segment with it
template <typename T>
__global__ __launch_bounds__(kMaskWaves * kWave) void nms_mask_tiles_seg(const T* __restrict__ dets,
                                                                         const int64_t* __restrict__ oidx,
                                                                         const int64_t* __restrict__ keys, int n,
                                                                         const int64_t* __restrict__ n_dev, int CB,
                                                                         double thr, ThrBand band, u64* __restrict__ mask) {
  __shared__ __attribute__((aligned(16))) T s_row[5][64];
  __shared__ long long s_key[64];
  __shared__ int s_cbmax;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int rb = blockIdx.x;
  const int row0 = rb * 64;
  n = live_boxes(n, n_dev);
  if (row0 >= n) return;
  if (threadIdx.x < 64) {
    const int r = row0 + lane;
    T x1 = 0, y1 = 0, x2 = 0, y2 = 0;
    long long k = 0;
    if (r < n) {
      const Box<T> b = load_box<T>(dets, oidx[r]);
      x1 = b.x1;
      y1 = b.y1;
      x2 = b.x2;
      y2 = b.y2;
      k = keys[r];
    }
    s_row[0][lane] = x1;
    s_row[1][lane] = y1;
    s_row[2][lane] = x2;
    s_row[3][lane] = y2;
    s_row[4][lane] = (x2 - x1) * (y2 - y1);
    s_key[lane] = k;
    {
      // column blocks that share a segment with this row block end with the segment of its last row; a
      // segment beyond the sweep limit is going to be redone by the global-order pipeline: skip its tiles
      const int64_t klast = keys[min(n - 1, row0 + 63)];
      const int cbm = (upper_bound_key_wave(keys, n, klast) - 1) >> 6;
      const int first_blk = upper_bound_key_wave(keys, n, klast - 1) >> 6;  // ids are integers: first p with key >= klast
      if (lane == 0) s_cbmax = (cbm - first_blk + 1 > kSegMaxBlocks) ? rb - 1 : cbm;
    }
  }
  __syncthreads();
  const int cbmax = s_cbmax;
  for (int cb = rb + wave; cb <= cbmax; cb += kMaskWaves) {
    const int j = cb * 64 + lane;
    T jx1 = 0, jy1 = 0, jx2 = 0, jy2 = 0;
    long long jkey = 0;
    const bool jvalid = j < n;
    if (jvalid) {
      const Box<T> b = load_box<T>(dets, oidx[j]);
      jx1 = b.x1;
      jy1 = b.y1;
      jx2 = b.x2;
      jy2 = b.y2;
      jkey = keys[j];
    }
    const T jarea = (jx2 - jx1) * (jy2 - jy1);
    const u64 mine = suppression_tile<T, 64>(&s_row[0][0], s_key, min(64, n - row0), jx1, jy1, jx2, jy2, jarea, jkey, jvalid,
                                             cb == rb, thr, band);
    mask[((size_t)rb * CB + (cb - rb)) * 64 + lane] = mine;
  }
}

// one workgroup per 64-box block in which at least one segment starts: it sweeps the blocks
// from its own to the end of the last segment that starts in it
__global__ __launch_bounds__(kSuper * kWave) void nms_sweep_seg(const u64* __restrict__ mask,
                                                                const int64_t* __restrict__ keys, int n,
                                                                const int64_t* __restrict__ n_dev, int CB,
                                                                u64* __restrict__ keepbits, int* __restrict__ err) {
  __shared__ u64 s_keepbits[kSegMaxBlocks];
  __shared__ int s_info[3];  // first start position, end position, number of blocks
  const int lane = threadIdx.x & 63;
  const int c_loc = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int B0 = blockIdx.x;
  n = live_boxes(n, n_dev);
  if (B0 * 64 >= n) return;
  if (threadIdx.x < 64) {
    const int p = B0 * 64 + lane;
    const bool start = p < n && (p == 0 || keys[p] != keys[p - 1]);
    const u64 starts = __ballot(start);
    if (starts == 0ull) {   // (wave-uniform)
      if (lane == 0) s_info[2] = 0;
    } else {
      const int end = upper_bound_key_wave(keys, n, keys[min(n - 1, B0 * 64 + 63)]);
      if (lane == 0) {
        s_info[0] = B0 * 64 + __builtin_ctzll(starts);
        s_info[1] = end;
        s_info[2] = ((end - 1) >> 6) + 1 - B0;
      }
    }
  }
  __syncthreads();
  const int nb = s_info[2];
  if (nb == 0) return;
  if (nb > kSegMaxBlocks) {
    if (threadIdx.x == 0) *err = 1;
    return;
  }
  const int first = s_info[0], end = s_info[1];
  for (int b0 = 0; b0 < nb; b0 += kSuper) {
    const int b1 = min(nb, b0 + kSuper);
    const int lb = b0 + c_loc;  // local block of this wave
    const int cb = B0 + lb;
    const bool have = lb < b1;
    u64 diag = 0ull, above[kSuper - 1];
    if (have) diag = mask[((size_t)cb * CB) * 64 + lane];
#pragma unroll
    for (int q = 0; q < kSuper - 1; ++q) {
      above[q] = 0ull;
      if (have && q < c_loc) above[q] = mask[((size_t)(B0 + b0 + q) * CB + (cb - (B0 + b0 + q))) * 64 + lane];
    }
    u64 acc = 0ull;
    if (have) {
      for (int rl = 0; rl < b0; ++rl) {
        const u64 w = mask[((size_t)(B0 + rl) * CB + (cb - (B0 + rl))) * 64 + lane];
        if ((s_keepbits[rl] >> lane) & 1ull) acc |= w;
      }
    }
    u64 rem = b0 > 0 ? wave_or64(acc) : 0ull;
    const int rows_here = have ? min(64, n - cb * 64) : 0;
    const u64 valid = rows_here >= 64 ? ~0ull : ((1ull << rows_here) - 1ull);
#pragma unroll
    for (int step = 0; step < kSuper; ++step) {
      if (step == c_loc && have) {
        u64 r = uniform64(rem);
        u64 active = uniform64(__ballot(diag != 0ull)) & ~r & valid;
        while (active) {
          const int k = __builtin_ctzll(active);
          r |= readlane64(diag, k);
          active &= ~(r | (1ull << k));
        }
        if (lane == 0) s_keepbits[lb] = ~r & valid;
      }
      __syncthreads();
      if (step < kSuper - 1 && have && step < c_loc) {
        const u64 kb = s_keepbits[b0 + step];
        const u64 contrib = ((kb >> lane) & 1ull) ? above[step] : 0ull;
        rem |= wave_or64(contrib);
      }
    }
  }
  // publish the keep bits of the rows this workgroup owns: positions [first, end)
  for (int lb = threadIdx.x; lb < nb; lb += kSuper * kWave) {
    const int p0 = (B0 + lb) * 64;
    u64 own = ~0ull;
    if (first > p0) own &= ~((1ull << (first - p0)) - 1ull);         // first - p0 < 64: only in the first block
    if (end < p0 + 64) own &= (1ull << (end - p0)) - 1ull;           // end - p0 >= 1
    const u64 bits = s_keepbits[lb] & own;
    if (bits) atomicOr(&keepbits[B0 + lb], bits);
  }
}

// emission in GLOBAL score order: g -> position -> keep bit, two-pass compaction
__device__ __forceinline__ bool seg_kept(const u64* __restrict__ keepbits, const int* __restrict__ invperm, int g, int n) {
  if (g >= n) return false;
  const int p = invperm[g];
  return (keepbits[p >> 6] >> (p & 63)) & 1ull;
}

__global__ __launch_bounds__(1024) void nms_seg_count(const u64* __restrict__ keepbits, const int* __restrict__ invperm, int n,
                                                      const int64_t* __restrict__ n_dev, int* __restrict__ counts) {
  __shared__ int s_w[16];
  n = live_boxes(n, n_dev);
  const int g = blockIdx.x * 1024 + threadIdx.x;
  const u64 bal = __ballot(seg_kept(keepbits, invperm, g, n));
  if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = __popcll(bal);
  __syncthreads();
  if (threadIdx.x == 0) {
    int t = 0;
    for (int w = 0; w < 16; ++w) t += s_w[w];
    counts[blockIdx.x] = t;
  }
}

__global__ __launch_bounds__(1024) void nms_seg_emit(const u64* __restrict__ keepbits, const int* __restrict__ invperm,
                                                     const int64_t* __restrict__ order, const int* __restrict__ counts, int n,
                                                     const int64_t* __restrict__ n_dev, const int* __restrict__ err,
                                                     const int* __restrict__ ext_err, int64_t* __restrict__ keep_out,
                                                     int64_t* __restrict__ num_keep) {
  __shared__ int s_w[16];
  __shared__ int s_part[16];
  n = live_boxes(n, n_dev);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // exclusive prefix of the chunk counts before this chunk
  int part = 0;
  for (int c = threadIdx.x; c < (int)blockIdx.x; c += 1024) part += counts[c];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o);
  if (lane == 0) s_part[wave] = part;
  const int g = blockIdx.x * 1024 + threadIdx.x;
  const bool kept = seg_kept(keepbits, invperm, g, n);
  const u64 bal = __ballot(kept);
  if (lane == 0) s_w[wave] = __popcll(bal);
  __syncthreads();
  int base = 0;
  for (int w = 0; w < 16; ++w) base += s_part[w];
  int before = 0;
  for (int w = 0; w < wave; ++w) before += s_w[w];
  if (kept) keep_out[base + before + __popcll(bal & ((1ull << lane) - 1ull))] = order[g];
  if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) {
    int tot = base;
    for (int w = 0; w < 16; ++w) tot += s_w[w];
    *num_keep = (*err || (ext_err && *ext_err)) ? -1 : tot;   // ext_err: the partition saw an id outside its promised range
  }
}


// ---------------------------------------------------------------------------------------
// Small batched NMS (n <= 4096 boxes, ids in [0, S), every segment <= 1024 boxes: the RPN /
// box-head / per-image sizes of a detector step) in TWO launches and no second sort:
//  A. nms_small_seg_tiles: workgroup (s, y) collects segment s from the score-ordered list
//     (stable; every workgroup of the segment repeats this cheap scan), and each of its waves
//     builds one upper-triangular 64x64 suppression tile of the segment -> tiles[s][t][64]
//     (136 tiles per segment at most, spread over 9 workgroups: the pair tests use the chip,
//     not one CU per segment); workgroup (s, 0) also stores the segment's global ranks.
//  B. nms_small_seg_sweep: workgroup s sweeps its segment (one super-block: no pull phase),
//     publishes one flag per box, and the LAST workgroup to finish (device-scope fence +
//     ticket) compacts the flags in global score order into keep_out / num_keep.
// Versus mask kernel + global sweep: no N x N mask, S resolve chains run concurrently.
// kSmallSegBoxes / kSmallSegBlocks / kSmallSegTiles: nms_device.h

struct SmallSegWorkspace {
  int* sync_words;       // [0] finished workgroups, [1] error flag
  int* gcnt;             // [S] boxes per segment
  int* glist;            // [S][1024] global rank of the i-th box of the segment
  int* flags;            // [n] keep flag per global rank
  u64* tiles;            // [S][136][64]
};
inline size_t small_seg_workspace_layout(int64_t n, int64_t S, char* base, SmallSegWorkspace* w) {
  size_t off = 0;
  auto take = [&](size_t bytes) {
    char* ptr = base ? base + off : nullptr;
    off += (bytes + 255) & ~(size_t)255;
    return ptr;
  };
  char* sw = take(2 * sizeof(int));
  char* gc = take((size_t)S * sizeof(int));
  char* gl = take((size_t)S * kSmallSegBoxes * sizeof(int));
  char* fl = take((size_t)n * sizeof(int));
  char* tl = take((size_t)S * kSmallSegTiles * 64 * sizeof(u64));
  if (w) {
    w->sync_words = reinterpret_cast<int*>(sw);
    w->gcnt = reinterpret_cast<int*>(gc);
    w->glist = reinterpret_cast<int*>(gl);
    w->flags = reinterpret_cast<int*>(fl);
    w->tiles = reinterpret_cast<u64*>(tl);
  }
  return off;
}

// Step A as TWO launches whose workgroups fit next to a chip-filling neighbour (round 5; the one-launch form that staged a whole
// segment in 21 KB of LDS per 1024-lane workgroup — it waited 100-195 us in the dispatcher under the RoIAlign forward, which owns
// 156 of the 160 KB of every CU — was deleted in round 6).  A1: nms_small_seg_collect records the segment's members (global ranks, in score order) and
// its size — 64 bytes of LDS.  B: nms_small_seg_tiles4, 256-lane workgroups of four tiles: a wave fetches its 64 column boxes
// straight into registers (rank -> order -> box), the at most three row blocks of a workgroup's four consecutive tiles are
// staged once in 3.75 KB of LDS.  Same pair arithmetic (suppression_tile), same tiles in ws.tiles, the sweep is unchanged.
template <typename T>
__global__ __launch_bounds__(1024) void nms_small_seg_collect(const int64_t* __restrict__ order, const int64_t* __restrict__ seg, int n,
                                                              const int64_t* __restrict__ n_dev, int S, SmallSegWorkspace ws) {
  __shared__ int s_wcnt[16];
  n = live_boxes(n, n_dev);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(tid >> 6));
  const int me = blockIdx.x;
  int* glist = ws.glist + (size_t)me * kSmallSegBoxes;
  int cnt = 0;
  bool bad = false;
  for (int base = 0; base < n; base += 1024) {
    const int g = base + tid;
    bool mine = false;
    if (g < n) {
      const int64_t sg = seg[order[g]];
      mine = sg == me;
      bad |= sg < 0 || sg >= S;
    }
    const u64 bal = __ballot(mine);
    if (lane == 0) s_wcnt[wave] = __popcll(bal);
    __syncthreads();
    int before = 0, total = 0;
    for (int w = 0; w < 16; ++w) {
      const int c = s_wcnt[w];
      before += w < wave ? c : 0;
      total += c;
    }
    const int pos = cnt + before + __popcll(bal & ((1ull << lane) - 1ull));
    if (mine && pos < kSmallSegBoxes) glist[pos] = g;
    cnt += total;
    __syncthreads();
  }
  const bool any_bad = me == 0 ? __syncthreads_or(bad) : false;
  if (tid == 0) {
    if (cnt > kSmallSegBoxes || any_bad) ws.sync_words[1] = 1;
    ws.gcnt[me] = min(cnt, kSmallSegBoxes);
  }
}

constexpr int kTiles4 = 4;   // tiles (= waves) per workgroup of nms_small_seg_tiles4

template <typename T>
__global__ __launch_bounds__(kTiles4 * 64) void nms_small_seg_tiles4(const T* __restrict__ dets, const int64_t* __restrict__ order,
                                                                     double thr, ThrBand band, SmallSegWorkspace ws) {
  __shared__ __attribute__((aligned(16))) T s_row[3][5][64];   // the row blocks of this workgroup's tiles, component-major
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int me = blockIdx.x;
  const int cnt = ws.gcnt[me];
  const int nb = (cnt + 63) >> 6, ntiles = nb * (nb + 1) / 2;
  const int t0 = blockIdx.y * kTiles4;
  if (t0 >= ntiles) return;   // the whole workgroup: the grid is sized for the longest possible segment
  const int* glist = ws.glist + (size_t)me * kSmallSegBoxes;
  auto row_of = [&](int t, int& rb, int& cb) {   // tile index -> (row block, column block) of the upper triangle, row-major
    rb = 0;
    int rem = t;
    while (rem >= nb - rb) {
      rem -= nb - rb;
      ++rb;
    }
    cb = rb + rem;
  };
  int rb0, cb0;
  row_of(t0, rb0, cb0);
  // four consecutive tiles span at most three row blocks (a row of the triangle has at least one tile, the last rows 3, 2, 1)
  if (wave < 3 && rb0 + wave < nb) {
    const int p = (rb0 + wave) * 64 + lane;
    T x1 = 0, y1 = 0, x2 = 0, y2 = 0;
    if (p < cnt) {
      const Box<T> b = load_box<T>(dets, order[glist[p]]);
      x1 = b.x1;
      y1 = b.y1;
      x2 = b.x2;
      y2 = b.y2;
    }
    s_row[wave][0][lane] = x1;
    s_row[wave][1][lane] = y1;
    s_row[wave][2][lane] = x2;
    s_row[wave][3][lane] = y2;
    s_row[wave][4][lane] = (x2 - x1) * (y2 - y1);
  }
  __syncthreads();
  const int t = t0 + wave;
  if (t >= ntiles) return;
  int rb, cb;
  row_of(t, rb, cb);
  const int j = cb * 64 + lane;
  const bool jvalid = j < cnt;
  T jx1 = 0, jy1 = 0, jx2 = 0, jy2 = 0;
  {
    const Box<T> b = load_box<T>(dets, order[glist[jvalid ? j : 0]]);
    jx1 = b.x1;
    jy1 = b.y1;
    jx2 = b.x2;
    jy2 = b.y2;
  }
  const T jarea = (jx2 - jx1) * (jy2 - jy1);
  const u64 mine = suppression_tile<T, 64>(&s_row[rb - rb0][0][0], nullptr, min(64, cnt - rb * 64), jx1, jy1, jx2, jy2, jarea, 0,
                                           jvalid, cb == rb, thr, band);
  ws.tiles[((size_t)me * kSmallSegTiles + t) * 64 + lane] = mine;
}

__global__ __launch_bounds__(kSuper * kWave) void nms_small_seg_sweep(const int64_t* __restrict__ order, int n,
                                                                      const int64_t* __restrict__ n_dev,
                                                                      SmallSegWorkspace ws, int64_t* __restrict__ keep_out,
                                                                      int64_t* __restrict__ num_keep) {
  static_assert(kSmallSegBlocks == kSuper, "one super-block per segment");
  n = live_boxes(n, n_dev);
  __shared__ u64 s_keepbits[kSmallSegBlocks];
  __shared__ int s_wcnt[16];
  __shared__ int s_flag;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(tid >> 6));
  const int me = blockIdx.x;
  const int cnt = ws.gcnt[me];
  const int nb = (cnt + 63) >> 6;
  const u64* tiles = ws.tiles + (size_t)me * kSmallSegTiles * 64;
  {
    auto tile_index = [nb](int rb, int cb) { return rb * nb - (rb * (rb - 1)) / 2 + (cb - rb); };
    const int c_loc = wave;
    const bool have = c_loc < nb;
    u64 diag = 0ull, above[kSuper - 1];
    if (have) diag = tiles[tile_index(c_loc, c_loc) * 64 + lane];
#pragma unroll
    for (int q = 0; q < kSuper - 1; ++q) {
      above[q] = 0ull;
      if (have && q < c_loc) above[q] = tiles[tile_index(q, c_loc) * 64 + lane];
    }
    u64 rem = 0ull;
    const int rows_here = have ? min(64, cnt - c_loc * 64) : 0;
    const u64 valid = rows_here >= 64 ? ~0ull : ((1ull << rows_here) - 1ull);
#pragma unroll
    for (int step = 0; step < kSuper; ++step) {
      if (step == c_loc && have) {
        u64 r = uniform64(rem);
        u64 active = uniform64(__ballot(diag != 0ull)) & ~r & valid;
        while (active) {
          const int k = __builtin_ctzll(active);
          r |= readlane64(diag, k);
          active &= ~(r | (1ull << k));
        }
        if (lane == 0) s_keepbits[c_loc] = ~r & valid;
      }
      __syncthreads();
      if (step < kSuper - 1 && have && step < c_loc) {
        const u64 kb = s_keepbits[step];
        const u64 contrib = ((kb >> lane) & 1ull) ? above[step] : 0ull;
        rem |= wave_or64(contrib);
      }
    }
  }
  // one flag per box of my segment, at its global rank
  const int* glist = ws.glist + (size_t)me * kSmallSegBoxes;
  // Published with agent-scope (write-through) stores and read back with agent-scope loads instead of release /
  // acquire fences: on the 8-XCD part a device-scope fence is a write-back / invalidate of the whole L2 slice,
  // which costs more than this kernel's actual work.
  for (int i = tid; i < cnt; i += kSuper * kWave)
    __hip_atomic_store(&ws.flags[glist[i]], (int)((s_keepbits[i >> 6] >> (i & 63)) & 1ull), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
  // last workgroup out compacts the flags in global score order
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // my flag stores are acknowledged
  __syncthreads();
  if (tid == 0) s_flag = atomicAdd(&ws.sync_words[0], 1) == (int)gridDim.x - 1;
  __syncthreads();
  if (!s_flag) return;
  int run = 0;
  for (int base = 0; base < n; base += 1024) {
    const int g = base + tid;
    const bool kept = g < n && __hip_atomic_load(&ws.flags[g], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
    const u64 bal = __ballot(kept);
    if (lane == 0) s_wcnt[wave] = __popcll(bal);
    __syncthreads();
    int before = 0, total = 0;
    for (int w = 0; w < 16; ++w) {
      const int c = s_wcnt[w];
      before += w < wave ? c : 0;
      total += c;
    }
    if (kept) keep_out[run + before + __popcll(bal & ((1ull << lane) - 1ull))] = order[g];
    run += total;
    __syncthreads();
  }
  if (tid == 0) *num_keep = *(volatile int*)&ws.sync_words[1] ? -1 : run;
}


// ---------------------------------------------------------------------------------------
// Score order for small inputs (n <= 4096, float32): order = aten::sort(scores, stable=True, descending=True)
// indices — NaN first, ties by ascending index, -0 == +0 — from 64-bit keys (order-preserving score bits << 32 | index): the
// unique index in the low word makes the rank of a key exact.  At these sizes torch's path is a radix-sort kernel plus an index
// arange plus two copies (~40 us of launches for 4000 scores).
// kSortMax, score_key: nms_device.h

// RANK COUNTING (round 5; the one-workgroup bitonic network over 32 KB of LDS it replaced was deleted in round 6): the keys are
// unique, so the position of element i is the number of keys below its own — n^2 compares (16 M at n = 4096) spread over n / 64
// workgroups instead of 78 compare-exchange passes of ONE workgroup.  Lane = element, the four waves of a workgroup count over one quarter of the keys
// each, 64 keys at a time through a 512-byte wave-private LDS slice (built once per 64 compares, read as broadcasts).  3 KB of
// LDS and 256 lanes per workgroup: it starts on any CU, also next to a launch that owns the CU's LDS (the bitonic kernel's 32 KB
// do not; HISTORY.md 6.0), and it is faster on an idle chip as well.  Identical output (the rank of a unique key is exact).
__global__ __launch_bounds__(256) void sort_scores_desc_rank(const float* __restrict__ scores, int n, int64_t* __restrict__ order) {
  __shared__ u64 s_keys[4][64];
  __shared__ int s_cnt[4][64];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int i = blockIdx.x * 64 + lane;
  const u64 ki = i < n ? score_key(scores[i], i) : ~0ull;
  const int per = ((n + 255) >> 8) << 6;            // keys per quarter: a multiple of 64
  const int j_begin = wave * per, j_end = min(n, j_begin + per);
  int cnt = 0;
  for (int j0 = j_begin; j0 < j_end; j0 += 64) {
    const int j = j0 + lane;
    s_keys[wave][lane] = j < n ? score_key(scores[j], j) : ~0ull;   // padding: above every real key, never counted
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int t = 0; t < 64; ++t) cnt += s_keys[wave][t] < ki ? 1 : 0;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
  s_cnt[wave][lane] = cnt;
  __syncthreads();
  if (wave == 0 && i < n) order[s_cnt[0][lane] + s_cnt[1][lane] + s_cnt[2][lane] + s_cnt[3][lane]] = i;
}

// ---------------------------------------------------------------------------------------
// Detector-step batched NMS (+ the padded top-k payload) as ONE launch (round 6).
// The chain it replaces for n <= 4096 boxes in <= 64 segments of <= 1024 boxes: memset, score sort -> collect -> tiles4 -> sweep
// (-> pack_detections): 5 kernels + a fill, 49 us of kernel time + 6 launch gaps.  Here the same steps are PHASES of one grid of
// 256-lane workgroups handed over through memory words (agent-scope stores / loads, the hand-off idiom of nms_push above):
// tile workgroups (segment s = blockIdx.x, j = blockIdx.y < G) and counting-only workgroups (j >= G).
//   A1  every tile workgroup: members of segment s as keys (score word << 32 | box index) in LDS, ascending box index — all G
//       workgroups of a segment build the same list (redundant work on an idle chip instead of a publish / poll round trip).
//   A3  workgroups j < 16: score order of the segment by RANK COUNTING — 64 members per workgroup (lane = member), its four waves
//       count over a quarter of the segment's keys each; the member's rank r is its place: box and index go to sbox[s][r] / sidx[s][r].
//   A2  counting-only workgroups: partial GLOBAL score ranks (over all n boxes, for the output order) — 64 boxes x a quarter of the
//       keys per workgroup, combined in LDS: part[q][box], q < 4.  16M key pairs at n = 4000: every SIMD of the chip counts.
//   B   every tile workgroup: waits for the publishers' flags; upper-triangular 64x64 suppression tiles t = 4j + wave (+ 4G ...) from
//       the sorted boxes — suppression_tile, the bit-exact reference predicate — to tiles[s][t]; raises its flag.
//   C   workgroup (s, 0): waits for the G tile flags; greedy sweep of the <= 16 blocks with 4 waves (wave w owns column blocks
//       w, w + 4, ...: the tiles above them in registers, a block's pending removals OR-reduced once, when its turn comes); waits
//       for the counting flags; slot[global rank] = kept ? (index + 1) | image << 13 : 0 for ITS boxes; takes a ticket.
//   D   the LAST sweeper compacts slot[] in rank order into keep_out / num_keep — the reference's output order: descending score
//       over ALL segments — writes the per-image padded top-`max_dets` rows (what tvmi_pack_detections_payload writes) in the same
//       pass, and puts the hand-over words back to zero.
// Counters shared by many producers are avoided (one flag word per producer): same-address agent-scope atomics retire ~150 ns
// apart.  Counting compares 32-bit score words with saturating subtracts (count_below_group): the condition-code round trip of a
// v_cmp -> v_addc chain cost ~45 cycles per key with one wave per SIMD.  Phase times on the MI355X at 4 x 1000 boxes (s_memrealtime,
// -DTVMI_STEP_TIMING): A1 3.9, A3 5.3, wait 2, B 5.6, tile loads 1, C 4.6, slots 4.3, ticket 1.5, D 11 = 39 us in ONE launch,
// against 49 us of kernels + 4.5 us fill + 6 gaps for the chain.
// Forward progress: workgroup (s, 0) and the publishers have the lowest workgroup ids of a launch capped at 512 workgroups (a
// fraction of the chip's resident capacity), so a poller never keeps its producer off the chip; every poll is bounded and a
// timeout surfaces as num_keep = -1 (as do an over-long segment and an id outside [0, S)).
// constants, workspace, hand-over helpers and the workgroup body: nms_step_device.h
// The hand-over words of the step kernel live in a block the LIBRARY owns, one per (device, stream): zero when it is created,
// and put back to zero by the last workgroup of every launch — launches on one stream run one after the other, so the next one
// finds it clean and the call needs no memset (a fill kernel of its own: 4.5 us on the MI355X, more than any phase of the step).
// A stream that is being captured into a graph gets the caller's workspace words + a zero-fill kernel node instead (no allocation
// and no event inside a capture).  A launch that gave up on a poll leaves the error word set: every later call on that stream
// reports num = -1.
// ONE step launch at a time per device: the workgroups of a launch wait for each other (low workgroup numbers for high ones),
// which is safe while the whole launch is resident — up to 512 workgroups against the 1,024 the chip holds of this kernel — and is
// NOT when two such launches on two streams share the chip: each can hold the slots the other's producers need until the bounded
// polls give up (seen as num = -1 on some of 8 streams in tests/test_gpu_parity.py once the process had other HIP streams).  So a
// launch on another stream than the device's previous step launch first waits for that stream (an event recorded there at the
// moment of the switch); launches that stay on one stream — the detector step — pay nothing.
struct StepSyncBlocks {
  std::mutex mu;
  std::vector<std::pair<std::pair<int, hipStream_t>, int*>> blocks;
  struct Last {
    hipStream_t stream = nullptr;
    bool any = false;
    hipEvent_t ev = nullptr;
  };
  Last last[64];
  int* get(hipStream_t s) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> lock(mu);
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &st) != hipSuccess || st != hipStreamCaptureStatusNone) {
      (void)hipGetLastError();
      return nullptr;
    }
    if (dev >= 0 && dev < 64) {
      Last& l = last[dev];
      if (l.any && l.stream != s) {   // a switch of streams: order this launch behind everything queued on the previous one
        hipStreamCaptureStatus pst = hipStreamCaptureStatusNone;   // (never record into a capture another thread has open there)
        const bool prev_idle = hipStreamIsCapturing(l.stream, &pst) == hipSuccess && pst == hipStreamCaptureStatusNone;
        if (prev_idle && !l.ev && hipEventCreateWithFlags(&l.ev, hipEventDisableTiming) != hipSuccess) l.ev = nullptr;
        if (prev_idle && l.ev && hipEventRecord(l.ev, l.stream) == hipSuccess) (void)hipStreamWaitEvent(s, l.ev, 0);
        (void)hipGetLastError();   // (a previous stream that no longer exists: nothing of it can still be running)
      }
      l.stream = s;
      l.any = true;
    }
    for (auto& b : blocks)
      if (b.first.first == dev && b.first.second == s) return b.second;
    int* p = nullptr;
    // zeroed ON THE STREAM of the launches that will use it: hipMemset (null stream) may return before the fill has run, torch's
    // streams do not wait for the null stream, and a fill that lands in the middle of the first launch wipes its flags — the polls
    // give up and the block stays poisoned (num = -1 on 2 of 8 new streams in round 6 once the process had other HIP streams)
    if (hipMalloc(reinterpret_cast<void**>(&p), kStepSyncWords * sizeof(int)) != hipSuccess ||
        hipMemsetAsync(p, 0, kStepSyncWords * sizeof(int), s) != hipSuccess) {
      (void)hipGetLastError();
      return nullptr;
    }
    blocks.push_back({{dev, s}, p});
    return p;
  }
};
inline StepSyncBlocks& step_sync_blocks() {
  static StepSyncBlocks* b = new StepSyncBlocks();   // never destroyed: the runtime may be gone at exit
  return *b;
}

__global__ __launch_bounds__(kStepThreads) void nms_step_fused(const float* __restrict__ dets, const float* __restrict__ scores,
                                                               const int64_t* __restrict__ seg, int n, int S, int G, double thr,
                                                               ThrBand band, StepWorkspace ws, int64_t* __restrict__ keep_out,
                                                               int64_t* __restrict__ num_keep, StepPack pk) {
  __shared__ StepShared sh;
  nms_step_body(sh, (int)blockIdx.x, (int)blockIdx.y, (int)gridDim.y, dets, scores, seg, n, S, G, thr, band, ws, keep_out, num_keep, pk);
}

struct SegWorkspace {
  u64* mask;
  u64* keepbits;
  int64_t* oidx;
  int* invperm;
  int* counts;
  int* err;
};
inline size_t seg_workspace_layout(int64_t n, char* base, SegWorkspace* w) {
  const size_t CB = (size_t)ceil_div(n, 64), NC = (size_t)ceil_div(n, 1024);
  size_t off = 0;
  auto take = [&](size_t bytes) {
    char* ptr = base ? base + off : nullptr;
    off += (bytes + 255) & ~(size_t)255;
    return ptr;
  };
  char* m = take(CB * std::min<size_t>(CB, (size_t)kSegMaxBlocks) * 64 * sizeof(u64));  // banded: tile (rb, cb) at row rb, slot cb - rb
  char* kb = take((CB + 1) * sizeof(u64));  // keep bits + (err, pad) right behind them: one memset
  char* oi = take((size_t)n * sizeof(int64_t));
  char* ip = take((size_t)n * sizeof(int));
  char* ct = take(NC * sizeof(int));
  if (w) {
    w->mask = reinterpret_cast<u64*>(m);
    w->keepbits = reinterpret_cast<u64*>(kb);
    w->err = reinterpret_cast<int*>(kb + CB * sizeof(u64));
    w->oidx = reinterpret_cast<int64_t*>(oi);
    w->invperm = reinterpret_cast<int*>(ip);
    w->counts = reinterpret_cast<int*>(ct);
  }
  return off;
}

template <typename T>
int launch_seg(const void* dets, const int64_t* order, const int64_t* keys, const int64_t* perm, int64_t n,
               const int64_t* n_dev, const int* ext_err, double thr, void* workspace, int64_t* keep_out, int64_t* num_keep,
               hipStream_t stream) {
  const int CB = (int)ceil_div(n, 64), NC = (int)ceil_div(n, 1024);
  // the mask is banded: a row block only meets column blocks of its own segments, at most kSegMaxBlocks ahead, so a row
  // of tiles is min(CB, kSegMaxBlocks) slots wide (n x 1 KB instead of n^2 / 8 bytes: 0.37 GB instead of 16 GB at 360k boxes)
  const int W = std::min(CB, kSegMaxBlocks);
  SegWorkspace w;
  seg_workspace_layout(n, static_cast<char*>(workspace), &w);
  hipError_t e = hipMemsetAsync(w.keepbits, 0, sizeof(u64) * ((size_t)CB + 1), stream);
  if (e != hipSuccess) return set_error((int)e, "tvmi_nms_segmented: memset");
  nms_seg_layout<<<dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, stream>>>(order, perm, (int)n, n_dev, w.oidx, w.invperm);
  nms_mask_tiles_seg<T><<<dim3((unsigned)CB), dim3(kMaskWaves * kWave), 0, stream>>>(static_cast<const T*>(dets), w.oidx, keys,
                                                                                  (int)n, n_dev, W, thr, thr_band(thr), w.mask);
  nms_sweep_seg<<<dim3((unsigned)CB), dim3(kSuper * kWave), 0, stream>>>(w.mask, keys, (int)n, n_dev, W, w.keepbits, w.err);
  nms_seg_count<<<dim3((unsigned)NC), dim3(1024), 0, stream>>>(w.keepbits, w.invperm, (int)n, n_dev, w.counts);
  nms_seg_emit<<<dim3((unsigned)NC), dim3(1024), 0, stream>>>(w.keepbits, w.invperm, order, w.counts, (int)n, n_dev, w.err,
                                                            ext_err, keep_out, num_keep);
  TVMI_RETURN_LAUNCH_STATUS("tvmi_nms_segmented");
}

}  // namespace

int set_nms_option(const char* name, int64_t value) {
  if (std::strcmp(name, "nms.replan_min_boxes") == 0) {
    g_replan_min_boxes.store(std::max<int64_t>(0, value), std::memory_order_relaxed);
    return 0;
  }
  if (std::strcmp(name, "nms.replan_divisor") == 0) {
    g_replan_divisor.store((int)std::max<int64_t>(2, std::min<int64_t>(value, 1024)), std::memory_order_relaxed);
    return 0;
  }
  if (std::strcmp(name, "nms.device_handoff") == 0) {
    g_device_handoff.store(value != 0, std::memory_order_relaxed);
    if (value != 0) g_handoff_broken.store(false, std::memory_order_relaxed);   // switching it on again re-arms it after a give-up
    return 0;
  }
  if (std::strcmp(name, "nms.handoff_lose_flag") == 0) {
    g_handoff_lose_flag.store(value != 0, std::memory_order_relaxed);
    return 0;
  }
  if (std::strcmp(name, "nms.mask_lds_bytes") == 0) {
    g_mask_lds_bytes.store((int)std::max<int64_t>(0, std::min<int64_t>(value, 60000)), std::memory_order_relaxed);
    return 0;
  }
  if (std::strcmp(name, "nms.replan_max") == 0) {
    g_replan_max.store((int)std::max<int64_t>(0, std::min<int64_t>(value, 16)), std::memory_order_relaxed);
    return 0;
  }
  if (std::strcmp(name, "nms.step_fused") == 0) {    // 1 (default): n <= 4096 / <= 64 segments as ONE launch (tvmi_nms_step)
    g_step_fused.store(value != 0, std::memory_order_relaxed);
    return 0;
  }
  return -1;
}

int get_nms_option(const char* name, int64_t* value) {
  if (std::strcmp(name, "nms.step_fused") == 0) {
    *value = g_step_fused.load(std::memory_order_relaxed) ? 1 : 0;
    return 0;
  }
  if (std::strcmp(name, "nms.replan_min_boxes") == 0) *value = (int64_t)g_replan_min_boxes.load(std::memory_order_relaxed);
  else if (std::strcmp(name, "nms.replan_divisor") == 0) *value = g_replan_divisor.load(std::memory_order_relaxed);
  else if (std::strcmp(name, "nms.device_handoff") == 0)
    *value = (g_device_handoff.load(std::memory_order_relaxed) && !g_handoff_broken.load(std::memory_order_relaxed)) ? 1 : 0;
  else if (std::strcmp(name, "nms.handoff_lose_flag") == 0) *value = g_handoff_lose_flag.load(std::memory_order_relaxed);
  else if (std::strcmp(name, "nms.mask_lds_bytes") == 0) *value = g_mask_lds_bytes.load(std::memory_order_relaxed);
  else if (std::strcmp(name, "nms.replan_max") == 0) *value = g_replan_max.load(std::memory_order_relaxed);
  else return -1;
  return 0;
}
}  // namespace tvmi

extern "C" size_t tvmi_nms_workspace_bytes(int64_t n) {
  if (n <= 0) return 0;
  const size_t CB = (size_t)tvmi::ceil_div(n, 64);
  if (CB <= (size_t)tvmi::kSmallCB) return CB * CB * 64 * sizeof(unsigned long long);
  return CB * CB * 64 * sizeof(unsigned long long) + tvmi::large_state_bytes((size_t)n);
}

namespace tvmi {
namespace {
int nms_entry(const void* dets, const int64_t* order, const int64_t* seg, int64_t n, double iou_threshold, tvmi_dtype dt,
              void* workspace, size_t workspace_bytes, int64_t* keep_out, int64_t* num_keep_out, void* stream, bool may_sync) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  TVMI_CHECK_ARG(n >= 0, "nms: negative box count");
  TVMI_CHECK_ARG(num_keep_out != nullptr, "nms: num_keep_out is null");
  if (n == 0) {
    hipError_t e = hipMemsetAsync(num_keep_out, 0, sizeof(int64_t), s);
    return e == hipSuccess ? 0 : set_error((int)e, "tvmi_nms: memset");
  }
  TVMI_CHECK_ARG(dets && order && keep_out && workspace, "nms: null pointer");
  TVMI_CHECK_ARG(n <= 1200000, "nms: more than 1.2M boxes is not supported by the bitmask path");
  TVMI_CHECK_ARG(workspace_bytes >= tvmi_nms_workspace_bytes(n), "nms: workspace too small");
  TVMI_CHECK_ARG(dt == TVMI_F32 || dt == TVMI_F64, "nms: dets must be float32 or float64");
  if (dt == TVMI_F32) return launch<float>(dets, order, seg, n, iou_threshold, workspace, keep_out, num_keep_out, s, may_sync);
  return launch<double>(dets, order, seg, n, iou_threshold, workspace, keep_out, num_keep_out, s, may_sync);
}
}  // namespace
}  // namespace tvmi

extern "C" int tvmi_nms(const void* dets, const int64_t* order, const int64_t* seg, int64_t n,
                        double iou_threshold, tvmi_dtype dt, void* workspace,
                        size_t workspace_bytes, int64_t* keep_out, int64_t* num_keep_out,
                        void* stream) {
  return tvmi::nms_entry(dets, order, seg, n, iou_threshold, dt, workspace, workspace_bytes, keep_out, num_keep_out, stream, false);
}
extern "C" int tvmi_nms_blocking(const void* dets, const int64_t* order, const int64_t* seg, int64_t n,
                                 double iou_threshold, tvmi_dtype dt, void* workspace,
                                 size_t workspace_bytes, int64_t* keep_out, int64_t* num_keep_out,
                                 void* stream) {
  return tvmi::nms_entry(dets, order, seg, n, iou_threshold, dt, workspace, workspace_bytes, keep_out, num_keep_out, stream, true);
}

extern "C" size_t tvmi_nms_segmented_workspace_bytes(int64_t n) {
  if (n <= 0) return 0;
  return tvmi::seg_workspace_layout(n, nullptr, nullptr);
}

namespace tvmi {
namespace {
int nms_segmented_entry(const void* dets, const int64_t* order, const int64_t* seg_keys, const int64_t* perm, int64_t n,
                        const int64_t* n_dev, const int* ext_err, double iou_threshold, tvmi_dtype dt, void* workspace,
                        size_t workspace_bytes, int64_t* keep_out, int64_t* num_keep_out, void* stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  TVMI_CHECK_ARG(n >= 0, "nms_segmented: negative box count");
  TVMI_CHECK_ARG(num_keep_out != nullptr, "nms_segmented: num_keep_out is null");
  if (n == 0) {
    hipError_t e = hipMemsetAsync(num_keep_out, 0, sizeof(int64_t), s);
    return e == hipSuccess ? 0 : set_error((int)e, "tvmi_nms_segmented: memset");
  }
  TVMI_CHECK_ARG(dets && order && seg_keys && perm && keep_out && workspace, "nms_segmented: null pointer");
  TVMI_CHECK_ARG(n <= 1200000, "nms_segmented: more than 1.2M boxes is not supported by the bitmask path");
  TVMI_CHECK_ARG(workspace_bytes >= tvmi_nms_segmented_workspace_bytes(n), "nms_segmented: workspace too small");
  TVMI_CHECK_ARG(dt == TVMI_F32 || dt == TVMI_F64, "nms_segmented: dets must be float32 or float64");
  if (dt == TVMI_F32)
    return launch_seg<float>(dets, order, seg_keys, perm, n, n_dev, ext_err, iou_threshold, workspace, keep_out, num_keep_out, s);
  return launch_seg<double>(dets, order, seg_keys, perm, n, n_dev, ext_err, iou_threshold, workspace, keep_out, num_keep_out, s);
}

int nms_small_segments_entry(const void* dets, const int64_t* order, const int64_t* seg, int64_t n, const int64_t* n_dev,
                             int64_t num_segments, double iou_threshold, tvmi_dtype dt, void* workspace,
                             size_t workspace_bytes, int64_t* keep_out, int64_t* num_keep_out, void* stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  TVMI_CHECK_ARG(n >= 0 && num_keep_out != nullptr, "nms_small_segments: bad arguments");
  if (n == 0) {
    hipError_t e = hipMemsetAsync(num_keep_out, 0, sizeof(int64_t), s);
    return e == hipSuccess ? 0 : set_error((int)e, "tvmi_nms_small_segments: memset");
  }
  TVMI_CHECK_ARG(dets && order && seg && keep_out && workspace, "nms_small_segments: null pointer");
  TVMI_CHECK_ARG(n <= 4096 && num_segments >= 1 && num_segments <= 1024,
                 "nms_small_segments: needs n <= 4096 and 1 <= num_segments <= 1024");
  TVMI_CHECK_ARG(workspace_bytes >= tvmi_nms_small_segments_workspace_bytes(n, num_segments),
                 "nms_small_segments: workspace too small");
  TVMI_CHECK_ARG(dt == TVMI_F32 || dt == TVMI_F64, "nms_small_segments: dets must be float32 or float64");
  SmallSegWorkspace w;
  small_seg_workspace_layout(n, num_segments, static_cast<char*>(workspace), &w);
  hipError_t e = hipMemsetAsync(w.sync_words, 0, 2 * sizeof(int), s);
  if (e != hipSuccess) return set_error((int)e, "tvmi_nms_small_segments: memset");
  // a segment of m boxes has ceil(m/64)*(ceil(m/64)+1)/2 tiles; m <= min(n, 1024).  Two launches whose workgroups fit next to a
  // launch that owns the LDS of every CU (see nms_small_seg_collect)
  const int nbmax = (int)std::min<int64_t>(kSmallSegBlocks, ceil_div(n, 64));
  const dim3 grid4((unsigned)num_segments, (unsigned)ceil_div(nbmax * (nbmax + 1) / 2, kTiles4));
  if (dt == TVMI_F32) {
    nms_small_seg_collect<float><<<dim3((unsigned)num_segments), dim3(1024), 0, s>>>(order, seg, (int)n, n_dev, (int)num_segments, w);
    nms_small_seg_tiles4<float><<<grid4, dim3(kTiles4 * 64), 0, s>>>(static_cast<const float*>(dets), order, iou_threshold,
                                                                   thr_band(iou_threshold), w);
  } else {
    nms_small_seg_collect<double><<<dim3((unsigned)num_segments), dim3(1024), 0, s>>>(order, seg, (int)n, n_dev, (int)num_segments, w);
    nms_small_seg_tiles4<double><<<grid4, dim3(kTiles4 * 64), 0, s>>>(static_cast<const double*>(dets), order, iou_threshold,
                                                                    thr_band(iou_threshold), w);
  }
  nms_small_seg_sweep<<<dim3((unsigned)num_segments), dim3(kSuper * kWave), 0, s>>>(order, (int)n, n_dev, w, keep_out, num_keep_out);
  TVMI_RETURN_LAUNCH_STATUS("tvmi_nms_small_segments");
}

// masked candidates -> inputs of the device-count forms: score -inf and the largest key for a masked-out candidate (both sorts
// put it behind every live one), and the number of live candidates.  n_live must be zero before the launch.
__global__ __launch_bounds__(256) void nms_mask_inputs_kernel(const float* __restrict__ scores, const int64_t* __restrict__ seg,
                                                              const uint8_t* __restrict__ valid, int64_t n,
                                                              float* __restrict__ scores_out, int64_t* __restrict__ seg_out,
                                                              int64_t* __restrict__ n_live) {
  __shared__ int s_cnt[4];
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const bool live = i < n && valid[i] != 0;
  if (i < n) {
    scores_out[i] = live ? scores[i] : -INFINITY;
    seg_out[i] = live ? seg[i] : INT64_MAX;
  }
  const u64 bal = __ballot(live);
  if ((threadIdx.x & 63) == 0) s_cnt[threadIdx.x >> 6] = __popcll(bal);
  __syncthreads();
  if (threadIdx.x == 0) {
    const int t = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
    if (t) atomicAdd(reinterpret_cast<unsigned long long*>(n_live), (unsigned long long)t);
  }
}
}  // namespace
}  // namespace tvmi

extern "C" int tvmi_nms_segmented(const void* dets, const int64_t* order, const int64_t* seg_keys, const int64_t* perm,
                                  int64_t n, double iou_threshold, tvmi_dtype dt, void* workspace, size_t workspace_bytes,
                                  int64_t* keep_out, int64_t* num_keep_out, void* stream) {
  return tvmi::nms_segmented_entry(dets, order, seg_keys, perm, n, nullptr, nullptr, iou_threshold, dt, workspace, workspace_bytes,
                                   keep_out, num_keep_out, stream);
}

extern "C" int tvmi_nms_segmented_devcount(const void* dets, const int64_t* order, const int64_t* seg_keys, const int64_t* perm,
                                           int64_t capacity, const int64_t* n_dev, const int* partition_flag, double iou_threshold,
                                           tvmi_dtype dt, void* workspace, size_t workspace_bytes, int64_t* keep_out,
                                           int64_t* num_keep_out, void* stream) {
  return tvmi::nms_segmented_entry(dets, order, seg_keys, perm, capacity, n_dev, partition_flag, iou_threshold, dt, workspace,
                                   workspace_bytes, keep_out, num_keep_out, stream);
}

extern "C" int tvmi_nms_mask_inputs(const float* scores, const int64_t* seg, const uint8_t* valid, int64_t n, float* scores_out,
                                    int64_t* seg_out, int64_t* n_live, void* stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  TVMI_CHECK_ARG(n >= 0 && n_live != nullptr, "nms_mask_inputs: bad arguments");
  hipError_t e = hipMemsetAsync(n_live, 0, sizeof(int64_t), s);
  if (e != hipSuccess) return tvmi::set_error((int)e, "tvmi_nms_mask_inputs: memset");
  if (n == 0) return 0;
  TVMI_CHECK_ARG(scores && seg && valid && scores_out && seg_out, "nms_mask_inputs: null pointer");
  tvmi::nms_mask_inputs_kernel<<<dim3((unsigned)tvmi::ceil_div(n, 256)), dim3(256), 0, s>>>(scores, seg, valid, n, scores_out, seg_out,
                                                                                         n_live);
  TVMI_RETURN_LAUNCH_STATUS("tvmi_nms_mask_inputs");
}

extern "C" size_t tvmi_nms_small_segments_workspace_bytes(int64_t n, int64_t num_segments) {
  if (n <= 0 || num_segments <= 0) return 0;
  return tvmi::small_seg_workspace_layout(n, num_segments, nullptr, nullptr);
}

extern "C" int tvmi_nms_small_segments(const void* dets, const int64_t* order, const int64_t* seg, int64_t n,
                                       int64_t num_segments, double iou_threshold, tvmi_dtype dt, void* workspace,
                                       size_t workspace_bytes, int64_t* keep_out, int64_t* num_keep_out, void* stream) {
  return tvmi::nms_small_segments_entry(dets, order, seg, n, nullptr, num_segments, iou_threshold, dt, workspace, workspace_bytes,
                                        keep_out, num_keep_out, stream);
}

extern "C" int tvmi_nms_small_segments_devcount(const void* dets, const int64_t* order, const int64_t* seg, int64_t capacity,
                                                const int64_t* n_dev, int64_t num_segments, double iou_threshold, tvmi_dtype dt,
                                                void* workspace, size_t workspace_bytes, int64_t* keep_out,
                                                int64_t* num_keep_out, void* stream) {
  TVMI_CHECK_ARG(n_dev != nullptr, "nms_small_segments_devcount: n_dev is null");
  return tvmi::nms_small_segments_entry(dets, order, seg, capacity, n_dev, num_segments, iou_threshold, dt, workspace,
                                        workspace_bytes, keep_out, num_keep_out, stream);
}

namespace tvmi {
int* nms_step_sync_block(hipStream_t stream) { return step_sync_blocks().get(stream); }
}  // namespace tvmi

extern "C" size_t tvmi_nms_step_workspace_bytes(int64_t n, int64_t num_segments) {
  if (n <= 0 || num_segments <= 0) return 0;
  return tvmi::step_workspace_layout(n, num_segments, nullptr, nullptr);
}

extern "C" int tvmi_nms_step(const float* dets, const float* scores, const int64_t* seg, int64_t n, int64_t num_segments,
                             double iou_threshold, void* workspace, size_t workspace_bytes, int64_t* keep_out,
                             int64_t* num_keep_out, const int64_t* image_idx, const int64_t* labels, int64_t num_images,
                             int64_t max_dets, float* payload, int64_t row_stride, int32_t* counts, int count_in_row, void* stream) {
  using namespace tvmi;
  hipStream_t s = static_cast<hipStream_t>(stream);
  StepArgs a;
  const int st = step_prepare(&a, dets, scores, seg, n, num_segments, iou_threshold, workspace, workspace_bytes, keep_out, num_keep_out,
                              image_idx, labels, num_images, max_dets, payload, row_stride, counts, count_in_row, s);
  if (st != 0) return st;
  nms_step_fused<<<dim3((unsigned)a.S, (unsigned)a.gdim_y), dim3(kStepThreads), 0, s>>>(a.dets, a.scores, a.seg, a.n, a.S, a.G, a.thr,
                                                                                         a.band, a.ws, a.keep_out, a.num_keep, a.pk);
#ifdef TVMI_STEP_TIMING
  {
    static int calls = 0;
    if (++calls % 10 == 0) {
      (void)hipStreamSynchronize(s);
      unsigned long long h[64];
      (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(tvmi::g_step_stamp), sizeof(h));
      fprintf(stderr, "[step timing, 10 ns ticks]");
      for (int i = 1; i <= 14; ++i) fprintf(stderr, " %d:%lld", i, (long long)(h[i] - h[0]));
      fprintf(stderr, " | counting(0,G) %lld..%lld (0,last) %lld..%lld", (long long)(h[21] - h[0]), (long long)(h[22] - h[0]), (long long)(h[23] - h[0]), (long long)(h[24] - h[0]));
      fprintf(stderr, " | worker(0,16):");
      for (int i = 16; i <= 20; ++i) fprintf(stderr, " %d:%lld", i, (long long)(h[i] - h[0]));
      fprintf(stderr, "\n");
    }
  }
#endif
  TVMI_RETURN_LAUNCH_STATUS("tvmi_nms_step");
}

extern "C" int tvmi_sort_scores_desc(const float* scores, int64_t n, int64_t* order, void* stream) {
  TVMI_CHECK_ARG(n >= 0 && n <= tvmi::kSortMax, "sort_scores_desc: 0 <= n <= 4096");
  if (n == 0) return 0;
  TVMI_CHECK_ARG(scores && order, "sort_scores_desc: null pointer");
  tvmi::sort_scores_desc_rank<<<dim3((unsigned)tvmi::ceil_div(n, 64)), dim3(256), 0, static_cast<hipStream_t>(stream)>>>(scores, (int)n, order);
  TVMI_RETURN_LAUNCH_STATUS("tvmi_sort_scores_desc");
}
