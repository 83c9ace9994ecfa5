// tvmi_core.hip — version / error plumbing of libtvmi_kernels.so.
#include <string.h>

#include <atomic>
#include <mutex>

#include "tvmi_common.h"

namespace tvmi {
namespace {
thread_local char g_last_error[256] = "";
}

int set_error(int code, const char* what) {
  const char* hip_msg = hipGetErrorString(static_cast<hipError_t>(code));
  snprintf(g_last_error, sizeof(g_last_error), "%s (%d: %s)", what ? what : "tvmi", code,
           hip_msg ? hip_msg : "?");
  return code;
}
}  // namespace tvmi

extern "C" int tvmi_version(void) { return TVMI_ABI_VERSION; }
extern "C" const char* tvmi_arch(void) { return "gfx950"; }
extern "C" const char* tvmi_last_error(void) { return tvmi::g_last_error; }

// `waiter` waits for everything enqueued on `signaler` so far — torch's Stream.wait_stream, with an event that releases to
// DEVICE scope (hipEventReleaseToDevice): both streams live on this device, and torch's events release to system scope (a
// write-back + invalidate meant for host readers) on every record.  The step of this path forks and joins its two streams
// once per step; the two hops were ~25 us of its 0.28 ms (kernel-trace: 32 us between the last RoIAlign launch of a step and
// the first launch of the next with two streams, 8 us with one).  Events come from a per-device ring (an event may be
// re-recorded while an older wait on it is pending: the wait took a snapshot).
namespace tvmi {
namespace {
constexpr int kRing = 256, kMaxDev = 16;
hipEvent_t g_ring[kMaxDev][kRing];
std::atomic<bool> g_ring_ready[kMaxDev];   // zero-initialised (static storage)
std::atomic<unsigned> g_ring_next[kMaxDev];
std::mutex g_ring_mutex;
std::atomic<int> g_event_scope{1};   // 0 system (torch's), 1 device, 2 no fence from the event itself
}  // namespace
int set_stream_option(int scope) {
  if (scope < 0 || scope > 2) return -1;
  std::lock_guard<std::mutex> lock(g_ring_mutex);
  for (int d = 0; d < kMaxDev; ++d)
    if (g_ring_ready[d].load()) {
      for (int i = 0; i < kRing; ++i) (void)hipEventDestroy(g_ring[d][i]);
      g_ring_ready[d].store(false);
    }
  g_event_scope = scope;
  return 0;
}
}  // namespace tvmi

extern "C" int tvmi_stream_wait_stream(void* waiter, void* signaler) {
  using namespace tvmi;
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess || dev < 0 || dev >= kMaxDev) return set_error(e != hipSuccess ? (int)e : (int)hipErrorInvalidDevice, "stream_wait_stream: device");
  if (!g_ring_ready[dev].load(std::memory_order_acquire)) {
    std::lock_guard<std::mutex> lock(g_ring_mutex);
    if (!g_ring_ready[dev].load()) {
      const int scope = g_event_scope.load();
      const unsigned flags = hipEventDisableTiming | (scope == 1 ? hipEventReleaseToDevice : scope == 2 ? hipEventDisableSystemFence : 0u);
      for (int i = 0; i < kRing; ++i) {
        e = hipEventCreateWithFlags(&g_ring[dev][i], flags);
        if (e != hipSuccess) return set_error((int)e, "stream_wait_stream: hipEventCreateWithFlags");
      }
      g_ring_ready[dev].store(true, std::memory_order_release);
    }
  }
  hipEvent_t ev = g_ring[dev][g_ring_next[dev].fetch_add(1) % kRing];
  e = hipEventRecord(ev, static_cast<hipStream_t>(signaler));
  if (e != hipSuccess) return set_error((int)e, "stream_wait_stream: hipEventRecord");
  e = hipStreamWaitEvent(static_cast<hipStream_t>(waiter), ev, 0);
  if (e != hipSuccess) return set_error((int)e, "stream_wait_stream: hipStreamWaitEvent");
  return 0;
}

// 0: events release to system scope (what torch's events do), 1 (default): device scope, 2: no fence from the event itself
extern "C" int tvmi_stream_event_scope(int scope) {
  return tvmi::set_stream_option(scope) == 0 ? 0 : tvmi::set_error((int)hipErrorInvalidValue, "stream_event_scope: 0, 1 or 2");
}
