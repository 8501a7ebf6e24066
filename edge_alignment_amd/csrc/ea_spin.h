// ea_spin.h — bounded polling of the pinned progress words a solve waits on (host side, no HIP in here so that the
// logic can be compiled and tested without a device: tests/lm_host_shim.cpp).
#pragma once
#include <chrono>
#include <sched.h>

namespace ea {

// One instance per wait.  poll() is called after every look at the progress words that found nothing new; it relaxes
// the core (a pause every poll, a sched_yield every 256) and reports true once `timeout_ms` have gone by since the last
// progress() -- a kernel that never lowers its flag then costs the caller an error code instead of a hung process.
// timeout_ms < 0: unbounded (the round-1 behaviour).
class SpinWait {
 public:
  explicit SpinWait(double timeout_ms) : timeout_ms_(timeout_ms) { progress(); }
  void progress() {
    polls_ = 0;
    last_ = std::chrono::steady_clock::now();
  }
  bool poll() {
    ++polls_;
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#endif
    if ((polls_ & 0xff) != 0) return false;
    sched_yield();
    if (timeout_ms_ < 0) return false;
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - last_).count();
    return ms > timeout_ms_;
  }
  double timeout_ms() const { return timeout_ms_; }

 private:
  double timeout_ms_;
  unsigned polls_ = 0;
  std::chrono::steady_clock::time_point last_;
};

// spin until *flag == 0; 0 = lowered, 1 = timed out
inline int spin_until_zero(const volatile int *flag, double timeout_ms) {
  SpinWait w(timeout_ms);
  while (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != 0)
    if (w.poll()) return 1;
  return 0;
}

}  // namespace ea
