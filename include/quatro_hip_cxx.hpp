// quatro_hip_cxx.hpp — the two C++ helpers shared by the drop-in headers (quatro.hpp, fpfh_manager.hpp,
// teaser/graph.h): the process-wide handle and the status -> exception mapping.
#pragma once
#include <cstdlib>
#include <functional>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>

#include "quatro_hip.h"

namespace quatro_hip {
// One process-wide handle (device 0) shared by every Quatro / FPFHManager object; created on first use with
// QUATRO_HIP_SLOTS stream slots (default 4; one slot = one set of device arenas, ~0.7 GB at the default limits).
// The function-local static is initialised exactly once even when several threads arrive together (C++11); a failed
// qtr_create throws out of the initialiser, so the next call tries again.
inline int default_slot_count() {
  static const int n = []() {
    const char* e = std::getenv("QUATRO_HIP_SLOTS");
    const int v = e ? std::atoi(e) : 4;
    return v < 1 ? 1 : (v > 16 ? 16 : v);
  }();
  return n;
}
inline qtr_handle* default_handle() {
  static qtr_handle* h = []() {
    qtr_handle* hh = nullptr;
    qtr_limits lim;
    qtr_default_limits(&lim);
    lim.n_slots = default_slot_count();
    const int rc = qtr_create(0, &lim, &hh);
    if (rc != QTR_OK) {
      std::string msg = hh ? qtr_last_error(hh) : "qtr_create failed";
      if (hh) qtr_destroy(hh);
      throw std::runtime_error("[quatro_hip] " + msg);
    }
    // The reference's classes have no stage timers (the demo brackets its calls with std::chrono): the drop-in objects
    // record no events — each record is a marker the GPU queue retires before the next launch, ~17 us per registration.
    // QUATRO_HIP_TIMING=1 keeps them for callers that read qtr_get_stage_times on default_handle() themselves.
    const char* keep = std::getenv("QUATRO_HIP_TIMING");
    if (!(keep && keep[0] == '1')) {
      qtr_set_stage_events(hh, 0);
      qtr_set_nn_event_stride(hh, 0);
    }
    return hh;
  }();
  return h;
}
// The C ABI wants calls on one slot serialised, and reference objects are independent of each other: two of them may
// legally be driven from two threads.  Every wrapper call leases a slot for its duration — the first free one, else it
// queues on the slot its thread hashes to — so independent objects proceed side by side on different slots instead of
// behind one process-wide mutex.  A wrapper that calls another wrapper on the same thread reuses the lease it holds.
class SlotLease {
 public:
  int slot;
  SlotLease() : slot(0), owner_(false) {
    int& held = held_slot();
    if (held >= 0) {  // nested call on this thread
      slot = held;
      return;
    }
    const int n = default_slot_count();
    for (int i = 0; i < n && !owner_; ++i)
      if (pool()[i].try_lock()) {
        slot = i;
        owner_ = true;
      }
    if (!owner_) {
      slot = static_cast<int>(std::hash<std::thread::id>()(std::this_thread::get_id()) % static_cast<size_t>(n));
      pool()[slot].lock();
      owner_ = true;
    }
    held = slot;
  }
  ~SlotLease() {
    if (owner_) {
      held_slot() = -1;
      pool()[slot].unlock();
    }
  }
  SlotLease(const SlotLease&) = delete;
  SlotLease& operator=(const SlotLease&) = delete;

 private:
  bool owner_;
  static std::mutex* pool() {
    static std::mutex m[16];
    return m;
  }
  static int& held_slot() {
    static thread_local int s = -1;
    return s;
  }
};
inline void check(qtr_handle* h, int rc) {
  if (rc == QTR_OK || rc == QTR_ERR_CLIQUE_TOO_SMALL) return;
  if (rc == QTR_ERR_BAD_ARG || rc == QTR_ERR_UNSUPPORTED) throw std::invalid_argument(qtr_last_error(h));
  throw std::runtime_error(qtr_last_error(h));
}
}  // namespace quatro_hip
