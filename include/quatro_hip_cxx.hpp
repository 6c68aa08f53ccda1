// quatro_hip_cxx.hpp — the two C++ helpers shared by the drop-in headers (quatro.hpp, fpfh_manager.hpp,
// teaser/graph.h): the process-wide handle and the status -> exception mapping.
#pragma once
#include <mutex>
#include <stdexcept>
#include <string>

#include "quatro_hip.h"

namespace quatro_hip {
// One process-wide handle (device 0) shared by every Quatro / FPFHManager object; created on first use.
// The function-local static is initialised exactly once even when several threads arrive together (C++11); a failed
// qtr_create throws out of the initialiser, so the next call tries again.
inline qtr_handle* default_handle() {
  static qtr_handle* h = []() {
    qtr_handle* hh = nullptr;
    const int rc = qtr_create(0, nullptr, &hh);
    if (rc != QTR_OK) {
      std::string msg = hh ? qtr_last_error(hh) : "qtr_create failed";
      if (hh) qtr_destroy(hh);
      throw std::runtime_error("[quatro_hip] " + msg);
    }
    return hh;
  }();
  return h;
}
// Every drop-in object forwards to slot 0 of that handle, and the C ABI wants same-slot calls serialised: reference
// objects are independent of each other, so two of them may legally be driven from two threads — the wrappers take
// this mutex around every slot-0 call (and around reading the error text that belongs to it).
inline std::recursive_mutex& default_slot_mutex() {
  static std::recursive_mutex m;
  return m;
}
inline void check(qtr_handle* h, int rc) {
  if (rc == QTR_OK || rc == QTR_ERR_CLIQUE_TOO_SMALL) return;
  if (rc == QTR_ERR_BAD_ARG || rc == QTR_ERR_UNSUPPORTED) throw std::invalid_argument(qtr_last_error(h));
  throw std::runtime_error(qtr_last_error(h));
}
}  // namespace quatro_hip
