// quatro_hip_cxx.hpp — the two C++ helpers shared by the drop-in headers (quatro.hpp, fpfh_manager.hpp,
// teaser/graph.h): the process-wide handle and the status -> exception mapping.
#pragma once
#include <stdexcept>
#include <string>

#include "quatro_hip.h"

namespace quatro_hip {
// One process-wide handle (device 0) shared by every Quatro / FPFHManager object; created on first use.
inline qtr_handle* default_handle() {
  static qtr_handle* h = nullptr;
  if (!h) {
    const int rc = qtr_create(0, nullptr, &h);
    if (rc != QTR_OK) {
      std::string msg = h ? qtr_last_error(h) : "qtr_create failed";
      if (h) qtr_destroy(h);
      h = nullptr;
      throw std::runtime_error("[quatro_hip] " + msg);
    }
  }
  return h;
}
inline void check(qtr_handle* h, int rc) {
  if (rc == QTR_OK || rc == QTR_ERR_CLIQUE_TOO_SMALL) return;
  if (rc == QTR_ERR_BAD_ARG || rc == QTR_ERR_UNSUPPORTED) throw std::invalid_argument(qtr_last_error(h));
  throw std::runtime_error(qtr_last_error(h));
}
}  // namespace quatro_hip
