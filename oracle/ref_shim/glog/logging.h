// TEST INFRASTRUCTURE (oracle/_ref build only).  glog is not in the reference tree.
// cuckoohash_map.hpp:43 only needs the include to resolve; batch_softmax_optimizer.cc:28,56 uses
// DCHECK_EQ and LOG(FATAL) << ...: the check is a no-op here, a FATAL message throws when it is
// destroyed (the drivers catch it).
#pragma once
#include <iostream>
#include <sstream>
#include <stdexcept>
namespace oracle_glog_shim {
struct FatalSink {
  std::ostringstream os;
  template <class T>
  FatalSink& operator<<(const T& v) {
    os << v;
    return *this;
  }
  ~FatalSink() noexcept(false) { throw std::runtime_error(os.str()); }
};
struct NullSink {
  template <class T>
  NullSink& operator<<(const T&) { return *this; }
};
}  // namespace oracle_glog_shim
#ifndef LOG
#define ORACLE_GLOG_SINK_FATAL ::oracle_glog_shim::FatalSink()
#define ORACLE_GLOG_SINK_ERROR ::oracle_glog_shim::NullSink()
#define ORACLE_GLOG_SINK_WARNING ::oracle_glog_shim::NullSink()
#define ORACLE_GLOG_SINK_INFO ::oracle_glog_shim::NullSink()
#define LOG(sev) ORACLE_GLOG_SINK_##sev
#endif
#ifndef DCHECK_EQ
#define DCHECK_EQ(a, b) ((void)0)
#endif
