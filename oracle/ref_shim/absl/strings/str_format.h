// TEST INFRASTRUCTURE (oracle/_ref build only).  Shadow of absl/strings/str_format.h for
// /root/reference/.../hash_filter/sliding_hash_filter.cc:17,180-186 (ValidateData's message): the
// absl sources are not under /root/reference.
#pragma once
#include <cstdio>
#include <string>
namespace absl {
template <class... A>
std::string StrFormat(const char* fmt, A... a) {
  char buf[512];
  std::snprintf(buf, sizeof(buf), fmt, a...);
  return std::string(buf);
}
}  // namespace absl
