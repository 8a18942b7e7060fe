// Minimal stand-in for the parts of Scanner's public C++ API that the imgproc kernels of
// scannertools use (scanner-research/scanner is not vendored under /root/reference and is not
// installed here).  The surface is reconstructed from every use in the reference tree
// (SURVEY.md section 8b).  A build against real Scanner puts its include directory first on the
// include path; these headers are only reached when it is absent.
#pragma once
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

namespace scanner {

using i8 = int8_t;   using u8 = uint8_t;
using i16 = int16_t; using u16 = uint16_t;
using i32 = int32_t; using u32 = uint32_t;
using i64 = int64_t; using u64 = uint64_t;
using f32 = float;   using f64 = double;

enum class DeviceType { CPU = 0, GPU = 1 };

struct DeviceHandle {
  DeviceType type;
  i32 id;
  bool operator==(const DeviceHandle& o) const { return type == o.type && id == o.id; }
  bool operator!=(const DeviceHandle& o) const { return !(*this == o); }
};
static const DeviceHandle CPU_DEVICE = {DeviceType::CPU, 0};

// Result message used by validate(): RESULT_ERROR(&valid_, "fmt", ...)
// (reference use: scannertools_cpp/imgproc/blur_kernel_cpu.cpp:29-33,44)
struct Result {
  bool ok = true;
  std::string message;
  bool success() const { return ok; }
  void set_success(bool s) { ok = s; }
  const std::string& msg() const { return message; }
  void set_msg(const std::string& m) { message = m; }
};

inline void result_error(Result* r, const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  r->set_success(false);
  r->set_msg(buf);
}
#define RESULT_ERROR(result__, ...) ::scanner::result_error((result__), __VA_ARGS__)

// glog-style logging as the reference kernels use it (LOG(FATAL) << ..., LOG_IF(FATAL, c), LOG(WARNING) << ...):
// every severity prints to stderr, FATAL aborts.
struct LogStream {
  std::string text;
  bool active;
  int severity;
  LogStream(bool a, int sev) : active(a), severity(sev) {}
  template <typename T> LogStream& operator<<(const T& v) { if (active) append(v); return *this; }
  void append(const char* s) { text += s; }
  void append(const std::string& s) { text += s; }
  template <typename T> void append(const T& v) { text += std::to_string(v); }
  ~LogStream() {
    if (!active) return;
    static const char* const names[4] = {"INFO", "WARNING", "ERROR", "FATAL"};
    // a severity outside the four is printed as such, it does not turn into FATAL
    if (severity >= 0 && severity <= 3) fprintf(stderr, "%s: %s\n", names[severity], text.c_str());
    else fprintf(stderr, "LOG(%d): %s\n", severity, text.c_str());
    if (severity == 3) abort();
  }
};
// glog's spelling, LOG(INFO) / LOG_IF(FATAL, cond), by token pasting: the severities are ST_LOG_* constants, and the bare
// words INFO / WARNING / ERROR / FATAL are NOT defined as macros (they would rewrite every later use of those identifiers --
// enum values, generated protobuf code -- in each translation unit that includes this header).
enum { ST_LOG_INFO = 0, ST_LOG_WARNING = 1, ST_LOG_ERROR = 2, ST_LOG_FATAL = 3 };
#ifndef LOG
#define LOG(severity) ::scanner::LogStream(true, ::scanner::ST_LOG_##severity)
#define LOG_IF(severity, cond) ::scanner::LogStream(static_cast<bool>(cond), ::scanner::ST_LOG_##severity)
#endif

}  // namespace scanner
