// Stand-in for scanner/util/profiler.h (Scanner itself is out of scope): the part of the worker's Profiler a kernel
// class touches -- add_interval(key, start, end) with now() time points, as the reference's kernels call it
// (scannertools_caffe_cpp/caffe_kernel.cpp:381-387, cpm2_input_kernel_gpu.cpp:92,153-155).  Scanner's own class writes the
// intervals into the job's profile; this one keeps them so that the mini engine (and a test) can read them back.
#pragma once
#include <chrono>
#include <mutex>
#include <string>
#include <vector>

namespace scanner {

using timepoint_t = std::chrono::time_point<std::chrono::high_resolution_clock>;
inline timepoint_t now() { return std::chrono::high_resolution_clock::now(); }

class Profiler {
 public:
  struct TaskRecord {
    std::string key;
    int64_t start, end;  // nanoseconds since the profiler was created
  };
  Profiler() : base_(now()) {}
  void add_interval(const std::string& key, timepoint_t start, timepoint_t end) {
    std::lock_guard<std::mutex> g(m_);
    records_.push_back({key, ns(start), ns(end)});
  }
  std::vector<TaskRecord> get_records() const {
    std::lock_guard<std::mutex> g(m_);
    return records_;
  }

 private:
  int64_t ns(timepoint_t t) const { return std::chrono::duration_cast<std::chrono::nanoseconds>(t - base_).count(); }
  timepoint_t base_;
  mutable std::mutex m_;
  std::vector<TaskRecord> records_;
};

}  // namespace scanner
