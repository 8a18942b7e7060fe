// Frame / FrameInfo subset (scanner/api/frame.h).  Reference uses:
// FrameInfo(h, w, c, FrameType) and new_frame(device, info)
// (optical_flow_kernel_cpu.cpp:32-34), new_frames (optical_flow_kernel_gpu.cpp:61-64),
// frame->width()/height()/channels()/data (blur_kernel_cpu.cpp:57-70).
#pragma once
#include "scanner/util/memory.h"

namespace scanner {

enum class FrameType { U8 = 0, F32 = 1, F64 = 2 };

inline size_t size_of_frame_type(FrameType t) { return t == FrameType::U8 ? 1 : (t == FrameType::F32 ? 4 : 8); }

class FrameInfo {
 public:
  FrameInfo() : type(FrameType::U8) { shape[0] = shape[1] = shape[2] = 0; }
  FrameInfo(int h, int w, int c, FrameType t) : type(t) { shape[0] = h; shape[1] = w; shape[2] = c; }
  bool operator==(const FrameInfo& o) const {
    return shape[0] == o.shape[0] && shape[1] == o.shape[1] && shape[2] == o.shape[2] && type == o.type;
  }
  bool operator!=(const FrameInfo& o) const { return !(*this == o); }
  size_t size() const { return (size_t)shape[0] * shape[1] * shape[2] * size_of_frame_type(type); }
  int height() const { return shape[0]; }
  int width() const { return shape[1]; }
  int channels() const { return shape[2]; }
  int shape[3];
  FrameType type;
};

class Frame {
 public:
  Frame(FrameInfo info, u8* buffer) : data(buffer), type(info.type) {
    shape[0] = info.shape[0]; shape[1] = info.shape[1]; shape[2] = info.shape[2];
  }
  FrameInfo as_frame_info() const { return FrameInfo(shape[0], shape[1], shape[2], type); }
  size_t size() const { return as_frame_info().size(); }
  int height() const { return shape[0]; }
  int width() const { return shape[1]; }
  int channels() const { return shape[2]; }
  u8* data;
  int shape[3];
  FrameType type;
};

Frame* new_frame(DeviceHandle device, FrameInfo info);
// n frames backed by ONE block allocation (the engine frees it when the last frame dies)
std::vector<Frame*> new_frames(DeviceHandle device, FrameInfo info, i32 num);
void delete_frame(DeviceHandle device, Frame* frame);  // shim: frees the Frame and drops its buffer ref

}  // namespace scanner
