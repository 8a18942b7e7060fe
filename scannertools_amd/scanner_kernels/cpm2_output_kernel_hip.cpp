// CPM2Output op for Scanner on MI355X (pose path, BASELINE config 5).
//
// Drop-in for the reference's kernel
//   CPM2OutputKernel  /root/reference/scannertools_caffe/scannertools_caffe_cpp/cpm2_output_kernel_cpu.cpp:112-816
// Same op declaration (frame inputs "cpm2_resized_map" and "cpm2_joints", bytes input
// "original_frame_info", bytes output "poses", protobuf_name("CPM2Args")), same output element: the
// people of a frame as serialize_proto_vector_of_vectors<scanner::Point> writes them.
//   DeviceType::CPU  host columns, as the reference registers it: candidate scoring and assembly on the
//                    host (cpm2_parse.h).
//   DeviceType::GPU  the heat maps stay where the network wrote them (57 float planes of the network's
//                    input size: 55 MB per 1080p frame at scale 0.34): st_cpm2_limb_scores() samples them in
//                    place for the whole batch, and only 19 x 64 x 64 scores and the joint candidates
//                    (14 KB) per frame come back for the sequential assembly.
#include <cstring>

#include "scanner/api/kernel.h"
#include "scanner/api/op.h"
#include "scanner/util/hip.h"
#include "scanner/util/memory.h"
#include "cpm2_parse.h"
#include "proto_lite.h"
#include "scannertools_hip.h"
#include "stage.h"

namespace scanner {

template <bool ON_GPU>
class CPM2OutputKernelHIPImpl : public BatchedKernel, public VideoKernel {
 public:
  CPM2OutputKernelHIPImpl(const KernelConfig& config)
    : BatchedKernel(config), device_(config.devices[0]), stage_(ON_GPU ? config.devices[0].id : 0) {
    std::vector<proto_lite::Field> fields;
    if (!proto_lite::parse(config.args.data(), config.args.size(), &fields)) {
      RESULT_ERROR(&valid_, "Could not parse CPM2Args");
      return;
    }
    for (auto& f : fields)
      if (f.number == 2 && f.wire == 5) scale_ = proto_lite::as_float(f);
    if (!(scale_ > 0.f)) {
      RESULT_ERROR(&valid_, "CPM2Output: scale must be positive, got %f", scale_);
      return;
    }
    if (ON_GPU) {
      if (device_.type != DeviceType::GPU) {
        RESULT_ERROR(&valid_, "CPM2OutputKernelHIP runs on DeviceType::GPU only");
        return;
      }
      int st = st_ctx_create(device_.id, &ctx_);
      if (st != ST_OK) RESULT_ERROR(&valid_, "st_ctx_create(%d) failed: %s", device_.id, st_status_string(st));
    }
  }
  ~CPM2OutputKernelHIPImpl() {
    if (ctx_) st_ctx_destroy(ctx_);
  }
  void validate(Result* result) override {
    result->set_msg(valid_.msg());
    result->set_success(valid_.success());
  }

  void new_frame_info() override {
    // cpm2_output_kernel_cpu.cpp:123-137; frame_info_ is the ORIGINAL frame's (third input column)
    int rh, rw;
    int st = st_cpm2_geometry(frame_info_.height(), frame_info_.width(), scale_, &rh, &rw, &net_input_height_, &net_input_width_);
    LOG_IF(FATAL, st != ST_OK) << "CPM2Output: bad original frame info";
  }

  void execute(const BatchedElements& input_columns, BatchedElements& output_columns) override {
    LOG_IF(FATAL, input_columns.size() != 3) << "CPM2Output takes three input columns";
    const i32 heatmap_idx = 0, joints_idx = 1, frame_info_idx = 2;
    const i32 input_count = (i32)num_rows(input_columns[0]);
    if (input_count == 0) return;
    check_frame_info(CPU_DEVICE, input_columns[frame_info_idx][0]);
    const int H = net_input_height_, W = net_input_width_, mp = params_.max_peaks;
    const size_t map_bytes = (size_t)W * H * cpm2::kMaps * sizeof(f32);
    const size_t peak_floats = (size_t)cpm2::kParts * (mp + 1) * 3, score_floats = (size_t)cpm2::kLimbs * mp * mp;
    for (i32 b = 0; b < input_count; ++b) {
      LOG_IF(FATAL, input_columns[heatmap_idx][b].as_const_frame()->size() != map_bytes)
          << "CPM2Output: heat map of " << input_columns[heatmap_idx][b].as_const_frame()->size() << " bytes, expected " << map_bytes;
      LOG_IF(FATAL, input_columns[joints_idx][b].as_const_frame()->size() < peak_floats * sizeof(f32))
          << "CPM2Output: joints frame too small";
    }
    scores_.resize(score_floats * input_count);
    peaks_.resize(peak_floats * input_count);
    if (ON_GPU) {
      std::vector<const float*> hm(input_count), pk(input_count);
      for (i32 b = 0; b < input_count; ++b) {
        hm[b] = (const float*)input_columns[heatmap_idx][b].as_const_frame()->data;
        pk[b] = (const float*)input_columns[joints_idx][b].as_const_frame()->data;
      }
      float* dev_scores = (float*)stage_.reserve(score_floats * sizeof(f32) * input_count);
      int st = st_cpm2_limb_scores(ctx_, hm.data(), pk.data(), input_count, H, W, mp, params_.inter_threshold,
                                   params_.inter_min_above, dev_scores);
      LOG_IF(FATAL, st != ST_OK) << "st_cpm2_limb_scores: " << st_ctx_last_error(ctx_);
      LOG_IF(FATAL, st_ctx_sync(ctx_) != ST_OK) << "st_ctx_sync: " << st_ctx_last_error(ctx_);
      stage_.download((u8*)scores_.data(), (const u8*)dev_scores, score_floats * sizeof(f32) * input_count);
      for (i32 b = 0; b < input_count; ++b)
        stage_.download((u8*)(peaks_.data() + peak_floats * b), (const u8*)pk[b], peak_floats * sizeof(f32));
    } else {
      for (i32 b = 0; b < input_count; ++b) {
        const float* heatmap = (const float*)input_columns[heatmap_idx][b].as_const_frame()->data;
        memcpy(peaks_.data() + peak_floats * b, input_columns[joints_idx][b].as_const_frame()->data, peak_floats * sizeof(f32));
        cpm2::limb_scores_host(heatmap, peaks_.data() + peak_floats * b, H, W, params_, scores_.data() + score_floats * b);
      }
    }
    std::vector<float> joints;
    std::vector<uint8_t> bytes;
    for (i32 b = 0; b < input_count; ++b) {
      const int people = cpm2::assemble(scores_.data() + score_floats * b, peaks_.data() + peak_floats * b, frame_info_.height(),
                                        frame_info_.width(), H, W, params_, &joints);
      cpm2::serialize_people(joints, people, &bytes);
      u8* buffer = new_buffer(device_, bytes.size());
      memcpy_buffer(buffer, device_, bytes.data(), CPU_DEVICE, bytes.size());
      insert_element(output_columns[0], buffer, bytes.size());
    }
  }

 private:
  DeviceHandle device_;
  DeviceStage stage_;
  f32 scale_ = 0.f;
  int net_input_width_ = 0, net_input_height_ = 0;
  cpm2::Params params_;
  Result valid_;
  st_ctx* ctx_ = nullptr;
  std::vector<float> scores_, peaks_;
};

using CPM2OutputKernelHost = CPM2OutputKernelHIPImpl<false>;
using CPM2OutputKernelHIP = CPM2OutputKernelHIPImpl<true>;

REGISTER_OP(CPM2Output)
    .frame_input("cpm2_resized_map")
    .frame_input("cpm2_joints")
    .input("original_frame_info")
    .output("poses")
    .protobuf_name("CPM2Args");

REGISTER_KERNEL(CPM2Output, CPM2OutputKernelHost).device(DeviceType::CPU).batch().num_devices(1);

REGISTER_KERNEL(CPM2Output, CPM2OutputKernelHIP).device(DeviceType::GPU).batch().num_devices(1);
}
