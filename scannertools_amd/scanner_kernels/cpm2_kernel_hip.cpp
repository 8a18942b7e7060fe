// CPM2 op for Scanner on MI355X (pose path, BASELINE config 5): the network between CPM2Input and CPM2Output.
//
// Drop-in for the reference's kernel
//   CPM2Kernel  /root/reference/scannertools_caffe/scannertools_caffe_cpp/cpm2_kernel.cpp:8-52
// Same op declaration (frame_input("cpm2_input") -> frame_output("cpm2_resized_map"), frame_output("cpm2_joints")),
// same arguments (CPM2Args{caffe_args = 1 {net_descriptor = 1 {model_path = 1, model_weights_path = 2, ...},
// batch_size = 2}, scale = 2}, scannertools_caffe.proto:1-48), registered on DeviceType::GPU and DeviceType::CPU
// with .batch() like the reference.  The reference is a CaffeKernel: it loads prototxt + caffemodel into Caffe, and
// net_config() (cpm2_kernel.cpp:13-29) points the fork's `resize` layer at the network input size.  Here the
// forward pass is pose_net.h (MFMA convolution kernels through the C ABI) followed by st_cpm2_resize_maps and
// st_cpm2_nms; the architecture is the COCO body model's (pose_net.h says where the layer list comes from), the
// weights are read from the caffemodel the arguments name.  Outputs:
//   cpm2_resized_map  FrameInfo(57, H, W, F32): 19 part heat maps, then 38 part-affinity planes, at the network
//                     input size (what cpm2_output_kernel_cpu.cpp:84-88 indexes)
//   cpm2_joints       FrameInfo(18, max_peaks + 1, 3, F32): per part [count, -, -] then (x, y, score) peaks
//                     (cpm2_output_kernel_cpu.cpp:481-499)
#include "scanner/api/kernel.h"
#include "scanner/api/op.h"
#include "scanner/util/hip.h"
#include "scanner/util/memory.h"
#include "pose_net.h"
#include "proto_lite.h"
#include "scannertools_hip.h"
#include "stage.h"

namespace scanner {
namespace {
constexpr int kMaxPeaks = 64;           // cpm2_output_kernel_cpu.cpp:760-761 (max_peaks_)
constexpr float kNmsThreshold = 0.05f;  // the model's nms_param ([EXT] pose_deploy_linevec.prototxt)

// CPM2Args.caffe_args.net_descriptor.model_weights_path (and .model_path, the deploy prototxt, when given)
bool parse_weights_path(const std::vector<u8>& args, std::string* path, std::string* prototxt) {
  std::vector<proto_lite::Field> top, caffe, net;
  if (!proto_lite::parse(args.data(), args.size(), &top)) return false;
  for (auto& f : top)
    if (f.number == 1 && f.wire == 2 && !proto_lite::parse((const uint8_t*)f.bytes.data(), f.bytes.size(), &caffe)) return false;
  for (auto& f : caffe)
    if (f.number == 1 && f.wire == 2 && !proto_lite::parse((const uint8_t*)f.bytes.data(), f.bytes.size(), &net)) return false;
  for (auto& f : net) {
    if (f.number == 2 && f.wire == 2) *path = f.bytes;
    if (f.number == 1 && f.wire == 2) *prototxt = f.bytes;
  }
  return true;
}
}  // namespace

template <bool STAGED>
class CPM2KernelHIPImpl : public BatchedKernel, public VideoKernel {
 public:
  CPM2KernelHIPImpl(const KernelConfig& config)
    : BatchedKernel(config), device_(config.devices[0]), gpu_(STAGED ? staging_device_id() : config.devices[0].id),
      stage_(gpu_) {
    std::string path, prototxt;
    if (!parse_weights_path(config.args, &path, &prototxt)) {
      RESULT_ERROR(&valid_, "Could not parse CPM2Args");
      return;
    }
    if (path.empty()) {
      RESULT_ERROR(&valid_, "CPM2: CPM2Args.caffe_args.net_descriptor.model_weights_path is empty");
      return;
    }
    if (!STAGED && device_.type != DeviceType::GPU) {
      RESULT_ERROR(&valid_, "CPM2KernelHIP runs on DeviceType::GPU only");
      return;
    }
    int st = st_ctx_create(gpu_, &ctx_);
    if (st != ST_OK) {
      RESULT_ERROR(&valid_, "st_ctx_create(%d) failed: %s (no CPU fallback exists)", gpu_, st_status_string(st));
      return;
    }
    if (hipSetDevice(gpu_) != hipSuccess) {
      RESULT_ERROR(&valid_, "CPM2: hipSetDevice(%d) failed", gpu_);
      return;
    }
    std::string err;
    if (!net_.load(path, &err, prototxt)) RESULT_ERROR(&valid_, "CPM2: %s", err.c_str());
    for (int c = 0; c < 57; ++c) chan_[c] = c < pose::kHeat ? pose::kOffHeat + c : pose::kOffPaf + (c - pose::kHeat);
  }
  ~CPM2KernelHIPImpl() {
    (void)hipSetDevice(gpu_);  // the network's buffers are freed by its destructor, on their device
    if (ctx_) st_ctx_destroy(ctx_);
  }
  void validate(Result* result) override {
    result->set_msg(valid_.msg());
    result->set_success(valid_.success());
  }

  void execute(const BatchedElements& input_columns, BatchedElements& output_columns) override {
    auto& in_col = input_columns[0];
    const i32 n = (i32)num_rows(in_col);
    if (n == 0) return;
    check_frame(device_, in_col[0]);
    // the input frame is CPM2Input's: FrameInfo(3, H, W, F32) (cpm2_input_kernel_gpu.cpp:112-113)
    LOG_IF(FATAL, frame_info_.shape[0] != 3 || frame_info_.type != FrameType::F32) << "CPM2 expects planar (3, H, W) F32 frames";
    const int H = frame_info_.shape[1], W = frame_info_.shape[2];
    LOG_IF(FATAL, H % 8 || W % 8) << "CPM2: the network input must be padded to a multiple of 8 (CPM2Input does)";
    HIP_CHECK(hipSetDevice(gpu_));
    FrameInfo map_info(57, H, W, FrameType::F32), joint_info(pose::kHeat - 1, kMaxPeaks + 1, 3, FrameType::F32);
    std::vector<Frame*> maps = new_frames(device_, map_info, n), joints = new_frames(device_, joint_info, n);
    const size_t in_bytes = frame_info_.size(), map_bytes = map_info.size(), joint_bytes = joint_info.size();
    src_.resize(n); map_ptr_.resize(n); joint_ptr_.resize(n);
    if (STAGED) {
      const size_t is = DeviceStage::align(in_bytes), ms = DeviceStage::align(map_bytes), js = DeviceStage::align(joint_bytes);
      u8* dev = stage_.reserve((is + ms + js) * n);
      for (i32 i = 0; i < n; ++i) {
        LOG_IF(FATAL, in_col[i].as_const_frame()->as_frame_info() != frame_info_) << "CPM2: frame shape changes inside a batch";
        stage_.upload(dev + is * i, in_col[i].as_const_frame()->data, in_bytes);
        src_[i] = (const float*)(dev + is * i);
        map_ptr_[i] = (float*)(dev + is * n + ms * i);
        joint_ptr_[i] = (float*)(dev + (is + ms) * n + js * i);
      }
    } else {
      for (i32 i = 0; i < n; ++i) {
        LOG_IF(FATAL, in_col[i].as_const_frame()->as_frame_info() != frame_info_) << "CPM2: frame shape changes inside a batch";
        src_[i] = (const float*)in_col[i].as_const_frame()->data;
        map_ptr_[i] = (float*)maps[i]->data;
        joint_ptr_[i] = (float*)joints[i]->data;
      }
    }
    std::string err;
    const auto net_start = now();  // caffe_kernel.cpp:381
    const float* final_maps = net_.forward(ctx_, src_.data(), n, H, W, &err);
    LOG_IF(FATAL, !final_maps) << "CPM2: " << err;
    int st = st_cpm2_resize_maps(ctx_, final_maps, n, H / 8, W / 8, pose::kCatPad, chan_, 57, H, W, map_ptr_.data());
    LOG_IF(FATAL, st != ST_OK) << "st_cpm2_resize_maps: " << st_ctx_last_error(ctx_);
    cmap_ptr_.assign(map_ptr_.begin(), map_ptr_.end());
    st = st_cpm2_nms(ctx_, cmap_ptr_.data(), n, H, W, pose::kHeat - 1, kMaxPeaks, kNmsThreshold, joint_ptr_.data());
    LOG_IF(FATAL, st != ST_OK) << "st_cpm2_nms: " << st_ctx_last_error(ctx_);
    st = st_ctx_sync(ctx_);
    LOG_IF(FATAL, st != ST_OK) << "st_ctx_sync: " << st_ctx_last_error(ctx_);
    // the network + its resize / nms layers, complete on the device (the reference brackets net->Forward() the same way and
    // notes that the interval is only meaningful with a synchronisation, caffe_kernel.cpp:384-387)
    if (profiler_) profiler_->add_interval("caffe:net", net_start, now());
    if (STAGED)
      for (i32 i = 0; i < n; ++i) {
        stage_.download(maps[i]->data, (const u8*)map_ptr_[i], map_bytes);
        stage_.download(joints[i]->data, (const u8*)joint_ptr_[i], joint_bytes);
      }
    for (i32 i = 0; i < n; ++i) {
      insert_frame(output_columns[0], maps[i]);
      insert_frame(output_columns[1], joints[i]);
    }
  }

 private:
  DeviceHandle device_;
  int gpu_;
  DeviceStage stage_;
  Result valid_;
  st_ctx* ctx_ = nullptr;
  pose::Net net_;
  int chan_[57];
  std::vector<const float*> src_, cmap_ptr_;
  std::vector<float*> map_ptr_, joint_ptr_;
};

}  // namespace scanner

// Model-file check without a GPU: number of layers whose weights were found with the architecture's sizes (92 = all),
// or -1 with the reason in `err`.
extern "C" __attribute__((visibility("default"))) int scannertools_caffe_check_model(const char* caffemodel, char* err, size_t err_len) {
  std::string msg;
  int matched = 0;
  const bool ok = caffemodel && scanner::pose::check_caffemodel(caffemodel, &matched, &msg);
  if (!caffemodel) msg = "null path";
  if (err && err_len) { strncpy(err, msg.c_str(), err_len - 1); err[err_len - 1] = 0; }
  return ok ? matched : -1;
}

// Deploy-description check without a GPU: 92 when the prototxt describes the network the kernels implement and, if
// `caffemodel` is given, the weights of every one of ITS layer names are in that file with the architecture's sizes;
// -1 with the reason in `err`.
extern "C" __attribute__((visibility("default"))) int scannertools_caffe_check_prototxt(const char* prototxt, const char* caffemodel, char* err,
                                                                                        size_t err_len) {
  std::string msg;
  int matched = -1;
  try {
    std::vector<std::string> names;
    if (!prototxt) {
      msg = "null path";
    } else if (scanner::pose::prototxt_layer_names(prototxt, &names, &msg)) {
      matched = (int)names.size();
      if (caffemodel) {
        std::map<std::string, scanner::pose::Blobs> blobs;
        const auto arch = scanner::pose::all_layers();
        if (!scanner::pose::read_caffemodel(caffemodel, &blobs, &msg)) {
          matched = -1;
        } else {
          for (size_t i = 0; i < arch.size() && matched >= 0; ++i) {
            auto it = blobs.find(names[i]);
            if (it == blobs.end() || it->second.w.size() != (size_t)arch[i].cout * arch[i].cin * arch[i].k * arch[i].k ||
                it->second.b.size() != (size_t)arch[i].cout) {
              msg = "caffemodel has no weights of the right size for prototxt layer " + names[i];
              matched = -1;
            }
          }
        }
      }
    }
  } catch (const std::exception& e) {
    msg = e.what();
    matched = -1;
  }
  if (err && err_len) { strncpy(err, msg.c_str(), err_len - 1); err[err_len - 1] = 0; }
  return matched;
}

namespace scanner {
using CPM2KernelHIP = CPM2KernelHIPImpl<false>;
using CPM2KernelHIPStaged = CPM2KernelHIPImpl<true>;

REGISTER_OP(CPM2).frame_input("cpm2_input").frame_output("cpm2_resized_map").frame_output("cpm2_joints").protobuf_name("CPM2Args");

REGISTER_KERNEL(CPM2, CPM2KernelHIPStaged).device(DeviceType::CPU).num_devices(1).batch();
REGISTER_KERNEL(CPM2, CPM2KernelHIP).device(DeviceType::GPU).num_devices(1).batch();
}
