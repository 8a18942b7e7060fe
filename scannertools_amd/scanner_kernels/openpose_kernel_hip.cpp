// OpenPose op for Scanner on MI355X (pose path, BASELINE config 5): frames -> people, body keypoints only.
//
// Stands in for the reference's kernel
//   OpenPoseKernel  /root/reference/scannertools_caffe/scannertools_caffe_cpp/openpose_kernel.cpp:20-229
// which hands whole frames to the OpenPose LIBRARY (op::Wrapper: COCO_18 body model, net input height 368,
// `pose_num_scales` scales `pose_scale_gap` apart, ScaleMode::ZeroToOne, no rendering; :88-123) and copies its results
// into one byte element per frame (:160-212).  Same op declaration (frame_input("frame") -> output("pose", Bytes,
// "PoseList"), protobuf_name("OpenPoseArgs")), same arguments (scannertools_caffe.proto:50-61), same element layout:
//   per person  [1 pose score][18 x (x, y, score)][70 x 3 face][21 x 3 left hand][21 x 3 right hand]   float32
//   no person   one float 0
// What runs here is this repository's pose chain -- CPM2Input's transform at scale 368 / frame height (x (1 - i gap)
// for scale i), the body network (pose_net.h), the scales' maps merged by st_cpm2_resize_merge_maps, st_cpm2_nms,
// st_cpm2_limb_scores and the assembly of cpm2_parse.h -- i.e. the algorithm of the reference's own CPM2 ops, not a
// restatement of the OpenPose library's internals, which are not in the reference tree ([EXT]; PARITY UNPINNED:
// OpenPose's resize / NMS / connection code differs in details such as sub-pixel peak refinement).  Coordinates are
// ZeroToOne like the reference's configuration: x / frame width, y / frame height.  The pose score is the mean of the
// 18 keypoint scores.  Hands and faces (compute_hands / compute_face) need networks that are not built: asking for
// them fails validate().  The model is read from <model_directory>/pose/coco/pose_iter_440000.caffemodel (OpenPose's
// layout, openpose_kernel.cpp:47-52); nothing is downloaded.
#include "scanner/api/kernel.h"
#include "scanner/api/op.h"
#include "scanner/util/hip.h"
#include "scanner/util/memory.h"
#include "cpm2_parse.h"
#include "pose_net.h"
#include "proto_lite.h"
#include "scannertools_hip.h"
#include "stage.h"

namespace scanner {
namespace {
constexpr int POSE_KEYPOINTS = 18, FACE_KEYPOINTS = 70, HAND_KEYPOINTS = 21, POSE_SCORES = 1;
constexpr int TOTAL_KEYPOINTS = POSE_KEYPOINTS + FACE_KEYPOINTS + 2 * HAND_KEYPOINTS;  // openpose_kernel.cpp:14-18
constexpr int kNetHeight = 368;                                                          // openpose_kernel.cpp:99
constexpr float kNmsThreshold = 0.05f;

struct OpenPoseArgs {  // scannertools_caffe.proto:50-61
  std::string model_directory;
  int pose_num_scales = 0;
  float pose_scale_gap = 0.f;
  bool compute_hands = false, compute_face = false;
};

bool parse_args(const std::vector<u8>& bytes, OpenPoseArgs* a) {
  std::vector<proto_lite::Field> fields;
  if (!proto_lite::parse(bytes.data(), bytes.size(), &fields)) return false;
  for (auto& f : fields) {
    if (f.number == 1 && f.wire == 2) a->model_directory = f.bytes;
    else if (f.number == 2 && f.wire == 0) a->pose_num_scales = (int)f.value;
    else if (f.number == 3 && f.wire == 5) a->pose_scale_gap = proto_lite::as_float(f);
    else if (f.number == 4 && f.wire == 0) a->compute_hands = f.value != 0;
    else if (f.number == 7 && f.wire == 0) a->compute_face = f.value != 0;
  }
  return true;
}
}  // namespace

template <bool STAGED>
class OpenPoseKernelHIPImpl : public BatchedKernel, public VideoKernel {
 public:
  OpenPoseKernelHIPImpl(const KernelConfig& config)
    : BatchedKernel(config), device_(config.devices[0]), gpu_(STAGED ? staging_device_id() : config.devices[0].id),
      stage_(gpu_) {
    if (!parse_args(config.args, &args_)) {
      RESULT_ERROR(&valid_, "Could not parse OpenPoseArgs");
      return;
    }
    if (args_.compute_hands || args_.compute_face) {
      RESULT_ERROR(&valid_, "OpenPose: compute_hands / compute_face need the hand and face networks, which this build does not have");
      return;
    }
    scales_ = args_.pose_num_scales < 1 ? 1 : args_.pose_num_scales;  // proto3 default 0 = one scale
    if (scales_ > 8 || !(args_.pose_scale_gap >= 0.f) || (scales_ > 1 && !(1.f - (scales_ - 1) * args_.pose_scale_gap > 0.05f))) {
      RESULT_ERROR(&valid_, "OpenPose: pose_num_scales = %d with pose_scale_gap = %f is outside what is supported (<= 8 scales, smallest above 0.05)",
                   scales_, args_.pose_scale_gap);
      return;
    }
    if (args_.model_directory.empty()) {
      RESULT_ERROR(&valid_, "OpenPose: OpenPoseArgs.model_directory is empty (the reference downloads the model there; this build does not)");
      return;
    }
    if (!STAGED && device_.type != DeviceType::GPU) {
      RESULT_ERROR(&valid_, "OpenPoseKernelHIP runs on DeviceType::GPU only");
      return;
    }
    int st = st_ctx_create(gpu_, &ctx_);
    if (st != ST_OK) {
      RESULT_ERROR(&valid_, "st_ctx_create(%d) failed: %s (no CPU fallback exists)", gpu_, st_status_string(st));
      return;
    }
    if (hipSetDevice(gpu_) != hipSuccess) {
      RESULT_ERROR(&valid_, "OpenPose: hipSetDevice(%d) failed", gpu_);
      return;
    }
    std::string err;
    // the deploy description beside the weights (OpenPose's own file layout), when the directory holds it
    std::string proto = args_.model_directory + "/pose/coco/pose_deploy_linevec.prototxt", probe;
    if (!pose::read_file(proto, &probe)) proto.clear();
    const std::string model = args_.model_directory + "/pose/coco/pose_iter_440000.caffemodel";
    if (!proto.empty()) {
      // The description was found by probing, not named by the caller (OpenPoseArgs has no field for it): a file this
      // reader cannot FOLLOW (parse or structure) must not make a model unusable that loads by the published layer names.
      // Only that failure falls back; a description that parses and then names weights the caffemodel lacks, or an upload
      // that fails, is an error (and nothing is loaded twice over a half-loaded net).
      std::vector<std::string> names;
      std::string perr;
      if (!pose::prototxt_layer_names(proto, &names, &perr)) {
        LOG(WARNING) << "OpenPose: ignoring " << proto << " (" << perr << "); weights are looked up by the published layer names";
        proto.clear();
      }
    }
    const bool loaded = net_.load(model, &err, proto);
    if (!loaded) RESULT_ERROR(&valid_, "OpenPose: %s", err.c_str());
    for (int c = 0; c < 57; ++c) chan_[c] = c < pose::kHeat ? pose::kOffHeat + c : pose::kOffPaf + (c - pose::kHeat);
  }
  ~OpenPoseKernelHIPImpl() {
    (void)hipSetDevice(gpu_);
    if (ctx_) st_ctx_destroy(ctx_);
  }
  void validate(Result* result) override {
    result->set_msg(valid_.msg());
    result->set_success(valid_.success());
  }

  void new_frame_info() override {
    const int H = frame_info_.height(), W = frame_info_.width();
    geom_.clear();
    for (int i = 0; i < scales_; ++i) {
      Geom g;
      // the network input height is a given (368 rows at scale 0, openpose_kernel.cpp:99; 368 (1 - i gap) rounded at
      // scale i): the transform's float scale is chosen so that its truncating size rule lands on exactly that height
      const int target = (int)lroundf((float)kNetHeight * (1.f - (float)i * args_.pose_scale_gap));
      int st = st_cpm2_scale_for_height(H, target, &g.scale);
      if (st == ST_OK) st = st_cpm2_geometry(H, W, g.scale, &g.rh, &g.rw, &g.nh, &g.nw);
      LOG_IF(FATAL, st != ST_OK || g.rh < 8 || g.rw < 8) << "OpenPose: a " << W << "x" << H << " frame at scale " << g.scale
                                                            << " leaves no network input";
      geom_.push_back(g);
    }
  }

  void execute(const BatchedElements& input_columns, BatchedElements& output_columns) override {
    auto& frame_col = input_columns[0];
    const i32 n = (i32)num_rows(frame_col);
    if (n == 0) return;
    check_frame(device_, frame_col[0]);
    LOG_IF(FATAL, frame_info_.channels() != 3 || frame_info_.type != FrameType::U8) << "OpenPose expects U8 frames with 3 channels";
    HIP_CHECK(hipSetDevice(gpu_));
    const int H = frame_info_.height(), W = frame_info_.width(), mp = params_.max_peaks;
    const Geom& g0 = geom_[0];
    // device scratch: [staged frames][network inputs of the current scale][merged maps][joints][limb scores]
    const size_t frame_bytes = DeviceStage::align(frame_info_.size());
    size_t in_bytes = 0;
    for (auto& g : geom_) in_bytes = std::max(in_bytes, DeviceStage::align((size_t)3 * g.nh * g.nw * sizeof(f32)));
    const size_t map_bytes = DeviceStage::align((size_t)57 * g0.nh * g0.nw * sizeof(f32));
    const size_t peak_floats = (size_t)cpm2::kParts * (mp + 1) * 3, score_floats = (size_t)cpm2::kLimbs * mp * mp;
    const size_t joint_bytes = DeviceStage::align(peak_floats * sizeof(f32)), score_bytes = DeviceStage::align(score_floats * sizeof(f32) * n);
    u8* dev = stage_.reserve(((STAGED ? frame_bytes : 0) + in_bytes + map_bytes + joint_bytes) * n + score_bytes);
    u8* p = dev;
    frames_.resize(n);
    for (i32 i = 0; i < n; ++i) {
      LOG_IF(FATAL, frame_col[i].as_const_frame()->as_frame_info() != frame_info_) << "OpenPose: frame shape changes inside a batch";
      if (STAGED) {
        stage_.upload(p, frame_col[i].as_const_frame()->data, frame_info_.size());
        frames_[i] = p;
        p += frame_bytes;
      } else {
        frames_[i] = frame_col[i].as_const_frame()->data;
      }
    }
    net_in_.resize(n); maps_.resize(n); joints_.resize(n);
    for (i32 i = 0; i < n; ++i) { net_in_[i] = (float*)p; p += in_bytes; }
    for (i32 i = 0; i < n; ++i) { maps_[i] = (float*)p; p += map_bytes; }
    for (i32 i = 0; i < n; ++i) { joints_[i] = (float*)p; p += joint_bytes; }
    float* dev_scores = (float*)p;

    // every scale: input transform + network; the stage buffers of scale s stay alive in slot s until the merge
    std::vector<const float*> srcs(scales_);
    std::vector<int> sh(scales_), sw(scales_);
    std::vector<float> eh(scales_), ew(scales_);
    std::string err;
    for (int s = 0; s < scales_; ++s) {
      const Geom& g = geom_[s];
      int st = st_cpm2_input_batch(ctx_, frames_.data(), n, H, W, g.scale, net_in_.data());
      LOG_IF(FATAL, st != ST_OK) << "st_cpm2_input_batch: " << st_ctx_last_error(ctx_);
      cin_.assign(net_in_.begin(), net_in_.end());
      srcs[s] = net_.forward(ctx_, cin_.data(), n, g.nh, g.nw, &err, s);
      LOG_IF(FATAL, !srcs[s]) << "OpenPose: " << err;
      sh[s] = g.nh / 8; sw[s] = g.nw / 8;
      // the part of scale s's maps that shows the frame, stretched over the part of the output that shows it
      eh[s] = s == 0 ? (float)sh[0] : (float)sh[0] * ((float)g.rh / (float)g0.rh);
      ew[s] = s == 0 ? (float)sw[0] : (float)sw[0] * ((float)g.rw / (float)g0.rw);
    }
    int st = st_cpm2_resize_merge_maps(ctx_, srcs.data(), sh.data(), sw.data(), eh.data(), ew.data(), scales_, n, pose::kCatPad, chan_, 57,
                                       g0.nh, g0.nw, maps_.data());
    LOG_IF(FATAL, st != ST_OK) << "st_cpm2_resize_merge_maps: " << st_ctx_last_error(ctx_);
    cmaps_.assign(maps_.begin(), maps_.end());
    st = st_cpm2_nms(ctx_, cmaps_.data(), n, g0.nh, g0.nw, cpm2::kParts, mp, kNmsThreshold, joints_.data());
    LOG_IF(FATAL, st != ST_OK) << "st_cpm2_nms: " << st_ctx_last_error(ctx_);
    cjoints_.assign(joints_.begin(), joints_.end());
    st = st_cpm2_limb_scores(ctx_, cmaps_.data(), cjoints_.data(), n, g0.nh, g0.nw, mp, params_.inter_threshold, params_.inter_min_above, dev_scores);
    LOG_IF(FATAL, st != ST_OK) << "st_cpm2_limb_scores: " << st_ctx_last_error(ctx_);
    LOG_IF(FATAL, st_ctx_sync(ctx_) != ST_OK) << "st_ctx_sync: " << st_ctx_last_error(ctx_);
    scores_.resize(score_floats * n);
    peaks_.resize(peak_floats * n);
    stage_.download((u8*)scores_.data(), (const u8*)dev_scores, score_floats * sizeof(f32) * n);
    for (i32 i = 0; i < n; ++i) stage_.download((u8*)(peaks_.data() + peak_floats * i), (const u8*)joints_[i], peak_floats * sizeof(f32));

    std::vector<float> kp, row;
    for (i32 i = 0; i < n; ++i) {
      // joints in ZeroToOne coordinates: x_net / (width of the resized frame), i.e. assemble()'s rescaling with a 1 x 1 "frame"
      const int people = cpm2::assemble(scores_.data() + score_floats * i, peaks_.data() + peak_floats * i, 1, 1, g0.rh, g0.rw, params_, &kp);
      row.assign(people > 0 ? (size_t)people * (POSE_SCORES + TOTAL_KEYPOINTS * 3) : 1, 0.f);  // openpose_kernel.cpp:175-180
      for (int q = 0; q < people; ++q) {
        float* out = row.data() + (size_t)q * (POSE_SCORES + TOTAL_KEYPOINTS * 3);
        float sum = 0.f;
        for (int j = 0; j < POSE_KEYPOINTS; ++j) sum += kp[((size_t)q * POSE_KEYPOINTS + j) * 3 + 2];
        out[0] = sum / (float)POSE_KEYPOINTS;
        memcpy(out + POSE_SCORES, kp.data() + (size_t)q * POSE_KEYPOINTS * 3, sizeof(float) * POSE_KEYPOINTS * 3);
      }
      const size_t size = row.size() * sizeof(float);
      u8* buffer = new_buffer(device_, size);
      memcpy_buffer(buffer, device_, (const u8*)row.data(), CPU_DEVICE, size);
      insert_element(output_columns[0], buffer, size);
    }
  }

 private:
  struct Geom {
    float scale;
    int rh, rw, nh, nw;
  };
  DeviceHandle device_;
  int gpu_;
  DeviceStage stage_;
  Result valid_;
  OpenPoseArgs args_;
  int scales_ = 1;
  st_ctx* ctx_ = nullptr;
  pose::Net net_;
  cpm2::Params params_;
  int chan_[57];
  std::vector<Geom> geom_;
  std::vector<const uint8_t*> frames_;
  std::vector<float*> net_in_, maps_, joints_;
  std::vector<const float*> cin_, cmaps_, cjoints_;
  std::vector<float> scores_, peaks_;
};

using OpenPoseKernelHIP = OpenPoseKernelHIPImpl<false>;
using OpenPoseKernelHIPStaged = OpenPoseKernelHIPImpl<true>;

REGISTER_OP(OpenPose).frame_input("frame").output("pose", ColumnType::Bytes, "PoseList").protobuf_name("OpenPoseArgs");

REGISTER_KERNEL(OpenPose, OpenPoseKernelHIPStaged).device(DeviceType::CPU).num_devices(1).batch();
REGISTER_KERNEL(OpenPose, OpenPoseKernelHIP).device(DeviceType::GPU).num_devices(1).batch();
}
