// Op registration subset (scanner/api/op.h).  Reference uses:
//   REGISTER_OP(Histogram).frame_input("frame").output("histogram", ColumnType::Bytes, "Histogram");
//   REGISTER_OP(OpticalFlow).frame_input("frame").frame_output("flow").stencil({0, 1});
// (histogram_kernel_cpu.cpp:52, optical_flow_kernel_cpu.cpp:51-54); other builder methods that
// appear in the tree are accepted too.
#pragma once
#include "scanner/util/common.h"

namespace scanner {

enum class ColumnType { Bytes = 0, Video = 1 };

struct OpColumn {
  std::string name;
  ColumnType type;
  std::string type_name;
};

struct OpRegistration {
  std::string name;
  std::vector<OpColumn> inputs, outputs;
  std::vector<i32> stencil;  // empty => {0}
  bool variadic_inputs = false;
  bool unbounded_state = false;
  i32 bounded_state = -1;
  std::string protobuf_name, stream_protobuf_name;
};

class OpBuilder {
 public:
  explicit OpBuilder(const std::string& name) { reg_.name = name; }
  ~OpBuilder();  // commits the registration
  OpBuilder& variadic_inputs() { reg_.variadic_inputs = true; return *this; }
  OpBuilder& input(const std::string& n, ColumnType t = ColumnType::Bytes) { reg_.inputs.push_back({n, t, ""}); return *this; }
  OpBuilder& frame_input(const std::string& n) { reg_.inputs.push_back({n, ColumnType::Video, ""}); return *this; }
  OpBuilder& output(const std::string& n, ColumnType t = ColumnType::Bytes, const std::string& type_name = "") {
    reg_.outputs.push_back({n, t, type_name});
    return *this;
  }
  OpBuilder& frame_output(const std::string& n) { reg_.outputs.push_back({n, ColumnType::Video, ""}); return *this; }
  OpBuilder& stencil(const std::vector<i32>& s) { reg_.stencil = s; return *this; }
  OpBuilder& bounded_state(i32 warmup = 0) { reg_.bounded_state = warmup; return *this; }
  OpBuilder& unbounded_state() { reg_.unbounded_state = true; return *this; }
  OpBuilder& protobuf_name(const std::string& n) { reg_.protobuf_name = n; return *this; }
  OpBuilder& stream_protobuf_name(const std::string& n) { reg_.stream_protobuf_name = n; return *this; }
  OpBuilder& per_element_output() { return *this; }

 private:
  OpRegistration reg_;
};

#define ST_SHIM_OPCAT2(a, b) a##b
#define ST_SHIM_OPCAT(a, b) ST_SHIM_OPCAT2(a, b)
#define REGISTER_OP(name__) \
  static ::scanner::OpBuilder ST_SHIM_OPCAT(op_registration_, __COUNTER__) __attribute__((unused)) = ::scanner::OpBuilder(#name__)

}  // namespace scanner
