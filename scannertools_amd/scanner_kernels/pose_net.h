// The CPM2 op's network (OpenPose COCO body model) as a sequence of C-ABI layer calls: what the Caffe forward pass
// behind CaffeKernel::execute computes for /root/reference/scannertools_caffe/scannertools_caffe_cpp/cpm2_kernel.cpp:8-52.
// C++ twin of scannertools_amd/pose_net.py (same layer list, same buffer layout, same call order, hence the same bits):
//   trunk   conv1_1 .. conv4_2 (VGG-19 head), conv4_3_CPM, conv4_4_CPM -> 128 feature channels at 1/8 resolution
//   stage 1 two branches (L1: 38 part-affinity planes, L2: 19 heat maps): 3 x (3x3, 128) + 1x1 512 + 1x1 out
//   stage 2..6 on concat(L1, L2, features) = 185 channels: 5 x (7x7, 128) + 1x1 128 + 1x1 out
// The layer list is the published pose_deploy_linevec.prototxt ([EXT]: the reference downloads prototxt and
// caffemodel at run time, openpose_kernel.cpp:35-78; neither is in its tree); weights come from the caffemodel the
// op's arguments name (CPM2Args.caffe_args.net_descriptor.model_weights_path).
// Layout: activations NHWC float32; the stage input lives in ONE 192-channel buffer [features 128 | L1 38 | L2 19 |
// 7 zero] the branches write their slices of (no concat pass); the first layer of stages 2..6 has its input
// channels permuted accordingly.
#pragma once
#include <hip/hip_runtime_api.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <exception>
#include <map>
#include <string>
#include <vector>

#include <sys/stat.h>

#include "proto_lite.h"
#include "scannertools_hip.h"

namespace scanner {
namespace pose {

constexpr int kPaf = 38, kHeat = 19, kFeat = 128, kCat = kPaf + kHeat + kFeat, kCatPad = 192;
constexpr int kOffFeat = 0, kOffPaf = kFeat, kOffHeat = kFeat + kPaf;

struct LayerSpec {
  std::string name;  // name in the prototxt / caffemodel
  int cin, cout, k, relu;
};

inline std::vector<LayerSpec> trunk_layers() {
  return {{"conv1_1", 3, 64, 3, 1},   {"conv1_2", 64, 64, 3, 1},      {"conv2_1", 64, 128, 3, 1},     {"conv2_2", 128, 128, 3, 1},
          {"conv3_1", 128, 256, 3, 1}, {"conv3_2", 256, 256, 3, 1},    {"conv3_3", 256, 256, 3, 1},    {"conv3_4", 256, 256, 3, 1},
          {"conv4_1", 256, 512, 3, 1}, {"conv4_2", 512, 512, 3, 1},    {"conv4_3_CPM", 512, 256, 3, 1}, {"conv4_4_CPM", 256, 128, 3, 1}};
}
// pooling follows these trunk layers (index into trunk_layers())
inline bool pool_after(int i) { return i == 1 || i == 3 || i == 7; }

inline std::vector<LayerSpec> branch_layers(int stage, int branch /*1 | 2*/) {
  const int out = branch == 1 ? kPaf : kHeat;
  std::vector<LayerSpec> l;
  char buf[64];
  if (stage == 1) {
    const int ci[5] = {128, 128, 128, 128, 512}, co[5] = {128, 128, 128, 512, out}, k[5] = {3, 3, 3, 1, 1};
    for (int i = 0; i < 5; ++i) {
      snprintf(buf, sizeof buf, "conv5_%d_CPM_L%d", i + 1, branch);
      l.push_back({buf, ci[i], co[i], k[i], i < 4});
    }
  } else {
    for (int i = 0; i < 7; ++i) {
      snprintf(buf, sizeof buf, "Mconv%d_stage%d_L%d", i + 1, stage, branch);
      l.push_back({buf, i == 0 ? kCat : 128, i == 6 ? out : 128, i < 5 ? 7 : 1, i < 6});
    }
  }
  return l;
}

// ---- caffemodel ------------------------------------------------------------------------------------------
// NetParameter wire format ([EXT] caffe.proto): layer = 100 (LayerParameter: name = 1, blobs = 7) or the V1
// `layers` = 2 (name = 4, blobs = 6); BlobProto: data = 5 (packed float).  Only the float payloads are needed:
// the shapes are the architecture's.
struct Blobs {
  std::vector<float> w, b;
};

// Whole file into memory; false for anything that is not a readable regular file of a plausible size (a directory
// opens with fopen() and reports LONG_MAX from ftell()).
inline bool read_file(const std::string& path, std::string* out) {
  struct stat sb;
  if (stat(path.c_str(), &sb) != 0 || !S_ISREG(sb.st_mode)) return false;
  constexpr long kMaxModelBytes = 1L << 32;  // the COCO body model is 209 MB
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) return false;
  bool ok = fseek(f, 0, SEEK_END) == 0;
  const long n = ok ? ftell(f) : -1;
  ok = ok && n >= 0 && n <= kMaxModelBytes && fseek(f, 0, SEEK_SET) == 0;
  if (ok) {
    out->resize((size_t)n);
    ok = n == 0 || fread(&(*out)[0], 1, out->size(), f) == out->size();
  }
  fclose(f);
  return ok;
}

inline bool blob_floats(const std::string& blob, std::vector<float>* out) {
  std::vector<proto_lite::Field> fs;
  if (!proto_lite::parse((const uint8_t*)blob.data(), blob.size(), &fs)) return false;
  out->clear();
  for (auto& f : fs) {
    if (f.number != 5) continue;
    if (f.wire == 2) {
      const size_t n = f.bytes.size() / 4, at = out->size();
      out->resize(at + n);
      memcpy(out->data() + at, f.bytes.data(), n * 4);
    } else if (f.wire == 5) {
      out->push_back(proto_lite::as_float(f));
    }
  }
  return true;
}

inline bool read_caffemodel_impl(const std::string& path, std::map<std::string, Blobs>* out, std::string* err) {
  std::string buf;
  if (!read_file(path, &buf)) { *err = "cannot read " + path; return false; }
  std::vector<proto_lite::Field> top;
  if (!proto_lite::parse((const uint8_t*)buf.data(), buf.size(), &top)) { *err = path + " is not a serialized NetParameter"; return false; }
  for (auto& f : top) {
    if (f.wire != 2 || (f.number != 100 && f.number != 2)) continue;
    const uint32_t name_field = f.number == 100 ? 1 : 4, blob_field = f.number == 100 ? 7 : 6;
    std::vector<proto_lite::Field> lf;
    if (!proto_lite::parse((const uint8_t*)f.bytes.data(), f.bytes.size(), &lf)) { *err = "malformed layer in " + path; return false; }
    std::string name;
    std::vector<const std::string*> blobs;
    for (auto& g : lf) {
      if (g.number == name_field && g.wire == 2) name = g.bytes;
      else if (g.number == blob_field && g.wire == 2) blobs.push_back(&g.bytes);
    }
    if (name.empty() || blobs.size() < 2) continue;
    Blobs bl;
    if (!blob_floats(*blobs[0], &bl.w) || !blob_floats(*blobs[1], &bl.b)) { *err = "malformed blob in layer " + name; return false; }
    (*out)[name] = std::move(bl);
  }
  return true;
}

// The file is untrusted input and these readers sit behind extern "C" entry points and kernel constructors: nothing
// may leave them as an exception (std::bad_alloc / std::length_error on a hostile length field would otherwise
// cross the C ABI and end the host process instead of becoming a validate() error).
inline bool read_caffemodel(const std::string& path, std::map<std::string, Blobs>* out, std::string* err) {
  try {
    return read_caffemodel_impl(path, out, err);
  } catch (const std::exception& e) {
    *err = "cannot parse " + path + ": " + e.what();
  } catch (...) {
    *err = "cannot parse " + path;
  }
  out->clear();
  return false;
}

inline std::vector<LayerSpec> all_layers() {
  std::vector<LayerSpec> all = trunk_layers();
  for (int st = 1; st <= 6; ++st)
    for (int br = 1; br <= 2; ++br)
      for (auto& l : branch_layers(st, br)) all.push_back(l);
  return all;
}

// ---- prototxt ----------------------------------------------------------------------------------------------
// The deploy description the reference hands Caffe next to the weights (CaffeArgs.net_descriptor.model_path; OpenPose reads
// <model_directory>/pose/coco/pose_deploy_linevec.prototxt).  Protobuf text format, read far enough to list the
// convolutions: layer { name: ".." type: "Convolution" convolution_param { num_output: N kernel_size: K } }.  The kernels
// implement ONE architecture; what the file is used for is (i) refusing a description of another network and (ii) the
// layer NAMES the weights are looked up by (scannertools_amd/pose_net.py: names_from_prototxt does the same check with
// the channel counts walked through the blobs).
struct ProtoLayer {
  std::string name, type;
  std::vector<std::string> bottoms, tops;
  int cout = 0, k = 1;
  bool conv() const { return type == "Convolution" || type == "CONVOLUTION"; }
  bool relu() const { return type == "ReLU" || type == "RELU"; }
  bool pool() const { return type == "Pooling" || type == "POOLING"; }
  bool concat() const { return type == "Concat" || type == "CONCAT"; }
};

// Every layer of the description in file order (name, type, blobs, convolution parameters) and the net-level `input:`.
inline bool prototxt_layers_impl(const std::string& text, std::vector<ProtoLayer>* out, std::string* net_input, std::string* err) {
  // tokens: identifiers / numbers, quoted strings, '{', '}', ':'; '#' starts a comment
  std::vector<std::string> tok;
  for (size_t i = 0; i < text.size();) {
    const char c = text[i];
    if (c == '#') { while (i < text.size() && text[i] != '\n') ++i; continue; }
    if (c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == ',' || c == ';') { ++i; continue; }
    if (c == '{' || c == '}' || c == ':') { tok.push_back(std::string(1, c)); ++i; continue; }
    if (c == '"' || c == '\'') {
      size_t j = i + 1;
      while (j < text.size() && text[j] != c) j += text[j] == '\\' ? 2 : 1;
      if (j >= text.size()) { *err = "unterminated string in the prototxt"; return false; }
      tok.push_back("\"" + text.substr(i + 1, j - i - 1));  // strings carry a leading quote mark
      i = j + 1;
      continue;
    }
    size_t j = i;
    while (j < text.size() && !strchr(" \t\n\r{}:#\"',;", text[j])) ++j;
    tok.push_back(text.substr(i, j - i));
    i = j;
  }
  // walk the nesting: path of open message fields; remember the fields of the current top-level layer
  std::vector<std::string> path;
  ProtoLayer cur;
  for (size_t i = 0; i < tok.size(); ++i) {
    const std::string& t = tok[i];
    if (t == "}") {
      if (path.empty()) { *err = "unbalanced '}' in the prototxt"; return false; }
      const bool layer_end = path.size() == 1 && (path[0] == "layer" || path[0] == "layers");
      path.pop_back();
      if (layer_end) {
        out->push_back(cur);
        cur = ProtoLayer();
      }
      continue;
    }
    if (t == "{" || t == ":") { *err = "unexpected '" + t + "' in the prototxt"; return false; }
    // a field name
    size_t j = i + 1;
    if (j < tok.size() && tok[j] == ":") ++j;
    if (j >= tok.size()) { *err = "field " + t + " has no value in the prototxt"; return false; }
    if (tok[j] == "{") {
      path.push_back(t);
      i = j;
      continue;
    }
    std::string v = tok[j];
    if (!v.empty() && v[0] == '"') v = v.substr(1);
    const bool in_layer = !path.empty() && (path[0] == "layer" || path[0] == "layers");
    if (path.empty() && t == "input" && net_input->empty()) *net_input = v;
    if (in_layer && path.size() == 1 && t == "name") cur.name = v;
    if (in_layer && path.size() == 1 && t == "type") cur.type = v;
    if (in_layer && path.size() == 1 && t == "bottom") cur.bottoms.push_back(v);
    if (in_layer && path.size() == 1 && t == "top") cur.tops.push_back(v);
    if (in_layer && path.size() == 2 && path[1] == "convolution_param") {
      if (t == "num_output") cur.cout = atoi(v.c_str());
      if (t == "kernel_size" || t == "kernel_h") cur.k = atoi(v.c_str());
    }
    i = j;
  }
  if (!path.empty()) { *err = "missing '}' in the prototxt"; return false; }
  return true;
}

// Names of the 92 convolutions of `path` in all_layers() order; false (with the first difference in *err) if the file is
// unreadable or describes another network.  The layers are identified by walking the blobs from the input, NOT by their
// position in the file: the published pose_deploy_linevec.prototxt ([EXT]) interleaves the two branches of a stage layer
// by layer (conv5_1_CPM_L1, conv5_1_CPM_L2, conv5_2_CPM_L1 ...) where all_layers() lists branch L1, then branch L2.
// Trunk = the single chain of convolutions / poolings from the input to the blob two convolutions read; a stage = the two
// chains behind that blob, told apart by their output counts (38 = L1, 19 = L2); the next stage reads the Concat of
// (L1, L2, features) in that order -- the order load() packs the weights for.
inline bool prototxt_layer_names(const std::string& path, std::vector<std::string>* names, std::string* err) try {
  std::string text;
  if (!read_file(path, &text)) { *err = "cannot read " + path; return false; }
  std::vector<ProtoLayer> layers;
  std::string cur;
  if (!prototxt_layers_impl(text, &layers, &cur, err)) { *err += " (" + path + ")"; return false; }
  const std::vector<LayerSpec> arch = all_layers();
  size_t nconv = 0;
  for (auto& l : layers) nconv += l.conv();
  if (nconv != arch.size()) {
    *err = "prototxt " + path + " describes " + std::to_string(nconv) + " convolutions, the kernels implement " + std::to_string(arch.size());
    return false;
  }
  std::map<std::string, std::string> alias, pool_of;   // a ReLU that is not in place renames its blob
  for (auto& l : layers)
    if (l.relu() && !l.bottoms.empty() && !l.tops.empty() && l.tops[0] != l.bottoms[0]) alias[l.tops[0]] = l.bottoms[0];
  auto blob = [&](std::string b) {
    for (int guard = 0; guard < 1000 && alias.count(b); ++guard) b = alias[b];
    return b;
  };
  std::map<std::string, std::vector<const ProtoLayer*>> readers;
  std::vector<const ProtoLayer*> concats;
  for (auto& l : layers) {
    if ((l.conv() || l.pool() || l.concat() || l.relu()) && (l.bottoms.empty() || l.tops.empty())) {
      *err = "prototxt layer " + l.name + " names no bottom / top blob (" + path + ")";
      return false;
    }
    if (l.conv()) readers[blob(l.bottoms[0])].push_back(&l);
    if (l.pool()) pool_of[blob(l.bottoms[0])] = l.tops[0];
    if (l.concat()) concats.push_back(&l);
    if (cur.empty() && l.type == "Input" && !l.tops.empty()) cur = l.tops[0];
  }
  auto matches = [&](const ProtoLayer& c, const LayerSpec& a) {
    if (c.cout == a.cout && c.k == a.k) return true;
    *err = "prototxt layer " + c.name + " is a " + std::to_string(c.k) + "x" + std::to_string(c.k) + " convolution with " + std::to_string(c.cout) +
           " outputs; the kernels implement " + a.name + " as " + std::to_string(a.k) + "x" + std::to_string(a.k) + " with " + std::to_string(a.cout);
    return false;
  };
  auto only_reader = [&](const std::string& b, const std::string& after) -> const ProtoLayer* {
    auto it = readers.find(b);
    const size_t n = it == readers.end() ? 0 : it->second.size();
    if (n == 1) return it->second[0];
    *err = "prototxt " + path + ": " + std::to_string(n) + " convolutions read blob " + b + " behind " + after + "; the kernels implement a single chain there";
    return nullptr;
  };
  names->clear();
  const std::vector<LayerSpec> trunk = trunk_layers();
  std::string last = "the input";
  for (size_t i = 0; i < trunk.size(); ++i) {
    const ProtoLayer* c = only_reader(cur, last);
    if (!c || !matches(*c, trunk[i])) return false;
    names->push_back(c->name);
    cur = c->tops[0];
    last = c->name;
    const bool pooled = pool_of.count(cur) && !readers.count(cur);
    if (pooled != pool_after((int)i)) {
      *err = "prototxt " + path + (pooled ? " pools behind " : " does not pool behind ") + last + "; the kernels " + (pooled ? "do not" : "do");
      return false;
    }
    if (pooled) cur = pool_of[cur];
  }
  const std::string feat = cur;
  std::string src = feat;
  for (int st = 1; st <= 6; ++st) {
    auto it = readers.find(src);
    const size_t nh = it == readers.end() ? 0 : it->second.size();
    if (nh != 2) {
      *err = "prototxt " + path + ": " + std::to_string(nh) + " convolutions read the input of stage " + std::to_string(st) + " (blob " + src + "); the kernels implement two branches";
      return false;
    }
    const size_t depth = branch_layers(st, 1).size();
    std::vector<const ProtoLayer*> chain[2];
    for (int b = 0; b < 2; ++b) {
      chain[b].push_back(it->second[b]);
      while (chain[b].size() < depth) {
        const ProtoLayer* nx = only_reader(chain[b].back()->tops[0], chain[b].back()->name);
        if (!nx) return false;
        chain[b].push_back(nx);
      }
    }
    const int o0 = chain[0].back()->cout, o1 = chain[1].back()->cout;
    if (!((o0 == kPaf && o1 == kHeat) || (o0 == kHeat && o1 == kPaf))) {
      *err = "prototxt " + path + ": the branches of stage " + std::to_string(st) + " end in " + std::to_string(o0) + " and " + std::to_string(o1) +
             " outputs; the kernels implement " + std::to_string(kPaf) + " (L1) and " + std::to_string(kHeat) + " (L2)";
      return false;
    }
    const std::vector<const ProtoLayer*>* by_branch[2] = {o0 == kPaf ? &chain[0] : &chain[1], o0 == kPaf ? &chain[1] : &chain[0]};
    for (int br = 1; br <= 2; ++br) {
      const std::vector<LayerSpec> spec = branch_layers(st, br);
      for (size_t i = 0; i < depth; ++i) {
        if (!matches(*(*by_branch[br - 1])[i], spec[i])) return false;
        names->push_back((*by_branch[br - 1])[i]->name);
      }
    }
    if (st < 6) {
      const std::vector<std::string> want = {by_branch[0]->back()->tops[0], by_branch[1]->back()->tops[0], feat};
      const ProtoLayer* cat = nullptr;
      for (auto* c : concats) {
        std::vector<std::string> got;
        for (auto& b : c->bottoms) got.push_back(blob(b));
        std::vector<std::string> a = got, w = want;
        std::sort(a.begin(), a.end());
        std::sort(w.begin(), w.end());
        if (a != w) continue;
        cat = c;
        if (got != want) {
          *err = "prototxt " + path + ": " + c->name + " concatenates (" + got[0] + ", " + got[1] + ", " + got[2] + "); the kernels implement the order (L1, L2, features)";
          return false;
        }
      }
      if (!cat) {
        *err = "prototxt " + path + ": no Concat of the two branches of stage " + std::to_string(st) + " and the features";
        return false;
      }
      src = cat->tops[0];
    }
  }
  return true;
} catch (const std::exception& e) {
  *err = "cannot parse " + path + ": " + e.what();
  return false;
}

// Does the file hold weights of the right sizes for every layer of the architecture?  (No GPU involved: what
// CPM2's validate() reports for a wrong or damaged model file, and a check a deployment can run up front.)
inline bool check_caffemodel(const std::string& path, int* matched, std::string* err) try {
  std::map<std::string, Blobs> blobs;
  if (matched) *matched = 0;
  if (!read_caffemodel(path, &blobs, err)) return false;
  for (auto& l : all_layers()) {
    auto it = blobs.find(l.name);
    if (it == blobs.end()) { *err = "caffemodel " + path + " has no weights for layer " + l.name; return false; }
    if (it->second.w.size() != (size_t)l.cout * l.cin * l.k * l.k || it->second.b.size() != (size_t)l.cout) {
      *err = "layer " + l.name + ": the file's blob sizes do not match the architecture";
      return false;
    }
    if (matched) ++*matched;
  }
  return true;
} catch (const std::exception& e) {
  *err = "cannot check " + path + ": " + e.what();
  return false;
}

// ---- the network on one GPU ---------------------------------------------------------------------------------
class Net {
 public:
  // SCANNERTOOLS_POSE_MATH=bf16x3 (read when the kernel instance is created) selects the split-bf16 arithmetic of
  // st_conv2d_nhwc_bf16x3 for every layer -- float32-grade accuracy on the bf16 matrix pipe, not bit-identical to the
  // default float32 instruction.  The reference's op arguments (CPM2Args / OpenPoseArgs) have no field for it.
  Net() {
    const char* m = getenv("SCANNERTOOLS_POSE_MATH");
    const std::string v = m ? m : "";
    bf16x3_ = v == "bf16x3";
    // anything but the two arithmetics is a configuration error, reported by load() (kernel validation): it must not
    // silently select float32
    if (!(v.empty() || v == "f32" || v == "bf16x3")) bad_math_ = v;
  }
  ~Net() { release(); }
  bool bf16x3() const { return bf16x3_; }

  // Loads the weights, packs them as [cout_pad][k][k][cin_pad] and uploads them to the current device.
  // prototxt (optional): the model's deploy description -- checked against the architecture, and the source of the names
  // the weights are looked up by (a model with other layer names and the same structure is usable)
  bool load(const std::string& caffemodel, std::string* err, const std::string& prototxt = std::string()) try {
    if (!bad_math_.empty()) {
      *err = "SCANNERTOOLS_POSE_MATH=" + bad_math_ + " is not an arithmetic of this build (f32, bf16x3)";
      return false;
    }
    release();  // a second load (or one after a failure part-way through) starts from nothing: no allocation is overwritten
    std::vector<std::string> file_names;
    if (!prototxt.empty() && !prototxt_layer_names(prototxt, &file_names, err)) return false;
    std::map<std::string, Blobs> blobs;
    if (!read_caffemodel(caffemodel, &blobs, err)) return false;
    size_t li = 0;
    for (auto& l : all_layers()) {
      const std::string& fname = file_names.empty() ? l.name : file_names[li];
      ++li;
      auto it = blobs.find(fname);
      if (it == blobs.end()) { *err = "caffemodel " + caffemodel + " has no weights for layer " + fname; return false; }
      const Blobs& bl = it->second;
      if (bl.w.size() != (size_t)l.cout * l.cin * l.k * l.k || bl.b.size() != (size_t)l.cout) {
        *err = "layer " + l.name + ": the file's blob sizes do not match the architecture";
        return false;
      }
      const int cip = l.cin == kCat ? kCatPad : (l.cin + 15) / 16 * 16, cop = (l.cout + 63) / 64 * 64;
      std::vector<float> wp((size_t)cop * l.k * l.k * cip, 0.f), bp(cop, 0.f);
      for (int o = 0; o < l.cout; ++o) {
        bp[o] = bl.b[o];
        for (int c = 0; c < l.cin; ++c) {
          // stage-input channel order of the prototxt is (L1 38, L2 19, features 128); the buffer holds the features first
          const int cb = l.cin == kCat ? (c < kPaf + kHeat ? kFeat + c : c - (kPaf + kHeat)) : c;
          for (int kk = 0; kk < l.k * l.k; ++kk)
            wp[((size_t)o * l.k * l.k + kk) * cip + cb] = bl.w[((size_t)o * l.cin + c) * l.k * l.k + kk];
        }
      }
      Packed& p = packed_[l.name];  // registered first, so that release() frees whatever part of it was allocated
      p.cin_pad = cip; p.cout_pad = cop;
      if (hipMalloc(&p.w, wp.size() * 4) != hipSuccess || hipMalloc(&p.b, bp.size() * 4) != hipSuccess ||
          hipMemcpy(p.w, wp.data(), wp.size() * 4, hipMemcpyHostToDevice) != hipSuccess ||
          hipMemcpy(p.b, bp.data(), bp.size() * 4, hipMemcpyHostToDevice) != hipSuccess) {
        *err = "out of device memory while uploading layer " + l.name;
        return false;
      }
    }
    return true;
  } catch (const std::exception& e) {
    *err = "cannot load " + caffemodel + ": " + e.what();
    return false;
  }

  // inputs: n device pointers to planar (3, H, W) float32 frames (CPM2Input's output); H, W multiples of 8.
  // Returns the stage buffer (n, H/8, W/8, 192) holding stage 6's outputs at channels kOffPaf.. / kOffHeat.. (device
  // memory owned by the net, valid until the next call), or nullptr with *err set.
  // slot: which pair of stage buffers receives the result -- a caller that runs several network scales per batch
  // (OpenPose's pose_num_scales) keeps every scale's output alive by giving each its own slot.
  const float* forward(st_ctx* ctx, const float* const* inputs, int n, int H, int W, std::string* err, int slot = 0) {
    if (n <= 0 || H % 8 || W % 8 || slot < 0 || slot >= kMaxSlots) { *err = "pose net: the network input must be a multiple of 8 in both dimensions"; return nullptr; }
    if (!reserve(n, H, W, slot, err)) return nullptr;
    float* const* cat_ = slots_[slot].cat;
    auto fail = [&](const char* what) { *err = std::string(what) + ": " + st_ctx_last_error(ctx); return (const float*)nullptr; };
    for (int i = 0; i < n; ++i)
      if (st_planar_to_nhwc_f32(ctx, inputs[i], 1, 3, H, W, a_ + (size_t)i * H * W * 16, 16) != ST_OK) return fail("st_planar_to_nhwc_f32");
    float *x = a_, *y = b_;
    int h = H, w = W, xc = 16;
    const auto trunk = trunk_layers();
    for (int i = 0; i < (int)trunk.size(); ++i) {
      const LayerSpec& l = trunk[i];
      Packed& p = packed_[l.name];
      if (i == (int)trunk.size() - 1) {
        // conv4_4_CPM: the features go straight into BOTH stage buffers (stage s reads buffer (s-1)&1)
        for (int t = 0; t < 2; ++t)
          if (conv(ctx, x, n, h, w, xc, xc, 0, p, l, cat_[t], kCatPad, kOffFeat) != ST_OK) return fail("st_conv2d_nhwc");
        break;
      }
      const int yc = (l.cout + 15) / 16 * 16;
      if (conv(ctx, x, n, h, w, xc, xc, 0, p, l, y, yc, 0) != ST_OK) return fail("st_conv2d_nhwc");
      std::swap(x, y);
      xc = yc;
      if (pool_after(i)) {
        if (st_maxpool2_nhwc_f32(ctx, x, n, h, w, xc, xc, y, xc) != ST_OK) return fail("st_maxpool2_nhwc_f32");
        std::swap(x, y);
        h /= 2; w /= 2;
      }
    }
    // The two branches of a stage have the same layer shapes: layer i of both goes out as ONE paired call (one launch where the
    // spatial-tile kernel runs: at a few frames per call a branch alone leaves CUs idle), each branch with its own temporaries
    // -- the same calls as scannertools_amd/pose_net.py: forward_raw, so the two drivers stay bit-identical.
    for (int st = 1; st <= 6; ++st) {
      const float* src = cat_[(st - 1) & 1];
      float* dst = cat_[st & 1];
      const std::vector<LayerSpec> layers[2] = {branch_layers(st, 1), branch_layers(st, 2)};
      const float* xin[2] = {src, src};
      int xs[2] = {kCatPad, kCatPad}, xoff[2] = {st == 1 ? kOffFeat : 0, st == 1 ? kOffFeat : 0};
      int xcin = st == 1 ? kFeat : kCatPad;
      for (int i = 0; i < (int)layers[0].size(); ++i) {
        st_conv_operands ops[2];
        for (int b = 0; b < 2; ++b) {
          const LayerSpec& l = layers[b][i];
          Packed& p = packed_[l.name];
          float* yout;
          int ys, yoff;
          if (i == (int)layers[b].size() - 1) { yout = dst; ys = kCatPad; yoff = b == 0 ? kOffPaf : kOffHeat; }
          else if (l.cout == 512) { yout = wide_[b]; ys = 512; yoff = 0; }
          else { yout = tmp_[b][i & 1]; ys = 128; yoff = 0; }
          const int st_w = prepare_weights(ctx, p, l);
          if (st_w != ST_OK) return fail("st_conv_pack_weights");
          ops[b] = st_conv_operands{xin[b], xs[b], xoff[b], bf16x3_ ? (const void*)p.w3 : (const void*)p.w, bf16x3_ ? nullptr : p.wt, p.b, l.cout,
                                    yout, ys, yoff};
          xin[b] = yout; xs[b] = ys; xoff[b] = 0;
        }
        const LayerSpec& l0 = layers[0][i];
        const int cop = packed_[l0.name].cout_pad;
        const int rc = bf16x3_ ? st_conv2d_nhwc_bf16x3_pair(ctx, n, h, w, xcin, l0.k, l0.k, cop, l0.relu, &ops[0], &ops[1])
                               : st_conv2d_nhwc_f32_pair(ctx, n, h, w, xcin, l0.k, l0.k, cop, l0.relu, &ops[0], &ops[1]);
        if (rc != ST_OK) return fail("st_conv2d_nhwc_pair");
        xcin = ops[0].y_stride;
      }
    }
    return cat_[6 & 1];
  }

 private:
  struct Packed {
    float* w = nullptr;
    float* b = nullptr;
    void* w3 = nullptr;  // the same weights as bf16 triples (bf16x3 arithmetic only; packed on first use)
    void* wt = nullptr;  // float32 weights in the spatial-tile kernel's operand order (float32 arithmetic, eligible layers; on first use)
    bool wt_tried = false;
    int cin_pad = 0, cout_pad = 0, k = 0;
  };

  // the layer's weights in the form(s) the selected arithmetic reads, packed on first use
  int prepare_weights(st_ctx* ctx, Packed& p, const LayerSpec& l) {
    if (!bf16x3_) {
      if (!p.wt_tried) {
        p.wt_tried = true;
        const long long nb = st_conv_f32_tile_bytes(p.cout_pad, l.k, l.k, p.cin_pad);
        if (nb > 0) {
          void* wt = nullptr;
          if (hipMalloc(&wt, (size_t)nb) != hipSuccess) return ST_ERR_HIP;
          const int st = st_conv_pack_weights_f32_tile(ctx, p.w, p.cout_pad, l.k, l.k, p.cin_pad, wt);
          if (st != ST_OK) {
            (void)hipFree(wt);
            return st;
          }
          p.wt = wt;
        }
      }
      return ST_OK;
    }
    if (!p.w3) {
      void* w3 = nullptr;
      const size_t w3_bytes = (size_t)st_conv_bf16x3_packed_bytes(p.cout_pad, l.k, l.k, p.cin_pad);
      if (hipMalloc(&w3, w3_bytes) != hipSuccess) return ST_ERR_HIP;
      const int st = st_conv_pack_weights_bf16x3_n(ctx, p.w, p.cout_pad, l.k, l.k, p.cin_pad, w3, w3_bytes);
      if (st != ST_OK) {  // an unpacked buffer must never be mistaken for packed weights by the next call
        (void)hipFree(w3);
        return st;
      }
      p.w3 = w3;
    }
    return ST_OK;
  }

  // one convolution layer in the selected arithmetic
  int conv(st_ctx* ctx, const float* x, int n, int h, int w, int cin, int xs, int xoff, Packed& p, const LayerSpec& l, float* y, int ys, int yoff) {
    const int st = prepare_weights(ctx, p, l);
    if (st != ST_OK) return st;
    if (!bf16x3_) return st_conv2d_nhwc_f32_tiled(ctx, x, n, h, w, cin, xs, xoff, p.w, p.wt, p.b, l.k, l.k, l.cout, p.cout_pad, l.relu, y, ys, yoff);
    return st_conv2d_nhwc_bf16x3(ctx, x, n, h, w, cin, xs, xoff, p.w3, p.b, l.k, l.k, l.cout, p.cout_pad, l.relu, y, ys, yoff);
  }

  bool reserve(int n, int H, int W, int slot, std::string* err) {
    const size_t big = (size_t)n * H * W * 64, small = (size_t)n * (H / 8) * (W / 8);
    auto alloc = [&](float** p, size_t floats) {
      if (*p) (void)hipFree(*p);
      *p = nullptr;
      return hipMalloc(p, floats * 4) == hipSuccess;
    };
    bool ok = true;
    // the largest trunk activation is conv1's (64 channels at full resolution); grow-only, shared by all slots
    if (big > cap_big_) { ok = ok && alloc(&a_, big) && alloc(&b_, big); cap_big_ = ok ? big : 0; }
    if (ok && small > cap_small_) {
      ok = true;
      for (int b = 0; b < 2 && ok; ++b) ok = alloc(&tmp_[b][0], small * 128) && alloc(&tmp_[b][1], small * 128) && alloc(&wide_[b], small * 512);
      cap_small_ = ok ? small : 0;
    }
    Slot& sl = slots_[slot];
    if (ok && small > sl.cap) {
      ok = alloc(&sl.cat[0], small * kCatPad) && alloc(&sl.cat[1], small * kCatPad);
      // the 7 pad channels of the stage buffers are read (against zero weights) and never written: zero them once
      ok = ok && hipMemset(sl.cat[0], 0, small * kCatPad * 4) == hipSuccess && hipMemset(sl.cat[1], 0, small * kCatPad * 4) == hipSuccess &&
           hipDeviceSynchronize() == hipSuccess;  // the layer calls run on the context's own (non-blocking) stream
      sl.cap = ok ? small : 0;
    }
    if (!ok) {
      *err = "pose net: out of device memory for a batch of " + std::to_string(n) + " frames";
      free_buffers();
      return false;
    }
    return true;
  }
  void free_buffers() {
    float** all[] = {&a_, &b_, &tmp_[0][0], &tmp_[0][1], &tmp_[1][0], &tmp_[1][1], &wide_[0], &wide_[1]};
    for (auto p : all) {
      if (*p) (void)hipFree(*p);
      *p = nullptr;
    }
    for (auto& sl : slots_) {
      for (auto& c : sl.cat) {
        if (c) (void)hipFree(c);
        c = nullptr;
      }
      sl.cap = 0;
    }
    cap_big_ = cap_small_ = 0;
  }
  void release() {
    free_buffers();
    for (auto& kv : packed_) {
      if (kv.second.w) (void)hipFree(kv.second.w);
      if (kv.second.b) (void)hipFree(kv.second.b);
      if (kv.second.w3) (void)hipFree(kv.second.w3);
      if (kv.second.wt) (void)hipFree(kv.second.wt);
    }
    packed_.clear();
  }

  std::map<std::string, Packed> packed_;
  bool bf16x3_ = false;
  std::string bad_math_;
  static constexpr int kMaxSlots = 8;
  struct Slot {
    float* cat[2] = {nullptr, nullptr};
    size_t cap = 0;
  };
  Slot slots_[kMaxSlots];
  float *a_ = nullptr, *b_ = nullptr, *tmp_[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}}, *wide_[2] = {nullptr, nullptr};   // temporaries per branch
  size_t cap_big_ = 0, cap_small_ = 0;
};

}  // namespace pose
}  // namespace scanner
