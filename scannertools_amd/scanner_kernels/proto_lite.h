// Minimal proto3 wire-format reader for the argument messages of the imgproc ops
// (/root/reference/scannertools/scannertools_cpp/imgproc/scannertools_imgproc.proto).  With a real
// Scanner build the generated scannertools_imgproc.pb.h classes can be used instead; this reader
// only keeps the op library free of a protoc step.  Unknown fields are skipped.
#pragma once
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

namespace scanner {
namespace proto_lite {

struct Field {
  uint32_t number = 0;
  uint32_t wire = 0;        // 0 varint, 1 fixed64, 2 length-delimited, 5 fixed32
  uint64_t value = 0;       // varint / fixed payload
  std::string bytes;        // length-delimited payload
};

// Returns false on a malformed buffer.
inline bool parse(const uint8_t* p, size_t n, std::vector<Field>* out) {
  size_t i = 0;
  auto varint = [&](uint64_t* v) {
    *v = 0;
    for (int shift = 0; shift < 64 && i < n; shift += 7) {
      const uint8_t b = p[i++];
      *v |= (uint64_t)(b & 0x7f) << shift;
      if (!(b & 0x80)) return true;
    }
    return false;
  };
  while (i < n) {
    uint64_t key;
    if (!varint(&key)) return false;
    Field f;
    f.number = (uint32_t)(key >> 3);
    f.wire = (uint32_t)(key & 7);
    switch (f.wire) {
      case 0: if (!varint(&f.value)) return false; break;
      case 1: if (n - i < 8) return false; memcpy(&f.value, p + i, 8); i += 8; break;
      case 5: { if (n - i < 4) return false; uint32_t v; memcpy(&v, p + i, 4); f.value = v; i += 4; break; }
      case 2: {
        uint64_t len;
        if (!varint(&len) || len > (uint64_t)(n - i)) return false;  // no `i + len`: a 64-bit length must not wrap
        f.bytes.assign((const char*)p + i, (size_t)len);
        i += (size_t)len;
        break;
      }
      default: return false;
    }
    out->push_back(f);
  }
  return true;
}

inline float as_float(const Field& f) {
  uint32_t v = (uint32_t)f.value;
  float r;
  memcpy(&r, &v, 4);
  return r;
}

}  // namespace proto_lite
}  // namespace scanner
