// Host side of the CPM2Output op: from the scored candidate pairs of every limb to people.
//
// Follows CPM2OutputKernel::connect_limbs_coco
// (/root/reference/scannertools_caffe/scannertools_caffe_cpp/cpm2_output_kernel_cpu.cpp:362-689) step by
// step -- the single-part special cases, the greedy one-to-one matching of a limb's candidates by
// descending score, the merging of limbs into people through shared joints, the final filter and the
// rescaling of the joints to the original frame -- with the same accumulator types (scores summed in
// double, joint coordinates in float).  Structure differs: a person is a small struct, not a row of
// doubles, and the dense part (ten part-affinity samples per candidate pair) is a separate function so
// that it can run where the heat maps are: limb_scores_host() for host-resident columns (what the
// reference's CPU kernel does), st_cpm2_limb_scores() (include/scannertools_hip.h) on the GPU.
// One deliberate difference: the reference orders candidates with std::sort, whose treatment of equal
// scores is unspecified; std::stable_sort is used here so that results are reproducible.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

namespace cpm2 {

constexpr int kParts = 18, kLimbs = 19, kMaps = 57;
// COCO_18 (cpm2_output_kernel_cpu.cpp:84-88)
static const int kLimbSeq[38] = {1, 2, 1, 5, 2, 3, 3, 4, 5, 6, 6, 7, 1, 8, 8, 9, 9, 10, 1, 11,
                                 11, 12, 12, 13, 1, 0, 0, 14, 14, 16, 0, 15, 15, 17, 2, 16, 5, 17};
static const int kMapIdx[38] = {31, 32, 39, 40, 33, 34, 35, 36, 41, 42, 43, 44, 19, 20, 21, 22, 23, 24, 25,
                                26, 27, 28, 29, 30, 47, 48, 49, 50, 53, 54, 51, 52, 55, 56, 37, 38, 45, 46};

struct Params {  // cpm2_output_kernel_cpu.cpp:786-801 (COCO values)
  int max_people = 96;
  int max_peaks = 64;
  int min_subset_cnt = 3;
  float min_subset_score = 0.4f;
  float inter_threshold = 0.050f;
  int inter_min_above = 9;
};

// scores[k][i-1][j-1] = mean affinity of candidate pair (i, j) of limb k, or -1 (:424-487)
inline void limb_scores_host(const float* heatmap, const float* peaks, int h, int w, const Params& p, float* scores) {
  const size_t plane = (size_t)h * w;
  const int peaks_offset = 3 * (p.max_peaks + 1);
  const int num_inter = 10;
  for (int k = 0; k < kLimbs; ++k) {
    const float* map_x = heatmap + (size_t)kMapIdx[2 * k] * plane;
    const float* map_y = heatmap + (size_t)kMapIdx[2 * k + 1] * plane;
    const float* candA = peaks + kLimbSeq[2 * k] * peaks_offset;
    const float* candB = peaks + kLimbSeq[2 * k + 1] * peaks_offset;
    const int nA = std::min(std::max((int)candA[0], 0), p.max_peaks), nB = std::min(std::max((int)candB[0], 0), p.max_peaks);
    float* out = scores + (size_t)k * p.max_peaks * p.max_peaks;
    for (int i = 1; i <= p.max_peaks; ++i)
      for (int j = 1; j <= p.max_peaks; ++j) {
        float res = -1.f;
        if (i <= nA && j <= nB) {
          const float s_x = candA[i * 3], s_y = candA[i * 3 + 1];
          const float d_x = candB[j * 3] - candA[i * 3], d_y = candB[j * 3 + 1] - candA[i * 3 + 1];
          const float norm_vec = std::sqrt(d_x * d_x + d_y * d_y);
          if (!(norm_vec < 1e-6)) {
            const float vec_x = d_x / norm_vec, vec_y = d_y / norm_vec;
            float sum = 0;
            int count = 0;
            for (int lm = 0; lm < num_inter; lm++) {
              int my = (int)std::round(s_y + lm * d_y / num_inter);
              int mx = (int)std::round(s_x + lm * d_x / num_inter);
              if (mx >= w) mx = w - 1;
              if (my >= h) my = h - 1;
              if (mx < 0) mx = 0;  // the reference aborts (CHECK_GE)
              if (my < 0) my = 0;
              const int idx = my * w + mx;
              const float score = vec_x * map_x[idx] + vec_y * map_y[idx];
              if (score > p.inter_threshold) { sum = sum + score; count++; }
            }
            if (count > p.inter_min_above) res = sum / count;
          }
        }
        out[(size_t)(i - 1) * p.max_peaks + (j - 1)] = res;
      }
  }
}

struct Person {
  int part[kParts];  // flat index of the joint's score in `peaks` (0 = joint absent)
  double score;
  int count;
};

// joints: people x 18 x (x, y, score); returns the number of people (:489-689)
inline int assemble(const float* scores, const float* peaks, int frame_h, int frame_w, int net_h, int net_w,
                    const Params& p, std::vector<float>* joints) {
  const int peaks_offset = 3 * (p.max_peaks + 1);
  std::vector<Person> people;
  auto single = [&](int part, int flat, float part_score) {
    Person q;
    memset(&q, 0, sizeof(q));
    q.part[part] = flat;
    q.count = 1;
    q.score = part_score;
    people.push_back(q);
  };
  struct Cand { int i, j; double score; };
  struct Conn { int a, b; double score; };
  for (int k = 0; k < kLimbs; ++k) {
    const int pa = kLimbSeq[2 * k], pb = kLimbSeq[2 * k + 1];
    const float* candA = peaks + pa * peaks_offset;
    const float* candB = peaks + pb * peaks_offset;
    const int nA = std::min(std::max((int)candA[0], 0), p.max_peaks), nB = std::min(std::max((int)candB[0], 0), p.max_peaks);
    if (nA == 0 && nB == 0) continue;
    if (nA == 0 || nB == 0) {
      // only one end of the limb was detected: its candidates become one-joint people unless a person has them already
      const int part = nA == 0 ? pb : pa, cnt = nA == 0 ? nB : nA;
      const float* cand = nA == 0 ? candB : candA;
      for (int i = 1; i <= cnt; ++i) {
        const int flat = part * peaks_offset + i * 3 + 2;
        bool seen = false;
        for (auto& q : people) seen = seen || q.part[part] == flat;
        if (!seen) single(part, flat, cand[i * 3 + 2]);
      }
      continue;
    }
    std::vector<Cand> temp;
    const float* sc = scores + (size_t)k * p.max_peaks * p.max_peaks;
    for (int i = 1; i <= nA; ++i)
      for (int j = 1; j <= nB; ++j) {
        const float s = sc[(size_t)(i - 1) * p.max_peaks + (j - 1)];
        if (s >= 0.f) temp.push_back(Cand{i, j, (double)s});
      }
    std::stable_sort(temp.begin(), temp.end(), [](const Cand& l, const Cand& r) { return l.score > r.score; });
    const int num = std::min(nA, nB);
    std::vector<char> usedA(nA, 0), usedB(nB, 0);
    std::vector<Conn> conns;
    for (auto& c : temp) {
      if ((int)conns.size() == num) break;
      if (!usedA[c.i - 1] && !usedB[c.j - 1]) {
        conns.push_back(Conn{pa * peaks_offset + c.i * 3 + 2, pb * peaks_offset + c.j * 3 + 2, (double)(float)c.score});
        usedA[c.i - 1] = usedB[c.j - 1] = 1;
      }
    }
    for (auto& c : conns) {
      int hits = 0;
      if (k != 0) {
        for (auto& q : people)
          if (q.part[pa] == c.a) {
            q.part[pb] = c.b;
            ++hits;
            q.count += 1;
            q.score = q.score + peaks[c.b] + c.score;
          }
      }
      if (hits == 0) {
        Person q;
        memset(&q, 0, sizeof(q));
        q.part[pa] = c.a;
        q.part[pb] = c.b;
        q.count = 2;
        q.score = peaks[c.a] + peaks[c.b] + c.score;  // float + float, then + double, as in the reference
        people.push_back(q);
      }
    }
  }
  joints->clear();
  int cnt = 0;
  for (auto& q : people) {
    if (q.count >= p.min_subset_cnt && (q.score / q.count) > p.min_subset_score) {
      for (int j = 0; j < kParts; ++j) {
        const int idx = q.part[j];
        if (idx) {
          joints->push_back(peaks[idx - 2] * frame_w / (float)net_w);
          joints->push_back(peaks[idx - 1] * frame_h / (float)net_h);
          joints->push_back(peaks[idx]);
        } else {
          joints->push_back(0.f); joints->push_back(0.f); joints->push_back(0.f);
        }
      }
      if (++cnt == p.max_people) break;
    }
  }
  return cnt;
}

// serialize_proto_vector_of_vectors<scanner::Point> (scanner/util/serialize.h, [EXT]: not in the
// reference tree, reconstructed): u64 people; per person u64 joints; per joint i32 byte size + the
// proto3 encoding of Point{float x = 1; float y = 2; float score = 3;} (zero fields omitted).
inline void serialize_people(const std::vector<float>& joints, int people, std::vector<uint8_t>* out) {
  out->clear();
  auto put = [&](const void* p, size_t n) { out->insert(out->end(), (const uint8_t*)p, (const uint8_t*)p + n); };
  const uint64_t np = (uint64_t)people;
  put(&np, 8);
  for (int q = 0; q < people; ++q) {
    const uint64_t nj = kParts;
    put(&nj, 8);
    for (int j = 0; j < kParts; ++j) {
      uint8_t buf[15];
      int32_t n = 0;
      for (int f = 0; f < 3; ++f) {
        const float v = joints[((size_t)q * kParts + j) * 3 + f];
        uint32_t bits;
        memcpy(&bits, &v, 4);
        if (bits != 0) {  // proto3 omits +0.0 only; -0.0 and NaN are written
          buf[n++] = (uint8_t)(((f + 1) << 3) | 5);
          memcpy(buf + n, &v, 4);
          n += 4;
        }
      }
      put(&n, 4);
      put(buf, (size_t)n);
    }
  }
}

}  // namespace cpm2
