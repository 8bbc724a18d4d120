// Deterministic input generators equivalent to the reference's gtest helpers,
// so our round-trip tests run on the same byte/float streams the reference's
// tests do (libstdc++ mt19937 + distributions are bit-reproducible here).
//   generate_symbols : dietgpu/ans/ANSTest.cu:18-31  (mt19937(10), exponential(lambda))
//   generate_normals : dietgpu/float/FloatTest.cu:110-120 (mt19937(10 + num), normal)
// TEST INFRASTRUCTURE ONLY.
#include <algorithm>
#include <cstdint>
#include <random>

extern "C" {

void generate_symbols(uint8_t* out, int num, float lambda) {
  std::mt19937 gen(10);
  std::exponential_distribution<float> dist(lambda);
  for (int i = 0; i < num; ++i) {
    float sample = std::min(dist(gen), 1.0f);
    out[i] = (uint8_t)(sample * 256.0);
  }
}

void generate_normals(float* out, int num) {
  std::mt19937 gen(10 + num);
  std::normal_distribution<float> dist;
  for (int i = 0; i < num; ++i) out[i] = dist(gen);
}

}
