// ASan / UBSan fuzz of FlannIndex::Load + FindWordIds on mutated vocabulary files (bit flips, overwritten sizes, deletions,
// truncations inside the FLANN section):  g++ -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=all -std=c++17 -o /tmp/f \
//   tools/fuzz_flann_loader.cc dagsfm_amd/host/flann_index.cc -pthread && /tmp/f tests/golden/vocab_flann_*.bin
#include "../dagsfm_amd/host/flann_index.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>
using namespace dagsfm_amd;
int main(int argc, char** argv) {
  int total = 0, loaded = 0;
  for (int a = 1; a < argc; ++a) {
    FILE* f = fopen(argv[a], "rb");
    fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
    std::vector<uint8_t> src(n); if (fread(src.data(), 1, n, f) != (size_t)n) return 2; fclose(f);
    uint64_t rows; memcpy(&rows, src.data(), 8);
    const size_t begin = 16 + rows * 128;
    std::mt19937 rng(a * 7919);
    std::vector<uint8_t> q(64 * 128);
    for (auto& b : q) b = rng();
    for (int trial = 0; trial < 4000; ++trial) {
      std::vector<uint8_t> b = src;
      const int kind = trial % 5;
      // mutate only inside / around the FLANN section
      size_t span = std::min<size_t>(b.size() - begin, 40000);
      size_t pos = begin + rng() % span;
      if (kind == 0) b[pos] ^= 1u << (rng() % 8);
      else if (kind == 1) { uint64_t v = rng(); v = (v << 20) ^ rng(); memcpy(&b[pos], &v, std::min<size_t>(8, b.size() - pos)); }
      else if (kind == 2) b.erase(b.begin() + pos, b.begin() + std::min(b.size(), pos + 1 + rng() % 100));
      else if (kind == 3) b.resize(pos);
      else { for (int k = 0; k < 16; ++k) b[begin + rng() % span] = rng(); }
      FlannIndex ix;
      size_t at = begin;
      std::vector<uint8_t> words(b.begin() + 16, b.begin() + std::min(b.size(), begin));
      if (words.size() != rows * 128) continue;
      ++total;
      if (ix.Load(b.data(), b.size(), &at, words.data(), (uint32_t)rows)) {
        ++loaded;
        std::vector<int32_t> ids(64 * 5);
        std::vector<float> d(64 * 5);
        ix.FindWordIds(q.data(), 64, 5, 32, 2, ids.data(), d.data());
        for (int32_t v : ids) if (!(v == FlannIndex::kInvalidWordId || (v >= 0 && (uint64_t)v < rows))) { printf("bad id %d\n", v); return 1; }
      }
    }
  }
  printf("mutated files %d, still loadable %d, no sanitizer report\n", total, loaded);
  return 0;
}
