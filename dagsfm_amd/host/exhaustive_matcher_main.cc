// exhaustive_matcher_main.cc -- `colmap exhaustive_matcher --database_path X` equivalent
// (/root/reference/src/exe/colmap.cc:305-329) on the MI355X path.
#include <cstdlib>
#include <iostream>
#include <string>

extern "C" int dsm_host_exhaustive_matcher_ex4(const char* database_path, int block_size, int use_prior_defaults,
                                               unsigned random_seed, double max_ratio, double max_distance, int cross_check,
                                               int min_num_inliers, int guided_matching, int multiple_models,
                                               const char* gpu_index, int async_write_back, unsigned flags, int match_slice_pairs);

int main(int argc, char** argv) {
  std::string db;
  int block = 1000;  // the reference's default of 50 bounds its host cache; here a large block amortises the per-call costs
  unsigned seed = 0;
  int guided = 0, multiple = 0, timing = 0, bulk = 0, overlap = 1, slice = -1, on_device = 0;  // slice < 0: SiftMatchingOptions' default
  std::string gpu_index = "-1";  // all visible devices
  // < 0: SiftMatchingOptions' default (on since round 5).  This executable's own switch for the tests / tools (the libraries
  // read no environment): DSM_ASYNC_WRITE_BACK=0 / =1 is --SiftMatching.async_write_back 0 / 1
  int async_write_back = -1;
  if (const char* e = std::getenv("DSM_ASYNC_WRITE_BACK")) async_write_back = std::atoi(e) != 0 ? 1 : 0;
  for (int i = 1; i + 1 < argc; i += 2) {
    const std::string k = argv[i];
    if (k == "--database_path") db = argv[i + 1];
    else if (k == "--ExhaustiveMatching.block_size") block = std::atoi(argv[i + 1]);
    else if (k == "--random_seed") seed = static_cast<unsigned>(std::strtoul(argv[i + 1], nullptr, 10));
    else if (k == "--SiftMatching.guided_matching") guided = std::atoi(argv[i + 1]);
    else if (k == "--SiftMatching.multiple_models") multiple = std::atoi(argv[i + 1]);
    else if (k == "--SiftMatching.gpu_index") gpu_index = argv[i + 1];
    else if (k == "--SiftMatching.async_write_back") async_write_back = std::atoi(argv[i + 1]);
    else if (k == "--timing") timing = std::atoi(argv[i + 1]);
    else if (k == "--SiftMatching.bulk_load_journal") bulk = std::atoi(argv[i + 1]);
    else if (k == "--SiftMatching.match_slice_pairs") slice = std::atoi(argv[i + 1]);
    else if (k == "--ExhaustiveMatching.overlap_setup") overlap = std::atoi(argv[i + 1]);
    else if (k == "--SiftMatching.assemble_on_device") on_device = std::atoi(argv[i + 1]);
  }
  if (db.empty()) {
    std::cerr << "usage: dsm_exhaustive_matcher --database_path database.db [--ExhaustiveMatching.block_size 1000] [--random_seed 0]"
                 " [--SiftMatching.guided_matching 0] [--SiftMatching.multiple_models 0] [--SiftMatching.gpu_index -1]"
                 " [--SiftMatching.async_write_back 1] [--SiftMatching.bulk_load_journal 0] [--SiftMatching.match_slice_pairs 32768]"
                 " [--ExhaustiveMatching.overlap_setup 1] [--SiftMatching.assemble_on_device 0] [--timing 0]\n";
    return 64;
  }
  return dsm_host_exhaustive_matcher_ex4(db.c_str(), block, 1, seed, 0, 0, 1, 15, guided, multiple, gpu_index.c_str(), async_write_back,
                                         (timing ? 1u : 0u) | (bulk ? 2u : 0u) | (overlap ? 0u : 4u) | (on_device ? 8u : 0u), slice);
}
