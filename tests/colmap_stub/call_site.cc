// The reference's call site, verbatim in shape (src/controllers/distributed_mapper_controller.cpp:506-520;
// incremental_mapper_controller.cc:450-471), compiled against the replaced SiftFeatureMatcher.
#include "feature/matching.h"

namespace colmap {
bool MatchImagePairs(Database* database_ptr, const std::vector<std::pair<image_t, image_t>>& image_pairs, size_t num_images) {
  Database& database = *database_ptr;
  SiftMatchingOptions options;
  FeatureMatcherCache cache(5 * num_images, &database);
  SiftFeatureMatcher matcher(options, &database, &cache);
  if (!matcher.Setup()) return false;
  cache.Setup();
  matcher.Match(image_pairs);
  return true;
}
}  // namespace colmap
