// /root/reference/src/base/database.h:336-364 (pair ids) -- the matcher itself touches the database only through
// FeatureMatcherCache
#pragma once
#include <cstddef>
#include "util/types.h"
namespace colmap {
class Database {
 public:
  const static size_t kMaxNumImages;
  static image_pair_t ImagePairToPairId(const image_t image_id1, const image_t image_id2);
};
}  // namespace colmap
