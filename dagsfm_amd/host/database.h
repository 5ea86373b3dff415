// database.h -- COLMAP/DAGSfM database.db access for the matching path (system sqlite3).
// Mirrors the parts of /root/reference/src/base/database.{h,cc} this path touches: schema
// (:1165-1262), pair ids (database.h:336-364), blob formats (:90-110), matches /
// two_view_geometries rows incl. DAGSfM's use of the F column for qvec and the E column for tvec
// (:681-751, :493-533).
#ifndef DAGSFM_AMD_HOST_DATABASE_H_
#define DAGSFM_AMD_HOST_DATABASE_H_

#include <exception>
#include <string>
#include <vector>

#include "types.h"

struct sqlite3;
struct sqlite3_stmt;

namespace dagsfm_amd {
struct StmtCache;
}

namespace dagsfm_amd {

class Database {
 public:
  const static int kSchemaVersion = 1;
  const static size_t kMaxNumImages = static_cast<size_t>(std::numeric_limits<int32_t>::max());

  Database();
  explicit Database(const std::string& path);
  ~Database();
  void Open(const std::string& path);
  void Close();

  static image_pair_t ImagePairToPairId(image_t image_id1, image_t image_id2);
  static void PairIdToImagePair(image_pair_t pair_id, image_t* image_id1, image_t* image_id2);
  static bool SwapImagePair(image_t image_id1, image_t image_id2);

  bool ExistsMatches(image_t image_id1, image_t image_id2) const;
  bool ExistsInlierMatches(image_t image_id1, image_t image_id2) const;
  // all pair ids of `matches` (inliers == false) or `two_view_geometries` (true), one query
  std::vector<image_pair_t> ReadPairIds(bool inliers) const;
  size_t NumMatchedImagePairs() const;
  size_t NumVerifiedImagePairs() const;

  std::vector<Camera> ReadAllCameras() const;
  std::vector<Image> ReadAllImages() const;
  FeatureKeypoints ReadKeypoints(image_t image_id) const;
  FeatureDescriptors ReadDescriptors(image_t image_id) const;
  FeatureMatches ReadMatches(image_t image_id1, image_t image_id2) const;
  TwoViewGeometry ReadTwoViewGeometry(image_t image_id1, image_t image_id2) const;

  camera_t WriteCamera(const Camera& camera) const;
  image_t WriteImage(const Image& image) const;
  void WriteKeypoints(image_t image_id, const FeatureKeypoints& keypoints) const;
  void WriteDescriptors(image_t image_id, const FeatureDescriptors& descriptors) const;
  void WriteMatches(image_t image_id1, image_t image_id2, const FeatureMatches& matches) const;
  void WriteTwoViewGeometry(image_t image_id1, image_t image_id2, const TwoViewGeometry& two_view_geometry) const;
  void DeleteMatches(image_t image_id1, image_t image_id2) const;
  void DeleteInlierMatches(image_t image_id1, image_t image_id2) const;

  void BeginTransaction() const;
  void EndTransaction() const;
  void RollbackTransaction() const;
  bool InTransaction() const;  // sqlite3_get_autocommit() == 0
  // Bulk-load setting for a run that only APPENDS rows (extension; measured in tools/sqlite_ceiling.py): the rollback journal
  // in memory instead of the write-ahead log, so that every page of the new rows is written once instead of twice (WAL +
  // checkpoint).  ROLLBACK keeps working; a process that dies inside the transaction can leave the file damaged where WAL
  // would not (synchronous=OFF, the reference's own setting, already gives up the power-loss case).  false restores WAL,
  // the mode the reference's Database::Open sets (database.cc:267-276).  No other connection may be open.
  void SetBulkLoadJournal(bool in_memory) const;

 private:
  void CreateTables() const;
  void Exec(const char* sql) const;
  sqlite3* database_ = nullptr;
  StmtCache* stmts_ = nullptr;  // prepared statements of the per-pair calls, compiled once
};

// RAII transaction like DatabaseTransaction, database.h:306-318
class DatabaseTransaction {
 public:
  explicit DatabaseTransaction(Database* database) : database_(database), exceptions_(std::uncaught_exceptions()) {
    database_->BeginTransaction();
  }
  // Leaving the scope normally commits, like the reference's destructor (database.h:306-318).  Leaving it because
  // an exception unwinds ROLLS BACK: the reference aborts the process where this code throws, so it never commits
  // half a block; neither does this.  END / ROLLBACK failing inside a destructor is swallowed (errors surface through
  // Commit()).
  ~DatabaseTransaction() noexcept {
    if (!done_) {
      try {
        if (std::uncaught_exceptions() > exceptions_)
          database_->RollbackTransaction();
        else
          database_->EndTransaction();
      } catch (...) {
      }
    }
  }
  void Commit() {
    done_ = true;
    try {
      database_->EndTransaction();
    } catch (...) {
      // a COMMIT that failed (SQLITE_BUSY, SQLITE_FULL) leaves the transaction open: close it without its rows, so that
      // the next BeginTransaction does not fail with "cannot start a transaction within a transaction"
      try {
        database_->RollbackTransaction();
      } catch (...) {
      }
      throw;
    }
  }
  void Rollback() {
    done_ = true;
    database_->RollbackTransaction();
  }

 private:
  Database* database_;
  int exceptions_;
  bool done_ = false;
};

}  // namespace dagsfm_amd
#endif
