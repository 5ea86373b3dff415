// database.cc -- see database.h.  Errors throw std::runtime_error (the reference aborts through
// SQLITE3_CALL / glog FATAL, /root/reference/src/util/sqlite3_utils.h).
#include "database.h"

#include <unordered_map>

#include <sqlite3.h>

#include <cmath>
#include <cstring>
#include <stdexcept>

namespace dagsfm_amd {
// A prepared statement that is compiled once per connection and re-used (the reference prepares all of its
// statements in Database::PrepareSQLStatements, database.cc:1121-1131): the per-pair Exists / Write calls of
// SiftFeatureMatcher::Match otherwise spend more time in sqlite3_prepare_v2 than in the insert itself.
struct StmtCache {
  std::unordered_map<const char*, sqlite3_stmt*> map;  // keyed by the address of the SQL literal
};
namespace {

void Check(int rc, sqlite3* db, const char* what) {
  if (rc != SQLITE_OK && rc != SQLITE_DONE && rc != SQLITE_ROW)
    throw std::runtime_error(std::string("sqlite3 error in ") + what + ": " + (db ? sqlite3_errmsg(db) : "?"));
}

struct Stmt {
  sqlite3* db;
  sqlite3_stmt* s = nullptr;
  bool cached = false;
  Stmt(sqlite3* d, const char* sql, StmtCache* cache = nullptr) : db(d) {
    if (cache) {
      auto it = cache->map.find(sql);
      if (it != cache->map.end()) {
        s = it->second;
      } else {
        Check(sqlite3_prepare_v2(db, sql, -1, &s, nullptr), db, sql);
        cache->map.emplace(sql, s);
      }
      cached = true;
    } else {
      Check(sqlite3_prepare_v2(db, sql, -1, &s, nullptr), db, sql);
    }
  }
  ~Stmt() {
    if (cached) {
      sqlite3_reset(s);
      sqlite3_clear_bindings(s);
    } else {
      sqlite3_finalize(s);
    }
  }
  int Step() {
    const int rc = sqlite3_step(s);
    Check(rc, db, "step");
    return rc;
  }
};

// QuaternionRotatePoint through Eigen::Quaterniond::_transformVector (pose.cc:122-128)
void QuaternionRotatePoint(const double q[4], const double v[3], double out[3]) {
  double n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  double w = 1.0, x = q[1], y = q[2], z = q[3];  // NormalizeQuaternion, pose.cc:82-91
  if (n != 0) {
    w = q[0] / n;
    x = q[1] / n;
    y = q[2] / n;
    z = q[3] / n;
  }
  double uv[3] = {y * v[2] - z * v[1], z * v[0] - x * v[2], x * v[1] - y * v[0]};
  for (int i = 0; i < 3; ++i) uv[i] += uv[i];
  const double c[3] = {y * uv[2] - z * uv[1], z * uv[0] - x * uv[2], x * uv[1] - y * uv[0]};
  for (int i = 0; i < 3; ++i) out[i] = v[i] + w * uv[i] + c[i];
}

}  // namespace

bool SiftMatchingOptions::Check() const {
  return max_ratio > 0.0 && max_distance > 0.0 && max_error > 0.0 && min_num_trials >= 0 && max_num_trials > 0 &&
         min_num_trials <= max_num_trials && min_inlier_ratio >= 0 && min_inlier_ratio <= 1 && min_num_inliers >= 0;
}

void TwoViewGeometry::Invert() {
  auto transpose = [](double* M) {
    std::swap(M[1], M[3]);
    std::swap(M[2], M[6]);
    std::swap(M[5], M[7]);
  };
  transpose(F);
  transpose(E);
  {  // H = H.inverse() (cofactors, one 1/det)
    const double* M = H;
    auto cof = [&](int i, int j) {
      const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
      return M[i1 * 3 + j1] * M[i2 * 3 + j2] - M[i1 * 3 + j2] * M[i2 * 3 + j1];
    };
    const double c00 = cof(0, 0), c10 = cof(1, 0), c20 = cof(2, 0);
    const double invdet = 1.0 / (c00 * M[0] + c10 * M[3] + c20 * M[6]);
    const double R[9] = {c00 * invdet,       c10 * invdet,       c20 * invdet,
                         cof(0, 1) * invdet, cof(1, 1) * invdet, cof(2, 1) * invdet,
                         cof(0, 2) * invdet, cof(1, 2) * invdet, cof(2, 2) * invdet};
    std::memcpy(H, R, sizeof(R));
  }
  // InvertPose, pose.cc:192-196
  const double oq[4] = {qvec[0], qvec[1], qvec[2], qvec[3]};
  const double ot[3] = {tvec[0], tvec[1], tvec[2]};
  qvec[0] = oq[0];
  qvec[1] = -oq[1];
  qvec[2] = -oq[2];
  qvec[3] = -oq[3];
  double rt[3];
  QuaternionRotatePoint(qvec, ot, rt);
  for (int i = 0; i < 3; ++i) tvec[i] = -rt[i];
  for (auto& m : inlier_matches) std::swap(m.point2D_idx1, m.point2D_idx2);
}

Database::Database() {}
Database::Database(const std::string& path) { Open(path); }
Database::~Database() { Close(); }

void Database::Exec(const char* sql) const {
  char* err = nullptr;
  if (sqlite3_exec(database_, sql, nullptr, nullptr, &err) != SQLITE_OK) {
    const std::string msg = err ? err : "?";
    sqlite3_free(err);
    throw std::runtime_error(std::string("sqlite3 exec failed: ") + msg + " in " + sql);
  }
}

void Database::Open(const std::string& path) {
  Close();
  // SQLITE_OPEN_NOMUTEX as database.cc:253-259; pragmas as :267-276
  Check(sqlite3_open_v2(path.c_str(), &database_, SQLITE_OPEN_READWRITE | SQLITE_OPEN_CREATE | SQLITE_OPEN_NOMUTEX, nullptr),
        database_, "open");
  Exec("PRAGMA synchronous=OFF;");
  Exec("PRAGMA journal_mode=WAL;");
  Exec("PRAGMA temp_store=MEMORY;");
  Exec("PRAGMA foreign_keys=ON;");
  CreateTables();
  stmts_ = new StmtCache();
  Exec("PRAGMA user_version = 3600;");  // COLMAP_VERSION_NUMBER, /root/reference/CMakeLists.txt:37
}

void Database::Close() {
  if (stmts_) {
    for (auto& kv : stmts_->map) sqlite3_finalize(kv.second);
    delete stmts_;
    stmts_ = nullptr;
  }
  if (database_) sqlite3_close_v2(database_);
  database_ = nullptr;
}

void Database::CreateTables() const {
  Exec("CREATE TABLE IF NOT EXISTS cameras (camera_id INTEGER PRIMARY KEY AUTOINCREMENT NOT NULL, model INTEGER NOT NULL,"
       " width INTEGER NOT NULL, height INTEGER NOT NULL, params BLOB, prior_focal_length INTEGER NOT NULL);");
  Exec("CREATE TABLE IF NOT EXISTS images (image_id INTEGER PRIMARY KEY AUTOINCREMENT NOT NULL, name TEXT NOT NULL UNIQUE,"
       " camera_id INTEGER NOT NULL, prior_qw REAL, prior_qx REAL, prior_qy REAL, prior_qz REAL, prior_tx REAL,"
       " prior_ty REAL, prior_tz REAL, CONSTRAINT image_id_check CHECK(image_id >= 0 and image_id < 2147483647),"
       " FOREIGN KEY(camera_id) REFERENCES cameras(camera_id));"
       "CREATE UNIQUE INDEX IF NOT EXISTS index_name ON images(name);");
  Exec("CREATE TABLE IF NOT EXISTS keypoints (image_id INTEGER PRIMARY KEY NOT NULL, rows INTEGER NOT NULL, cols INTEGER NOT NULL,"
       " data BLOB, FOREIGN KEY(image_id) REFERENCES images(image_id) ON DELETE CASCADE);");
  Exec("CREATE TABLE IF NOT EXISTS descriptors (image_id INTEGER PRIMARY KEY NOT NULL, rows INTEGER NOT NULL, cols INTEGER NOT NULL,"
       " data BLOB, FOREIGN KEY(image_id) REFERENCES images(image_id) ON DELETE CASCADE);");
  Exec("CREATE TABLE IF NOT EXISTS matches (pair_id INTEGER PRIMARY KEY NOT NULL, rows INTEGER NOT NULL, cols INTEGER NOT NULL,"
       " data BLOB);");
  Exec("CREATE TABLE IF NOT EXISTS two_view_geometries (pair_id INTEGER PRIMARY KEY NOT NULL, rows INTEGER NOT NULL,"
       " cols INTEGER NOT NULL, data BLOB, config INTEGER NOT NULL, F BLOB, E BLOB, H BLOB);");
}

image_pair_t Database::ImagePairToPairId(image_t image_id1, image_t image_id2) {
  if (SwapImagePair(image_id1, image_id2)) return static_cast<image_pair_t>(kMaxNumImages) * image_id2 + image_id1;
  return static_cast<image_pair_t>(kMaxNumImages) * image_id1 + image_id2;
}
void Database::PairIdToImagePair(image_pair_t pair_id, image_t* image_id1, image_t* image_id2) {
  *image_id2 = static_cast<image_t>(pair_id % kMaxNumImages);
  *image_id1 = static_cast<image_t>((pair_id - *image_id2) / kMaxNumImages);
}
bool Database::SwapImagePair(image_t image_id1, image_t image_id2) { return image_id1 > image_id2; }

static bool ExistsRow(sqlite3* db, StmtCache* cache, const char* sql, image_pair_t pair_id) {
  Stmt st(db, sql, cache);
  sqlite3_bind_int64(st.s, 1, static_cast<sqlite3_int64>(pair_id));
  return st.Step() == SQLITE_ROW;
}
bool Database::ExistsMatches(image_t a, image_t b) const {
  static const char* const kSql = "SELECT 1 FROM matches WHERE pair_id = ?;";
  return ExistsRow(database_, stmts_, kSql, ImagePairToPairId(a, b));
}
bool Database::ExistsInlierMatches(image_t a, image_t b) const {
  static const char* const kSql = "SELECT 1 FROM two_view_geometries WHERE pair_id = ?;";
  return ExistsRow(database_, stmts_, kSql, ImagePairToPairId(a, b));
}
static size_t CountRows(sqlite3* db, const char* sql) {
  Stmt st(db, sql);
  st.Step();
  return static_cast<size_t>(sqlite3_column_int64(st.s, 0));
}
std::vector<image_pair_t> Database::ReadPairIds(bool inliers) const {
  Stmt st(database_, inliers ? "SELECT pair_id FROM two_view_geometries;" : "SELECT pair_id FROM matches;");
  std::vector<image_pair_t> ids;
  while (st.Step() == SQLITE_ROW) ids.push_back(static_cast<image_pair_t>(sqlite3_column_int64(st.s, 0)));
  return ids;
}
size_t Database::NumMatchedImagePairs() const { return CountRows(database_, "SELECT COUNT(*) FROM matches WHERE rows > 0;"); }
size_t Database::NumVerifiedImagePairs() const {
  return CountRows(database_, "SELECT COUNT(*) FROM two_view_geometries WHERE rows > 0;");
}

std::vector<Camera> Database::ReadAllCameras() const {
  std::vector<Camera> cams;
  Stmt st(database_, "SELECT camera_id, model, width, height, params, prior_focal_length FROM cameras;");
  while (st.Step() == SQLITE_ROW) {
    Camera c;
    c.camera_id = static_cast<camera_t>(sqlite3_column_int64(st.s, 0));
    c.model_id = static_cast<int>(sqlite3_column_int64(st.s, 1));
    c.width = static_cast<size_t>(sqlite3_column_int64(st.s, 2));
    c.height = static_cast<size_t>(sqlite3_column_int64(st.s, 3));
    const size_t nb = static_cast<size_t>(sqlite3_column_bytes(st.s, 4));
    // (the reference copies num_params_bytes into num_params_bytes / 8 doubles and then CHECKs Camera::VerifyParams,
    // database.cc ReadCameraRow; here an odd blob or a parameter count that is not the model's is an exception)
    static const size_t kNumParams[11] = {3, 4, 4, 5, 8, 8, 12, 5, 4, 5, 12};  // camera_models.h:187-349, ids 0 .. 10
    if (nb % sizeof(double) != 0 || (c.model_id >= 0 && c.model_id <= 10 && nb / sizeof(double) != kNumParams[c.model_id]))
      throw std::runtime_error("camera " + std::to_string(c.camera_id) + ": the params blob does not hold the parameters of camera model " +
                               std::to_string(c.model_id));
    c.params.resize(nb / sizeof(double));
    if (nb) std::memcpy(c.params.data(), sqlite3_column_blob(st.s, 4), nb);
    c.prior_focal_length = sqlite3_column_int64(st.s, 5) != 0;
    cams.push_back(c);
  }
  return cams;
}

std::vector<Image> Database::ReadAllImages() const {
  std::vector<Image> images;
  Stmt st(database_, "SELECT image_id, name, camera_id FROM images;");
  while (st.Step() == SQLITE_ROW) {
    Image im;
    im.image_id = static_cast<image_t>(sqlite3_column_int64(st.s, 0));
    im.name = reinterpret_cast<const char*>(sqlite3_column_text(st.s, 1));
    im.camera_id = static_cast<camera_t>(sqlite3_column_int64(st.s, 2));
    images.push_back(im);
  }
  return images;
}

FeatureKeypoints Database::ReadKeypoints(image_t image_id) const {
  // FeatureKeypointsFromBlob, database.cc:60-88: 2, 4 or 6 float columns
  static const char* const kSql = "SELECT rows, cols, data FROM keypoints WHERE image_id = ?;";
  Stmt st(database_, kSql, stmts_);
  sqlite3_bind_int64(st.s, 1, image_id);
  FeatureKeypoints kps;
  if (st.Step() != SQLITE_ROW) return kps;
  // (the reference CHECKs the blob size against rows x cols, database.cc:60-88 -> BlobToMatrix; here a malformed row is an
  // exception the caller reports, never a read past the blob)
  const sqlite3_int64 rows64 = sqlite3_column_int64(st.s, 0), cols64 = sqlite3_column_int64(st.s, 1);
  const size_t nb = static_cast<size_t>(sqlite3_column_bytes(st.s, 2));
  if (rows64 < 0 || (cols64 != 2 && cols64 != 4 && cols64 != 6) || static_cast<uint64_t>(rows64) * static_cast<uint64_t>(cols64) * 4 != nb)
    throw std::runtime_error("keypoints of image " + std::to_string(image_id) + ": blob size does not match rows x cols (2, 4 or 6 float columns)");
  const size_t rows = static_cast<size_t>(rows64);
  const size_t cols = static_cast<size_t>(cols64);
  const float* d = static_cast<const float*>(sqlite3_column_blob(st.s, 2));
  kps.resize(rows);
  for (size_t i = 0; i < rows; ++i) {
    const float* r = d + i * cols;
    kps[i].x = r[0];
    kps[i].y = r[1];
    if (cols == 4) {  // FeatureKeypoint(x, y, scale, orientation), feature/types.cc
      const float scale = r[2], ori = r[3];
      kps[i].a11 = scale * std::cos(ori);
      kps[i].a12 = -scale * std::sin(ori);
      kps[i].a21 = scale * std::sin(ori);
      kps[i].a22 = scale * std::cos(ori);
    } else if (cols == 6) {
      kps[i].a11 = r[2];
      kps[i].a12 = r[3];
      kps[i].a21 = r[4];
      kps[i].a22 = r[5];
    } else if (cols != 2) {
      throw std::runtime_error("Keypoint format not supported");
    }
  }
  return kps;
}

FeatureDescriptors Database::ReadDescriptors(image_t image_id) const {
  static const char* const kSql = "SELECT rows, cols, data FROM descriptors WHERE image_id = ?;";
  Stmt st(database_, kSql, stmts_);
  sqlite3_bind_int64(st.s, 1, image_id);
  FeatureDescriptors d;
  if (st.Step() != SQLITE_ROW) return d;
  const sqlite3_int64 rows64 = sqlite3_column_int64(st.s, 0), cols64 = sqlite3_column_int64(st.s, 1);
  const size_t nb = static_cast<size_t>(sqlite3_column_bytes(st.s, 2));
  // an image without features may be stored as a 0 x 0 (or 0 x anything) matrix with an empty blob: the reference's
  // ReadDynamicMatrixBlob only CHECKs rows * cols * size == num_bytes (database.cc:60-77) and returns it
  if (rows64 == 0 && nb == 0 && cols64 >= 0) return d;
  if (rows64 < 0 || cols64 != 128 || static_cast<uint64_t>(rows64) * 128u != nb)  // FeatureDescriptors: rows x 128 uint8
    throw std::runtime_error("descriptors of image " + std::to_string(image_id) + ": blob size does not match rows x 128");
  d.rows = static_cast<size_t>(rows64);
  d.cols = static_cast<size_t>(cols64);
  d.data.resize(nb);
  if (nb) std::memcpy(d.data.data(), sqlite3_column_blob(st.s, 2), nb);
  return d;
}

static FeatureMatches MatchesFromBlob(sqlite3_stmt* s, int col_rows, int col_data) {
  const sqlite3_int64 rows64 = sqlite3_column_int64(s, col_rows);
  const size_t nb = static_cast<size_t>(sqlite3_column_bytes(s, col_data));
  const sqlite3_int64 cols64 = sqlite3_column_int64(s, col_rows + 1);  // (rows, cols, data are adjacent columns of both tables)
  if (rows64 < 0 || cols64 != 2 || static_cast<uint64_t>(rows64) * 8u != nb)  // rows x 2 uint32 (FeatureMatchesFromBlob, database.cc:103-122)
    throw std::runtime_error("matches blob: size does not match rows x 2 uint32");
  const size_t rows = static_cast<size_t>(rows64);
  FeatureMatches m(rows);
  const uint32_t* d = static_cast<const uint32_t*>(sqlite3_column_blob(s, col_data));
  for (size_t i = 0; i < rows; ++i) {
    m[i].point2D_idx1 = d[2 * i];
    m[i].point2D_idx2 = d[2 * i + 1];
  }
  return m;
}

FeatureMatches Database::ReadMatches(image_t a, image_t b) const {
  static const char* const kSql = "SELECT rows, cols, data FROM matches WHERE pair_id = ?;";
  Stmt st(database_, kSql, stmts_);
  sqlite3_bind_int64(st.s, 1, static_cast<sqlite3_int64>(ImagePairToPairId(a, b)));
  FeatureMatches m;
  if (st.Step() != SQLITE_ROW) return m;
  m = MatchesFromBlob(st.s, 0, 2);
  if (SwapImagePair(a, b))
    for (auto& x : m) std::swap(x.point2D_idx1, x.point2D_idx2);
  return m;
}

TwoViewGeometry Database::ReadTwoViewGeometry(image_t a, image_t b) const {
  // database.cc:493-533: qvec comes back from the F column, tvec from the E column
  static const char* const kSql = "SELECT rows, cols, data, config, F, E, H FROM two_view_geometries WHERE pair_id = ?;";
  Stmt st(database_, kSql, stmts_);
  sqlite3_bind_int64(st.s, 1, static_cast<sqlite3_int64>(ImagePairToPairId(a, b)));
  TwoViewGeometry t;
  if (st.Step() != SQLITE_ROW) return t;
  t.inlier_matches = MatchesFromBlob(st.s, 0, 2);
  t.config = static_cast<int>(sqlite3_column_int64(st.s, 3));
  if (sqlite3_column_bytes(st.s, 4) == 32) std::memcpy(t.qvec, sqlite3_column_blob(st.s, 4), 32);
  if (sqlite3_column_bytes(st.s, 5) == 24) std::memcpy(t.tvec, sqlite3_column_blob(st.s, 5), 24);
  if (SwapImagePair(a, b)) t.Invert();
  return t;
}

camera_t Database::WriteCamera(const Camera& c) const {
  Stmt st(database_, "INSERT INTO cameras(camera_id, model, width, height, params, prior_focal_length) VALUES(?, ?, ?, ?, ?, ?);");
  if (c.camera_id)
    sqlite3_bind_int64(st.s, 1, c.camera_id);
  else
    sqlite3_bind_null(st.s, 1);
  sqlite3_bind_int64(st.s, 2, c.model_id);
  sqlite3_bind_int64(st.s, 3, static_cast<sqlite3_int64>(c.width));
  sqlite3_bind_int64(st.s, 4, static_cast<sqlite3_int64>(c.height));
  sqlite3_bind_blob(st.s, 5, c.params.data(), static_cast<int>(c.params.size() * sizeof(double)), SQLITE_STATIC);
  sqlite3_bind_int64(st.s, 6, c.prior_focal_length ? 1 : 0);
  st.Step();
  return static_cast<camera_t>(sqlite3_last_insert_rowid(database_));
}

image_t Database::WriteImage(const Image& im) const {
  Stmt st(database_, "INSERT INTO images(image_id, name, camera_id) VALUES(?, ?, ?);");
  if (im.image_id)
    sqlite3_bind_int64(st.s, 1, im.image_id);
  else
    sqlite3_bind_null(st.s, 1);
  sqlite3_bind_text(st.s, 2, im.name.c_str(), -1, SQLITE_STATIC);
  sqlite3_bind_int64(st.s, 3, im.camera_id);
  st.Step();
  return static_cast<image_t>(sqlite3_last_insert_rowid(database_));
}

void Database::WriteKeypoints(image_t image_id, const FeatureKeypoints& kps) const {
  std::vector<float> blob(kps.size() * 6);
  for (size_t i = 0; i < kps.size(); ++i) {
    float* r = blob.data() + 6 * i;
    r[0] = kps[i].x; r[1] = kps[i].y; r[2] = kps[i].a11; r[3] = kps[i].a12; r[4] = kps[i].a21; r[5] = kps[i].a22;
  }
  Stmt st(database_, "INSERT INTO keypoints(image_id, rows, cols, data) VALUES(?, ?, ?, ?);");
  sqlite3_bind_int64(st.s, 1, image_id);
  sqlite3_bind_int64(st.s, 2, static_cast<sqlite3_int64>(kps.size()));
  sqlite3_bind_int64(st.s, 3, 6);
  sqlite3_bind_blob(st.s, 4, blob.data(), static_cast<int>(blob.size() * sizeof(float)), SQLITE_STATIC);
  st.Step();
}

void Database::WriteDescriptors(image_t image_id, const FeatureDescriptors& d) const {
  Stmt st(database_, "INSERT INTO descriptors(image_id, rows, cols, data) VALUES(?, ?, ?, ?);");
  sqlite3_bind_int64(st.s, 1, image_id);
  sqlite3_bind_int64(st.s, 2, static_cast<sqlite3_int64>(d.rows));
  sqlite3_bind_int64(st.s, 3, static_cast<sqlite3_int64>(d.cols));
  sqlite3_bind_blob(st.s, 4, d.data.data(), static_cast<int>(d.data.size()), SQLITE_STATIC);
  st.Step();
}

static std::vector<uint32_t> MatchesToBlob(const FeatureMatches& m, bool swap) {
  std::vector<uint32_t> blob(m.size() * 2);
  for (size_t i = 0; i < m.size(); ++i) {
    blob[2 * i] = swap ? m[i].point2D_idx2 : m[i].point2D_idx1;
    blob[2 * i + 1] = swap ? m[i].point2D_idx1 : m[i].point2D_idx2;
  }
  return blob;
}

void Database::WriteMatches(image_t a, image_t b, const FeatureMatches& matches) const {
  const std::vector<uint32_t> blob = MatchesToBlob(matches, SwapImagePair(a, b));
  static const char* const kSql = "INSERT INTO matches(pair_id, rows, cols, data) VALUES(?, ?, ?, ?);";
  Stmt st(database_, kSql, stmts_);
  sqlite3_bind_int64(st.s, 1, static_cast<sqlite3_int64>(ImagePairToPairId(a, b)));
  sqlite3_bind_int64(st.s, 2, static_cast<sqlite3_int64>(matches.size()));
  sqlite3_bind_int64(st.s, 3, 2);
  sqlite3_bind_blob(st.s, 4, blob.data(), static_cast<int>(blob.size() * sizeof(uint32_t)), SQLITE_STATIC);
  st.Step();
}

void Database::WriteTwoViewGeometry(image_t a, image_t b, const TwoViewGeometry& tvg_in) const {
  TwoViewGeometry swapped;
  const TwoViewGeometry* t = &tvg_in;
  if (SwapImagePair(a, b)) {
    swapped = tvg_in;
    swapped.Invert();
    t = &swapped;
  }
  const std::vector<uint32_t> blob = MatchesToBlob(t->inlier_matches, false);
  static const char* const kSql = "INSERT INTO two_view_geometries(pair_id, rows, cols, data, config, F, E, H) VALUES(?, ?, ?, ?, ?, ?, ?, ?);";
  Stmt st(database_, kSql, stmts_);
  sqlite3_bind_int64(st.s, 1, static_cast<sqlite3_int64>(ImagePairToPairId(a, b)));
  sqlite3_bind_int64(st.s, 2, static_cast<sqlite3_int64>(t->inlier_matches.size()));
  sqlite3_bind_int64(st.s, 3, 2);
  sqlite3_bind_blob(st.s, 4, blob.data(), static_cast<int>(blob.size() * sizeof(uint32_t)), SQLITE_STATIC);
  sqlite3_bind_int64(st.s, 5, t->config);
  // DAGSfM: qvec goes into the F column, tvec into the E column, H is never bound (database.cc:733-747);
  // zero-length blobs when there are no inliers
  if (!t->inlier_matches.empty()) {
    sqlite3_bind_blob(st.s, 6, t->qvec, 32, SQLITE_STATIC);
    sqlite3_bind_blob(st.s, 7, t->tvec, 24, SQLITE_STATIC);
  } else {
    sqlite3_bind_zeroblob(st.s, 6, 0);
    sqlite3_bind_zeroblob(st.s, 7, 0);
  }
  sqlite3_bind_null(st.s, 8);
  st.Step();
}

static void DeleteRow(sqlite3* db, StmtCache* cache, const char* sql, image_pair_t pair_id) {
  Stmt st(db, sql, cache);
  sqlite3_bind_int64(st.s, 1, static_cast<sqlite3_int64>(pair_id));
  st.Step();
}
void Database::DeleteMatches(image_t a, image_t b) const {
  static const char* const kSql = "DELETE FROM matches WHERE pair_id = ?;";
  DeleteRow(database_, stmts_, kSql, ImagePairToPairId(a, b));
}
void Database::DeleteInlierMatches(image_t a, image_t b) const {
  static const char* const kSql = "DELETE FROM two_view_geometries WHERE pair_id = ?;";
  DeleteRow(database_, stmts_, kSql, ImagePairToPairId(a, b));
}

void Database::BeginTransaction() const { Exec("BEGIN TRANSACTION;"); }
void Database::EndTransaction() const { Exec("END TRANSACTION;"); }
bool Database::InTransaction() const { return database_ != nullptr && sqlite3_get_autocommit(database_) == 0; }
void Database::SetBulkLoadJournal(bool in_memory) const { Exec(in_memory ? "PRAGMA journal_mode=MEMORY;" : "PRAGMA journal_mode=WAL;"); }

void Database::RollbackTransaction() const {
  // nothing to roll back outside a transaction (sqlite would answer "cannot rollback - no transaction is active")
  if (database_ != nullptr && sqlite3_get_autocommit(database_)) return;
  Exec("ROLLBACK TRANSACTION;");
}

}  // namespace dagsfm_amd
