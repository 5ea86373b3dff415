"""Test helper: builds a COLMAP/DAGSfM database.db with python's sqlite3 (blob formats as in
/root/reference/scripts/python/database.py:113-140, 196-226) and reads result rows back."""
import sqlite3

import numpy as np

MAX_IMAGE_ID = 2 ** 31 - 1

SCHEMA = """
CREATE TABLE IF NOT EXISTS cameras (camera_id INTEGER PRIMARY KEY AUTOINCREMENT NOT NULL, model INTEGER NOT NULL,
    width INTEGER NOT NULL, height INTEGER NOT NULL, params BLOB, prior_focal_length INTEGER NOT NULL);
CREATE TABLE IF NOT EXISTS images (image_id INTEGER PRIMARY KEY AUTOINCREMENT NOT NULL, name TEXT NOT NULL UNIQUE,
    camera_id INTEGER NOT NULL, prior_qw REAL, prior_qx REAL, prior_qy REAL, prior_qz REAL, prior_tx REAL, prior_ty REAL,
    prior_tz REAL, CONSTRAINT image_id_check CHECK(image_id >= 0 and image_id < 2147483647),
    FOREIGN KEY(camera_id) REFERENCES cameras(camera_id));
CREATE TABLE IF NOT EXISTS keypoints (image_id INTEGER PRIMARY KEY NOT NULL, rows INTEGER NOT NULL, cols INTEGER NOT NULL,
    data BLOB, FOREIGN KEY(image_id) REFERENCES images(image_id) ON DELETE CASCADE);
CREATE TABLE IF NOT EXISTS descriptors (image_id INTEGER PRIMARY KEY NOT NULL, rows INTEGER NOT NULL, cols INTEGER NOT NULL,
    data BLOB, FOREIGN KEY(image_id) REFERENCES images(image_id) ON DELETE CASCADE);
CREATE TABLE IF NOT EXISTS matches (pair_id INTEGER PRIMARY KEY NOT NULL, rows INTEGER NOT NULL, cols INTEGER NOT NULL, data BLOB);
CREATE TABLE IF NOT EXISTS two_view_geometries (pair_id INTEGER PRIMARY KEY NOT NULL, rows INTEGER NOT NULL,
    cols INTEGER NOT NULL, data BLOB, config INTEGER NOT NULL, F BLOB, E BLOB, H BLOB);
"""


def pair_id(a, b):
    a, b = min(a, b), max(a, b)
    return a * MAX_IMAGE_ID + b


def create(path, images, focal=800.0, width=1000, height=750, prior=True, kp_cols=6):
    """images: list of (descriptors u8 [n,128], keypoints f32 [n,2]); image ids are 1..N, one shared SIMPLE_PINHOLE camera."""
    con = sqlite3.connect(path)
    con.executescript(SCHEMA)
    params = np.array([focal, width / 2.0, height / 2.0], dtype=np.float64)
    con.execute("INSERT INTO cameras VALUES (?, ?, ?, ?, ?, ?)", (1, 0, width, height, params.tobytes(), int(prior)))
    for i, (desc, kp) in enumerate(images):
        iid = i + 1
        con.execute("INSERT INTO images(image_id, name, camera_id) VALUES (?, ?, ?)", (iid, "img%04d.jpg" % iid, 1))
        kp = np.asarray(kp, dtype=np.float32)
        kp = kp.reshape(len(kp), -1) if len(kp) else np.zeros((0, kp_cols), np.float32)
        k = np.zeros((len(kp), kp_cols), dtype=np.float32)
        if kp.shape[1] == kp_cols:  # x, y and the affine shape a11, a12, a21, a22 as given
            k[:] = kp
        else:
            k[:, :2] = kp[:, :2]
            if kp_cols == 6:
                k[:, 2] = 1.0
                k[:, 5] = 1.0
        con.execute("INSERT INTO keypoints VALUES (?, ?, ?, ?)", (iid, k.shape[0], kp_cols, k.tobytes()))
        d = np.ascontiguousarray(desc, dtype=np.uint8)
        con.execute("INSERT INTO descriptors VALUES (?, ?, ?, ?)", (iid, d.shape[0], 128, d.tobytes()))
    con.commit()
    con.close()


def read_results(path):
    con = sqlite3.connect(path)
    matches = {}
    for pid, rows, cols, data in con.execute("SELECT pair_id, rows, cols, data FROM matches"):
        matches[pid] = np.frombuffer(data or b"", dtype=np.uint32).reshape(rows, 2) if rows else np.zeros((0, 2), np.uint32)
    tvgs = {}
    for pid, rows, cols, data, config, F, E, H in con.execute(
            "SELECT pair_id, rows, cols, data, config, F, E, H FROM two_view_geometries"):
        tvgs[pid] = dict(inliers=np.frombuffer(data or b"", dtype=np.uint32).reshape(rows, 2) if rows else np.zeros((0, 2), np.uint32),
                         config=config, F=F, E=E, H=H)
    con.close()
    return matches, tvgs
