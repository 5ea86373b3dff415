"""Pure-numpy restatement of what dsm_retrieval_index / dsm_retrieval_matches hold and deliver (tests only): the inverted
files (entries sorted by word, image, feature), the IDF weights and, per query, the candidate tuples for a given set of
retrieved images.  The word search comes from the oracle (exact, like the device's); the Hamming signatures are the float
sums left to right the oracle and the device both define."""
import numpy as np


def signatures(proj, thr_rows, desc):
    """bit i of feature f = (sum_j proj[i][j] * float(d[j]), left to right in float) > thr_rows[f][i]"""
    acc = np.zeros((len(desc), 64), np.float32)
    d = desc.astype(np.float32)
    for j in range(128):
        acc = (acc + (proj[:, j][None, :] * d[:, j][:, None]).astype(np.float32)).astype(np.float32)
    bits = acc > thr_rows
    return (bits.astype(np.uint64) << np.arange(64, dtype=np.uint64)[None, :]).sum(axis=1).astype(np.uint64)


def emulate(words, proj, thr, descs, k, orc):
    """Returns (tuples(q, retrieved image list) -> [m, 5] uint32 as dsm_get_retrieval_matches orders them, idf [W] f32)."""
    ent = []
    for img, d in enumerate(descs):
        if len(d) == 0:
            continue
        w = orc.find_word_ids(d, 1)[:, 0]
        sig = signatures(proj, thr[w], d)
        for f in range(len(d)):
            ent.append((int(w[f]), img, f, int(sig[f])))
    ent.sort(key=lambda e: (e[0], e[1], e[2]))
    e_word = np.array([e[0] for e in ent], np.int64)
    e_img = np.array([e[1] for e in ent], np.int64)
    e_feat = np.array([e[2] for e in ent], np.int64)
    e_sig = np.array([e[3] for e in ent], np.uint64)
    n_img_total = sum(1 for d in descs if len(d))
    idf = np.zeros(len(words), np.float32)
    for w in np.unique(e_word):
        idf[w] = np.float32(np.log(n_img_total / len(np.unique(e_img[e_word == w]))))
    start = np.searchsorted(e_word, np.arange(len(words) + 1))

    def tuples(q, top):
        d = descs[q]
        out = []
        if len(d) == 0:
            return np.zeros((0, 5), np.uint32)
        wid = orc.find_word_ids(d, k)
        topset = set(int(t) for t in top)
        for i in range(len(d)):
            for n in range(k):
                w = int(wid[i, n])
                if w == 0x7fffffff:
                    continue
                s, e = start[w], start[w + 1]
                if s == e:
                    continue
                bq = signatures(proj, thr[w][None, :], d[i:i + 1])[0]
                for p in range(s, e):
                    if int(e_img[p]) in topset:
                        h = bin(int(bq) ^ int(e_sig[p])).count("1")
                        if h <= 24:
                            out.append((i, int(e_img[p]), int(e_feat[p]), (w << 8) | h, p))
        return np.array(out, np.uint32).reshape(-1, 5)
    return tuples, idf
