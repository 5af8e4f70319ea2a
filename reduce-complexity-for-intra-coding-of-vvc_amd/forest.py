"""Flattened random forest for the FAST_ALGORITHM partition classifier.

The reference pickles an sklearn RandomForestClassifier (BIN/TEST.py: joblib.load("Partition_32.pkl").predict(x)) that is not in its
repository.  The device needs plain arrays, so a forest travels as the concatenated sklearn `tree_` arrays of its trees:
    root[n_trees]                  first node of every tree
    feature, left, right [n_nodes] int32, children -1 at leaves (child indices already offset into the concatenation)
    threshold[n_nodes]             float64; the row goes left when float32(feature value) <= threshold, as in sklearn
    value[n_nodes, n_classes]      float64 class distribution of the node (tree_.value normalised per node)
    classes[n_classes]             label of each column: 0 no split, 1 QT, 2 BT_H, 3 BT_V, 4 TT_H, 5 TT_V
predict = classes[argmax(sum over trees, in tree order, of value[leaf])] — ForestClassifier.predict with n_jobs=1."""
import numpy as np

N_FEATURES = 26
KEYS = ("root", "feature", "threshold", "left", "right", "value", "classes")


def forest_from_sklearn(clf):
    roots, feat, thr, left, right, val = [], [], [], [], [], []
    base = 0
    for est in clf.estimators_:
        t = est.tree_
        n = t.node_count
        roots.append(base)
        feat.append(np.where(t.children_left < 0, 0, t.feature).astype(np.int32))
        thr.append(t.threshold.astype(np.float64))
        left.append(np.where(t.children_left < 0, -1, t.children_left + base).astype(np.int32))
        right.append(np.where(t.children_right < 0, -1, t.children_right + base).astype(np.int32))
        v = t.value[:, 0, :].astype(np.float64)
        val.append(v / v.sum(axis=1, keepdims=True))
        base += n
    return dict(root=np.array(roots, np.int32), feature=np.concatenate(feat), threshold=np.concatenate(thr), left=np.concatenate(left),
                right=np.concatenate(right), value=np.ascontiguousarray(np.concatenate(val)), classes=np.asarray(clf.classes_, np.int32))


def save_forest(path, forest, **meta):
    np.savez_compressed(path, **{k: forest[k] for k in KEYS}, **meta)


def load_forest(path):
    g = np.load(path)
    f = {k: np.ascontiguousarray(g[k]) for k in KEYS}
    check_forest(f)
    return f


def check_forest(f):
    n = len(f["feature"])
    assert f["root"].dtype == np.int32 and f["feature"].dtype == np.int32 and f["left"].dtype == np.int32 and f["right"].dtype == np.int32
    assert f["threshold"].dtype == np.float64 and f["value"].dtype == np.float64 and f["classes"].dtype == np.int32
    assert len(f["threshold"]) == n and len(f["left"]) == n and len(f["right"]) == n and f["value"].shape == (n, len(f["classes"]))
    assert 1 <= len(f["classes"]) <= 8 and ((f["classes"] >= 0) & (f["classes"] <= 5)).all()
    inner = f["left"] >= 0
    assert (f["right"][inner] >= 0).all() and (f["left"][inner] < n).all() and (f["right"][inner] < n).all()
    assert ((f["feature"][inner] >= 0) & (f["feature"][inner] < N_FEATURES)).all() and ((f["root"] >= 0) & (f["root"] < n)).all()


def predict_numpy(f, rows):
    """Plain restatement used by host-side tests (not a product path: the encoder evaluates the forest on the device)."""
    rows = np.asarray(rows, np.int32).reshape(-1, N_FEATURES)
    out = np.zeros(len(rows), np.int32)
    for i, r in enumerate(rows):
        acc = np.zeros(len(f["classes"]))
        x = r.astype(np.float32).astype(np.float64)
        for t in f["root"]:
            n = int(t)
            while f["left"][n] >= 0:
                n = int(f["left"][n]) if x[f["feature"][n]] <= f["threshold"][n] else int(f["right"][n])
            acc += f["value"][n]
        out[i] = f["classes"][int(np.argmax(acc))]
    return out
