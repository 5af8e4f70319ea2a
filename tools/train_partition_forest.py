#!/usr/bin/env python3
"""Train the partition forest the FAST_ALGORITHM path asks (counterpart of the fork authors' off-line training: GET_TRAINING_SET dump →
sklearn RandomForestClassifier → Partition_<QP>.pkl, none of which is in the reference repository).

Training rows come from the CPU oracle's full-RDO search over seeded synthetic pictures: for every luma node that qualifies for the
classifier, the 26 features and the partition the exhaustive search chose there.  Runs in the development container only (needs
sklearn and the oracle); the product loads the resulting arrays (forests/partition_qp<QP>.npz) and never imports sklearn.

  python tools/train_partition_forest.py [--qp 32] [--trees 24] [--depth 12] [--pictures 12] [--balance 0.5] [--device]
Also writes tests/golden/forest.npz: feature rows of held-out pictures with sklearn's own predict() on them (the golden vector for
the oracle's and the device's forest inference)."""
import argparse
import importlib
import os
import sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def rows_of(pkg, O, W, H, qp, seed, texture):
    out = []
    O.compress_frame(pkg.synth_frame(W, H, seed % 5, 8, seed, chroma_texture=texture), W, H, pkg.slice_params(qp), chroma=0, tools=O.TOOLS_DEFAULT | (1 << 4), training_rows=out)      # luma tree with every built luma tool: MRL, MTS, CU reuse
    r = out[0]
    return r[r[:, 27] >= 0]


def rows_of_device(pkg, W, H, qp, seeds, texture, tools):
    """The same rows from the device's own dump (vvcx_enable_training_dump): the plain full search with the given tool set over one batch of pictures on an MI355X,
    one tile per CTU.  Minutes where the oracle needs hours, and with every tool of the cfg in the labels."""
    import torch
    ctw, cth = (W + 127) // 128, (H + 127) // 128
    enc = pkg.VvcxEncoder(W, H, 8, tile_cols=ctw, tile_rows=cth, tools=tools, max_frames=len(seeds))
    sp = pkg.slice_params(qp, dep_quant=bool(tools & 0x40))
    enc.set_slice(sp["qp"], sp["qp_c"], sp["lam"], sp["dist_weight"])
    enc.enable_training_dump(len(seeds) * ctw * cth * 6000)
    dev = []
    for i, seed in enumerate(seeds):
        org = [torch.from_numpy(np.ascontiguousarray(p)).cuda() for p in pkg.synth_frame(W, H, seed % 5, 8, seed, chroma_texture=texture if i % 2 else 0.0)]
        dev.append((org, [torch.zeros_like(t) for t in org]))
    enc.bind_frames([([t.data_ptr() for t in o], [t.data_ptr() for t in r], [t.shape[1] for t in o]) for o, r in dev])
    enc.compress_bound_frames()
    r = enc.training_rows()
    enc.close()
    return r[r[:, 27] >= 0]


def _rows_job(W, H, qp, seed, texture):
    import oracle_lib as O
    pkg = importlib.import_module("reduce-complexity-for-intra-coding-of-vvc_amd")
    return rows_of(pkg, O, W, H, qp, seed, texture)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--qp", type=int, default=32)
    ap.add_argument("--trees", type=int, default=24)
    ap.add_argument("--depth", type=int, default=12)
    ap.add_argument("--pictures", type=int, default=12)
    ap.add_argument("--balance", type=float, default=0.5)
    ap.add_argument("--device", action="store_true", help="rows from the device's training dump (needs an MI355X): 1080p pictures, tools 0xfff; writes forests/partition_qp<QP>_device.npz "
                    "and leaves the shipped forest and its golden vector alone")
    ap.add_argument("--dump-rows", type=str, default=None, help="with --device: only dump the rows (a bounded random sample: train / held) to this .npz and stop - the GPU box's part of "
                    "the job; the forest is then trained from the file where sklearn time is free (--rows)")
    ap.add_argument("--rows", type=str, default=None, help="train from a --dump-rows file (device rows, tools 0xfff) and write the SHIPPED forest forests/partition_qp<QP>.npz "
                    "(+ the golden vector of the QP 32 forest)")
    ap.add_argument("--max-rows", type=int, default=400000)
    args = ap.parse_args()
    if args.dump_rows:
        pkg = importlib.import_module("reduce-complexity-for-intra-coding-of-vvc_amd")
        train = rows_of_device(pkg, 1920, 1080, args.qp, [5000 + i for i in range(args.pictures)], 0.5, 0xfff)
        held = rows_of_device(pkg, 1920, 1080, args.qp, [7000, 7001], 0.5, 0xfff)
        g = np.random.default_rng(args.qp)
        n_all = len(train)
        if len(train) > args.max_rows:
            train = train[g.permutation(len(train))[:args.max_rows]]
        if len(held) > args.max_rows // 8:
            held = held[g.permutation(len(held))[:args.max_rows // 8]]
        np.savez_compressed(args.dump_rows, train=train.astype(np.int32), held=held.astype(np.int32), qp=np.array([args.qp]), rows_produced=np.array([n_all]), tools=np.array([0xfff]))
        print("dumped", len(train), "of", n_all, "training rows and", len(held), "held-out rows to", args.dump_rows)
        return
    from sklearn.ensemble import RandomForestClassifier
    import sklearn
    import oracle_lib as O
    pkg = importlib.import_module("reduce-complexity-for-intra-coding-of-vvc_amd")
    from concurrent.futures import ProcessPoolExecutor
    if args.rows:
        d = np.load(args.rows)
        assert int(d["qp"][0]) == args.qp, "the rows were dumped at another QP"
        train, held = d["train"], d["held"]
    elif args.device:
        train = rows_of_device(pkg, 1920, 1080, args.qp, [5000 + i for i in range(args.pictures)], 0.5, 0xfff)
        held = rows_of_device(pkg, 1920, 1080, args.qp, [7000, 7001], 0.5, 0xfff)
    else:
        with ProcessPoolExecutor(max_workers=6) as ex:
            jobs = [ex.submit(_rows_job, 256, 256, args.qp, 5000 + i, 0.5 * (i % 2)) for i in range(args.pictures)] + [ex.submit(_rows_job, 256, 128, args.qp, 7000 + i, 0.5) for i in range(2)]
            parts = [j.result() for j in jobs]
        train = np.concatenate(parts[:args.pictures]); held = np.concatenate(parts[args.pictures:])
    print("training rows", len(train), "label histogram", np.bincount(train[:, 27], minlength=6), "held-out rows", len(held))
    # most visited nodes are small ones where "no split" wins: weight the classes (balanced weights to the power --balance) so that the
    # forest does not collapse to class 0.  0.5 measured +1 % RD cost for 1.7x less search on a held-out picture; 1.0: +10 % for 3.5x
    cnt = np.bincount(train[:, 27], minlength=6).astype(float)
    weight = {c: float((cnt.sum() / (6 * max(cnt[c], 1.0))) ** args.balance) for c in range(6)}
    clf = RandomForestClassifier(n_estimators=args.trees, max_depth=args.depth, min_samples_leaf=4, random_state=0, n_jobs=1, class_weight=weight)
    clf.fit(train[:, :26], train[:, 27])
    pred = clf.predict(held[:, :26])
    print("held-out accuracy %.3f" % float((pred == held[:, 27]).mean()), "predicted histogram", np.bincount(pred, minlength=6))
    forest = pkg.forest_from_sklearn(clf)
    path = os.path.join(ROOT, "reduce-complexity-for-intra-coding-of-vvc_amd", "forests", "partition_qp%d%s.npz" % (args.qp, "_device" if args.device else ""))
    pkg.save_forest(path, forest, qp=np.array([args.qp]), sklearn_version=np.array([sklearn.__version__]), training_rows=np.array([len(train)]),
                    row_source=np.array(["device dump, tools 0xfff, 1080p" if (args.rows or args.device) else "oracle, luma tree, MRL + MTS, 256x256"]))
    print("wrote", path, "trees", len(forest["root"]), "nodes", len(forest["feature"]))
    order = np.random.default_rng(0).permutation(len(held))
    sel = np.concatenate([order[pred[order] == c][:120] for c in range(6)])         # up to 120 rows per predicted class
    if args.qp == 32 and not args.device:                   # the golden vector belongs to the shipped QP 32 forest (oracle rows or --rows: both write the shipped file)
        np.savez_compressed(os.path.join(ROOT, "tests", "golden", "forest.npz"), rows=held[sel, :26].astype(np.int32), sklearn_predict=pred[sel].astype(np.int32),
                            qp=np.array([args.qp]))


if __name__ == "__main__":
    main()
