"""Fusion of the two streams: the reference's ``Sheet03/combinedModel.py`` call surface.

``combineDescriptors`` is the inner join of the two per-video descriptor CSVs (Sheet03/combinedModel.py:9-26).
The reference then fits ``svm.LinearSVC`` on the CPU (``:34-35``, liblinear training: out of scope) and
calls ``predict`` (``:38``); ``linearSvmPredict`` is that predict as argmax(X W^T + b).
"""
import numpy as np
import pandas as pd

from .parameters import VIDEO_DESCRIPTOR_DIM


def combineDescriptors(spatialCsv, temporalCsv):
    """Late fusion input (Sheet03/combinedModel.py:9-26): the two per-video descriptor CSVs written by
    ``saveVideoDescriptors`` (no header row; column 0 = video name, column 1 = label, then the descriptor) are
    inner-joined on the video name.  Returns ``(X, y)``: ``X`` float64 ``[N, 2D]`` with the spatial descriptor in
    columns ``[0, D)`` and the temporal one in ``[D, 2D)``; ``y`` = the label column of the SPATIAL file.  Row order
    and the treatment of repeated names are pandas' inner merge with the spatial table on the left (videos in the
    spatial file's order; a name that occurs m times on one side and n times on the other gives m*n rows)."""
    D = VIDEO_DESCRIPTOR_DIM
    tables = []
    for path in (spatialCsv, temporalCsv):
        t = pd.read_csv(path, header=None)
        if t.shape[1] != D + 2:
            raise ValueError("combineDescriptors: %s has %d columns, expected name, label and %d descriptor values"
                             % (path, t.shape[1], D))
        tables.append(t)
    left, right = tables
    left.columns = ["name", "label"] + ["s%d" % i for i in range(D)]
    right.columns = ["name", "label_t"] + ["t%d" % i for i in range(D)]
    both = left.merge(right, how="inner", on="name")
    X = both[["s%d" % i for i in range(D)] + ["t%d" % i for i in range(D)]].values
    return X, both["label"].values


def linearSvmPredict(descriptors, coef, intercept, classes):
    """``LinearSVC.predict`` (Sheet03/combinedModel.py:38) on the GPU: classes[argmax(X coef^T + intercept)]
    (one-vs-rest; a single coefficient row is the binary case).  ``coef`` / ``intercept`` / ``classes``
    are a fitted ``LinearSVC``'s ``coef_`` / ``intercept_`` / ``classes_`` (fitting stays on the CPU)."""
    from . import fusion
    return fusion.linear_svm_predict(descriptors, coef, intercept, classes)


def accuracy(preds, labels):
    """The reference's accuracy loop (Sheet03/combinedModel.py:39-43), in percent."""
    acc = 0
    for (pred, actual) in zip(preds, labels):
        if pred == actual:
            acc = acc + 1
    return (acc * 100.0) / len(labels)


def main():
    """The fusion script (Sheet03/combinedModel.py:29-43): join the train and the test descriptor CSVs of both
    streams, fit ``LinearSVC`` on the CPU (liblinear; fitting is outside the GPU path), keep the fitted model in
    ``SVM_FILE``, predict the test videos on the GPU and print the accuracy."""
    import joblib  # (the reference's ``sklearn.externals.joblib`` no longer exists)
    from sklearn import svm

    from .parameters import (SPATIAL_TEST_CSV_LOC, SPATIAL_TRAIN_CSV_LOC, SVM_FILE, TEMPORAL_TEST_CSV_LOC,
                             TEMPORAL_TRAIN_CSV_LOC)
    trainX, trainY = combineDescriptors(SPATIAL_TRAIN_CSV_LOC, TEMPORAL_TRAIN_CSV_LOC)
    testX, testY = combineDescriptors(SPATIAL_TEST_CSV_LOC, TEMPORAL_TEST_CSV_LOC)
    clf = svm.LinearSVC()
    clf.fit(trainX, trainY)
    joblib.dump(clf, SVM_FILE)
    preds = linearSvmPredict(testX, clf.coef_, clf.intercept_, clf.classes_)
    print("accuracy = %f percent" % accuracy(preds, testY))


if __name__ == "__main__":
    main()
