"""Fusion of the two streams: the reference's ``Sheet03/combinedModel.py`` call surface.

``combineDescriptors`` is the inner join of the two per-video descriptor CSVs (Sheet03/combinedModel.py:9-26).
The reference then fits ``svm.LinearSVC`` on the CPU (``:34-35``, liblinear training: out of scope) and
calls ``predict`` (``:38``); ``linearSvmPredict`` is that predict as argmax(X W^T + b).
"""
import numpy as np
import pandas as pd

from .parameters import VIDEO_DESCRIPTOR_DIM


def combineDescriptors(spatialCsv, temporalCsv):
    """-> (descriptors float64 [N, 2*VIDEO_DESCRIPTOR_DIM] = spatial || temporal, labels = spatial labels),
    rows in the order of the spatial CSV restricted to videos present in both (inner merge on the name)."""
    headers = ["vidname", "label"]
    for dim in range(VIDEO_DESCRIPTOR_DIM):
        headers.append("dim" + str(dim))
    dfSpatial = pd.read_csv(spatialCsv, names=headers)
    dfTemporal = pd.read_csv(temporalCsv, names=headers)
    dfMerged = pd.merge(dfSpatial, dfTemporal, on="vidname", how="inner", suffixes=("_s", "_t"))
    spatialHeaders = [headers[i] + "_s" for i in range(2, len(headers))]
    temporalHeaders = [headers[i] + "_t" for i in range(2, len(headers))]
    allHeaders = spatialHeaders + temporalHeaders
    descriptors = dfMerged[allHeaders].values
    labels = dfMerged["label_s"].values
    return descriptors, labels


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
