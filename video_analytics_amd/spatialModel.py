"""Spatial stream: the reference's ``Sheet03/spatialModel.py`` call surface over the gfx950 kernels.

``SpatialDataset`` keeps the reference's indexing and file layout bit for bit; ``SpatialNetwork``
keeps the constructor signature and ``validate()`` (the inference hot loop,
Sheet03/spatialModel.py:197-231).  The forward pass runs in libva_hip.so (``vgg.Vgg16Stream``).
``train`` / ``resume`` / ``save`` are the training loop and checkpoint bookkeeping: out of scope of
this inference path and raise NotImplementedError.
"""
from __future__ import division

import os
import random  # noqa: F401  (the dataset draws from the global random state, like the reference)

import torch
from PIL import Image
from torch.utils.data import Dataset

from . import fusion, synth, vgg
from .parameters import *  # noqa: F401,F403
from .parameters import (FRAME_EXTN, NORM_MEANS_TF, NORM_STDS_TF, SPATIAL_TEST_CSV_LOC, VIDEO_INPUT_FRAME_COUNT)
from .utils import AverageMeter, ToTensor, saveVideoDescriptors, spatialFrameIndex, videoInfo


def _read_label_dict(actionLabelLoc):
    """``"<int> <name>"`` lines -> {name: int} (Sheet03/spatialModel.py:47-53)."""
    d = {}
    with open(actionLabelLoc, "r") as actionLabelFile:
        for line in actionLabelFile:
            val, key = line.split(" ")
            d[key.strip()] = int(val)
    return d


class SpatialDataset(Dataset):
    """One random frame per video per invocation (Sheet03/spatialModel.py:21-81)."""

    def __init__(self, videoListLoc, rootDir, imageTransforms=None, frameSampleSize=VIDEO_INPUT_FRAME_COUNT,
                 mode="train", actionLabelLoc=None):
        super(SpatialDataset, self).__init__()
        self.rootDir = rootDir if rootDir.endswith("/") else rootDir + "/"
        self.imageTransforms = imageTransforms
        self.frameSampleSize = frameSampleSize  # accepted and ignored, as in the reference (quirk 9)
        self.mode = mode
        with open(videoListLoc, "r") as videoListFile:
            self.videoList = [line for line in videoListFile]
        if actionLabelLoc is None:
            raise ValueError("Action label dictionary required!")
        self.actionLabelDict = _read_label_dict(actionLabelLoc)

    def __len__(self):
        return len(self.videoList)

    def __getitem__(self, index):
        _, videoName, actionLabel, actionCategory, _, _ = videoInfo(self.videoList[index], self.mode)
        if self.mode == "test":
            actionLabel = self.actionLabelDict[actionCategory]
        frameDir = self.rootDir + actionCategory + "/" + videoName + "/"
        nFrames = len([frameName for frameName in os.listdir(frameDir)])
        frameName = spatialFrameIndex(nFrames)
        img = Image.open(frameDir + str(frameName) + FRAME_EXTN)
        if self.imageTransforms is not None:
            loadedFrame = self.imageTransforms(img)
        else:
            loadedFrame = ToTensor()(img)
        actionLabel = int(actionLabel)  # the raw 1-based class index (quirk 4)
        return loadedFrame, actionLabel, videoName


class SpatialNetwork(object):
    """Wrapper of the spatial stream (Sheet03/spatialModel.py:85-283), inference part."""

    C_IN = 3

    def __init__(self, nActionClasses, nEpochs, lr, momentumVal, descriptorDim, trainLoader, testLoader, lrMilestones,
                 ckpLoc, gpu=False, weights=None, seed=1):
        """Same positional arguments as the reference.  ``weights``: dict(conv_w, conv_b, fc_w, fc_b) of
        float32 tensors (OIHW / [out,in]); None = deterministic random init of the same architecture
        (the reference's ImageNet download, ``models.vgg16(pretrained=True)``, is impossible offline)."""
        super(SpatialNetwork, self).__init__()
        self.nActionClasses = nActionClasses
        self.nEpochs = nEpochs
        self.lr = lr
        self.trainLoader = trainLoader
        self.totalTrain = len(self.trainLoader.dataset) if trainLoader is not None else 0
        self.testLoader = testLoader
        self.totalTest = len(self.testLoader.dataset) if testLoader is not None else 0
        self.lrMilestones = lrMilestones
        self.descriptorDim = descriptorDim
        self.gpu = gpu
        if not torch.cuda.is_available():
            raise RuntimeError("SpatialNetwork needs an MI355X GPU: the hot path has no CPU fallback")
        self.device = torch.device("cuda", torch.cuda.current_device())
        if weights is None:
            weights = synth.synth_vgg16_weights(c_in=self.C_IN, n_classes=nActionClasses, desc_dim=descriptorDim,
                                                seed=seed, device=self.device)
        self.model = self._build(weights)
        self.startEpoch = 0
        self.epoch = 0
        self.highestPrecision = 0.0
        self.isBest = False
        self.ckpLoc = ckpLoc if ckpLoc is None or ckpLoc.endswith("/") else ckpLoc + "/"
        self.features = self.model.features
        self.classify = self.model.classify
        self.trainDict = {}
        self.testDict = {}
        self.testMeters = fusion.DescriptorMeters(descriptorDim, self.device)  # persists across epochs (quirk 7)

    def _build(self, weights):
        return vgg.Vgg16Stream(weights["conv_w"], weights["conv_b"], weights["fc_w"], weights["fc_b"],
                               self.nActionClasses, self.descriptorDim, NORM_MEANS_TF, NORM_STDS_TF,
                               device=self.device.index)

    def validate(self):
        """Sheet03/spatialModel.py:197-231: returns (correct / totalTest, summed per-batch mean CE)."""
        correct = 0
        loss = 0
        pending = []
        for iBatch, (data, labels, videoNames) in enumerate(self.testLoader):
            ip = data.to(self.device, non_blocking=True)
            op = self.features(ip)
            featureVectors, op = self.classify(op)
            pending.append(vgg.validate_batch(op, labels))  # [mean CE, n correct] on the device, no sync
            # per-video running means (Sheet03/spatialModel.py:223-228) accumulate in HBM: no copy per batch
            self.testMeters.update(featureVectors, videoNames, labels)
        for t in pending:
            v = t.cpu()
            loss = loss + v[0]
            correct += int(v[1].item())
        self.testDict = self.testMeters.as_dict()
        print("Validation for epoch %d: total = %d, correct = %d, loss = %f"
              % (self.epoch, self.totalTest, correct, float(loss)))
        return (correct / self.totalTest), loss

    def execute(self):
        """Inference-only ``execute``: one validation pass + the per-video descriptor CSV
        (Sheet03/spatialModel.py:274,283)."""
        precision, loss = self.validate()
        saveVideoDescriptors(self.testDict, SPATIAL_TEST_CSV_LOC, self.gpu)
        return precision, loss

    def train(self):
        raise NotImplementedError("training is outside the inference hot path (SURVEY.md section 8f)")

    def resume(self):
        raise NotImplementedError("checkpoint resume is training bookkeeping (SURVEY.md section 8f)")

    def save(self):
        raise NotImplementedError("checkpoint save is training bookkeeping (SURVEY.md section 8f)")
