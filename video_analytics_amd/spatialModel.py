"""Spatial stream: the reference's ``Sheet03/spatialModel.py`` call surface over the gfx950 kernels.

``SpatialDataset`` keeps the reference's indexing and file layout bit for bit; ``SpatialNetwork``
keeps the constructor signature and ``validate()`` (the inference hot loop,
Sheet03/spatialModel.py:197-231).  The forward pass runs in libva_hip.so (``vgg.Vgg16Stream``).
``train`` / ``resume`` / ``save`` / ``execute`` (Sheet03/spatialModel.py:157-194, 234-283) run the training
step on the GPU as well (``Vgg16Stream.train_step``: SURVEY.md section 8f rank 4).
"""
from __future__ import division

import os
import time
import random  # noqa: F401  (the dataset draws from the global random state, like the reference)

import torch
from PIL import Image
from torch.utils.data import Dataset

from . import fusion, synth, vgg
from .parameters import *  # noqa: F401,F403
from .parameters import (FRAME_EXTN, NORM_MEANS_TF, NORM_STDS_TF, SPATIAL_TEST_CSV_LOC, VIDEO_INPUT_FRAME_COUNT)
from .utils import (AverageMeter, ToTensor, checkAndMakeDirectories, makeCheckpoint, multiStepLr, savePerformance,
                    saveVideoDescriptors, spatialFrameIndex, videoInfo)


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
    """Wrapper of the spatial stream (Sheet03/spatialModel.py:85-283)."""

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
        self.momentumVal = momentumVal
        self.schedulerLastEpoch = 0  # MultiStepLR(..., last_epoch=-1) steps once on construction
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
        self.classifierList = self.model.classifier_list()  # the reference's indexable surface (:128-129), over the fused head
        self.classifierLen = len(self.classifierList)
        self.trainDict = {}
        self.testDict = {}
        self.testMeters = fusion.DescriptorMeters(descriptorDim, self.device)  # persists across epochs (quirk 7)
        self.trainMeters = fusion.DescriptorMeters(descriptorDim, self.device)

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

    # ---- training: Sheet03/spatialModel.py:157-194, 234-283 ----

    CKP_FILE, BEST_FILE = SPATIAL_CKP_FILE, SPATIAL_BEST_FILE
    TRAIN_CSV, TEST_CSV, PERFORMANCE_CSV = SPATIAL_TRAIN_CSV_LOC, SPATIAL_TEST_CSV_LOC, SPATIAL_PERFORMANCE_LOC

    def currentLr(self):
        """The optimiser's learning rate: MultiStepLR over what the reference feeds it (see ``multiStepLr``)."""
        return multiStepLr(self.lr, self.lrMilestones, self.schedulerLastEpoch)

    def train(self):
        """Train for an epoch (Sheet03/spatialModel.py:157-194): every batch is one fused forward / backward /
        SGD step on the GPU (``va_vgg16_train_step``); the train-mode descriptors are collated per video
        (quirk 7 of SURVEY.md: Dropout is active in them)."""
        if self.trainLoader is None:
            raise ValueError("train(): no trainLoader")
        startTime = time.time()
        lr = self.currentLr()
        pending = []
        for iBatch, (data, labels, videoNames) in enumerate(self.trainLoader):
            ip = data.to(self.device, non_blocking=True)
            # Dropout masks: the reference draws them from torch's global generator; here they are a pure function
            # of (epoch, batch index) so that a run can be reproduced (and re-derived on the CPU by the tests)
            stats, featureVectors = self.model.train_step(ip, labels, lr, self.momentumVal, self.epoch * 1000003 + iBatch)
            pending.append(stats)
            self.trainMeters.update(featureVectors, videoNames, labels)
        # (Sheet03/spatialModel.py:190 clips the gradient norm AFTER the last optimizer.step() of the epoch: it never
        # changes an update, so there is nothing to do here.)
        self.trainDict = self.trainMeters.as_dict()
        self.lastTrainStats = [t.cpu() for t in pending]
        print("Epoch %d completed in %f seconds" % (self.epoch, time.time() - startTime))
        self.save()

    def state(self):
        """What the reference checkpoints (Sheet03/spatialModel.py:256-261): epoch, model.state_dict() (``module.``
        keys), highestPrecision, optimizer.state_dict() (momentum buffers in parameter order + the param group)."""
        params = self.model.export_state()
        names = list(vgg.state_dict_from_weights(params).keys())
        opt = {"state": {}, "param_groups": [{"lr": self.currentLr(), "initial_lr": self.lr, "momentum": self.momentumVal,
                                              "dampening": 0, "weight_decay": 0, "nesterov": False,
                                              "params": list(range(len(names)))}]}
        if getattr(self.model, "_train_ready", False):
            mom = list(vgg.state_dict_from_weights(self.model.export_state(momentum=True)).values())
            opt["state"] = {i: {"momentum_buffer": t.cpu()} for i, t in enumerate(mom)}
        return {"epoch": self.epoch,
                "model": type(vgg.state_dict_from_weights(params))((k, v.cpu()) for k, v in vgg.state_dict_from_weights(params).items()),
                "highestPrecision": self.highestPrecision, "optimizer": opt,
                "schedulerLastEpoch": self.schedulerLastEpoch}

    def save(self):
        """Sheet03/spatialModel.py:252-262."""
        if self.ckpLoc is None:
            return
        checkAndMakeDirectories(self.ckpLoc)
        makeCheckpoint(self.state(), self.isBest, self.ckpLoc + self.CKP_FILE, self.ckpLoc + self.BEST_FILE)

    def resume(self):
        """Sheet03/spatialModel.py:234-250: continue from ``ckpLoc + CKP_FILE`` if it exists."""
        resumeLoc = None if self.ckpLoc is None else self.ckpLoc + self.CKP_FILE
        if not (resumeLoc and os.path.isfile(resumeLoc)):
            print("No checkpoints found; starting from scratch!")
            return False
        print("Resuming training from checkpoint file: %s" % resumeLoc)
        checkpoint = torch.load(resumeLoc, map_location="cpu", weights_only=True)
        self.startEpoch = checkpoint["epoch"] + 1
        self.highestPrecision = checkpoint["highestPrecision"]
        self.model.import_state(vgg.weights_from_state_dict(checkpoint["model"]))
        st = checkpoint["optimizer"]["state"]
        if len(st):
            names = list(checkpoint["model"].keys())
            self.model.import_state(vgg.weights_from_state_dict({n: st[i]["momentum_buffer"] for i, n in enumerate(names)}),
                                    momentum=True)
        # the reference rebuilds MultiStepLR with last_epoch = startEpoch (Sheet03/spatialModel.py:248)
        self.schedulerLastEpoch = self.startEpoch
        print("Loaded checkpoint: starting from epoch: %d" % self.startEpoch)
        return True

    def execute(self):
        """All epochs, each = train + validate (Sheet03/spatialModel.py:265-283).  Without a trainLoader this is
        the inference-only pass: one validation + the per-video descriptor CSV."""
        if self.trainLoader is None:
            precision, loss = self.validate()
            saveVideoDescriptors(self.testDict, self.TEST_CSV, self.gpu)
            return precision, loss
        self.resume()
        precision, loss = None, None
        for self.epoch in range(self.startEpoch, self.nEpochs):
            self.train()
            precision, loss = self.validate()
            if precision > self.highestPrecision:
                self.highestPrecision = precision
                self.isBest = True  # (never reset in the reference either: every later save also refreshes the best file)
            self.schedulerLastEpoch = float(loss)  # scheduler.step(loss): the loss lands where the epoch belongs (quirk)
            self.save()
            savePerformance(precision, float(loss), self.PERFORMANCE_CSV)
            saveVideoDescriptors(self.trainDict, self.TRAIN_CSV, self.gpu)
            saveVideoDescriptors(self.testDict, self.TEST_CSV, self.gpu)
        return precision, loss


def main(weights=None):
    """The spatial-stream script (Sheet03/spatialModel.py:286-298): optional frame extraction, the train and test
    datasets over the frame directories of ``parameters.py`` (the SAME random transforms for test as for train), their
    loaders, the network, ``execute()``.  ``weights``: see ``SpatialNetwork`` (the reference downloads ImageNet weights)."""
    from . import parameters as P
    from .utils import convertVideosToFrames, getDataLoader, getTransforms
    if P.CONVERT:
        convertVideosToFrames(P.DATA_DIR, P.FRAMES_DIR_TEST, P.VIDEOLIST_TEST, mode="test")
    tf = getTransforms()
    trainSet = SpatialDataset(P.VIDEOLIST_TRAIN, P.FRAMES_DIR_TRAIN, tf, frameSampleSize=2, actionLabelLoc=P.ACTIONLABEL_FILE)
    testSet = SpatialDataset(P.VIDEOLIST_TEST, P.FRAMES_DIR_TEST, tf, mode="test", actionLabelLoc=P.ACTIONLABEL_FILE)
    net = SpatialNetwork(P.NACTION_CLASSES, P.NEPOCHS, P.INITIAL_LR, P.MOMENTUM_VAL, P.VIDEO_DESCRIPTOR_DIM,
                         getDataLoader(trainSet, batchSize=P.SPATIAL_BATCH_SIZE),
                         getDataLoader(testSet, batchSize=P.SPATIAL_BATCH_SIZE), P.MILESTONES_LR, P.CHECKPOINT_DIR,
                         gpu=True, weights=weights)  # gpu=True is hard-coded in the reference as well (:297)
    return net.execute()


if __name__ == "__main__":
    main()
