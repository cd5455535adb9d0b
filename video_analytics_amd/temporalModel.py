"""Temporal stream: the reference's ``Sheet03/temporalModel.py`` call surface over the gfx950 kernels.

``TemporalDataset`` reads the precomputed ``flow_x_%04d.jpg`` / ``flow_y_%04d.jpg`` images exactly as
the reference does (Sheet03/temporalModel.py:67-92); ``flowVolumesFromFrames`` is the MI355X-native
replacement of that offline step (frames -> TV-L1 -> 2L-channel flow volume on the GPU).
"""
from __future__ import division

import os

import torch
from PIL import Image
from torch.utils.data import Dataset

from . import flow as vflow
from . import synth, vgg
from .parameters import *  # noqa: F401,F403
from .parameters import (FRAME_EXTN, TEMPORAL_TEST_CSV_LOC, VIDEO_INPUT_FLOW_COUNT, X_PREFIX_FLOW, Y_PREFIX_FLOW)
from .spatialModel import SpatialNetwork, _read_label_dict
from .utils import ToTensor, flowFileName, saveVideoDescriptors, temporalFlowIndices, videoInfo


class TemporalDataset(Dataset):
    """2L flow frames from a random start per video per invocation (Sheet03/temporalModel.py:22-92)."""

    def __init__(self, videoListLoc, rootDir, imageTransforms=None, flowSampleSize=VIDEO_INPUT_FLOW_COUNT,
                 mode="train", actionLabelLoc=None):
        super(TemporalDataset, self).__init__()
        self.rootDir = rootDir if rootDir.endswith("/") else rootDir + "/"
        self.imageTransforms = imageTransforms
        self.flowSampleSize = flowSampleSize
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
        actionLabel = int(actionLabel)
        flowDir = self.rootDir + actionCategory + "/" + videoName + "/"
        nFiles = len([fName for fName in os.listdir(flowDir)])
        _, order = temporalFlowIndices(nFiles, self.flowSampleSize)
        flowFrames = [flowDir + flowFileName(X_PREFIX_FLOW if ax == "x" else Y_PREFIX_FLOW, idx, FRAME_EXTN)
                      for (ax, idx) in order]
        # the transform is applied to each of the 2L images independently: 2L independent random
        # crops / flips (Sheet03/temporalModel.py:86, quirk 2)
        tf = self.imageTransforms if self.imageTransforms is not None else ToTensor()
        loadedFrames = [tf(Image.open(frame)) for frame in flowFrames]
        flowVolume = torch.squeeze(torch.stack(loadedFrames, dim=0))  # [2L,1,H,W] -> [2L,H,W]
        return flowVolume, actionLabel, videoName


def flowVolumesFromFrames(gray, flowSampleSize=VIDEO_INPUT_FLOW_COUNT, tvl1_params=None, bound=vflow.FLOW_BOUND):
    """gray CUDA u8/f32 ``[B, L+1, H, W]`` -> flow volumes f32 ``[B, 2L, H, W]`` on the GPU: TV-L1 on the
    L consecutive pairs, 8-bit quantisation, ToTensor+Normalize, x/y interleave -- the tensor
    ``TemporalDataset.__getitem__`` would have assembled from the upstream tool's JPEGs."""
    if gray.dim() != 4 or gray.shape[1] != flowSampleSize + 1:
        raise ValueError("flowVolumesFromFrames: gray must be [B,%d,H,W]" % (flowSampleSize + 1))
    B, _, H, W = gray.shape
    fl = vflow.tvl1_flow(gray, tvl1_params)
    return vflow.flow_to_stack(fl, bound=bound).view(B, 2 * flowSampleSize, H, W)


class TemporalNetwork(SpatialNetwork):
    """Wrapper of the motion stream (Sheet03/temporalModel.py:96-312)."""

    def __init__(self, nActionClasses, flowSampleSize, nEpochs, lr, momentumVal, descriptorDim, trainLoader, testLoader,
                 lrMilestones, ckpLoc, gpu=False, weights=None, seed=2):
        self.flowSampleSize = flowSampleSize
        self.C_IN = 2 * flowSampleSize
        if weights is None and torch.cuda.is_available():
            dev = torch.device("cuda", torch.cuda.current_device())
            weights = synth.synth_vgg16_weights(c_in=self.C_IN, n_classes=nActionClasses, desc_dim=descriptorDim,
                                                seed=seed, device=dev)
            weights["conv_w"][0] = self.__copyFirstLayer__(weights["conv_w"][0].to(dev))
        super(TemporalNetwork, self).__init__(nActionClasses, nEpochs, lr, momentumVal, descriptorDim, trainLoader,
                                              testLoader, lrMilestones, ckpLoc, gpu=gpu, weights=weights, seed=seed)

    def __copyFirstLayer__(self, w_rgb):
        """Sheet03/temporalModel.py:149-162: RGB-mean kernel replicated over the 2L input channels
        (the first-layer bias is NOT copied: whatever bias the weights carry is kept)."""
        return vgg.copy_first_layer(w_rgb, 2 * self.flowSampleSize)

    def _build(self, weights):
        if weights["conv_w"][0].shape[1] == 3:
            weights = dict(weights)
            weights["conv_w"] = [self.__copyFirstLayer__(weights["conv_w"][0].to(self.device))] + list(weights["conv_w"][1:])
        return vgg.Vgg16Stream(weights["conv_w"], weights["conv_b"], weights["fc_w"], weights["fc_b"],
                               self.nActionClasses, self.descriptorDim, device=self.device.index)

    CKP_FILE, BEST_FILE = MOTION_CKP_FILE, MOTION_BEST_FILE
    TRAIN_CSV, TEST_CSV, PERFORMANCE_CSV = TEMPORAL_TRAIN_CSV_LOC, TEMPORAL_TEST_CSV_LOC, TEMPORAL_PERFORMANCE_LOC


def main(weights=None):
    """The motion-stream script (Sheet03/temporalModel.py:315-324): datasets over the flow-image directory, loaders,
    network, ``execute()``."""
    from . import parameters as P
    from .utils import getDataLoader, getTransforms
    tf = getTransforms()
    L = P.VIDEO_INPUT_FLOW_COUNT
    trainSet = TemporalDataset(P.VIDEOLIST_TRAIN, P.FLOW_DATA_DIR, tf, flowSampleSize=L, actionLabelLoc=P.ACTIONLABEL_FILE)
    testSet = TemporalDataset(P.VIDEOLIST_TEST, P.FLOW_DATA_DIR, tf, flowSampleSize=L, mode="test",
                              actionLabelLoc=P.ACTIONLABEL_FILE)
    net = TemporalNetwork(P.NACTION_CLASSES, L, P.NEPOCHS, P.INITIAL_LR, P.MOMENTUM_VAL, P.VIDEO_DESCRIPTOR_DIM,
                          getDataLoader(trainSet, batchSize=P.TEMPORAL_BATCH_SIZE),
                          getDataLoader(testSet, batchSize=P.TEMPORAL_BATCH_SIZE), P.MILESTONES_LR, P.CHECKPOINT_DIR,
                          gpu=True, weights=weights)
    return net.execute()


if __name__ == "__main__":
    main()
