"""Host-side helpers of the two-stream path: the reference's ``Sheet03/utils.py`` surface.

Pure-Python string/indexing logic is bit-exact with the reference lines cited per function.
The image transforms restate the torchvision ~0.2 classes the reference composes
(``getTransforms``, Sheet03/utils.py:137-151) on numpy/torch, because torchvision is not part of this
stack; they draw from Python's global ``random`` module in the same order as that torchvision
generation (crop top, crop left, flip).  Training bookkeeping (``makeCheckpoint``, ``savePerformance``) and
frame extraction (``extractEveryNthFrame``, ``convertVideosToFrames``) are provided too; the reference decodes
with ``cv2.VideoCapture``, absent here, so ``iterVideoFrames`` reads RIFF/AVI Motion-JPEG itself and refuses
other codecs by name.
"""
from __future__ import division

import csv
import os
import shutil
import random

import numpy as np
import torch
from PIL import Image
from torch.utils.data import DataLoader

from .parameters import *  # noqa: F401,F403  (the reference star-imports its config the same way)
from .parameters import (COLOR_JITTERS, CROP_SIZE_TF, HORIZONTAL_FLIP_TF, NORM_MEANS_TF, NORM_STDS_TF,
                         NWORKERS_LOADER, SHUFFLE_LOADER, TEMPORAL_BATCH_SIZE)


def checkAndMakeDirectories(*args):
    """Create missing directories; item i is True if directory i already existed
    (Sheet03/utils.py:14-26)."""
    exists = [True] * len(args)
    for i, arg in enumerate(args):
        if not os.path.exists(arg):
            exists[i] = False
            os.makedirs(arg)
    return exists


def makeCheckpoint(modelState, isBest, ckpLoc, bestModel):
    """Save the training state; copy it to ``bestModel`` when ``isBest`` (Sheet03/utils.py:29-35)."""
    torch.save(modelState, ckpLoc)
    if isBest:
        shutil.copyfile(ckpLoc, bestModel)


def savePerformance(precision, loss, csvLoc):
    """Append ``precision,loss`` of one epoch (Sheet03/utils.py:198-205)."""
    with open(csvLoc, "a") as csvFile:
        csvFile.write(str(precision) + "," + str(loss) + "\n")


def multiStepLr(baseLr, milestones, lastEpoch, gamma=0.1):
    """``MultiStepLR.get_lr()`` of torch 0.4 (the reference's scheduler, Sheet03/spatialModel.py:119):
    baseLr * gamma ** bisect_right(milestones, lastEpoch).  ``lastEpoch`` is whatever was passed to
    ``scheduler.step(...)`` -- the reference passes the validation LOSS (Sheet03/spatialModel.py:278, quirk)."""
    import bisect
    return baseLr * gamma ** bisect.bisect_right(sorted(milestones), lastEpoch)


def iterVideoFrames(videoLoc):
    """Decoded frames (PIL RGB images) of a video file, in order.  The reference decodes with ``cv2.VideoCapture``
    (Sheet03/utils.py:57-68); OpenCV is not in this image, so the container is parsed here: RIFF/AVI with
    Motion-JPEG streams (``MJPG``: every ``##dc`` / ``##db`` chunk of the ``movi`` list is one JPEG, decoded with
    PIL).  Any other codec (UCF-101 ships XviD) raises ValueError naming its fourcc: there is no decoder for it
    here."""
    import io
    import struct
    with open(videoLoc, "rb") as f:
        data = f.read()
    if len(data) < 12 or data[:4] != b"RIFF" or data[8:12] != b"AVI ":
        raise ValueError("iterVideoFrames: %s is not a RIFF/AVI file" % videoLoc)
    fourcc = None
    k = data.find(b"strh")
    while k >= 0:  # the video stream header: 'strh' <size> 'vids' <handler fourcc>
        if data[k + 8:k + 12] == b"vids":
            fourcc = data[k + 12:k + 16]
            break
        k = data.find(b"strh", k + 4)
    if fourcc is None or fourcc.upper() not in (b"MJPG", b"JPEG"):
        raise ValueError("iterVideoFrames: no decoder for video codec %r of %s in this image (OpenCV is absent; "
                         "Motion-JPEG AVI files are supported)" % (fourcc, videoLoc))
    k = data.find(b"movi")
    if k < 0:
        raise ValueError("iterVideoFrames: %s has no movi list" % videoLoc)
    end = k - 4 + 8 + struct.unpack("<I", data[k - 4:k])[0]
    k += 4
    while k + 8 <= min(end, len(data)):
        cid, size = data[k:k + 4], struct.unpack("<I", data[k + 4:k + 8])[0]
        if cid == b"LIST":  # 'rec ' groups: descend
            k += 12
            continue
        if cid[2:4] in (b"dc", b"db") and size > 0:
            yield Image.open(io.BytesIO(data[k + 8:k + 8 + size])).convert("RGB")
        k += 8 + size + (size & 1)


def extractEveryNthFrame(videoLoc, N):
    """Every N-th frame (0, N, 2N, ...) of the video at ``videoLoc`` (Sheet03/utils.py:51-69); ValueError if the
    file does not exist."""
    if not (os.path.exists(videoLoc) and os.path.isfile(videoLoc)):
        raise ValueError("Video does not exist: %s" % (videoLoc))
    frameList = []
    for frameIdx, frame in enumerate(iterVideoFrames(videoLoc)):
        if frameIdx % N == 0:
            frameList.append(frame)
    return frameList


def convertVideosToFrames(rootDir, saveDir, videoListLoc, sampleRate=VIDEO_FRAME_SAMPLE_RATE, mode="train"):
    """Sheet03/utils.py:95-121: every ``sampleRate``-th frame of each listed video goes to
    ``saveDir/<category>/<videoName>/<i>.jpg`` (i = 0, 1, ... counts the SAMPLED frames: the names
    ``SpatialDataset.__getitem__`` opens, Sheet03/spatialModel.py:75-77).  A video whose frame directory already
    exists is skipped, like in the reference (it takes an existing directory as 'frames written already')."""
    if not rootDir.endswith("/"):
        rootDir = rootDir + "/"
    if not saveDir.endswith("/"):
        saveDir = saveDir + "/"
    with open(videoListLoc, "r") as videoListFile:
        for line in videoListFile:
            videoLoc, videoName, _, actionCategory, _, _ = videoInfo(line, mode)
            videoLoc = rootDir + videoLoc
            frameDir = saveDir + actionCategory + "/" + videoName
            if all(checkAndMakeDirectories(frameDir)):
                continue
            frameList = extractEveryNthFrame(videoLoc, sampleRate)
            for i in range(len(frameList)):
                frameList[i].save(frameDir + "/" + str(i) + FRAME_EXTN, quality=95)  # cv2.imwrite's default JPEG quality
    return


def videoInfo(line, mode):
    """Parse one video-list line (Sheet03/utils.py:73-91).

    Returns (videoLoc, videoName, actionLabel or None, actionCategory, ngroup, nclip).
    train lines are ``"Cat/v_Cat_gNN_cNN.avi K"`` (exactly one space), test lines have no label.
    Raises ValueError exactly where the reference's tuple unpacking does (wrong number of
    ``" "``, ``"/"`` or ``"_"`` separated parts).
    """
    actionLabel = None
    if mode == "train":
        videoLoc, actionLabel = line.split(" ")
        actionLabel = actionLabel.strip()
    else:
        videoLoc = line
    videoLoc = videoLoc.strip()
    actionCategory, videoName = videoLoc.split("/")
    actionCategory = actionCategory.strip()
    videoName = videoName[:videoName.rfind(".")]
    _, _, ngroup, nclip = videoName.split("_")
    return videoLoc, videoName, actionLabel, actionCategory, ngroup, nclip


def spatialFrameIndex(nFrames, r=None):
    """Index of the frame ``SpatialDataset.__getitem__`` opens: ``random.randint(0, nFrames-1)``,
    both ends inclusive (Sheet03/spatialModel.py:74-77).  ``r`` overrides the draw (tests)."""
    if nFrames < 1:
        raise ValueError("empty range for randrange() (0, %d, %d)" % (nFrames, nFrames))
    return random.randint(0, nFrames - 1) if r is None else r


def temporalFlowIndices(nFiles, flowSampleSize, r=None):
    """Start index and the 2L interleaved (prefix, index) pairs ``TemporalDataset.__getitem__`` opens
    (Sheet03/temporalModel.py:78-83).

    ``nFlows = nFiles / 2`` is a TRUE division in the reference (``from __future__ import division``)
    and Python 2's ``randint`` rejects a non-integral float bound, so an odd file count raises
    ValueError; the start lies in [1, nFlows - L] (the last flow index nFlows is never read).
    Order: x_s, y_s, x_{s+1}, y_{s+1}, ... (exactly 2L entries).
    """
    nFlows = nFiles / 2
    if nFlows != int(nFlows):
        raise ValueError("non-integer stop for randrange()")
    nFlows = int(nFlows)
    hi = nFlows - flowSampleSize
    if hi < 1:
        raise ValueError("empty range for randrange() (1, %d, %d)" % (hi + 1, hi))
    start = random.randint(1, hi) if r is None else r
    order = []
    for idx in range(start, start + flowSampleSize):
        order.append(("x", idx))
        order.append(("y", idx))
    return start, order


def flowFileName(prefix, idx, extn=".jpg"):
    """``flow_x_0007.jpg`` naming: prefix + 4-digit zero-padded 1-based index (Sheet03/temporalModel.py:80-81)."""
    return prefix + str(idx).zfill(4) + extn


def flowToImages(flow, bound=20.0):
    """flow ``[N,2,H,W]`` float32 (torch or numpy) -> uint8 ``[N,2,H,W]``: the 8-bit flow image
    convention the reference's flow JPEGs follow, ``q = rint(clamp(255*(v+bound)/(2*bound), 0, 255))``
    (the quantisation step of ``va_flow_to_stack``)."""
    f = flow.detach().cpu().numpy() if isinstance(flow, torch.Tensor) else np.asarray(flow)
    f = f.astype(np.float32)
    t = (np.float32(255.0) * (f + np.float32(bound))) / np.float32(2.0 * bound)
    return np.rint(np.clip(t, 0.0, 255.0)).astype(np.uint8)


def saveFlowImages(flow, flowDir, bound=20.0, firstIndex=1, quality=95):
    """Write the flow of the N consecutive pairs of one video as the files the reference reads:
    ``flow_x_%04d.jpg`` / ``flow_y_%04d.jpg``, 1-based, single-channel 'L' JPEGs
    (Sheet03/temporalModel.py:78-81, Sheet03/parameters.py:38-39).  Returns the file count."""
    from PIL import Image
    from .parameters import FRAME_EXTN, X_PREFIX_FLOW, Y_PREFIX_FLOW
    checkAndMakeDirectories(flowDir)
    q = flowToImages(flow, bound)
    for k in range(q.shape[0]):
        for prefix, plane in ((X_PREFIX_FLOW, 0), (Y_PREFIX_FLOW, 1)):
            Image.fromarray(q[k, plane], mode="L").save(os.path.join(flowDir, flowFileName(prefix, firstIndex + k, FRAME_EXTN)),
                                                         quality=quality)
    return 2 * q.shape[0]


# ----------------------------------------------------------------------------- transforms ------

class RandomCrop(object):
    """``transforms.RandomCrop(224)``: top = randint(0, h-th), left = randint(0, w-tw)."""

    def __init__(self, size):
        self.size = (int(size), int(size))

    def __call__(self, img):
        h, w = img.shape[0], img.shape[1]
        th, tw = self.size
        if w == tw and h == th:
            return img
        if h < th or w < tw:
            raise ValueError("Required crop size %s is larger than input image size %s" % ((th, tw), (h, w)))
        i = random.randint(0, h - th)
        j = random.randint(0, w - tw)
        return img[i:i + th, j:j + tw]


class RandomHorizontalFlip(object):
    def __init__(self, p=0.5):
        self.p = p

    def __call__(self, img):
        if random.random() < self.p:
            return img[:, ::-1]
        return img


class ColorJitter(object):
    """``ColorJitter(0,0,0,0)`` (Sheet03/parameters.py:21) is the identity and draws no random numbers."""

    def __init__(self, brightness=0, contrast=0, saturation=0, hue=0):
        if brightness or contrast or saturation or hue:
            raise NotImplementedError("only the reference's ColorJitter(0,0,0,0) (identity) is supported")

    def __call__(self, img):
        return img


class ToTensor(object):
    """u8 HWC (or HW) -> float32 CHW / 255."""

    def __call__(self, img):
        a = np.ascontiguousarray(img)
        if a.ndim == 2:
            a = a[:, :, None]
        t = torch.from_numpy(a.transpose(2, 0, 1).copy())
        return t.to(torch.float32).div(255)


class Normalize(object):
    """Per-channel (t - mean)/std over ``zip(tensor, mean, std)``: a 1-channel flow image therefore
    uses only the FIRST mean/std pair (0.485 / 0.229) -- SURVEY.md a5, appendix quirk 3."""

    def __init__(self, mean, std):
        self.mean, self.std = list(mean), list(std)

    def __call__(self, t):
        out = t.clone()
        for c, (m, s) in enumerate(zip(self.mean, self.std)):
            if c >= out.shape[0]:
                break
            out[c] = (out[c] - m) / s
        return out


class Compose(object):
    def __init__(self, transforms):
        self.transforms = list(transforms)

    def __call__(self, img):
        img = np.asarray(img)
        for t in self.transforms:
            img = t(img)
        return img


def getTransforms(cropSize=CROP_SIZE_TF, hortizontalFlip=HORIZONTAL_FLIP_TF, normMeans=NORM_MEANS_TF,
                  normStds=NORM_STDS_TF, jitter=COLOR_JITTERS):
    """Sheet03/utils.py:137-151.  The crop is the literal 224 of the reference whatever ``cropSize``
    says (``:143``); the same random transforms are used at test time (Sheet03/spatialModel.py:293-294)."""
    imgTrans = []
    if cropSize:
        imgTrans.append(RandomCrop(224))
    if hortizontalFlip:
        imgTrans.append(RandomHorizontalFlip())
    if jitter:
        imgTrans.append(ColorJitter(*jitter))
    imgTrans.append(ToTensor())
    if normMeans and normStds:
        imgTrans.append(Normalize(mean=normMeans, std=normStds))
    return Compose(imgTrans)


def getDataLoader(dataset, batchSize=TEMPORAL_BATCH_SIZE, nWorkers=NWORKERS_LOADER, shuffle=SHUFFLE_LOADER):
    """Sheet03/utils.py:125-133: default collate, last batch partial, shuffle on (also for test)."""
    return DataLoader(dataset=dataset, batch_size=batchSize, shuffle=shuffle, num_workers=nWorkers)


class AverageMeter(object):
    """Running mean (Sheet03/utils.py:154-171)."""

    def __init__(self):
        self.reset()

    def reset(self):
        self.val = 0
        self.avg = 0
        self.sum = 0
        self.count = 0

    def update(self, val, n=1):
        self.val = val
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count


def saveVideoDescriptors(videoDescDict, csvLoc, gpu=False):
    """One CSV row per video: ``name,label,`` then the descriptor as float64 reprs
    (Sheet03/utils.py:174-195; the width is VIDEO_DESCRIPTOR_DIM = 256, not the docstring's 4096)."""
    try:
        os.remove(csvLoc)
    except OSError:
        pass
    with open(csvLoc, "a") as csvFile:
        writer = csv.writer(csvFile, delimiter=",")
        for videoName in videoDescDict.keys():
            label = videoDescDict[videoName][1]
            videoLabel = label.cpu().numpy() if isinstance(label, torch.Tensor) else np.asarray(label)
            csvFile.write(videoName + "," + str(videoLabel) + ",")
            videoDesc = videoDescDict[videoName][0].avg
            videoDesc = videoDesc.detach().cpu().numpy().astype(float)
            writer.writerow(videoDesc)
