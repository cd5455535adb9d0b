"""Host logic of the path (pure Python, CPU): bit-exact known-answer tests derived by hand from the
reference lines cited in video_analytics_amd/utils.py (the reference ships no tests: SURVEY.md section 4)."""
import csv
import os
import random

import numpy as np
import pytest
import torch

from video_analytics_amd import utils as U
from video_analytics_amd import combinedModel, parameters

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_parameters_match_reference_values():
    # Sheet03/parameters.py:2-21
    assert parameters.VIDEO_INPUT_FLOW_COUNT == 10 and parameters.SPATIAL_BATCH_SIZE == 60
    assert parameters.TEMPORAL_BATCH_SIZE == 32 and parameters.NWORKERS_LOADER == 4 and parameters.SHUFFLE_LOADER is True
    assert parameters.NORM_MEANS_TF == [0.485, 0.456, 0.406] and parameters.NORM_STDS_TF == [0.229, 0.224, 0.225]
    assert parameters.NACTION_CLASSES == 101 and parameters.VIDEO_DESCRIPTOR_DIM == 256
    assert parameters.X_PREFIX_FLOW == "flow_x_" and parameters.Y_PREFIX_FLOW == "flow_y_" and parameters.FRAME_EXTN == ".jpg"
    assert parameters.COLOR_JITTERS == [0, 0, 0, 0] and parameters.CROP_SIZE_TF == 224


def test_every_reference_constant_is_kept():
    import json
    ref = json.load(open(os.path.join(GOLD, "reference_parameters.json")))
    assert len(ref) == 43
    site_specific = {"DATA_DIR", "FLOW_DATA_DIR", "FRAMES_DIR_TRAIN", "FRAMES_DIR_TEST", "VIDEOLIST_TRAIN", "VIDEOLIST_TEST",
                     "ACTIONLABEL_FILE", "CHECKPOINT_DIR"}  # absolute paths of the authors' machine (:26-33)
    for name, value in ref.items():
        assert hasattr(parameters, name), name
        if name in site_specific:
            assert os.path.basename(getattr(parameters, name).rstrip("/")) == os.path.basename(value.rstrip("/")), name
        else:
            assert getattr(parameters, name) == value, name


def test_videoinfo_known_answers_from_the_reference_lists():
    test_lines = open(os.path.join(GOLD, "demoTest.txt")).readlines()
    train_lines = open(os.path.join(GOLD, "demoTrain.txt")).readlines()
    assert len(test_lines) == 951 and len(train_lines) == 2409
    assert U.videoInfo(test_lines[0], "test") == (
        "ApplyEyeMakeup/v_ApplyEyeMakeup_g01_c01.avi", "v_ApplyEyeMakeup_g01_c01", None, "ApplyEyeMakeup", "g01", "c01")
    assert U.videoInfo(test_lines[950], "test") == (
        "CuttingInKitchen/v_CuttingInKitchen_g07_c04.avi", "v_CuttingInKitchen_g07_c04", None, "CuttingInKitchen", "g07", "c04")
    assert U.videoInfo(train_lines[0], "train") == (
        "ApplyEyeMakeup/v_ApplyEyeMakeup_g08_c01.avi", "v_ApplyEyeMakeup_g08_c01", "1", "ApplyEyeMakeup", "g08", "c01")
    # every line of both lists parses; labels are the raw 1-based indices 1..25 (quirk 4)
    labels = set()
    for l in train_lines:
        labels.add(int(U.videoInfo(l, "train")[2]))
    assert labels == set(range(1, 26))
    cats = {U.videoInfo(l, "test")[3] for l in test_lines}
    assert len(cats) == 25


def test_videoinfo_error_behaviour():
    with pytest.raises(ValueError):
        U.videoInfo("A/v_A_g01_c01.avi", "train")  # no label: tuple unpacking fails as in the reference
    with pytest.raises(ValueError):
        U.videoInfo("A/B/v_A_g01_c01.avi", "test")  # three path parts
    with pytest.raises(ValueError):
        U.videoInfo("A/v_A_extra_g01_c01.avi", "test")  # five '_' parts


def test_spatial_frame_index_is_inclusive_on_both_ends():
    random.seed(0)
    seen = {U.spatialFrameIndex(4) for _ in range(400)}
    assert seen == {0, 1, 2, 3}
    assert U.spatialFrameIndex(17, r=16) == 16
    with pytest.raises(ValueError):
        U.spatialFrameIndex(0)


def test_temporal_flow_indices():
    # 60 files = 30 flows per axis, L = 10: start in [1, 20]; the last index 30 is never read (quirk 8)
    random.seed(1)
    starts = {U.temporalFlowIndices(60, 10)[0] for _ in range(2000)}
    assert starts == set(range(1, 21))
    start, order = U.temporalFlowIndices(60, 10, r=7)
    assert start == 7 and len(order) == 20
    assert order[:4] == [("x", 7), ("y", 7), ("x", 8), ("y", 8)] and order[-1] == ("y", 16)
    assert U.flowFileName("flow_x_", 7) == "flow_x_0007.jpg" and U.flowFileName("flow_y_", 123) == "flow_y_0123.jpg"
    with pytest.raises(ValueError):
        U.temporalFlowIndices(61, 10)  # odd file count: non-integral float bound in py2's randint
    with pytest.raises(ValueError):
        U.temporalFlowIndices(20, 10)  # nFlows - L = 0: empty range


def test_transform_random_draw_order_and_ranges():
    img = np.arange(240 * 320 * 3, dtype=np.uint32).reshape(240, 320, 3).astype(np.uint8)
    random.seed(5)
    i, j, f = random.randint(0, 16), random.randint(0, 96), random.random()
    random.seed(5)
    out = U.Compose([U.RandomCrop(224), U.RandomHorizontalFlip()])(img)
    ref = img[i:i + 224, j:j + 224]
    if f < 0.5:
        ref = ref[:, ::-1]
    assert np.array_equal(out, ref)
    # UCF frames are 320x240: crop offsets top in [0,16], left in [0,96]
    random.seed(0)
    tops, lefts = set(), set()
    for _ in range(3000):
        s = random.getstate()
        tops.add(random.randint(0, 16)); lefts.add(random.randint(0, 96))
        random.setstate(s)
        U.RandomCrop(224)(img)
    assert tops == set(range(17)) and max(lefts) == 96 and min(lefts) == 0


def test_totensor_normalize_three_channel_and_single_channel_rule():
    img = np.array([[[0, 128, 255]]], dtype=np.uint8)  # 1x1 RGB
    t = U.Compose([U.ToTensor(), U.Normalize(parameters.NORM_MEANS_TF, parameters.NORM_STDS_TF)])(img)
    exp = [(0 / 255 - 0.485) / 0.229, (128 / 255 - 0.456) / 0.224, (255 / 255 - 0.406) / 0.225]
    assert t.shape == (3, 1, 1) and np.allclose(t.flatten().numpy(), exp, atol=1e-6)
    g = np.array([[100]], dtype=np.uint8)  # 'L' flow image: only the first mean/std pair applies
    t1 = U.Compose([U.ToTensor(), U.Normalize(parameters.NORM_MEANS_TF, parameters.NORM_STDS_TF)])(g)
    assert t1.shape == (1, 1, 1) and abs(float(t1) - (100 / 255 - 0.485) / 0.229) < 1e-6


def test_gettransforms_uses_literal_224_and_identity_jitter():
    tf = U.getTransforms(cropSize=100)
    assert isinstance(tf.transforms[0], U.RandomCrop) and tf.transforms[0].size == (224, 224)
    random.seed(3)
    out = tf(np.zeros((240, 320, 3), dtype=np.uint8))
    assert out.shape == (3, 224, 224) and out.dtype == torch.float32
    with pytest.raises(ValueError):
        tf(np.zeros((100, 100, 3), dtype=np.uint8))


def test_average_meter_and_descriptor_csv(tmp_path):
    m = U.AverageMeter()
    m.update(torch.tensor([1.0, 3.0])); m.update(torch.tensor([3.0, 5.0]))
    assert m.count == 2 and torch.equal(m.avg, torch.tensor([2.0, 4.0]))
    d = {"v_A_g01_c01": (U.AverageMeter(), torch.tensor(3)), "v_B_g01_c02": (U.AverageMeter(), torch.tensor(7))}
    d["v_A_g01_c01"][0].update(torch.arange(256, dtype=torch.float32) * 0.5)
    d["v_B_g01_c02"][0].update(torch.ones(256))
    p = str(tmp_path / "desc.csv")
    U.saveVideoDescriptors(d, p)
    rows = list(csv.reader(open(p)))
    assert len(rows) == 2 and len(rows[0]) == 258
    assert rows[0][0] == "v_A_g01_c01" and rows[0][1] == "3" and float(rows[0][3]) == 0.5
    assert rows[1][0] == "v_B_g01_c02" and rows[1][1] == "7" and rows[1][2] == "1.0"


def _write_csv(path, rows):
    with open(path, "w") as f:
        for name, label, val in rows:
            f.write(name + "," + str(label) + "," + ",".join(repr(float(val + k)) for k in range(256)) + "\n")


def test_combine_descriptors_join_order_and_layout(tmp_path):
    s, t = str(tmp_path / "s.csv"), str(tmp_path / "t.csv")
    _write_csv(s, [("v_a", 1, 0.0), ("v_b", 2, 1000.0), ("v_c", 3, 2000.0)])
    _write_csv(t, [("v_c", 3, -2000.0), ("v_a", 1, -1.0), ("v_x", 9, 5.0)])  # different order, one unmatched each
    X, y = combinedModel.combineDescriptors(s, t)
    assert X.shape == (2, 512) and list(y) == [1, 3]  # spatial order, inner join, spatial labels
    assert X[0, 0] == 0.0 and X[0, 255] == 255.0 and X[0, 256] == -1.0 and X[1, 0] == 2000.0 and X[1, 256] == -2000.0
    assert combinedModel.accuracy(np.array([1, 3]), y) == 100.0 and combinedModel.accuracy(np.array([1, 2]), y) == 50.0


def test_datasets_read_the_reference_directory_layout(tmp_path):
    from PIL import Image
    from video_analytics_amd.parameters import NORM_MEANS_TF, NORM_STDS_TF
    from video_analytics_amd.spatialModel import SpatialDataset
    from video_analytics_amd.temporalModel import TemporalDataset
    lst = tmp_path / "list.txt"
    lst.write_text("Archery/v_Archery_g01_c01.avi\nBiking/v_Biking_g02_c03.avi\n")
    lab = tmp_path / "classInd.txt"
    lab.write_text("1 Archery\n2 Biking\n")

    def flow_base(ax, i):  # every flow image has its own gray level: the file a channel came from can be read off it
        return (8 if ax == "x" else 128) + 8 * i

    ramp = (np.arange(320) // 32).astype(np.uint8)  # + a step pattern along x that shows where an image was cropped / flipped
    for cat, vid in (("Archery", "v_Archery_g01_c01"), ("Biking", "v_Biking_g02_c03")):
        fd = tmp_path / "frames" / cat / vid
        fd.mkdir(parents=True)
        for i in range(3):
            Image.fromarray(np.full((240, 320, 3), 40 * i + 10, dtype=np.uint8)).save(str(fd / ("%d.jpg" % i)), quality=100)
        wd = tmp_path / "flow" / cat / vid
        wd.mkdir(parents=True)
        for i in range(1, 14):
            for ax in ("x", "y"):
                img = np.broadcast_to(flow_base(ax, i) + ramp[None, :], (240, 320)).astype(np.uint8)
                Image.fromarray(img, mode="L").save(str(wd / ("flow_%s_%04d.png" % (ax, i))), format="PNG")
                os.rename(str(wd / ("flow_%s_%04d.png" % (ax, i))), str(wd / ("flow_%s_%04d.jpg" % (ax, i))))  # lossless content, reference name
    tf = U.getTransforms()
    random.seed(0)
    ds = SpatialDataset(str(lst), str(tmp_path / "frames"), tf, mode="test", actionLabelLoc=str(lab))
    assert len(ds) == 2
    x, label, name = ds[1]
    assert x.shape == (3, 224, 224) and label == 2 and name == "v_Biking_g02_c03"
    # --- TemporalDataset (Sheet03/temporalModel.py:67-92): which files, in which order, one crop / flip per image
    opened = []

    class Recording(object):  # the reference's transform chain, noting every image it is handed
        def __call__(self, img):
            opened.append(os.path.basename(img.filename))
            return tf(img)

    dt = TemporalDataset(str(lst), str(tmp_path / "flow"), Recording(), flowSampleSize=10, mode="test", actionLabelLoc=str(lab))
    for seed in (0, 1, 2, 3):
        random.seed(seed)
        start = random.randint(1, 13 - 10)  # the dataset's FIRST draw: start in [1, nFlows - L] = [1, 3]
        random.seed(seed)
        del opened[:]
        v, label, name = dt[0]
        assert v.shape == (20, 224, 224) and label == 1 and name == "v_Archery_g01_c01"
        want = [("flow_%s_%04d.jpg" % (ax, start + k)) for k in range(10) for ax in ("x", "y")]
        assert opened == want  # x_s, y_s, x_{s+1}, y_{s+1}, ...: exactly 2L files, none beyond start + L - 1
        q = v * NORM_STDS_TF[0] + NORM_MEANS_TF[0]  # undo the single-channel Normalize -> q / 255
        profiles = set()
        for k in range(10):
            for j, ax in enumerate(("x", "y")):
                row = (q[2 * k + j, 0] * 255.0).round().to(torch.int64) - flow_base(ax, start + k)
                assert 0 <= int(row.min()) and int(row.max()) <= 9, (k, ax)  # channel 2k + j IS flow_<ax>_<start + k>
                assert torch.equal((q[2 * k + j] * 255.0).round().to(torch.int64) - flow_base(ax, start + k),
                                   row[None, :].expand(224, 224))  # (columns only: the step pattern)
                profiles.add(tuple(row.tolist()))
        assert len(profiles) >= 8  # 20 independent (left offset, flip) draws, not one crop for the whole volume (quirk 2)
    with pytest.raises(ValueError):
        SpatialDataset(str(lst), str(tmp_path / "frames"), tf, mode="test")  # no label file: Sheet03/spatialModel.py:46


def test_flow_images_follow_the_reference_file_layout(tmp_path):
    from PIL import Image
    flow = torch.tensor([[-25.0, -20.0, 0.0], [20.0, 30.0, 0.07843]]).reshape(1, 2, 1, 3).repeat(3, 1, 8, 4)
    q = U.flowToImages(flow)
    assert q.dtype == np.uint8 and q.shape == (3, 2, 8, 12)
    assert q[0, 0, 0, :3].tolist() == [0, 0, 128] and q[0, 1, 0, :3].tolist() == [255, 255, 128]
    d = str(tmp_path / "Archery" / "v_Archery_g01_c01")
    assert U.saveFlowImages(flow, d) == 6
    names = sorted(os.listdir(d))
    assert names == ["flow_x_0001.jpg", "flow_x_0002.jpg", "flow_x_0003.jpg", "flow_y_0001.jpg", "flow_y_0002.jpg", "flow_y_0003.jpg"]
    im = Image.open(os.path.join(d, "flow_y_0002.jpg"))
    assert im.mode == "L" and im.size == (12, 8)
    assert np.abs(np.asarray(im).astype(int) - q[1, 1].astype(int)).max() <= 6  # JPEG is lossy
    # the reference's index arithmetic on this directory: 6 files -> nFlows = 3
    assert U.temporalFlowIndices(len(names), 2, r=1)[1] == [("x", 1), ("y", 1), ("x", 2), ("y", 2)]


def test_weights_from_reference_style_state_dict():
    from video_analytics_amd import vgg
    cfg = [(0, 3, 64), (2, 64, 64), (5, 64, 128), (7, 128, 128), (10, 128, 256), (12, 256, 256), (14, 256, 256),
           (17, 256, 512), (19, 512, 512), (21, 512, 512), (24, 512, 512), (26, 512, 512), (28, 512, 512)]
    sd = {}
    for i, ci, co in cfg:
        sd["module.features.%d.weight" % i] = torch.zeros(co, ci, 3, 3)
        sd["module.features.%d.bias" % i] = torch.full((co,), float(i))
    for i, (fo, fi) in zip([0, 3, 6, 9], [(4096, 25088), (4096, 4096), (256, 4096), (101, 256)]):
        sd["module.classifier.%d.weight" % i] = torch.zeros(1, 1).expand(fo, fi)
        sd["module.classifier.%d.bias" % i] = torch.zeros(fo)
    w = vgg.weights_from_state_dict(sd)
    assert [tuple(t.shape) for t in w["conv_w"]][4] == (256, 128, 3, 3) and float(w["conv_b"][12][0]) == 28.0
    assert tuple(w["fc_w"][2].shape) == (256, 4096) and len(w["fc_b"]) == 4
    with pytest.raises(ValueError):
        vgg.weights_from_state_dict({"features.0.weight": torch.zeros(1)})


def test_multistep_lr_and_the_scheduler_step_loss_quirk():
    # MultiStepLR([10, 20], gamma 0.1) as torch 0.4 computes it from whatever step() received
    assert U.multiStepLr(0.1, [10, 20], 0) == 0.1 and U.multiStepLr(0.1, [10, 20], 9) == 0.1
    assert abs(U.multiStepLr(0.1, [10, 20], 10) - 0.01) < 1e-12 and abs(U.multiStepLr(0.1, [10, 20], 20) - 0.001) < 1e-12
    # the reference passes the validation loss (Sheet03/spatialModel.py:278): a loss of 4.7 keeps lr, 25.3 cuts it twice
    assert U.multiStepLr(0.1, [10, 20], 4.7) == 0.1 and abs(U.multiStepLr(0.1, [10, 20], 25.3) - 0.001) < 1e-12


def test_state_dict_layout_round_trip():
    from video_analytics_amd import vgg
    w = dict(conv_w=[torch.full((1,), float(i)) for i in range(13)], conv_b=[torch.full((1,), 100.0 + i) for i in range(13)],
             fc_w=[torch.full((1,), 200.0 + i) for i in range(4)], fc_b=[torch.full((1,), 300.0 + i) for i in range(4)])
    sd = vgg.state_dict_from_weights(w)
    keys = list(sd.keys())
    assert keys[0] == "module.features.0.weight" and keys[1] == "module.features.0.bias" and keys[2] == "module.features.2.weight"
    assert keys[26] == "module.classifier.0.weight" and keys[-1] == "module.classifier.9.bias" and len(keys) == 34
    back = vgg.weights_from_state_dict(sd)
    for k in w:
        assert all(torch.equal(a, b) for a, b in zip(w[k], back[k]))


def test_checkpoint_and_performance_files(tmp_path):
    ck, best = str(tmp_path / "c.pth.tar"), str(tmp_path / "b.pth.tar")
    U.makeCheckpoint({"epoch": 3, "t": torch.arange(3)}, False, ck, best)
    assert os.path.isfile(ck) and not os.path.isfile(best)
    U.makeCheckpoint({"epoch": 4, "t": torch.arange(3)}, True, ck, best)
    assert torch.load(best, weights_only=True)["epoch"] == 4
    perf = str(tmp_path / "perf.csv")
    U.savePerformance(0.25, 3.5, perf); U.savePerformance(0.5, 2.5, perf)
    assert open(perf).read() == "0.25,3.5\n0.5,2.5\n"


def _write_mjpeg_avi(path, frames, fourcc=b"MJPG"):
    """Minimal RIFF/AVI writer (test fixture): one video stream, one '00dc' chunk per JPEG frame."""
    import io
    import struct
    chunks = b""
    for fr in frames:
        buf = io.BytesIO()
        fr.save(buf, format="JPEG", quality=95)
        d = buf.getvalue()
        chunks += b"00dc" + struct.pack("<I", len(d)) + d + (b"\0" if len(d) & 1 else b"")
    movi = b"LIST" + struct.pack("<I", 4 + len(chunks)) + b"movi" + chunks
    strh = b"strh" + struct.pack("<I", 56) + b"vids" + fourcc + b"\0" * 48
    strl = b"LIST" + struct.pack("<I", 4 + len(strh)) + b"strl" + strh
    hdrl = b"LIST" + struct.pack("<I", 4 + len(strl)) + b"hdrl" + strl
    body = b"AVI " + hdrl + movi
    with open(path, "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", len(body)) + body)


def test_frame_extraction_every_nth_frame_and_directory_layout(tmp_path):
    from PIL import Image
    root, save = tmp_path / "videos", tmp_path / "frames"
    (root / "ApplyEyeMakeup").mkdir(parents=True)
    frames = [Image.new("RGB", (32, 24), (10 * i, 255 - 10 * i, 5 * i)) for i in range(23)]
    _write_mjpeg_avi(str(root / "ApplyEyeMakeup" / "v_ApplyEyeMakeup_g01_c01.avi"), frames)
    got = U.extractEveryNthFrame(str(root / "ApplyEyeMakeup" / "v_ApplyEyeMakeup_g01_c01.avi"), 10)
    assert len(got) == 3  # frames 0, 10, 20
    assert abs(got[1].getpixel((5, 5))[0] - 100) <= 3 and abs(got[2].getpixel((5, 5))[0] - 200) <= 3
    with pytest.raises(ValueError):
        U.extractEveryNthFrame(str(root / "nope.avi"), 10)
    lst = tmp_path / "list.txt"
    lst.write_text("ApplyEyeMakeup/v_ApplyEyeMakeup_g01_c01.avi\n")
    U.convertVideosToFrames(str(root), str(save), str(lst), mode="test")
    d = save / "ApplyEyeMakeup" / "v_ApplyEyeMakeup_g01_c01"
    assert sorted(os.listdir(str(d))) == ["0.jpg", "1.jpg", "2.jpg"]  # the names SpatialDataset opens
    # an existing frame directory means "already converted": nothing is rewritten
    os.remove(str(d / "2.jpg"))
    U.convertVideosToFrames(str(root), str(save), str(lst), mode="test")
    assert sorted(os.listdir(str(d))) == ["0.jpg", "1.jpg"]
    # codecs without a decoder in this image fail loudly (UCF-101 itself is XviD)
    _write_mjpeg_avi(str(root / "x.avi"), frames[:2], fourcc=b"XVID")
    with pytest.raises(ValueError, match="XVID"):
        U.extractEveryNthFrame(str(root / "x.avi"), 1)
