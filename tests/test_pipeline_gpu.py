"""End-to-end GPU parity: committed goldens, the full two-stream clip pipeline against the oracle,
the reference-shaped Network.validate(), and size-independent properties at BASELINE.json's full
batch (32 clips = 320 TV-L1 pairs)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
TOL = 1e-3


def test_tvl1_against_committed_goldens():
    from video_analytics_amd import flow as vflow
    g = np.load(os.path.join(GOLD, "tvl1_64x48.npz"))
    gray = torch.from_numpy(g["gray"]).cuda()
    fixed = vflow.tvl1_flow(gray, epsilon=0.0, iters=30, warps=3)
    assert np.array_equal(fixed.cpu().numpy(), g["flow_fixed"])
    eps = vflow.tvl1_flow(gray, epsilon=0.01, iters=300)
    assert np.array_equal(eps.cpu().numpy(), g["flow_eps"])
    assert np.array_equal(vflow.flow_to_stack(fixed).cpu().numpy(), g["stack_fixed"])


def test_vgg_against_committed_goldens():
    from video_analytics_amd import pipeline, synth, vgg
    g = np.load(os.path.join(GOLD, "vgg_small.npz"))
    for name, c_in, seed in (("s", 3, 1), ("t", 20, 2)):
        w = pipeline.build_stream_weights(c_in, seed, torch.device("cuda", 0))
        m = vgg.Vgg16Stream(w["conv_w"], w["conv_b"], w["fc_w"], w["fc_b"], 101, 256)
        u = synth.hash_uniform(100 + seed, 77, 4 * c_in * 224 * 224).reshape(4, c_in, 224, 224)
        x = torch.from_numpy(u * 4.0 - 2.0).cuda()
        feat, desc, logits = m.forward(x, want_feat=True)
        assert np.abs(logits.cpu().numpy() - g["logits_" + name]).max() < TOL
        assert np.abs(desc.cpu().numpy() - g["desc_" + name]).max() < TOL
        fs = feat.double().sum(dim=(1, 2, 3)).cpu().numpy()
        assert np.abs(fs - g["feat_sum_" + name]).max() < 1e-3 * np.abs(g["feat_sum_" + name]).max()
        # the split call surface of the reference: classify(features(x)) == forward(x)
        d2, l2 = m.classify(m.features(x))
        assert torch.equal(d2, desc) and torch.equal(l2, logits)
        m.close()


def test_two_stream_clip_pipeline_matches_oracle(oracle_tvl1):
    from oracle import vgg_oracle
    from video_analytics_amd import _ffi, pipeline, synth
    from video_analytics_amd.parameters import NORM_MEANS_TF, NORM_STDS_TF
    n = 2
    rgb, gray, _ = synth.synth_clips(n, seed=12)
    kw = dict(epsilon=0.0, iters=40, warps=2)
    fl = oracle_tvl1.tvl1_flow(gray.numpy(), oracle_tvl1.default_params(**kw), nthreads=8)
    st = torch.from_numpy(oracle_tvl1.flow_to_stack(fl).reshape(n, 20, 224, 224))
    ws = synth.synth_vgg16_weights(c_in=3, seed=1)
    wt = synth.synth_vgg16_weights(c_in=20, seed=2)
    wt["conv_w"][0] = vgg_oracle.copy_first_layer(wt["conv_w"][0], 20)
    _, ds, ls = vgg_oracle.forward(vgg_oracle.normalize_u8(rgb, NORM_MEANS_TF, NORM_STDS_TF), ws["conv_w"], ws["conv_b"], ws["fc_w"], ws["fc_b"])
    _, dt, lt = vgg_oracle.forward(st, wt["conv_w"], wt["conv_b"], wt["fc_w"], wt["fc_b"])
    pipe = pipeline.TwoStreamPipeline(device=0, tvl1_params=_ffi.default_tvl1_params(**kw))
    assert torch.equal(pipe.flow_volume(gray.cuda()).cpu(), st)  # quantised flow volume: bit-exact
    out = pipe.run_batch(rgb.cuda(), gray.cuda())
    for got, ref in ((out["logits_s"], ls), (out["logits_t"], lt), (out["desc_s"], ds), (out["desc_t"], dt)):
        assert float((got.cpu() - ref).abs().max()) < TOL
    pipe.close()


class _MemDataset(torch.utils.data.Dataset):
    def __init__(self, x, labels):
        self.x, self.labels = x, labels

    def __len__(self):
        return len(self.labels)

    def __getitem__(self, i):
        return self.x[i], int(self.labels[i]), "v_Clip_g%02d_c%02d" % (i // 4 + 1, i % 4 + 1)


def test_network_validate_matches_reference_semantics():
    """SpatialNetwork/TemporalNetwork.validate(): accuracy, summed per-batch mean CE and per-video
    descriptors equal the oracle's restatement of Sheet03/spatialModel.py:197-231."""
    from oracle import vgg_oracle
    from torch.utils.data import DataLoader
    from video_analytics_amd import synth
    from video_analytics_amd.spatialModel import SpatialNetwork
    from video_analytics_amd.temporalModel import TemporalNetwork
    for Net, c_in, seed, extra in ((SpatialNetwork, 3, 1, ()), (TemporalNetwork, 20, 2, (10,))):
        w = synth.synth_vgg16_weights(c_in=c_in, seed=seed)
        if c_in != 3:
            w["conv_w"][0] = vgg_oracle.copy_first_layer(w["conv_w"][0], c_in)
        u = synth.hash_uniform(50 + seed, 9, 5 * c_in * 224 * 224).reshape(5, c_in, 224, 224)
        x = torch.from_numpy(u * 4.0 - 2.0)
        _, desc_r, log_r = vgg_oracle.forward(x, w["conv_w"], w["conv_b"], w["fc_w"], w["fc_b"])
        labels = log_r.argmax(1).clone()
        labels[3] = (labels[3] + 1) % 101  # one wrong label
        loader = DataLoader(_MemDataset(x, labels), batch_size=2, shuffle=False, num_workers=0)
        loss_r, corr_r = 0.0, 0
        for lo in range(0, 5, 2):
            l, c = vgg_oracle.validate_batch(log_r[lo:lo + 2], labels[lo:lo + 2])
            loss_r += float(l); corr_r += c
        args = (101,) + extra + (1, 0.1, 0.9, 256, None, loader, [10, 20], None)
        net = Net(*args, gpu=True, weights={k: [t.clone() for t in v] for k, v in w.items()})
        acc, loss = net.validate()
        assert acc == corr_r / 5 == 0.8
        assert abs(float(loss) - loss_r) < 1e-3
        assert len(net.testDict) == 5
        d = net.testDict["v_Clip_g01_c03"][0].avg
        assert float((d - desc_r[2]).abs().max()) < TOL
        net.model.close()


def test_full_batch_properties():
    """BASELINE configs[1] size (32 clips, 320 pairs, full 5x5x300 schedule): results do not depend on
    the temporal-blocking depth or on the batch a clip is in; static clips give exactly zero flow."""
    from video_analytics_amd import _ffi, pipeline, synth
    from video_analytics_amd import flow as vflow
    rgb, gray, true_flow = synth.synth_clips(32, seed=0)
    gray_d = gray.cuda()
    gray_d[5] = gray_d[5, 0:1]  # clip 5: a static scene
    fa = vflow.tvl1_flow(gray_d, epsilon=0.0, block_iters=5)
    fb = vflow.tvl1_flow(gray_d, epsilon=0.0, block_iters=12)
    assert torch.equal(fa, fb)
    assert float(fa[50:60].abs().max()) == 0.0
    alone = vflow.tvl1_flow(gray_d[7:8], epsilon=0.0)
    assert torch.equal(alone, fa[70:80])
    # the flow tracks the synthetic motion field away from the borders
    c = slice(32, -32)
    err = (fa[0:10, :, c, c].cpu() - true_flow[0][:, c, c]).abs().mean()
    assert float(err) < 0.6, float(err)  # sanity only: mean |flow - synthetic field| in px
    pipe = pipeline.TwoStreamPipeline(device=0, tvl1_params=_ffi.default_tvl1_params(epsilon=0.0))
    stack = vflow.flow_to_stack(fa).view(32, 20, 224, 224)
    out = pipe.run_batch(rgb.cuda(), flow_stack=stack)
    one = pipe.run_batch(rgb[9:10].cuda(), flow_stack=stack[9:10])
    assert torch.equal(one["logits_s"][0], out["logits_s"][9]) and torch.equal(one["logits_t"][0], out["logits_t"][9])
    assert bool(torch.isfinite(out["logits_t"]).all())
    pipe.close()


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_precomputed_flow_volumes_run_the_two_models_side_by_side(dtype):
    # with flow volumes given, the temporal model runs on a second stream beside the spatial one: same scores as the two
    # models called one after the other, batch after batch (each model owns its workspace; an event guards its reuse)
    from video_analytics_amd import pipeline, synth
    rgb, _, _ = synth.synth_clips(6, seed=21)
    g = torch.Generator().manual_seed(5)
    stack = torch.randn(6, 20, 224, 224, generator=g)
    pipe = pipeline.TwoStreamPipeline(device=0, cnn_dtype=dtype)
    rgb_d, stack_d = rgb.cuda(), stack.cuda()
    ref_s = pipe.spatial.forward(rgb_d)[2].clone()
    ref_t = pipe.temporal.forward(stack_d)[2].clone()
    torch.cuda.synchronize()
    outs = [pipe.submit(rgb_d, flow_stack=stack_d) for _ in range(3)]
    pipe.wait()
    for o in outs:
        assert torch.equal(o["logits_s"], ref_s) and torch.equal(o["logits_t"], ref_t)
    pipe.close()


def test_concurrent_flow_streams_give_identical_results():
    from video_analytics_amd import flow as vflow, synth
    _, gray, _ = synth.synth_clips(5, seed=6, n_gray=4)
    g = gray.cuda()
    p = dict(epsilon=0.0, iters=30, warps=2)
    from video_analytics_amd import _ffi
    ref = vflow.tvl1_flow(g, _ffi.default_tvl1_params(**p))
    for n in (2, 3, 8):
        out = vflow.tvl1_flow_concurrent(g, _ffi.default_tvl1_params(**p), n_streams=n)
        assert out.shape == ref.shape and torch.equal(out, ref)
    eps = vflow.tvl1_flow_concurrent(g, _ffi.default_tvl1_params(epsilon=0.01), n_streams=2)
    assert torch.equal(eps, vflow.tvl1_flow(g, _ffi.default_tvl1_params(epsilon=0.01)))


def test_sharded_sweep_equals_single_batch_results():
    """BASELINE config 4 logic at world_size 1: a 70-clip sweep in batches of 32 (ragged last batch)
    returns, in global clip order, exactly the scores of running those clips directly."""
    from video_analytics_amd import _ffi, pipeline, sweep, synth
    pipe = pipeline.TwoStreamPipeline(device=0, tvl1_params=_ffi.default_tvl1_params(epsilon=0.0, iters=10, warps=1, nscales=2))

    def make_batch(lo, hi):
        rgb, gray, _ = synth.synth_clips(hi - lo, seed=4, first_clip=lo)
        return rgb.cuda(), gray.cuda()

    scores = sweep.run_sweep(pipe, 70, make_batch, batch_size=32)
    assert scores.shape == (70, 2, 101)
    rgb, gray = make_batch(64, 70)
    ref = pipe.run_batch(rgb, gray)
    assert torch.equal(scores[64:70, 0], ref["logits_s"]) and torch.equal(scores[64:70, 1], ref["logits_t"])
    acc = sweep.score(scores, scores[:, 0].argmax(1))
    assert acc["spatial"] == 1.0 and 0.0 <= acc["fused"] <= 1.0
    pipe.close()


def test_pipelined_batches_equal_unpipelined_results():
    """TwoStreamPipeline.submit() overlaps batch i's flow quantisation + temporal CNN with batch i + 1's TV-L1 (separate
    streams, pipeline-owned double buffers guarded by events).  Five different batches submitted back to back (more
    than the buffer depth, ragged last batch) must give exactly what run_batch gives one batch at a time -- also with
    a single TV-L1 stream and with a buffer depth of 1."""
    from video_analytics_amd import _ffi, pipeline, synth
    kw = dict(epsilon=0.0, iters=25, warps=2, nscales=3)
    dev = torch.device("cuda", 0)
    sizes = [8, 8, 8, 8, 5]
    batches = []
    for i, n in enumerate(sizes):
        rgb, gray, _ = synth.synth_clips(n, seed=20, first_clip=8 * i, device=dev)
        batches.append((rgb, gray))
    pipe = pipeline.TwoStreamPipeline(device=0, tvl1_params=_ffi.default_tvl1_params(**kw))
    ref = []
    for rgb, gray in batches:
        r = pipe.run_batch(rgb, gray)
        ref.append({k: r[k].clone() for k in ("logits_s", "logits_t", "desc_s", "desc_t")})
    torch.cuda.synchronize()
    for streams, depth in ((2, 2), (1, 2), (2, 1), (3, 3)):
        p2 = pipeline.TwoStreamPipeline(device=0, tvl1_params=_ffi.default_tvl1_params(**kw), flow_streams=streams, depth=depth)
        outs = [p2.submit(rgb, gray) for rgb, gray in batches]
        p2.wait()
        torch.cuda.synchronize()
        for o, r in zip(outs, ref):
            for k in r:
                assert torch.equal(o[k], r[k]), (streams, depth, k)
        p2.close()
    pipe.close()


def test_config1_demoTest_clip_through_dataset_and_validate(tmp_path):
    """BASELINE config 1 end to end: the first lines of the reference's own Sheet03/demoTest.txt -> SpatialDataset over
    a frame directory in the reference's layout (<root>/<category>/<video>/<i>.jpg, Sheet03/spatialModel.py:64-81:
    random frame, random crop, random flip, ToTensor, Normalize) -> DataLoader -> SpatialNetwork.validate()
    (Sheet03/spatialModel.py:197-231) on the GPU, against the torch-CPU oracle evaluated on the very tensors the
    dataset produced: accuracy exact, summed per-batch mean cross-entropy and every per-video descriptor within 1e-3.
    The frames are synthetic JPEGs (the dataset itself is absent from the reference); the random draws are replayed
    by re-seeding Python's global generator, which is what the reference's transforms draw from."""
    import random
    from PIL import Image
    from oracle import vgg_oracle
    from video_analytics_amd import synth, utils as U
    from video_analytics_amd.spatialModel import SpatialDataset, SpatialNetwork
    lines = open(os.path.join(GOLD, "demoTest.txt")).readlines()[:3]
    assert lines[0] == "ApplyEyeMakeup/v_ApplyEyeMakeup_g01_c01.avi\n"
    lst = tmp_path / "demoTest_head.txt"
    lst.write_text("".join(lines))
    (tmp_path / "classInd.txt").write_text("1 ApplyEyeMakeup\n2 ApplyLipstick\n3 Archery\n")
    rng = np.random.default_rng(1)
    for k, line in enumerate(lines):
        _, videoName, _, category, _, _ = U.videoInfo(line, "test")
        fd = tmp_path / "frames" / category / videoName
        fd.mkdir(parents=True)
        for i in range(4 + k):  # UCF-101 frames are 320x240
            img = rng.integers(0, 255, (30, 40, 3), dtype=np.uint8).repeat(8, axis=0).repeat(8, axis=1)
            Image.fromarray(img).save(str(fd / ("%d.jpg" % i)), quality=90)
    ds = SpatialDataset(str(lst), str(tmp_path / "frames"), U.getTransforms(), mode="test",
                        actionLabelLoc=str(tmp_path / "classInd.txt"))
    loader = U.getDataLoader(ds, batchSize=2, nWorkers=0, shuffle=False)
    random.seed(2218)
    batches = [(d.clone(), l.clone(), list(n)) for d, l, n in loader]
    assert [n for _, _, ns in batches for n in ns] == ["v_ApplyEyeMakeup_g01_c01", "v_ApplyEyeMakeup_g01_c02", "v_ApplyEyeMakeup_g01_c03"]
    assert [int(v) for _, l, _ in batches for v in l] == [1, 1, 1]  # raw 1-based labels
    w = synth.synth_vgg16_weights(c_in=3, seed=1)
    loss_r, corr_r, desc_r = 0.0, 0, {}
    for d, l, ns in batches:
        assert d.shape[1:] == (3, 224, 224) and d.dtype == torch.float32
        _, desc, logits = vgg_oracle.forward(d, w["conv_w"], w["conv_b"], w["fc_w"], w["fc_b"])
        lb, cb = vgg_oracle.validate_batch(logits, l)
        loss_r += float(lb)
        corr_r += cb
        for i, n in enumerate(ns):
            desc_r[n] = desc[i]
    net = SpatialNetwork(101, 1, 0.1, 0.9, 256, None, loader, [10, 20], None, gpu=True,
                         weights={k: [t.clone() for t in v] for k, v in w.items()})
    random.seed(2218)  # the same frame / crop / flip draws again
    acc, loss = net.validate()
    assert acc == corr_r / 3
    assert abs(float(loss) - loss_r) < TOL
    assert list(net.testDict.keys()) == list(desc_r.keys())
    for n, ref in desc_r.items():
        meter, label = net.testDict[n]
        assert int(label) == 1 and meter.count == 1
        assert float((meter.avg - ref).abs().max()) < TOL
    # and the descriptor CSV the fusion step reads (Sheet03/utils.py:174-195)
    p = str(tmp_path / "spatial_test.csv")
    U.saveVideoDescriptors(net.testDict, p, True)
    rows = open(p).read().strip().split("\n")
    assert len(rows) == 3 and rows[0].startswith("v_ApplyEyeMakeup_g01_c01,1,") and len(rows[0].split(",")) == 258
    net.model.close()


def test_temporal_file_path_gpu_flow_to_jpegs_to_dataset_to_validate(tmp_path):
    """SURVEY 8f row 1 end to end, the temporal twin of the config-1 test: synthetic 320x240 gray frames -> ``va_tvl1_flow``
    on the GPU -> ``utils.saveFlowImages`` into the reference's layout ``<root>/<category>/<video>/flow_{x,y}_%04d.jpg``
    (1-based, 8-bit single-channel JPEGs: Sheet03/temporalModel.py:76-81, parameters.py:27,38-39) -> the file-backed
    ``TemporalDataset`` (random start in [1, nFlows - L], x/y interleave, one random crop and flip PER IMAGE, ToTensor,
    single-channel Normalize: :67-92) -> ``DataLoader`` -> ``TemporalNetwork.validate()`` (:226-260) on the GPU, against
    the torch-CPU oracle evaluated on the very tensors the dataset produced.  The random draws are replayed by re-seeding
    Python's global generator; what the files hold is checked against the flow the GPU produced."""
    import random
    from PIL import Image
    from oracle import vgg_oracle
    from video_analytics_amd import flow as vflow, synth, utils as U
    from video_analytics_amd.temporalModel import TemporalDataset, TemporalNetwork
    lines = open(os.path.join(GOLD, "demoTest.txt")).readlines()[:3]
    lst = tmp_path / "demoTest_head.txt"
    lst.write_text("".join(lines))
    (tmp_path / "classInd.txt").write_text("1 ApplyEyeMakeup\n2 ApplyLipstick\n3 Archery\n")
    L, n_frames = 10, (13, 12, 14)  # 12, 11, 13 flow pairs per video: start in [1, 2], [1, 1], [1, 3]
    flows = {}
    for k, line in enumerate(lines):
        _, videoName, _, category, _, _ = U.videoInfo(line, "test")
        # one clip of n frames at UCF-101's 320x240 (the texture / warp generator of the benchmark)
        _, gray, _ = synth.synth_clips(1, seed=40 + k, H=240, W=320, n_gray=n_frames[k])
        fl = vflow.tvl1_flow(gray.cuda(), epsilon=0.0, iters=20, warps=2, nscales=3)  # [n - 1, 2, 240, 320]
        assert fl.shape == (n_frames[k] - 1, 2, 240, 320) and bool(torch.isfinite(fl).all())
        fd = tmp_path / "flows" / category / videoName
        assert U.saveFlowImages(fl, str(fd)) == 2 * (n_frames[k] - 1)
        flows[videoName] = U.flowToImages(fl)  # what the files should hold, up to JPEG
        names = sorted(os.listdir(str(fd)))
        assert names[0] == "flow_x_0001.jpg" and names[-1] == "flow_y_%04d.jpg" % (n_frames[k] - 1) and len(names) == 2 * (n_frames[k] - 1)
        img = Image.open(str(fd / "flow_y_0003.jpg"))
        assert img.mode == "L" and img.size == (320, 240)
        err = np.abs(np.asarray(img, dtype=np.int32) - flows[videoName][2, 1].astype(np.int32))
        assert err.mean() < 1.5 and err.max() <= 24  # JPEG quality 95 of a smooth field
    ds = TemporalDataset(str(lst), str(tmp_path / "flows"), U.getTransforms(), flowSampleSize=L, mode="test",
                         actionLabelLoc=str(tmp_path / "classInd.txt"))
    loader = U.getDataLoader(ds, batchSize=2, nWorkers=0, shuffle=False)
    random.seed(2218)
    batches = [(d.clone(), l.clone(), list(n)) for d, l, n in loader]
    assert [n for _, _, ns in batches for n in ns] == ["v_ApplyEyeMakeup_g01_c01", "v_ApplyEyeMakeup_g01_c02", "v_ApplyEyeMakeup_g01_c03"]
    assert [int(v) for _, l, _ in batches for v in l] == [1, 1, 1]
    for d, _, _ in batches:
        assert d.shape[1:] == (20, 224, 224) and d.dtype == torch.float32
        # (q / 255 - 0.485) / 0.229 with q in [0, 255] (the single-channel Normalize rule): the flow is small against the
        # bound of 20 px, so every value sits near the image of q = 127.5
        assert float(d.min()) >= (0.0 - 0.485) / 0.229 - 1e-6 and float(d.max()) <= (1.0 - 0.485) / 0.229 + 1e-6
        assert abs(float(d.mean()) - (0.5 - 0.485) / 0.229) < 0.2
    w = synth.synth_vgg16_weights(c_in=20, seed=2)
    w["conv_w"][0] = vgg_oracle.copy_first_layer(w["conv_w"][0], 20)
    loss_r, corr_r, desc_r = 0.0, 0, {}
    for d, l, ns in batches:
        _, desc, logits = vgg_oracle.forward(d, w["conv_w"], w["conv_b"], w["fc_w"], w["fc_b"])
        lb, cb = vgg_oracle.validate_batch(logits, l)
        loss_r += float(lb)
        corr_r += cb
        for i, n in enumerate(ns):
            desc_r[n] = desc[i]
    net = TemporalNetwork(101, L, 1, 0.1, 0.9, 256, None, loader, [10, 20], None, gpu=True,
                          weights={k: [t.clone() for t in v] for k, v in w.items()})
    random.seed(2218)  # the same start / crop / flip draws again
    acc, loss = net.validate()
    assert acc == corr_r / 3
    assert abs(float(loss) - loss_r) < TOL
    assert list(net.testDict.keys()) == list(desc_r.keys())
    for n, ref in desc_r.items():
        meter, label = net.testDict[n]
        assert int(label) == 1 and meter.count == 1
        assert float((meter.avg - ref).abs().max()) < TOL
    p = str(tmp_path / "temporal_test.csv")
    U.saveVideoDescriptors(net.testDict, p, True)
    rows = open(p).read().strip().split("\n")
    assert len(rows) == 3 and rows[0].startswith("v_ApplyEyeMakeup_g01_c01,1,") and len(rows[0].split(",")) == 258
    net.model.close()
