"""Training step against torch autograd + torch.optim.SGD on the CPU."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

# Update of a tensor vs the oracle's, relative to the largest update in that tensor.  Not tighter because backward
# passes through DECISIONS of the forward pass: the max-pool gradient goes to the arg-max of each window and the ReLU
# gradient to the positive elements, and a few elements per 100 000 sit closer to such a decision than the fp32
# difference between the two forward passes (measured here: 2 of 100 352 windows of conv4_3 pick another position in
# the first case, 1 of 154 420 elements of conv5_2 flips its ReLU mask in the second; everywhere else the
# gradients agree to 2e-5): every layer below such a flip inherits a ~2e-3..7e-3 relative difference.
TOL_UPDATE = 1e-2


def _relerr(a, b):
    return float((a - b).abs().max()) / max(float(b.abs().max()), 1e-30)


@pytest.mark.parametrize("c_in,B", [(3, 2), (20, 3)])
def test_two_sgd_steps_match_autograd(c_in, B):
    """Two steps (the second one, restarted from the oracle's state, exercises the momentum buffers): loss, hits,
    train-mode descriptors, every parameter and every momentum buffer against the oracle.  The update of a
    tensor is compared relative to the largest update in that tensor."""
    from oracle import train_oracle, vgg_oracle
    from video_analytics_amd import synth, vgg
    torch.set_num_threads(8)
    w = synth.synth_vgg16_weights(c_in=3, seed=4)
    if c_in != 3:
        w["conv_w"][0] = vgg_oracle.copy_first_layer(w["conv_w"][0], c_in)
    lr, mu = 1e-4, 0.9  # small enough that the random-init network does not blow up in the second step
    ora = train_oracle.TrainOracle(w, lr, mu)
    m = vgg.Vgg16Stream(w["conv_w"], w["conv_b"], w["fc_w"], w["fc_b"], 101, 256)
    w0 = {k: [t.clone() for t in v] for k, v in w.items()}
    for step in range(2):
        if step == 1:
            # second step from the SAME state (parameters and momentum buffers of the oracle after step 0), so that it
            # checks the momentum arithmetic and import_state rather than the divergence of two chaotic trajectories
            m.import_state(ora.weights())
            m.import_state(ora.momentum(), momentum=True)
            w0 = ora.weights()
        u = synth.hash_uniform(70 + step, c_in, B * c_in * 224 * 224).reshape(B, c_in, 224, 224)
        x = torch.from_numpy(u * 4.0 - 2.0)
        labels = torch.tensor([(7 * i + 3 * step + 1) % 101 for i in range(B)], dtype=torch.int64)
        loss_r, corr_r, desc_r, _ = ora.step(x, labels, seed=1000 + step)
        stats, desc = m.train_step(x.cuda(), labels.cuda(), lr, mu, 1000 + step)
        stats = stats.cpu()
        assert abs(float(stats[0]) - loss_r) < 2e-4 * max(1.0, abs(loss_r)), (float(stats[0]), loss_r)
        assert int(stats[1]) == corr_r
        assert _relerr(desc.cpu(), desc_r) < 1e-3
        got, ref = m.export_state(), ora.weights()
        gotm, refm = m.export_state(momentum=True), ora.momentum()
        worst = []
        for k in ("conv_w", "conv_b", "fc_w", "fc_b"):
            for i, (g, r, o, gm, rm) in enumerate(zip(got[k], ref[k], w0[k], gotm[k], refm[k])):
                e_upd = _relerr(g.cpu() - o, r - o)
                e_mom = _relerr(gm.cpu(), rm)
                worst.append((max(e_upd, e_mom), step, k, i))
                print("step %d %-6s %2d: update err %.2e  momentum err %.2e" % (step, k, i, e_upd, e_mom))
        assert max(worst)[0] < TOL_UPDATE, max(worst)
        tight = [e for e, _, k, i in worst if k.startswith("fc")]  # no pooling / ReLU decision below the classifier's own
        assert max(tight) < 5e-4, max(tight)
    m.close()


def test_export_import_round_trip():
    from video_analytics_amd import synth, vgg
    w = synth.synth_vgg16_weights(c_in=3, seed=5)
    m = vgg.Vgg16Stream(w["conv_w"], w["conv_b"], w["fc_w"], w["fc_b"], 101, 256)
    got = m.export_state()
    for k in w:
        for a, b in zip(got[k], w[k]):
            assert torch.equal(a.cpu(), b)
    w2 = synth.synth_vgg16_weights(c_in=3, seed=6)
    m.import_state(w2)
    got = m.export_state()
    for k in w2:
        for a, b in zip(got[k], w2[k]):
            assert torch.equal(a.cpu(), b)
    m.train_init()
    m.import_state(w, momentum=True)
    gotm = m.export_state(momentum=True)
    for k in w:
        for a, b in zip(gotm[k], w[k]):
            assert torch.equal(a.cpu(), b)
    x = torch.zeros(1, 20, 224, 224, device="cuda")
    with pytest.raises(ValueError):
        m.train_step(x, torch.zeros(1, dtype=torch.int64), 0.1, 0.9, 0)  # 20 channels into a 3-channel model
    with pytest.raises(ValueError):
        m.train_step(x[:, :3], torch.zeros(2, dtype=torch.int64), 0.1, 0.9, 0)  # labels do not match the batch
    with pytest.raises(ValueError):
        m.import_state(dict(w, conv_w=w["conv_w"][:12]))
    m.close()
