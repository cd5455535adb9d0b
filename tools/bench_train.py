"""Training-step throughput of one VGG-16 stream: forward + backward + SGD at batch 32.
Run on the GPU box: python tools/bench_train.py [c_in] [batch]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from video_analytics_amd import synth, vgg
c_in = int(sys.argv[1]) if len(sys.argv) > 1 else 3
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
w = synth.synth_vgg16_weights(c_in=3, seed=1, device="cuda")
if c_in != 3:
    w["conv_w"][0] = vgg.copy_first_layer(w["conv_w"][0].cuda(), c_in)
m = vgg.Vgg16Stream(w["conv_w"], w["conv_b"], w["fc_w"], w["fc_b"], 101, 256)
x = torch.randn(B, c_in, 224, 224, device="cuda")
y = torch.randint(0, 101, (B,), device="cuda")
for i in range(2):
    m.train_step(x, y, 1e-5, 0.9, i)
torch.cuda.synchronize(); t = time.perf_counter()
N = 5
for i in range(N):
    stats, _ = m.train_step(x, y, 1e-5, 0.9, 10 + i)
torch.cuda.synchronize(); dt = (time.perf_counter() - t) / N
# forward conv FLOPs per clip: 30.69 G (c_in 3) / 31.68 G (c_in 20); backward = data + weight gradients ~ 2x forward
# (the first layer has no data gradient); classifier 0.24 G forward, 0.48 G backward
fwd = (30.693e9 if c_in == 3 else 31.676e9) + 0.2412e9
print("train step c_in=%d B=%d: %.1f ms = %.1f clips/s, ~%.1f TFLOP/s (3x forward FLOPs), loss %.4f"
      % (c_in, B, dt * 1e3, B / dt, 3 * fwd * B / dt / 1e12, float(stats[0])))
