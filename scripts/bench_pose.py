"""Pose path, config 5: 1080p frames -> CPM2Input -> CPM2 (network, resize, nms; random weights) -> limb scores; frames/s and TFLOP/s."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scannertools_amd import _native, pose_net
from scannertools_amd.hip import HipContext, cpm2_geometry

n = int(os.environ.get("N", 8))
h, w, scale = 1080, 1920, 368 / 1080.
ctx = HipContext(0)
net = pose_net.PoseNet(ctx, seed=1, math=os.environ.get("MATH", "f32"))
frames = torch.randint(0, 256, (n, h, w, 3), dtype=torch.uint8, device="cuda")
_, _, nh, nw = cpm2_geometry(h, w, scale)
fl = pose_net.flops(nh, nw)
for _ in range(2):
    maps, joints = net.detect(ctx.cpm2_input(frames, scale)); limb = ctx.cpm2_limb_scores(maps, joints)
torch.cuda.synchronize()
ctx.timing_enable([_native.K_CONV]); ctx.timing_reset()
reps = 3
t0 = time.perf_counter()
for _ in range(reps):
    maps, joints = net.detect(ctx.cpm2_input(frames, scale)); limb = ctx.cpm2_limb_scores(maps, joints)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
nl, ms = ctx.timing_read(_native.K_CONV)
print(os.environ.get("MATH", "f32"), "batch %d: %.2f ms per batch, %.1f frames/s, %.1f TFLOP/s end to end; conv+pool kernels %.2f ms per batch = %.1f TFLOP/s (%.0f %% of the 157.3 TFLOP/s f32 matrix peak); net input %dx%d, %.1f GFLOP per frame"
      % (n, dt * 1e3, n / dt, n * fl / dt / 1e12, ms / reps, n * fl / (ms / reps * 1e-3) / 1e12, 100 * n * fl / (ms / reps * 1e-3) / 157.3e12, nw, nh, fl / 1e9))
