"""Per-layer-shape timing of st_conv2d_nhwc_f32 at the pose network's shapes (batch N frames of 368x656)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scannertools_amd import _native
from scannertools_amd.hip import HipContext
n = int(os.environ.get("N", 16))
ctx = HipContext(0)
shapes = [(368, 656, 16, 64, 3), (368, 656, 64, 64, 3), (184, 328, 64, 128, 3), (184, 328, 128, 128, 3), (92, 164, 128, 256, 3),
          (92, 164, 256, 256, 3), (46, 82, 256, 512, 3), (46, 82, 512, 512, 3), (46, 82, 512, 256, 3), (46, 82, 256, 128, 3),
          (46, 82, 128, 128, 3), (46, 82, 128, 512, 1), (46, 82, 512, 38, 1), (46, 82, 192, 128, 7), (46, 82, 128, 128, 7),
          (46, 82, 128, 128, 1), (46, 82, 128, 19, 1)]
tot_ms = tot_fl = 0
for (h, w, ci, co, k) in shapes:
    cop = (co + 63) // 64 * 64
    x = torch.randn((n, h, w, ci), device="cuda")
    wt = torch.randn((cop, k, k, ci), device="cuda") * 0.05
    b = torch.zeros((cop,), device="cuda")
    y = torch.empty((n, h, w, (co + 3) // 4 * 4), device="cuda")
    def run():
        ctx._bind()
        ctx._check(ctx._L.st_conv2d_nhwc_f32(ctx._h, ctypes.c_void_p(x.data_ptr()), n, h, w, ci, ci, 0, ctypes.c_void_p(wt.data_ptr()),
                                             ctypes.c_void_p(b.data_ptr()), k, k, co, cop, 1, ctypes.c_void_p(y.data_ptr()), y.shape[3], 0))
    run(); torch.cuda.synchronize()
    ctx.timing_enable([_native.K_CONV]); ctx.timing_reset()
    for _ in range(3): run()
    nl, ms = ctx.timing_read(_native.K_CONV)
    ms /= 3
    fl = 2.0 * n * h * w * ci * co * k * k
    print("%4dx%-4d cin %3d cout %3d k %d: %8.3f ms  %6.1f TFLOP/s (useful; cout padded to %d)" % (h, w, ci, co, k, ms, fl / ms / 1e9, cop))
