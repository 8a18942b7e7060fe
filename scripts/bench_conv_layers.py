"""Per-layer-shape timing of the convolution kernels at the pose network's shapes (batch N frames of 368x656), with each
shape's count in the 92-layer network and its share of the stack's time: MATH=f32 (default) | bf16x3, ST_CONV_TILE=0 for the
per-tap bf16x3 kernel everywhere."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scannertools_amd import _native
from scannertools_amd.hip import HipContext
n = int(os.environ.get("N", 16))
ctx = HipContext(0)
math = os.environ.get("MATH", "f32")
# (h, w, cin, cout, k, layers of this shape in the network)
shapes = [(368, 656, 16, 64, 3, 1), (368, 656, 64, 64, 3, 1), (184, 328, 64, 128, 3, 1), (184, 328, 128, 128, 3, 1), (92, 164, 128, 256, 3, 1),
          (92, 164, 256, 256, 3, 3), (46, 82, 256, 512, 3, 1), (46, 82, 512, 512, 3, 1), (46, 82, 512, 256, 3, 1), (46, 82, 256, 128, 3, 1),
          (46, 82, 128, 128, 3, 6), (46, 82, 128, 512, 1, 2), (46, 82, 512, 38, 1, 2), (46, 82, 192, 128, 7, 10), (46, 82, 128, 128, 7, 40),
          (46, 82, 128, 128, 1, 10), (46, 82, 128, 19, 1, 10)]
rows = []
for (h, w, ci, co, k, cnt) in shapes:
    cop = (co + 63) // 64 * 64
    x = torch.randn((n, h, w, ci), device="cuda")
    wt = torch.randn((cop, k, k, ci), device="cuda") * 0.05
    b = torch.zeros((cop,), device="cuda")
    y = torch.empty((n, h, w, (co + 3) // 4 * 4), device="cuda")
    if math == "bf16x3":
        w3 = torch.empty((ctx._L.st_conv_bf16x3_packed_bytes(cop, k, k, ci),), dtype=torch.uint8, device="cuda")
        ctx._bind()
        ctx._check(ctx._L.st_conv_pack_weights_bf16x3(ctx._h, ctypes.c_void_p(wt.data_ptr()), cop, k, k, ci, ctypes.c_void_p(w3.data_ptr())))
    wtile = None
    if math == "f32" and ctx._L.st_conv_f32_tile_bytes(cop, k, k, ci) > 0:
        wtile = torch.empty((ctx._L.st_conv_f32_tile_bytes(cop, k, k, ci),), dtype=torch.uint8, device="cuda")
        ctx._bind()
        ctx._check(ctx._L.st_conv_pack_weights_f32_tile(ctx._h, ctypes.c_void_p(wt.data_ptr()), cop, k, k, ci, ctypes.c_void_p(wtile.data_ptr())))
    def run():
        ctx._bind()
        if math == "bf16x3":
            ctx._check(ctx._L.st_conv2d_nhwc_bf16x3(ctx._h, ctypes.c_void_p(x.data_ptr()), n, h, w, ci, ci, 0, ctypes.c_void_p(w3.data_ptr()),
                                                    ctypes.c_void_p(b.data_ptr()), k, k, co, cop, 1, ctypes.c_void_p(y.data_ptr()), y.shape[3], 0))
            return
        ctx._check(ctx._L.st_conv2d_nhwc_f32_tiled(ctx._h, ctypes.c_void_p(x.data_ptr()), n, h, w, ci, ci, 0, ctypes.c_void_p(wt.data_ptr()),
                                             ctypes.c_void_p(wtile.data_ptr()) if wtile is not None else None,
                                             ctypes.c_void_p(b.data_ptr()), k, k, co, cop, 1, ctypes.c_void_p(y.data_ptr()), y.shape[3], 0))
    run(); torch.cuda.synchronize()
    ctx.timing_enable([_native.K_CONV]); ctx.timing_reset()
    for _ in range(3): run()
    nl, ms = ctx.timing_read(_native.K_CONV)
    ms /= 3
    fl = 2.0 * n * h * w * ci * co * k * k
    rows.append((h, w, ci, co, k, cnt, ms, fl, cop))
tot = sum(r[5] * r[6] for r in rows)
for (h, w, ci, co, k, cnt, ms, fl, cop) in rows:
    print("%4dx%-4d cin %3d cout %3d k %d x%-2d: %8.3f ms  %6.1f TFLOP/s (useful; cout padded to %d)  %5.1f %% of the stack"
          % (h, w, ci, co, k, cnt, ms, fl / ms / 1e9, cop, 100 * cnt * ms / tot))
print("%s, batch %d: %.2f ms for the 92 convolutions = %.1f frames/s, %.1f TFLOP/s useful"
      % (math, n, tot, n / tot * 1e3, sum(r[5] * r[7] for r in rows) / tot / 1e9))
