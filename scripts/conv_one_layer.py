"""One convolution shape a few times (profiling target): H W CIN COUT K from the command line, batch from $N."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scannertools_amd import _native
from scannertools_amd.hip import HipContext
h, w, ci, co, k = (int(v) for v in sys.argv[1:6])
n = int(os.environ.get("N", 32))
reps = int(os.environ.get("REPS", 3))
ctx = HipContext(0)
cop = (co + 63) // 64 * 64
x = torch.randn((n, h, w, ci), device="cuda")
wt = torch.randn((cop, k, k, ci), device="cuda") * 0.05
b = torch.zeros((cop,), device="cuda")
y = torch.empty((n, h, w, (co + 3) // 4 * 4), device="cuda")
math = os.environ.get("MATH", "f32")
ctx._bind()
if math == "bf16x3":
    w3 = torch.empty((ctx._L.st_conv_bf16x3_packed_bytes(cop, k, k, ci),), dtype=torch.uint8, device="cuda")
    ctx._check(ctx._L.st_conv_pack_weights_bf16x3(ctx._h, ctypes.c_void_p(wt.data_ptr()), cop, k, k, ci, ctypes.c_void_p(w3.data_ptr())))
    torch.cuda.synchronize()
wtile = None
if math == "f32" and ctx._L.st_conv_f32_tile_bytes(cop, k, k, ci) > 0:
    wtile = torch.empty((ctx._L.st_conv_f32_tile_bytes(cop, k, k, ci),), dtype=torch.uint8, device="cuda")
    ctx._check(ctx._L.st_conv_pack_weights_f32_tile(ctx._h, ctypes.c_void_p(wt.data_ptr()), cop, k, k, ci, ctypes.c_void_p(wtile.data_ptr())))
    torch.cuda.synchronize()
ctx.timing_enable([_native.K_CONV]); ctx.timing_reset()
for _ in range(reps):
    ctx._bind()
    if math == "bf16x3":
        ctx._check(ctx._L.st_conv2d_nhwc_bf16x3(ctx._h, ctypes.c_void_p(x.data_ptr()), n, h, w, ci, ci, 0, ctypes.c_void_p(w3.data_ptr()),
                                                ctypes.c_void_p(b.data_ptr()), k, k, co, cop, 1, ctypes.c_void_p(y.data_ptr()), y.shape[3], 0))
    else:
        ctx._check(ctx._L.st_conv2d_nhwc_f32_tiled(ctx._h, ctypes.c_void_p(x.data_ptr()), n, h, w, ci, ci, 0, ctypes.c_void_p(wt.data_ptr()),
                                             ctypes.c_void_p(wtile.data_ptr()) if wtile is not None else None,
                                             ctypes.c_void_p(b.data_ptr()), k, k, co, cop, 1, ctypes.c_void_p(y.data_ptr()), y.shape[3], 0))
torch.cuda.synchronize()
nl, ms = ctx.timing_read(_native.K_CONV)
fl = 2.0 * n * h * w * ci * co * k * k
print(math, "%dx%d cin %d cout %d k %d N %d: %.3f ms/launch %.1f TFLOP/s" % (h, w, ci, co, k, n, ms / nl, fl / (ms / nl) / 1e9))
