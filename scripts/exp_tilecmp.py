import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, oracle
from util import texture_stream
from scannertools_amd.hip import HipContext
ctx = HipContext(0)
for (h, w) in ((203, 317), (256, 320), (480, 640)):
    frames, _ = texture_stream(h, 4, h, w)
    fr = torch.from_numpy(frames).cuda()
    got = ctx.optical_flow(fr, pairs=[(0, 1), (1, 2), (2, 3)]).cpu().numpy()
    for i in range(3):
        ref = oracle.optical_flow_rgb(frames[i], frames[i + 1])
        d = np.abs(got[i] - ref)
        idx = np.unravel_index(d.argmax(), d.shape)
        print(os.environ.get("ST_ITER_TILE", "default"), (h, w), i, "rel", np.linalg.norm(got[i] - ref) / np.linalg.norm(ref), "max", d.max(), "at", idx)
