"""Host-fed (DeviceType.CPU-registered, GPU-backed) kernels: frames in host memory -> results in
host memory, through the Scanner-style kernel classes.  PCIe-inclusive rates (never bench.py's value)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from scannertools_amd.engine import CacheMode, Client, DeviceType, NamedStream, NamedVideoStream, PerfParams
n, h, w = int(os.environ.get("N", 128)), 1080, 1920
frames = np.random.default_rng(0).integers(0, 256, (n, h, w, 3), dtype=np.uint8)
if os.environ.get("PIN", "1") != "0":
    # Scanner hands CPU kernels frames from its own (page-locked, when GPUs are present) allocator;
    # PIN=0 feeds pageable numpy memory instead
    import torch
    frames = torch.from_numpy(frames).pin_memory().numpy()
sc = Client()
sc.ingest_frames("v", frames)
frame = sc.io.Input([NamedVideoStream(sc, "v")])
for name, op, batch in (("Histogram", lambda: sc.ops.Histogram(frame=frame, device=DeviceType.CPU, batch=64), 64),
                        ("OpticalFlow", lambda: sc.ops.OpticalFlow(frame=frame, device=DeviceType.CPU, batch=32), 32)):
    for rep in range(2):
        out = NamedStream(sc, name)
        t0 = time.perf_counter()
        sc.execute_seconds = 0.0
        sc.run(sc.io.Output(op(), [out]), PerfParams.estimate(), cache_mode=CacheMode.Overwrite)
        dt = time.perf_counter() - t0
    gb = n * 3 * h * w / 1e9
    ke = sc.execute_seconds
    print("%-12s host-fed: kernel execute() %.1f frames/s (%.2f GB/s of input frames, batch %d); with the Python "
          "engine's result copies %.1f frames/s" % (name, n / ke, gb / ke, batch, n / dt))
