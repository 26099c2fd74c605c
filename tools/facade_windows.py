import sys, time, torch
sys.path.insert(0, "tensor-stream_amd")
import tensor_stream as ts
C5 = dict(width=640, height=360, resize_type=3, pixel_format=2, planes_pos=0, normalization=True)
r = ts.TensorStreamConverter("synthetic://3840x2160?seed=9&frames=0&fps=100000&pool=3", max_consumers=64, cuda_device=0, buffer_size=10, framerate_mode=ts.FrameRate.FAST)
r.initialize(); r.start()
names = [f"c{i}" for i in range(64)]
for _ in range(20): r.read_many(names, **C5)
torch.cuda.synchronize()
out = []
for w in range(12):
    t0 = time.perf_counter()
    for _ in range(100): r.read_many(names, **C5)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out.append(round(64 * 100 / dt * 15206400 / 8e12, 3))
r.stop()
print(out)
