import sys, os
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tensor-stream_amd"))
import tensor_stream as ts
vpp = ts.VideoProcessor(device=0)
w, h, dst = 640, 360, (320, 240)
dbg = torch.zeros(65536, dtype=torch.uint8, device="cuda")
os.environ["TSVPP_DEBUG_PTR"] = hex(dbg.data_ptr())
y = (np.arange(w)[None, :] % 256 + np.zeros((h, 1))).astype(np.uint8); uv = np.full((h//2, w), 50, np.uint8)
fp = ts.FrameParameters(width=dst[0], height=dst[1], resize_type=2, pixel_format=0, planes_pos=1, normalization=False)
print(ts.describe(fp, w, h, pitch=w))
got = vpp.Convert(torch.from_numpy(y).cuda(), torch.from_numpy(uv).cuda(), fp)
torch.cuda.synchronize()
L = dbg.cpu().numpy()
i32 = L.view(np.int32)
planes = 32 * 512 + 16 * 512
hy = planes + 16; huv = hy + 128 * 36; xtab = huv + 128 * 28 + 16; cxtab = xtab + 128 * 16; ytab = cxtab + 64 * 16; cytab = ytab + 16 * 16; rby = cytab + 8 * 16
print("rby", i32[rby // 4: rby // 4 + 32].tolist())
print("xtab[0..9]", i32[xtab // 4: xtab // 4 + 40].reshape(10, 4).tolist())
print("ytab", i32[ytab // 4: ytab // 4 + 64].reshape(16, 4).tolist())
print("staged row0 bytes 0..40", L[0:40].tolist()); print("staged row1 bytes 0..40", L[512:552].tolist())
for pc in (0, 32, 64, 96, 1, 33, 65, 97):
    print("H col pc", pc, L[hy + pc * 36: hy + pc * 36 + 36].tolist())
