import sys, os
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tensor-stream_amd"))
import tensor_stream as ts
from oracle import oracle as O
vpp = ts.VideoProcessor(device=0)
def case(name, y, uv, dst, fourcc=0):
    h, w = y.shape
    fp = ts.FrameParameters(width=dst[0], height=dst[1], resize_type=2, pixel_format=fourcc, planes_pos=1, normalization=False)
    d = ts.describe(fp, w, h, pitch=w)
    print(name, d['kernel'], d['shape'], d['rpt'], d['dma'], d['lds'])
    got = vpp.Convert(torch.from_numpy(y).cuda(), torch.from_numpy(uv).cuda(), fp).cpu().numpy().ravel()
    ref, ow, oh = O.convert(y, uv, dst=dst, resize_type=2, fourcc=fourcc, planes=1, normalization=False)
    return got[:ow*oh].reshape(oh, ow).astype(int), ref[:ow*oh].reshape(oh, ow).astype(int)
w, h, dst = 640, 360, (320, 240)
y = (np.arange(h)[:, None] * 2 % 256 + np.zeros((1, w))).astype(np.uint8); uv = np.full((h//2, w), 50, np.uint8)
g, r = case("vgrad", y, uv, dst)
print(" got col0 rows0-23", g[:24, 0].tolist()); print(" ref col0 rows0-23", r[:24, 0].tolist())
print(" rows where all cols equal across row?", [(int(i), np.unique(g[i]).tolist()[:4]) for i in range(16)])
y = (np.arange(w)[None, :] % 256 + np.zeros((h, 1))).astype(np.uint8)
g, r = case("hgrad", y, uv, dst)
for i in (0, 1, 2, 3, 16, 17):
    print(f" row{i} got", g[i, :40].tolist()); print(f" row{i} ref", r[i, :40].tolist())
print(" mismatch count per row (first 40 rows)", (g != r).sum(1)[:40].tolist())
print(" mismatch count per col (first 64 cols)", (g != r).sum(0)[:64].tolist())
