"""GPU: the other FourCC outputs -- Y800, NV12, UYVY, YUV444, HSV -- against the
reference's own golden dumps and against the oracle, with and without crop / resize in front."""
import numpy as np
import pytest
import torch

from util import synth_nv12, ulp_diff

pytestmark = pytest.mark.gpu
Y800, RGB24, BGR24, NV12, UYVY, YUV444, HSV = range(7)


def run(vpp, y, uv, fourcc, norm, crop=(0, 0, 0, 0), dst=(0, 0), rt=0, width=None):
    import tensor_stream as ts
    fp = ts.FrameParameters(width=dst[0], height=dst[1], crop_coords=crop, resize_type=rt, pixel_format=fourcc, normalization=norm)
    out = vpp.Convert(torch.from_numpy(y).cuda(), torch.from_numpy(uv).cuda(), fp, width=width)
    torch.cuda.synchronize()
    return out.cpu().numpy()


@pytest.mark.parametrize("name,fcc", [("Y800", Y800), ("NV12", NV12), ("UYVY", UYVY), ("YUV444", YUV444), ("HSV", HSV)])
def test_reference_golden_files(vpp, golden, oracle, name, fcc):
    """reference tests/src/VPPTests.cpp:387-512: normalised fp32 dumps of its 320x240 frame."""
    got = run(vpp, golden["Y"], golden["UVp"], fcc, True)
    assert got.dtype == np.float32 and got.shape == oracle.shape_for(fcc, 1, 320, 240)
    assert np.array_equal(got.ravel().view(np.uint32), golden[name])


@pytest.mark.parametrize("fcc", [Y800, NV12, UYVY, YUV444, HSV])
@pytest.mark.parametrize("norm", [False, True])
@pytest.mark.parametrize("crop,dst,rt", [((0, 0, 0, 0), (0, 0), 0),
                                         ((0, 0, 0, 0), (480, 360), 1),
                                         ((121, 65, 601, 401), (0, 0), 0),
                                         ((120, 64, 600, 400), (300, 200), 3),
                                         ((0, 0, 0, 0), (362, 202), 2)])
def test_formats_vs_oracle(vpp, oracle, fcc, norm, crop, dst, rt):
    y, uv = synth_nv12(1080, 608, seed=fcc * 10 + rt, pitch=1088)
    got = run(vpp, y, uv, fcc, norm, crop, dst, rt, width=1080)
    ref, ow, oh = oracle.convert(y, uv, crop=crop, dst=dst, resize_type=rt, fourcc=fcc, normalization=norm, nthreads=8, width=1080)
    assert got.shape == oracle.shape_for(fcc, 1, ow, oh)
    got = got.ravel()
    assert got.dtype == ref.dtype and got.size == ref.size
    if got.dtype == np.uint8:
        assert np.array_equal(got, ref)
    else:
        assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), f"max ulp {ulp_diff(np.abs(got), np.abs(ref))}"


def test_batch_of_two_pass_format(vpp, oracle):
    import tensor_stream as ts
    frames = [synth_nv12(640, 360, seed=50 + i) for i in range(3)]
    ys = torch.from_numpy(np.stack([f[0] for f in frames])).cuda()
    uvs = torch.from_numpy(np.stack([f[1] for f in frames])).cuda()
    fp = ts.FrameParameters(width=320, height=180, resize_type=1, pixel_format=YUV444, normalization=True)
    out = vpp.convert_batch(ys, uvs, fp)
    torch.cuda.synchronize()
    for i in range(3):
        ref, _, _ = oracle.convert(frames[i][0], frames[i][1], dst=(320, 180), resize_type=1, fourcc=YUV444, normalization=True)
        assert np.array_equal(out[i].cpu().numpy().ravel().view(np.uint32), ref.view(np.uint32))
