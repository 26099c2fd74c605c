"""The reference's CRC-32 goldens for this path (av_crc(AV_CRC_32_IEEE, -1, ...) of the uint8 output of frame 0 of
tests/resources/bbb_1080x608_420_10.h264).  tests/test_reference_crcs.py replays them on
`tests/golden/bbb_1080x608_frame0.nv12` (tight 1080x608 NV12, 984 960 bytes; decoded by tests/golden/make_bbb_frame0.py), whose two
planes are themselves pinned by reference tests/src/DecoderTests.cpp:63-65.

Fields: (source file:line, fourcc, planes, (dst_w, dst_h), resize, crop (l,t,r,b), accepted CRCs)
fourcc: 0 Y800 1 RGB24 2 BGR24 3 NV12 4 UYVY 5 YUV444; planes: 0 PLANAR 1 MERGED; resize: 0 NEAREST 1 BILINEAR 2 BICUBIC 3 AREA
"""
INPUT_PLANE_CRCS = {"Y": 3265466497, "UV": 2183362287}  # tests/src/DecoderTests.cpp:63-65

GOLDENS = [
    ("VPPTests.cpp:138", 1, 1, (1080, 608), 0, (0, 0, 0, 0), (2225932432,)),
    ("VPPTests.cpp:145", 1, 0, (1080, 608), 0, (0, 0, 0, 0), (3151499217,)),
    ("VPPTests.cpp:152", 1, 1, (540, 304), 0, (0, 0, 0, 0), (3545075074,)),
    ("VPPTests.cpp:159", 1, 1, (2160, 1216), 0, (0, 0, 0, 0), (97423732,)),
    ("VPPTests.cpp:166", 2, 1, (1080, 608), 0, (0, 0, 0, 0), (2467105116,)),
    ("VPPTests.cpp:173", 2, 0, (1080, 608), 0, (0, 0, 0, 0), (3969775694,)),
    ("VPPTests.cpp:180", 0, 0, (1080, 608), 0, (0, 0, 0, 0), (3265466497,)),
    ("VPPTests.cpp:187", 4, 1, (1080, 608), 0, (0, 0, 0, 0), (1323730732,)),
    ("VPPTests.cpp:194", 4, 1, (720, 480), 0, (0, 0, 0, 0), (1564587937,)),
    ("VPPTests.cpp:201", 5, 1, (1080, 608), 0, (0, 0, 0, 0), (1110927649,)),
    ("VPPTests.cpp:208", 5, 1, (720, 480), 0, (0, 0, 0, 0), (449974214,)),
    ("VPPTests.cpp:215", 3, 0, (1080, 608), 0, (0, 0, 0, 0), (2957341121,)),
    ("VPPTests.cpp:222", 3, 0, (720, 480), 0, (0, 0, 0, 0), (1200915282,)),
    ("VPPTests.cpp:231", 3, 0, (0, 0), 0, (0, 0, 320, 240), (3435719157,)),
    ("VPPTests.cpp:240", 3, 0, (0, 0), 0, (320, 240, 720, 480), (1515981907,)),
    ("VPPTests.cpp:249", 3, 0, (0, 0), 0, (400, 240, 720, 480), (655388614,)),
    ("VPPTests.cpp:258", 3, 0, (0, 0), 0, (640, 360, 1080, 608), (602193072,)),
    ("VPPTests.cpp:265", 3, 0, (720, 480), 0, (0, 0, 320, 240), (1764198598,)),
    ("VPPTests.cpp:273", 3, 0, (720, 480), 0, (160, 120, 480, 360), (1834204062,)),
    ("VPPTests.cpp:281", 3, 0, (720, 480), 0, (400, 240, 720, 480), (1750083777,)),
    ("VPPTests.cpp:289", 3, 0, (480, 320), 0, (0, 0, 720, 480), (3477030875,)),
    ("VPPTests.cpp:297", 3, 0, (480, 320), 0, (480, 340, 1080, 608), (2394953726,)),
    ("PythonTests.cpp:147", 2, 1, (540, 304), 0, (0, 0, 0, 0), (201454032,)),
    ("PythonTests.cpp:184", 1, 1, (480, 360), 0, (0, 0, 0, 0), (3234932936,)),
    ("PythonTests.cpp:192", 1, 1, (1920, 1080), 0, (0, 0, 0, 0), (867059050,)),
    ("PythonTests.cpp:196", 1, 1, (480, 360), 1, (0, 0, 0, 0), (1166179972,)),
    ("PythonTests.cpp:200", 1, 1, (540, 304), 1, (0, 0, 0, 0), (2257004891,)),
    ("PythonTests.cpp:204", 1, 1, (1920, 1080), 1, (0, 0, 0, 0), (930427804,)),
    ("PythonTests.cpp:208", 1, 1, (480, 360), 2, (0, 0, 0, 0), (4261607874, 1267073424)),  # the reference accepts two (Windows / Linux)
    ("PythonTests.cpp:212", 1, 1, (540, 304), 2, (0, 0, 0, 0), (4169518778,)),
    ("PythonTests.cpp:216", 1, 1, (1920, 1080), 2, (0, 0, 0, 0), (2402019758,)),
    ("PythonTests.cpp:220", 1, 1, (480, 360), 3, (0, 0, 0, 0), (3175240744,)),
    ("PythonTests.cpp:224", 1, 1, (540, 304), 3, (0, 0, 0, 0), (2257004891,)),
    ("PythonTests.cpp:228", 1, 1, (1920, 1080), 3, (0, 0, 0, 0), (2026855,)),
    ("PythonTests.cpp:232", 1, 1, (1920, 1080), 3, (0, 0, 320, 240), (2884432201,)),
    ("PythonTests.cpp:236", 1, 1, (1920, 1080), 3, (320, 240, 720, 480), (2674082046,)),
    ("PythonTests.cpp:240", 1, 1, (1920, 1080), 3, (720, 480, 1080, 608), (4006833449,)),
    ("PythonTests.cpp:244", 1, 1, (320, 240), 3, (120, 60, 960, 540), (1183295093,)),
]
