// Host build of the thread-tile code of tensor-stream_amd/csrc/vpp_bicubic_r32_core.h (g++, the four hardware operations emulated): resizes a whole
// NV12 frame tile by tile exactly as the kernel's threads do -- same loads, same edge fix-ups, same masks -- so that the CPU suite can compare
// it with the oracle (tests/test_bicubic_r32_cpu.py).  Test infrastructure.
#define BC_HOST_BOUNDS
#include <stddef.h>
#include <string.h>
#include "../../tensor-stream_amd/csrc/vpp_bicubic_r32_core.h"

using namespace tsvpp;

template <int P2>
static int run(const uint8_t *y, const uint8_t *uv, int pitch_y, int pitch_uv, int src_w, int src_h, uint8_t *out) {
    const int dst_w = src_w * 2 / P2, dst_h = src_h * 2 / P2;
    if (dst_w * P2 != src_w * 2 || dst_h * P2 != src_h * 2 || (dst_w & 7) || (dst_h & 3)) return -2;
    uint8_t *oy = out, *ouv = out + (size_t)dst_w * dst_h;
    using G = BcGeom<P2>;
    for (int n4 = 0; n4 < dst_h / 4; n4++)
        for (int q = 0; q < dst_w / 8; q++) {
            const bool first = q == 0, last = G::RUN * (q + 1) == src_w, last_row = 4 * (n4 + 1) == dst_h;
            uint32_t ey[G::NYR][P2 + 2], xy[G::NYR][2], ec[G::NCR][P2 + 2], xc[G::NCR][2];
            // (the planes' last rows end at their width, not at the pitch: nothing past the last sample may be read)
            bc_host_lo = y; bc_host_hi = y + (size_t)(src_h - 1) * pitch_y + src_w;
            bc_load_rows<P2, G::NYR>(y, pitch_y, 2 * P2 * n4 - 1, src_h, q, first, last, ey);
            bc_host_lo = uv; bc_host_hi = uv + (size_t)(src_h / 2 - 1) * pitch_uv + src_w;
            bc_load_rows<P2, G::NCR>(uv, pitch_uv, P2 * n4 - 1, src_h / 2, q, first, last, ec);
            bc_fix_rows<P2, false, G::NYR>(ey, xy, first, last);
            bc_fix_rows<P2, true, G::NCR>(ec, xc, first, last);
            uint32_t ylo[4], yhi[4], clo[2], chi[2];
            bc_tile<P2, true>(ey, xy, ec, xc, last_row, ylo, yhi, clo, chi);
            for (int r = 0; r < 4; r++) {
                memcpy(oy + (size_t)(4 * n4 + r) * dst_w + 8 * q, &ylo[r], 4);
                memcpy(oy + (size_t)(4 * n4 + r) * dst_w + 8 * q + 4, &yhi[r], 4);
            }
            for (int rc = 0; rc < 2; rc++) {
                memcpy(ouv + (size_t)(2 * n4 + rc) * dst_w + 8 * q, &clo[rc], 4);
                memcpy(ouv + (size_t)(2 * n4 + rc) * dst_w + 8 * q + 4, &chi[rc], 4);
            }
        }
    return bc_host_oob ? -9 : 0;
}

extern "C" int bicubic_r32_host(int p2, const uint8_t *y, const uint8_t *uv, int pitch_y, int pitch_uv, int src_w, int src_h, uint8_t *out) {
    return p2 == 3 ? run<3>(y, uv, pitch_y, pitch_uv, src_w, src_h, out) : p2 == 4 ? run<4>(y, uv, pitch_y, pitch_uv, src_w, src_h, out) : -2;
}
