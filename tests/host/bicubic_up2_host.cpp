// Host build of the thread-tile code of tensor-stream_amd/csrc/vpp_bicubic_up2_core.h (g++, the hardware operations emulated): resizes a whole NV12
// frame tile by tile exactly as the kernel's threads do -- same loads, same edge fix-ups, same masks -- so that the CPU suite can compare it with the
// oracle (tests/test_bicubic_up2_cpu.py).  Test infrastructure.
#define BC_HOST_BOUNDS
#include <stddef.h>
#include <string.h>
#include "../../tensor-stream_amd/csrc/vpp_bicubic_up2_core.h"

using namespace tsvpp;

extern "C" int bicubic_up2_host(const uint8_t *y, const uint8_t *uv, int pitch_y, int pitch_uv, int src_w, int src_h, uint8_t *out) {
    const int dst_w = 2 * src_w, dst_h = 2 * src_h;
    if ((dst_w & 7) || (dst_h & 3)) return -2;
    uint8_t *oy = out, *ouv = out + (size_t)dst_w * dst_h;
    const int ntiles = dst_h / 4;
    for (int n4 = 0; n4 < ntiles; n4++)
        for (int q = 0; q < dst_w / 8; q++) {
            const bool first = q == 0, last = 4 * (q + 1) == src_w;
            uint32_t ey[B2_NYR][3], xay[B2_NYR][2], xby[B2_NYR], ec[B2_NCR][3], xac[B2_NCR][2], xbc[B2_NCR];
            // (the planes' last rows end at their width, not at the pitch: nothing past the last sample may be read)
            bc_host_lo = y; bc_host_hi = y + (size_t)(src_h - 1) * pitch_y + src_w;
            u2_load_rows<B2_NYR>(y, pitch_y, 2 * n4 - 2, src_h, q, first, last, ey);
            bc_host_lo = uv; bc_host_hi = uv + (size_t)(src_h / 2 - 1) * pitch_uv + src_w;
            u2_load_rows<B2_NCR>(uv, pitch_uv, n4 - 2, src_h / 2, q, first, last, ec);
            b2_fix_rows<false, B2_NYR>(ey, xay, xby, first, last);
            b2_fix_rows<true, B2_NCR>(ec, xac, xbc, first, last);
            uint32_t ylo[4], yhi[4], clo[2], chi[2];
            b2_tile<true>(ey, xay, xby, ec, xac, xbc, first, n4 == 0, n4 == ntiles - 1, n4 == ntiles - 2, ylo, yhi, clo, chi);
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
