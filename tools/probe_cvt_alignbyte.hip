// Micro-probe (gfx950): rounding of v_cvt_pk_u8_f32 under the default and the round-toward-zero FP mode, and which bits of
// the shift operand v_alignbyte_b32 uses.   hipcc --offload-arch=gfx950 -O2 -o tools/bin/probe_cvt_alignbyte tools/probe_cvt_alignbyte.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void probe(const float *in, uint32_t *out_rne, uint32_t *out_rtz, float *sum_rtz, uint32_t *ab, int n) {
    const int i = threadIdx.x;
    if (i >= n) return;
    const float x = in[i];
    uint32_t a, b;
    asm volatile("v_cvt_pk_u8_f32 %0, %1, 0, 0" : "=v"(a) : "v"(x));
    float s;
    asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 3\n\t"
                 "v_cvt_pk_u8_f32 %0, %2, 0, 0\n\t"
                 "v_add_f32 %1, %2, %2\n\t"
                 "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 0"
                 : "=&v"(b), "=&v"(s) : "v"(x));
    out_rne[i] = a;
    out_rtz[i] = b;
    sum_rtz[i] = s;
    // is the mode switch effective for the very next VALU instruction, both ways?  1 + 1.5 * 2^-24: RNE -> 1 + 2^-23, RTZ -> 1
    float one = 1.0f, eps = 8.940696716308594e-08f, r_rtz, r_rne;
    asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 3\n\t"
                 "v_add_f32 %0, %2, %3\n\t"
                 "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 0\n\t"
                 "v_add_f32 %1, %2, %3"
                 : "=&v"(r_rtz), "=&v"(r_rne) : "v"(one), "v"(eps));
    if (i == 0) { sum_rtz[30] = r_rtz; sum_rtz[31] = r_rne; }
    ab[i] = __builtin_amdgcn_alignbyte(0x77665544u, 0x33221100u, (uint32_t)i);
}
int main() {
    const float h[] = { 0.0f, 0.4f, 0.5f, 0.6f, 1.5f, 2.5f, 2.7f, 3.5f, 254.5f, 254.9f, 255.0f, 255.4f, 255.5f, 255.9f, 256.0f, 300.0f, -0.5f, -3.0f, 127.99999f, 128.5f };
    const int n = sizeof(h) / sizeof(h[0]);
    float *in, *s; uint32_t *a, *b, *ab;
    hipMalloc(&in, sizeof(h)); hipMalloc(&a, 4 * n); hipMalloc(&b, 4 * n); hipMalloc(&s, 4 * 32); hipMalloc(&ab, 4 * n);
    hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
    probe<<<1, 64>>>(in, a, b, s, ab, n);
    uint32_t ha[32], hb[32], hab[32]; float hs[32];
    hipMemcpy(ha, a, 4 * n, hipMemcpyDeviceToHost); hipMemcpy(hb, b, 4 * n, hipMemcpyDeviceToHost);
    hipMemcpy(hs, s, 4 * 32, hipMemcpyDeviceToHost); hipMemcpy(hab, ab, 4 * n, hipMemcpyDeviceToHost);
    for (int i = 0; i < n; i++) printf("x=%11.6f  cvt_pk_u8 rne-mode=%3u  rtz-mode=%3u   alignbyte(shift=%2d)=%08x\n", h[i], ha[i] & 255, hb[i] & 255, i, hab[i]);
    printf("1 + 1.5*2^-24 right after setreg(RTZ): %.9g (RTZ gives 1)   right after setreg(RNE): %.9g (RNE gives 1.00000012)\n", hs[30], hs[31]);
    return 0;
}
