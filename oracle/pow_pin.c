/* One-off pin (test infrastructure): for EVERY float w in [0,1) -- the whole domain of the
 * bicubic weights, which are float fractions widened to double (reference src/Resize.cu:321-326) --
 * count where libm pow(w,2) != w*w and pow(w,3) != (w*w)*w bit-for-bit in double.
 * w has <=24 significant bits, so w*w is exact and (w*w)*w is the correctly rounded cube.
 * Result on glibc 2.35 (this image): squares 0 differences, cubes 2 582 422 of 1 065 353 216
 * (0.24 %, all 1 ulp): libm pow is not correctly rounded, so "pow()" does not define the
 * reference's bicubic bit-for-bit; oracle and HIP kernel both use the exact products.  Build & run:
 *   gcc -O2 -fopenmp -ffp-contract=off oracle/pow_pin.c -o /tmp/pow_pin -lm && /tmp/pow_pin
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
int main(void) {
    long bad2 = 0, bad3 = 0, n = 0;
    const uint32_t one = 0x3f800000u;
#pragma omp parallel for reduction(+ : bad2, bad3, n) schedule(static)
    for (uint32_t bits = 0; bits < one; bits++) {
        float f;
        memcpy(&f, &bits, 4);
        double w = (double)f;
        double p2 = pow(w, 2), p3 = pow(w, 3);
        double m2 = w * w, m3 = m2 * w;
        bad2 += memcmp(&p2, &m2, 8) != 0;
        bad3 += memcmp(&p3, &m3, 8) != 0;
        n++;
    }
    printf("checked %ld floats in [0,1): pow(w,2)!=w*w: %ld, pow(w,3)!=(w*w)*w: %ld\n", n, bad2, bad3);
    return 0;
}
