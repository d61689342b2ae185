/* Check of the division-by-constant form used by the reprojection gather (csrc/reproject.hip):
 *   q = x * rc;  r = fma(-q, c, x);  q' = fma(r, rc, q)      with rc = RN(1 / c)
 * against the IEEE quotient RN(x / c), over float bit patterns lo, lo + stride, ... < hi.
 * Inputs whose quotient is denormal (|x| < 1e-37) are skipped, signs of zero are not compared.
 *   gcc -O2 -mfma -ffp-contract=off tools/div_const_check.c -lm -o /tmp/divc
 *   /tmp/divc 12 1          all 2^32 patterns for c = 12 (about a minute on one core)
 * Exhaustive runs (stride 1) done for c = 3, 5, 6, 7, 9 ... 16, 18, 20, 24, 32 and 255: no mismatch. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
int main(int argc, char** argv) {
  if (argc < 3) { fprintf(stderr, "usage: %s <c> <stride>\n", argv[0]); return 2; }
  const float c = (float)atof(argv[1]);
  const uint64_t stride = strtoull(argv[2], 0, 10);
  const volatile float one = 1.0f;
  const float rc = one / c;
  uint64_t bad = 0, n = 0;
  for (uint64_t u = 0; u < (1ull << 32); u += stride) {
    uint32_t b = (uint32_t)u;
    float x;
    memcpy(&x, &b, 4);
    if (isnan(x) || isinf(x) || fabsf(x) < 1e-37f) continue;
    const float ref = x / c;
    const float q = x * rc;
    const float q2 = fmaf(fmaf(-q, c, x), rc, q);
    uint32_t a1, a2;
    memcpy(&a1, &ref, 4); memcpy(&a2, &q2, 4);
    ++n;
    if (a1 != a2 && !(isinf(ref) && isinf(q2) && (ref > 0) == (q2 > 0))) ++bad;
  }
  printf("c=%g stride=%llu checked=%llu mismatches=%llu\n", c, (unsigned long long)stride, (unsigned long long)n,
         (unsigned long long)bad);
  return bad != 0;
}
