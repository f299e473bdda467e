// exhaustive check: for every 32-bit x, RN(x / d) == fma(fma(-q0,d,x), r, q0) with q0 = x*r, r = RN(1/d)
#include <stdio.h>
#include <stdint.h>
#include <math.h>
#include <omp.h>
int main() {
  const double ds[3] = {30269.0, 30307.0, 30323.0};
  for (int k = 0; k < 3; k++) {
    const double d = ds[k], r = 1.0 / d;
    long bad = 0;
    #pragma omp parallel for reduction(+:bad) schedule(static)
    for (uint64_t x = 0; x < (1ull << 32); x++) {
      double xd = (double)x;
      double q0 = xd * r;
      double rem = fma(-q0, d, xd);
      double q1 = fma(rem, r, q0);
      if (q1 != xd / d) bad++;
    }
    printf("d=%.0f mismatches=%ld\n", d, bad);
  }
  return 0;
}
